"""Drop-in for the reference's model/sg_render.py: same public names and signatures, HIP kernels underneath.

render_with_all_sg / render_with_sg (sg_render.py:304-565), get_diffuse_visibility (:111-195),
get_specular_visibility (:198-301), hemisphere_int / lambda_trick / norm_axis (:62-108), envmap helpers (:9-59).

Extensions (keyword-only, ignored by reference callers):
  draws     dict of explicit uniform draws replacing the torch.rand calls ('dvis_theta','dvis_phi' [C,L,nsamp] or
            [L,nsamp]; 'svis_theta_dir','svis_phi_dir','svis_theta_ind','svis_phi_ind' [n,8]);
  chunk_id  int32 [n] + n_chunks: several 1024-pixel chunks in one call, each keeping its own draws and its own
            batch-global minimum in the specular cone (sg_render.py:222), i.e. exactly what the reference computes
            when it renders the chunks one after another;
  stats     dict receiving 'diffuse_vis_evals' (device counter tensor).
"""
import math

import numpy as np
import torch

from . import ops
from .nets import VisNetwork
from .octree_tracing import OctreeVisModel

TINY_NUMBER = 1e-6
OCTREE_VIS_BATCH = 2000000      # pairs per VisModel call of the reference (sg_render.py:158): one lock-step cast each
# Arithmetic of the hidden layers of the fused light-visibility kernel:
#   "f16x6"      (default) exact fp32 operands as three halves each, the six partial products of weight >= 2^-22 on the f16
#                MFMA in three fp32 accumulators (csrc/vis_diffuse_x6.hip): not narrower than the reference's fp32;
#   "fp32"       f32-input MFMA (bitwise an fp32 fma chain): the same results to summation order at ~half the speed;
#   "f16x3-auto" split precision (hi/lo half pairs = 22-bit operands, three products, ~2^-22 relative error; parity-tested
#                throughput mode, tests/test_precision_gpu.py anchors it on float64) with the kernel generation picked by
#                launch size: "f16x3-v3" (global tile list + persistent grid) for small launches, "f16x3-v2" (one point per
#                workgroup) for whole views -- bit-identical to each other; "f16x3" = first generation.
import os as _os
from .precision import vis_precision as _vis_precision
VIS_PRECISION = _vis_precision()


# ----------------------------------------------------------------------------------------- small public helpers
def norm_axis(x):
    return ops.normalize3(x.reshape(-1, 3).float().contiguous(), TINY_NUMBER, 0).reshape(x.shape)


def hemisphere_int(lambda_val, cos_beta):
    """Integral of exp(lambda (w.p - 1)) over the hemisphere whose pole makes angle beta with the lobe axis p
    (sg_render.py:62-81, the Meder & Bruederlin fit).  Public helper for callers of the reference's module; the renderer
    itself evaluates this inside rb_sg_shade.  Plain element-wise torch on the inputs' device."""
    lam = lambda_val + TINY_NUMBER
    r = 1.0 / lam
    t = lam.sqrt() * (1.6988 + 10.8438 * r) / (1.0 + 6.2201 * r + 10.2415 * r * r)
    ea = torch.exp(-t)
    up = cos_beta >= 0
    eb = torch.exp(-t * cos_beta.clamp(min=0.0))
    s_up = (1.0 - ea * eb) / (1.0 - ea + eb - ea * eb)
    b = torch.exp(t * cos_beta.clamp(max=0.0))
    s_dn = (b - ea) / ((1.0 - ea) * (b + 1.0))
    s = up.to(s_up.dtype) * s_up + (~up).to(s_up.dtype) * s_dn           # blend, not where(): NaN propagates like the reference
    e1 = torch.exp(-lam)
    lower = 2.0 * np.pi / lam * (e1 - torch.exp(-2.0 * lam))
    upper = 2.0 * np.pi / lam * (1.0 - e1)
    return lower * (1.0 - s) + upper * s


def lambda_trick(lobe1, lambda1, mu1, lobe2, lambda2, mu2):
    """Product of two SGs as one SG, arranged for lambda1 << lambda2 (sg_render.py:84-104).  Public helper, see above."""
    q = lambda1 / lambda2
    a1 = lobe1 / (torch.norm(lobe1, dim=-1, keepdim=True) + TINY_NUMBER)
    a2 = lobe2 / (torch.norm(lobe2, dim=-1, keepdim=True) + TINY_NUMBER)
    c = (a1 * a2).sum(-1, keepdim=True)
    m = torch.minimum(torch.sqrt(q * q + 1.0 + 2.0 * q * c), q + 1.0)
    return (q / m) * a1 + (1.0 / m) * a2, lambda2 * m, mu1 * mu2 * torch.exp(lambda2 * (m - q - 1.0))


def render_envmap_sg(lgtSGs, viewdirs):
    """sum_k mu_k exp(lambda_k (d.lobe_k - 1)) (sg_render.py:26-42)."""
    shape = list(viewdirs.shape[:-1]) + [3]
    d = viewdirs.to(lgtSGs.device).reshape(-1, 3).float().contiguous()
    return ops.envmap_sg(lgtSGs.detach().float().contiguous(), d).reshape(shape)


def compute_envmap(lgtSGs, H, W, upper_hemi=False):
    phi, theta = torch.meshgrid([torch.linspace(0.0, np.pi / 2.0 if upper_hemi else np.pi, H),
                                 torch.linspace(1.0 * np.pi, -1.0 * np.pi, W)], indexing="ij")
    viewdirs = torch.stack([torch.cos(theta) * torch.sin(phi), torch.sin(theta) * torch.sin(phi), torch.cos(phi)], -1)
    return render_envmap_sg(lgtSGs, viewdirs).reshape(H, W, 3)


def render_envmap(envmap, viewdirs):
    """Bilinear lat-long lookup (sg_render.py:45-59): envmap [H,W,3], viewdirs [n,3] -> [n,3]."""
    d = viewdirs.reshape(-1, 3).float().contiguous()
    return ops.envmap_lookup(envmap.to(d.device).float().contiguous(), d)


# ----------------------------------------------------------------------------------------- visibility
def _rand(shape, device):
    return torch.rand(*shape, device=device)


# The partition of the points at hand into lock-step chunks, for callers that cannot say (a runner's own get_sg_render calls
# render_with_all_sg with the reference's arguments): set by IDRNetwork._shade around the hook call when it passes the hit
# points of several chunks at once, so that every chunk keeps its own sample tables and specular minimum.
_BATCH_CTX = None


def _chunks(chunk_id, n_chunks, n=None):
    if chunk_id is None and _BATCH_CTX is not None and n == _BATCH_CTX[2]:
        return _BATCH_CTX[0], _BATCH_CTX[1]
    return (None, 1) if chunk_id is None else (chunk_id, int(n_chunks))


def get_diffuse_visibility(points, normals, VisModel, lgtSGLobes, lgtSGLambdas, nsamp=8, testing=False, thr=1.0,
                           bounding=False, argmax_vis=False, *, draws=None, chunk_id=None, n_chunks=1, stats=None):
    """-> vis [L, n] (sg_render.py:111-195).  lgtSGLobes [L,3], lgtSGLambdas [L,1].  bounding=True (sg_render.py:185-186; no
    caller in the reference) returns the per-sample visibilities [L, nsamp, n] before the lobe-weighted mean: the fused kernels never
    hold them, so that branch evaluates VisModel on the culled pairs (the MLP kernel on points / directions) and scatters."""
    dev = points.device
    L = lgtSGLobes.shape[0]
    n = points.shape[0]
    cid, C = _chunks(chunk_id, n_chunks, n)
    lgt = torch.zeros(L, 7, device=dev)
    lgt[:, :3] = lgtSGLobes
    lgt[:, 3:4] = lgtSGLambdas
    if draws is not None and "dvis_theta" in draws:
        u_t, u_p = draws["dvis_theta"], draws["dvis_phi"]
    else:
        u_t, u_p = _rand((C, L, nsamp), dev), _rand((C, L, nsamp), dev)
    if bounding:
        dirs, wdir, wsum = ops.dvis_dirs(lgt, u_t.to(dev), u_p.to(dev), thr, direct=True)
        return _diffuse_vis_generic(points, normals, VisModel, dirs, wdir, wsum, cid, C, L, nsamp, argmax_vis, per_sample=True)
    return _diffuse_vis_core(points, normals, VisModel, lgt, u_t, u_p, thr, argmax_vis, cid, C, stats, direct=True).t()


def _diffuse_vis_core(points, normals, VisModel, lgt, u_t, u_p, thr, argmax_vis, cid, C, stats, direct=False):
    """-> [n, L].  lgt [L,7]: raw light SGs (direct=False, render_with_sg's chain: lobe normalised twice, |lambda|) or the
    lobes / lambdas of a direct get_diffuse_visibility call (direct=True: normalised once, lambda as given)."""
    dev = points.device
    if u_t.dim() == 2:
        u_t, u_p = u_t[None], u_p[None]
    assert u_t.shape[0] == C
    _, L, nsamp = u_t.shape
    dirs, wdir, wsum = ops.dvis_dirs(lgt, u_t.to(dev), u_p.to(dev), thr, direct=direct)
    if isinstance(VisModel, VisNetwork):
        sp = VisModel.packed_split()
        if ops.SDF_FUSED_PE:       # first-layer halves straight from the points / directions (encoding fused)
            A = ops.linear_pe10_256(points.float().contiguous(), sp["point"])
            Bd = ops.linear_pe10_256(dirs, sp["dir"])
        else:
            A = ops.linear_64_256(ops.feat_pe10(points.float().contiguous()), sp["point"])
            Bd = ops.linear_64_256(ops.feat_pe10(dirs), sp["dir"])
        cnt = None
        if stats is not None:
            cnt = stats.setdefault("diffuse_vis_evals", torch.zeros(1, dtype=torch.int64, device=dev))
        return ops.dvis_fused(normals.float().contiguous(), cid, A, Bd, dirs, wdir, wsum, sp, L, nsamp, argmax_vis, cnt,
                              precision=VIS_PRECISION)
    if isinstance(VisModel, OctreeVisModel):
        # traced visibility (trace_vis, train_pbr.py:409-410): fused cull + lock-step secondary cast in the reference's
        # 2 M-pair batches per chunk, no pairs materialised (csrc/octree_vis.hip)
        cnt = None
        if stats is not None:
            cnt = stats.setdefault("diffuse_vis_evals", torch.zeros(1, dtype=torch.int64, device=dev))
        tree = VisModel.ray_tracer.sdf_octree
        return ops.dvis_octree(tree.tables, points.float().contiguous(), normals.float().contiguous(), cid, C, dirs, wdir, wsum,
                               L, nsamp, argmax_vis, cnt, batch_pairs=OCTREE_VIS_BATCH, max_iter=tree.max_iter)
    return _diffuse_vis_generic(points, normals, VisModel, dirs, wdir, wsum, cid, C, L, nsamp, argmax_vis)


def _diffuse_vis_generic(points, normals, VisModel, dirs, wdir, wsum, cid, C, L, nsamp, argmax_vis, per_sample=False):
    """Any other VisModel callable (e.g. OctreeVisModel): directions/weights from the HIP kernel, the callable is
    evaluated on the culled pairs in 2M batches like the reference."""
    n = points.shape[0]
    LS = L * nsamp
    d = dirs.reshape(C, LS, 3)
    c = torch.zeros(n, dtype=torch.long, device=points.device) if cid is None else cid.long()
    dd = d[c]                                                   # [n, LS, 3]
    front = (normals.unsqueeze(1) * dd).sum(-1) > TINY_NUMBER
    pi_, di_ = front.nonzero(as_tuple=True)
    logits = torch.zeros(pi_.shape[0], 2, device=points.device)
    for s in range(0, pi_.shape[0], OCTREE_VIS_BATCH):          # sg_render.py:158: batches of 2 000 000 pairs
        e = s + OCTREE_VIS_BATCH
        logits[s:e] = VisModel(points[pi_[s:e]], dd[pi_[s:e], di_[s:e]])
    pv = logits.argmax(-1).float() if argmax_vis else torch.softmax(logits, -1)[..., 1]
    vis = torch.zeros(n, LS, device=points.device)
    vis[front] = pv
    if per_sample:                                              # bounding=True: [L, nsamp, n] (sg_render.py:177,185-186)
        return vis.reshape(n, L, nsamp).permute(1, 2, 0)
    w = wdir.reshape(C, L, nsamp)[c]
    return (vis.reshape(n, L, nsamp) * w).sum(-1) / wsum.reshape(C, L)[c]


def _specular_vis_core(points, normals, viewdirs, VisModel, roughness, u_t, u_p, testing, inv, argmax_vis, cid, C, lobes=None,
                       lambdas=None):
    """-> bvis [n] for the warped BRDF lobe of each point: recomputed on the device from (normal, view, roughness) -- render_with_sg's
    own chain --, or the caller's lobes [n,3] / lambdas [n] when those are given (get_specular_visibility's reference signature)."""
    n, nsamp = u_t.shape
    if lobes is not None:
        dirs, wts, front = ops.spec_vis_sample_lobes(normals, viewdirs, lobes, lambdas, cid, C, u_t, u_p)
    else:
        dirs, wts, front = ops.spec_vis_sample(normals, viewdirs, roughness, cid, C, u_t, u_p)
    if isinstance(VisModel, VisNetwork):
        logits = VisModel.logits_from_points(points.float().contiguous(), dirs, rep=nsamp)
    elif isinstance(VisModel, OctreeVisModel) and cid is not None and C > 1:
        # several chunks in one call: each chunk's n_c * nsamp rays are their own lock-step batch, as in the reference's
        # per-chunk calls (group boundaries from the ascending chunk ids, computed on the device)
        gs = torch.searchsorted(cid.long(), torch.arange(C + 1, device=cid.device)) * nsamp
        logits = VisModel.forward_groups(points.unsqueeze(1).expand(-1, nsamp, 3).reshape(-1, 3), dirs, gs).contiguous()
    else:
        logits = VisModel(points.unsqueeze(1).expand(-1, nsamp, 3).reshape(-1, 3), dirs).float().contiguous()
    return ops.spec_vis_reduce(logits, front, wts, n, nsamp, inv, argmax_vis, testing)


def get_specular_visibility(points, normals, viewdirs, VisModel, lgtSGLobes, lgtSGLambdas, nsamp=24, multi_view=False,
                            testing=False, inv=False, argmax_vis=False, *, roughness=None, draws=None):
    """sg_render.py:198-301 with the reference's own signature: the cone around the reflected view direction opens by the PASSED
    lgtSGLambdas ([n,1] | [n]: clip 0.1..50, batch-global minimum, :219-223) and the samples are weighted by the PASSED lgtSGLobes
    [n,3] as they are (:281) -- single view, or multi_view=True with viewdirs / lobes / lambdas [V,n,.] (:227-231, 247-258).
    `roughness=` (keyword-only, not in the reference) is the path render_with_sg itself takes: lobe and lambda are then recomputed on
    the device from (normal, view, roughness) and the passed ones are not read (they may be None).  `draws=(u_theta, u_phi)` [n,nsamp]
    pins the two torch.rand draws (:224-225)."""
    n = points.shape[0]
    dev = points.device
    f = lambda t: t.to(dev).float().contiguous()
    if roughness is None and (lgtSGLobes is None or lgtSGLambdas is None):
        raise ValueError("get_specular_visibility needs lgtSGLobes / lgtSGLambdas (the reference's arguments) or roughness=")
    if draws is None:
        u_t, u_p = _rand((n, nsamp), dev), _rand((n, nsamp), dev)
    else:
        u_t, u_p = f(draws[0]), f(draws[1])
    if multi_view:
        # viewdirs [V,n,3] -> [V,n]: one batch of V n rows, the draws [n,nsamp] shared by the views, always the arg-max of the logits
        # (sg_render.py:227-231, 247-258: inv / argmax_vis are not read in this branch)
        V = viewdirs.shape[0]
        args = (f(points).repeat(V, 1), f(normals).repeat(V, 1), f(viewdirs).reshape(V * n, 3).contiguous(), VisModel)
        if roughness is not None:
            return _specular_vis_core(*args, f(roughness).reshape(-1).repeat(V), u_t.repeat(V, 1), u_p.repeat(V, 1), testing, False, True,
                                      None, 1).reshape(V, n)
        lob = f(lgtSGLobes).expand(V, n, 3).reshape(V * n, 3).contiguous()
        lam = f(lgtSGLambdas).reshape(-1, n).expand(V, n).reshape(V * n).contiguous()
        return _specular_vis_core(*args, None, u_t.repeat(V, 1), u_p.repeat(V, 1), testing, False, True, None, 1, lobes=lob,
                                  lambdas=lam).reshape(V, n)
    if roughness is not None:
        return _specular_vis_core(f(points), f(normals), f(viewdirs), VisModel, f(roughness).reshape(-1), u_t, u_p, testing, inv, argmax_vis,
                                  None, 1)
    return _specular_vis_core(f(points), f(normals), f(viewdirs), VisModel, None, u_t, u_p, testing, inv, argmax_vis, None, 1,
                              lobes=f(lgtSGLobes).reshape(n, 3), lambdas=f(lgtSGLambdas).reshape(n))


def _kl_divergence(x, mu):
    """utils/utils.py:14-17."""
    rho_hat = torch.mean(x, 0)
    rho = torch.full_like(rho_hat, mu)
    return torch.mean(rho * torch.log(rho / (rho_hat + 1e-4)) + (1 - rho) * torch.log((1 - rho) / (1 - rho_hat + 1e-4)))


# ----------------------------------------------------------------------------------------- shading
def render_with_sg(points, normal, viewdirs, lgtSGs, specular_reflectance, roughness, diffuse_albedo, comp_vis=True,
                   VisModel=None, fun_spec=False, lin_diff=False, testing=False, indir_integral=None, metallic=None,
                   diffuse_vis=None, prefit=False, argmax_vis=False, *, draws=None, chunk_id=None, n_chunks=1,
                   stats=None):
    """sg_render.py:343-565.  viewdirs [n,3], or [V,n,3] for the MULTI_VIEW form (_render_with_sg_multi_view).  lgtSGs [n,M,7] (or [M,7]).  fun_spec=True (sg_render.py:413,544-551): the specular term
    comes back as a function of a roughness tensor (its specular visibility is sampled at every call, like the reference's closure;
    `draws=` pins the samples), `sg_rgb` is then the diffuse term alone."""
    if viewdirs.dim() == 3:
        return _render_with_sg_multi_view(points, normal, viewdirs, lgtSGs, specular_reflectance, roughness, diffuse_albedo, comp_vis,
                                          VisModel, fun_spec, lin_diff, testing, indir_integral, metallic, diffuse_vis, prefit,
                                          argmax_vis, draws, chunk_id, n_chunks, stats)
    dev = points.device
    n = points.shape[0]
    cid, C = _chunks(chunk_id, n_chunks, n)
    draws = draws or {}
    pts = points.float().contiguous()
    nrm = normal.float().contiguous()
    vd = viewdirs.float().contiguous()
    rough = roughness.float().contiguous().reshape(-1)
    f0 = specular_reflectance          # stays on the device: rb_sg_shade reads the scalar itself
    shared = lgtSGs.dim() == 2 or (lgtSGs.stride(0) == 0)
    lgt_first = (lgtSGs if lgtSGs.dim() == 2 else lgtSGs[0]).float().contiguous()
    light_vis = None
    supervise = torch.zeros((), device=dev)
    if comp_vis:
        nsamp = 32 if diffuse_vis is None else 8          # sg_render.py:389
        u_t = draws.get("dvis_theta")
        u_p = draws.get("dvis_phi")
        if u_t is None:
            L = lgt_first.shape[0]
            u_t, u_p = _rand((C, L, nsamp), dev), _rand((C, L, nsamp), dev)
        # first row's light for every point (sg_render.py:388-390)
        light_vis = _diffuse_vis_core(pts, nrm, VisModel, lgt_first, u_t, u_p, 1.0, argmax_vis, cid, C, stats)
        if diffuse_vis is not None:
            # CESR: the shadow net's prediction replaces the sampled visibility in shading except during warm-up;
            # the sampled one only feeds the KL supervision term (sg_render.py:393-403; a loss, evaluated with torch ops)
            pred = diffuse_vis.reshape(-1, lgt_first.shape[0]).float().contiguous()
            factor = {"warmup": 0.1, "project": 0.2}.get(prefit, 1.0)
            supervise = _kl_divergence((light_vis - pred).abs(), 0.01) * factor
            if prefit != "warmup":
                light_vis = pred
    lgt = lgt_first if shared else lgtSGs.float().contiguous()

    def shade(rough_, spec_draws):
        u_t, u_p = spec_draws.get("svis_theta"), spec_draws.get("svis_phi")
        if u_t is None:
            u_t, u_p = _rand((n, 8), dev), _rand((n, 8), dev)
        bvis = _specular_vis_core(pts, nrm, vd, VisModel, rough_, u_t.to(dev), u_p.to(dev), testing, not comp_vis, argmax_vis,
                                  cid, C)
        return ops.sg_shade(nrm, vd, lgt, f0, rough_, diffuse_albedo, bvis, light_vis=light_vis, metallic=metallic,
                            indir_integral=indir_integral, lin_diff=lin_diff, want_shadow=True)

    if fun_spec:
        # the diffuse term and the shadow do not depend on the specular visibility: one shading pass with bvis = 1 gives them
        _, _, diff, shadow = ops.sg_shade(nrm, vd, lgt, f0, rough, diffuse_albedo, torch.ones(n, device=dev), light_vis=light_vis,
                                          metallic=metallic, indir_integral=indir_integral, lin_diff=lin_diff, want_shadow=True)

        def specular_rgb_fn(roughness, draws=None):
            return shade(roughness.float().contiguous().reshape(-1), draws or {})[1]

        return {"sg_rgb": diff, "sg_specular_rgb": specular_rgb_fn, "sg_diffuse_rgb": diff, "vis_shadow": shadow,
                "supervise": supervise}
    rgb, spec, diff, shadow = shade(rough, draws)
    return {"sg_rgb": rgb, "sg_specular_rgb": spec, "sg_diffuse_rgb": diff, "vis_shadow": shadow, "supervise": supervise}


def _render_with_sg_multi_view(points, normal, viewdirs, lgtSGs, specular_reflectance, roughness, diffuse_albedo, comp_vis, VisModel,
                               fun_spec, lin_diff, testing, indir_integral, metallic, diffuse_vis, prefit, argmax_vis, draws, chunk_id,
                               n_chunks, stats):
    """MULTI_VIEW form (sg_render.py:356, 375-378, 465-470; get_specular_visibility's multi_view branches, :227-231, 247-258): viewdirs
    [V,n,3].  Light visibility, diffuse term, shadow and supervision do not see the view (computed once, [n,.]); the specular term of
    every view comes from ONE batch of V n rows (row v n + i = point i seen from view v) through the single-view kernels: one set of
    draws [n,16] serves all views, the cone range takes the smallest sharpness of the whole batch, and the sampled visibility is always
    the arg-max of the logits, never inverted (the reference's branch ignores inv / argmax_vis) -> sg_specular_rgb, sg_rgb [V,n,3].
    (With exactly three views the reference's `torch.cross(z_axis, ref_dir)` -- no dim= -- crosses along the VIEW axis, the first of size
    3: this mirror always crosses along xyz.)"""
    if n_chunks != 1:
        raise ValueError("multi-view shading is one lock-step batch (n_chunks = 1)")
    V, n = viewdirs.shape[0], points.shape[0]
    dev = points.device
    base = render_with_sg(points, normal, viewdirs[0], lgtSGs, specular_reflectance, roughness, diffuse_albedo, comp_vis=comp_vis,
                          VisModel=VisModel, fun_spec=True, lin_diff=lin_diff, testing=testing, indir_integral=indir_integral,
                          metallic=metallic, diffuse_vis=diffuse_vis, prefit=prefit, argmax_vis=argmax_vis, draws=draws, stats=stats)
    pts = points.float().repeat(V, 1)
    nrm = normal.float().repeat(V, 1)
    vd = viewdirs.float().reshape(V * n, 3).contiguous()
    alb = diffuse_albedo.float().repeat(V, 1)
    met = metallic.float().reshape(n, -1).repeat(V, 1) if metallic is not None else None
    shared = lgtSGs.dim() == 2 or lgtSGs.stride(0) == 0
    lgt = ((lgtSGs if lgtSGs.dim() == 2 else lgtSGs[0]) if shared else lgtSGs.repeat(V, 1, 1)).float().contiguous()

    def specular_rgb_fn(roughness, draws=draws):
        draws = draws or {}
        rough = roughness.float().reshape(-1).repeat(V)
        u_t, u_p = draws.get("svis_theta"), draws.get("svis_phi")
        if u_t is None:
            u_t, u_p = _rand((n, 16), dev), _rand((n, 16), dev)      # nsamp = 16 in this branch (:467)
        bvis = _specular_vis_core(pts, nrm, vd, VisModel, rough, u_t.to(dev).repeat(V, 1), u_p.to(dev).repeat(V, 1), testing, False,
                                  True, None, 1)
        return ops.sg_shade(nrm, vd, lgt, specular_reflectance, rough, alb, bvis, metallic=met, lin_diff=lin_diff)[1].reshape(V, n, 3)

    if fun_spec:
        return dict(base, sg_specular_rgb=specular_rgb_fn)
    spec = specular_rgb_fn(roughness)
    return dict(base, sg_rgb=spec + base["sg_diffuse_rgb"], sg_specular_rgb=spec)


def render_with_all_sg(points, normal, viewdirs, lgtSGs, specular_reflectance, roughness, diffuse_albedo,
                       indir_integral=None, indir_lgtSGs=None, VisModel=None, fun_spec=False, lin_diff=False,
                       testing=False, metallic=None, diffuse_vis=None, prefit=False, argmax_vis=False, *, draws=None,
                       chunk_id=None, n_chunks=1, stats=None):
    """sg_render.py:304-337: direct pass (light visibility) + indirect pass (per-point lobes, inverted specular
    visibility, diffuse := indir_integral)."""
    draws = draws or {}
    d_dir = {k: draws[k] for k in ("dvis_theta", "dvis_phi") if k in draws}
    if "svis_theta_dir" in draws:
        d_dir.update(svis_theta=draws["svis_theta_dir"], svis_phi=draws["svis_phi_dir"])
    ret = render_with_sg(points, normal, viewdirs, lgtSGs, specular_reflectance, roughness, diffuse_albedo,
                         comp_vis=True, VisModel=VisModel, fun_spec=fun_spec, lin_diff=lin_diff, testing=testing,
                         metallic=metallic, diffuse_vis=diffuse_vis, prefit=prefit, argmax_vis=argmax_vis, draws=d_dir,
                         chunk_id=chunk_id, n_chunks=n_chunks, stats=stats)
    zeros = torch.zeros_like(points)
    ind = {"sg_rgb": zeros, "sg_diffuse_rgb": zeros, "sg_specular_rgb": zeros}
    if indir_lgtSGs is not None:
        d_ind = {}
        if "svis_theta_ind" in draws:
            d_ind.update(svis_theta=draws["svis_theta_ind"], svis_phi=draws["svis_phi_ind"])
        ind = render_with_sg(points, normal, viewdirs, indir_lgtSGs, specular_reflectance, roughness, diffuse_albedo,
                             comp_vis=False, VisModel=VisModel, fun_spec=fun_spec, lin_diff=lin_diff, testing=testing,
                             indir_integral=indir_integral, metallic=metallic, diffuse_vis=None, argmax_vis=argmax_vis,
                             draws=d_ind, chunk_id=chunk_id, n_chunks=n_chunks)
    ret.update({"indir_rgb": ind["sg_rgb"], "indir_diffuse_rgb": ind["sg_diffuse_rgb"],
                "indir_specular_rgb": ind["sg_specular_rgb"]})
    return ret
