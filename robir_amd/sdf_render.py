"""Drop-in for the reference's model/sdf_render.py (NeuS ray-march with hierarchical sampling, :263-374) and for
NormalTrainRunner.get_neus_surface (training/train_normal.py:239-286) on the HIP kernels.

render_neus(rays, model, cos_anneal_ratio, n_samples=64, n_importance=64, n_outside=32, up_sample_steps=4, ...) keeps the reference's
signature and supports n_outside = 0 (what every stage-2 caller passes; the default 32 raises), lindisp = False, and both sampling modes: is_eval / perturb = 0 (deterministic) and
perturb > 0 (the default of the only stage-2 caller, wrap_renderer, :397-399: one torch.rand([R,1]) shift per ray; `t_rand=` pins it).
`model` is robir_amd.nets.NeuSModel (sdf / sdf+feat+gradient / colour run on the MFMA kernels) or any object with the ISDF methods.
The building blocks are public under the reference's names and signatures: sample_pdf, up_sample, cat_z_vals, render_core (render_neus
is written in terms of them), render_core_outside (out of scope: raises), wrap_renderer, the IComp / ISDF protocol classes."""
import collections
import ctypes

import torch

from . import ops
from ._lib import call, ptr, stream_ptr

c_long, c_int, c_float = ctypes.c_long, ctypes.c_int, ctypes.c_float

Rays = collections.namedtuple("Rays", ("origins", "directions", "viewdirs", "radii", "lossmult", "near", "far"))


def _f(t):
    return t.float().contiguous()


class IComp:
    """model/sdf_render.py:10-16: what render_core_outside needs from a model."""

    def radius(self) -> float:
        raise NotImplementedError

    def background(self, x, dirs):
        raise NotImplementedError


class ISDF(IComp):
    """model/sdf_render.py:19-34: the protocol render_core / render_neus march against (robir_amd.nets.NeuSModel implements it on the
    MFMA kernels; any other object with these methods takes the generic path of render_core)."""

    def sdf(self, x):
        raise NotImplementedError

    def sdf_and_feat(self, x):
        raise NotImplementedError

    def color(self, x, gradients, dirs, feature_vector):
        raise NotImplementedError

    def grad(self, x):
        raise NotImplementedError

    def dev(self, x):
        raise NotImplementedError


def _is_native(model):
    """robir_amd's NeuSModel with its own sdf / colour entry points: value, gradient and colour come from the fused kernels."""
    from .nets import NeuSModel
    return isinstance(model, NeuSModel) and "color" not in vars(model) and "sdf" not in vars(model)


def _sdf_values(model, pts):
    """[M] signed distances of pts [M,3]."""
    if _is_native(model):
        return model.sdf_network.eval_points(pts, full=False)[0]
    return _f(model.sdf(pts).reshape(-1))


def _inv_s(model, dev):
    if hasattr(model, "inv_s"):
        return float(model.inv_s())
    return float(model.dev(torch.zeros([1, 3], device=dev))[:, :1].clip(1e-6, 1e6))


def _ray_points(o, d, zz, want_dirs=False):
    R, n = zz.shape
    pts = torch.empty(R * n, 3, device=o.device)
    dd = torch.empty(R * n, 3, device=o.device) if want_dirs else None
    call("rb_ray_points", ptr(o), ptr(d), ptr(zz), c_long(R), c_int(n), ptr(pts), ptr(dd), stream_ptr())
    return pts, dd


def sample_pdf(bins, weights, n_samples, det=False, *, u=None):
    """model/sdf_render.py:37-67: inverse-CDF samples of the piecewise-constant pdf `weights` [R,n-1] over `bins` [R,n] -> [R,n_samples].
    det=False draws u = torch.rand([R, n_samples]) on the device like the reference; `u=` (keyword-only, not in the reference) replays
    recorded draws."""
    bins, weights = _f(bins), _f(weights)
    R = bins.shape[0]
    if u is None:
        if det:
            u = torch.linspace(0. + 0.5 / n_samples, 1. - 0.5 / n_samples, steps=n_samples).to(bins.device)
        else:
            u = torch.rand([R, n_samples], device=bins.device)
    return ops.sample_pdf(bins, weights, _f(u.to(bins.device)))[0]


def up_sample(rays_o, rays_d, z_vals, sdf, n_importance, inv_s, sphere_radius=1.0):
    """model/sdf_render.py:70-114: importance samples [R,n_importance] at a fixed inv_s (one fused kernel: interval weights from the
    min(cos, prev_cos) slope estimate + the deterministic inverse CDF)."""
    o, d, z = _f(rays_o), _f(rays_d), _f(z_vals)
    R, n = z.shape
    per = int(n_importance)
    u = torch.linspace(0.0 + 0.5 / per, 1.0 - 0.5 / per, steps=per).to(o.device)
    wtmp = torch.empty(R, n, device=o.device)
    zn = torch.empty(R, per, device=o.device)
    call("rb_neus_upsample", ptr(o), ptr(d), ptr(z), ptr(_f(sdf.reshape(R, n))), c_long(R), c_int(n), c_int(per), c_float(float(inv_s)),
         c_float(float(sphere_radius)), ptr(u), ptr(wtmp), ptr(zn), stream_ptr())
    return zn


def cat_z_vals(model, rays_o, rays_d, z_vals, new_z_vals, sdf, last=False, *, assume_sorted=False):
    """model/sdf_render.py:117-132: merge the new depths into the sorted list, the SDFs following their depths (the new samples are
    evaluated unless last=True, in which case `sdf` is returned untouched like the reference does).  assume_sorted (keyword-only, not in
    the reference): new_z_vals is already ascending per ray (up_sample's output) -- skips the sort."""
    o, d, z, zn = _f(rays_o), _f(rays_d), _f(z_vals), _f(new_z_vals)
    R, n = z.shape
    per = zn.shape[1]
    if not assume_sorted:
        zn = torch.sort(zn, dim=-1)[0].contiguous()
    sn = None if last else _sdf_values(model, _ray_points(o, d, zn)[0]).reshape(R, per).contiguous()
    z2 = torch.empty(R, n + per, device=o.device)
    s2 = None if last else torch.empty(R, n + per, device=o.device)
    s_old = None if last else _f(sdf.reshape(R, n))
    call("rb_neus_merge", ptr(z), ptr(s_old), c_int(n), ptr(zn), ptr(sn), c_int(per), c_long(R), ptr(z2), ptr(s2), stream_ptr())
    return z2, (sdf if last else s2)


def render_core_outside(rays_o, rays_d, z_vals, sample_dist, model, background_rgb=None):
    raise NotImplementedError("render_core_outside: the NeRF++ background pass (model/sdf_render.py:135-172) runs only for n_outside > 0; every "
                              "configuration and every stage-2 caller uses n_outside = 0 -- OUT OF SCOPE (SURVEY.md section 2 row 6)")


def _core(o, d, z, sample_dist, model, near, far, white_bkgd, stage1_alpha, cos_anneal_ratio, need_grad_error):
    """Shared body of render_core / render_neus: mid-points, network evaluations, compositing (rb_neus_finish)."""
    R, n = z.shape
    dev = o.device
    radius = float(model.radius())
    inv_s = _inv_s(model, dev)
    zmid = torch.empty(R, n, device=dev)
    call("rb_neus_mid_z", ptr(z), c_long(R), c_int(n), c_float(sample_dist), ptr(zmid), stream_ptr())
    pts, dirs = _ray_points(o, d, zmid, want_dirs=True)
    native = _is_native(model)
    pruned = native and not need_grad_error and not stage1_alpha
    if pruned:
        net = model.sdf_network
        sdf_mid = net.eval_points(pts, full=False)[0]                    # [M]: the value the full pass would produce, bit for bit
        w0 = torch.empty(R, n, device=dev)
        keep = torch.empty(R * n, dtype=torch.uint8, device=dev)
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        call("rb_neus_weights", ptr(sdf_mid), ptr(pts), c_long(R), c_int(n), c_float(inv_s), c_float(radius), ptr(w0),
             ptr(keep), ptr(cnt), stream_ptr())
        idx = keep.nonzero()[:, 0]                                       # (the one host sync of this path)
        out = torch.zeros(R * n, 1, device=dev)
        out[:, 0] = sdf_mid
        grad = torch.zeros(R * n, 3, device=dev)
        col = torch.zeros(R * n, 3, device=dev)
        if idx.numel() > 0:
            pk, dk = pts[idx].contiguous(), dirs[idx].contiguous()
            ok, gk = net.eval_points(pk, full=True, grad=True)
            grad[idx] = gk
            col[idx] = model.color_network(pk, gk, dk, ok[:, 1:])
        sdf_stride = 1
    elif native:
        out, grad = model.sdf_network.eval_points(pts, full=True, grad=True)           # [M,257], [M,3]
        col = model.color_network(pts, grad, dirs, out[:, 1:])
        sdf_stride = 257
    else:
        # the ISDF protocol (model/sdf_render.py:19-34, :199-202) on a caller's object -- e.g. wrap_renderer's NeuSModel with `color`
        # replaced by a texture function
        sdf_v, feat = model.sdf_and_feat(pts)
        grad = _f(model.grad(pts).reshape(-1, 3))
        col = _f(model.color(pts, grad, dirs, feat).reshape(-1, 3))
        out = _f(sdf_v.reshape(-1, 1))
        sdf_stride = 1
    rgb, dist, acc = torch.empty(R, 3, device=dev), torch.empty(R, device=dev), torch.empty(R, device=dev)
    nrm, w = torch.empty(R, 3, device=dev), torch.empty(R, n, device=dev)
    gerr = torch.zeros(2, device=dev)
    call("rb_neus_finish", ptr(out), c_long(sdf_stride), ptr(col), ptr(grad), ptr(pts), ptr(zmid), ptr(near), ptr(far), c_long(R),
         c_int(n), c_float(inv_s), c_float(radius), c_int(1 if white_bkgd else 0),
         ptr(z if stage1_alpha else None), ptr(d if stage1_alpha else None), c_float(sample_dist),
         c_float(float(cos_anneal_ratio)), ptr(rgb), ptr(dist), ptr(acc), ptr(nrm), ptr(w), ptr(gerr), stream_ptr())
    ge = gerr[0] / (gerr[1] + 1e-5)
    if pruned:
        ge = torch.full((), float("nan"), device=dev)        # not evaluated on the skipped samples
    return {"rgb": rgb, "dist": dist, "acc": acc, "grad_error": ge, "grad": nrm, "weights": w, "zmid": zmid, "pts": pts, "out": out,
            "sdf_stride": sdf_stride, "gradients": grad, "inv_s": inv_s, "radius": radius}


def render_core(rays_o, rays_d, z_vals, sample_dist, model, background_alpha=None, background_sampled_color=None, background_rgb=None,
                cos_anneal_ratio=0.0):
    """model/sdf_render.py:175-260: composite the samples of sorted depths z_vals [R,n] along the rays.  `model`: robir_amd's NeuSModel
    (fused kernels) or any ISDF implementation.  background_alpha / background_sampled_color (the n_outside > 0 NeRF++ blend) are OUT
    OF SCOPE; background_rgb: None or ones([1,3]) (the white background of render_neus) or zeros."""
    if background_alpha is not None or background_sampled_color is not None:
        raise NotImplementedError("render_core with a NeRF++ background (n_outside > 0) is OUT OF SCOPE (SURVEY.md section 2 row 6)")
    o, d, z = _f(rays_o), _f(rays_d), _f(z_vals)
    R, n = z.shape
    dev = o.device
    white = 0
    if background_rgb is not None:
        b = background_rgb.reshape(-1)
        if bool((b == 1).all()):
            white = 1
        elif not bool((b == 0).all()):
            raise NotImplementedError("render_core: background_rgb is None, all ones (render_neus' white background) or all zeros")
    big = torch.full((R,), 3.0e38, device=dev)
    c = _core(o, d, z, float(sample_dist), model, -big, big, white, False, cos_anneal_ratio, True)
    dists, cdf, inside = ops.neus_core_aux(c["out"], c["sdf_stride"], c["pts"], z, R, n, c["inv_s"], c["radius"], float(sample_dist))
    inv = torch.full((R * n, 1), c["inv_s"], device=dev)
    return {
        "color": c["rgb"],
        "sdf": c["out"][:, :1],
        "dists": dists,
        "gradients": c["gradients"].reshape(R, n, 3),
        "s_val": 1.0 / inv,
        "mid_z_vals": c["zmid"],
        "weights": c["weights"],
        "cdf": cdf,
        "gradient_error": c["grad_error"],
        "inside_sphere": inside,
    }


def render_neus(rays, model, cos_anneal_ratio, n_samples=64, n_importance=64, n_outside=32, up_sample_steps=4,
                white_bkgd=True, lindisp=False, perturb=1.0, is_eval=False, stage1_alpha=False, need_grad_error=True, *, t_rand=None):
    """model/sdf_render.py:263-374 (stage 2; `cos_anneal_ratio` is ignored there).  n_outside must be passed as 0 -- like every caller of the
    reference does (:397-399; neus/config/render.gin:12); the signature's default 32 asks for the NeRF++ background and raises.
    stage1_alpha=True renders with the
    stage-1 render_core instead (neus/volume_render/sdf_render.py:172-190: alpha from the cos-annealed half-section
    extrapolation of the SDF), i.e. what NeuS stage-1 checkpoints were trained against; see render_neus_stage1.

    need_grad_error=False (stage 2 only): the caller does not use the eikonal term `grad_error` (no stage-2 caller of the
    reference does, model/sdf_render.py:376-420).  The weights are then computed first from an SDF-only pass, and gradient +
    colour are evaluated only for samples whose weight is not exactly zero -- rgb / dist / acc / grad / weights are bit-identical
    (0 * colour adds nothing), `grad_error` is NaN.  Pays off for trained sharpness (inv_s in the hundreds: the transmittance
    underflows to 0 a few samples behind the surface); costs one sync to size the compacted batch."""
    if is_eval:
        perturb = 0                                                      # sdf_render.py:273-274
    if n_outside != 0 or lindisp:
        raise NotImplementedError(f"HIP render_neus: n_outside={n_outside}, lindisp={lindisp} -- pass n_outside=0 (the NeRF++ background of "
                                  "n_outside > 0 is OUT OF SCOPE, SURVEY.md section 2 row 6; no configuration uses it) and lindisp=False")
    o, d = _f(rays.origins), _f(rays.directions)
    near, far = _f(rays.near).reshape(-1), _f(rays.far).reshape(-1)
    R, dev = o.shape[0], o.device
    sample_dist = 2.0 / n_samples
    radius = float(model.radius())
    S = stream_ptr

    lin = torch.linspace(0.0, 1.0, n_samples).to(dev)
    z = torch.empty(R, n_samples, device=dev)
    call("rb_neus_coarse_z", ptr(near), ptr(far), ptr(lin), c_long(R), c_int(n_samples), ptr(z), S())
    if perturb > 0:
        # sdf_render.py:293-295: t_rand = torch.rand([R,1]) - 0.5; z_vals += t_rand * 2.0 / n_samples.  `t_rand=` (keyword-only, not in
        # the reference) is that uniform draw [R,1] BEFORE the -0.5, for callers that replay recorded draws
        u = _f(torch.rand([R, 1], device=dev) if t_rand is None else t_rand.to(dev)).reshape(-1)
        assert u.shape[0] == R, (u.shape, R)
        call("rb_neus_jitter_z", ptr(u), c_long(R), c_int(n_samples), ptr(z), S())

    if n_importance > 0:
        sdf = _sdf_values(model, _ray_points(o, d, z)[0]).reshape(R, n_samples)
        for i in range(up_sample_steps):
            zn = up_sample(o, d, z, sdf, n_importance // up_sample_steps, 64 * 2 ** i, radius)
            z, sdf = cat_z_vals(model, o, d, z, zn, sdf, last=(i + 1 == up_sample_steps), assume_sorted=True)
    c = _core(o, d, z.contiguous(), sample_dist, model, near, far, white_bkgd, stage1_alpha, cos_anneal_ratio,
              need_grad_error or stage1_alpha)
    if stage1_alpha:        # the stage-1 function's own result dict (neus/volume_render/sdf_render.py:358-365)
        return {"rgb": c["rgb"], "dist": c["dist"], "acc": c["acc"], "sim_or_grad": c["grad_error"], "weights": c["weights"], "means": c["zmid"]}
    return {k: c[k] for k in ("rgb", "dist", "acc", "grad_error", "grad", "weights")}


def wrap_renderer(my_sdf_model, color_fn, model_input, near=1.0, far=6.0, is_eval=False):
    """model/sdf_render.py:377-426 (no caller in the reference tree): a NeuS march over model_input['points'] / ['dirs'] whose colour is
    `color_fn(x / 2)` instead of the colour network, returned in IDRNetwork.forward's key layout.  Like the reference this REPLACES
    `neus.color` on the model and puts it in eval mode."""
    rays_o, rays_d = model_input["points"].reshape(-1, 3), model_input["dirs"].reshape(-1, 3)
    rays_o = rays_o * 2.0
    ones = torch.ones_like(rays_o[..., :1])
    zeros_rgb = torch.zeros_like(rays_o)
    rays = Rays(rays_o, rays_d, rays_d, ones * 0.001, ones, ones * near, ones * far)
    neus = my_sdf_model.implicit_network.neus_model

    def color(x, gradients, dirs, feature_vector):
        return color_fn(x * 0.5)

    neus.color = color
    neus.eval()
    with torch.no_grad():       # the kernels carry no autograd either way (forward-only build)
        ret = render_neus(rays, neus, 1.0, n_samples=32, n_importance=32, n_outside=0, up_sample_steps=2)
    expand3 = lambda v: v[..., None].expand(-1, 3)      # noqa: E731
    out = {"points": rays_o, "sdf_output": ones, "network_object_mask": ones, "object_mask": ones}
    out.update({
        "bg_rgb": zeros_rgb, "sg_rgb": ret["rgb"], "indir_rgb": zeros_rgb,
        "sg_diffuse_rgb": expand3(ret["dist"]), "sg_specular_rgb": expand3(ret["acc"]),
        "indir_diffuse_rgb": zeros_rgb, "indir_specular_rgb": zeros_rgb,
        "normals": ret["grad"], "diffuse_albedo": ret["rgb"], "roughness": zeros_rgb, "surface_mask": ret["acc"] > 0.8,
        "vis_shadow": zeros_rgb, "random_xi_roughness": zeros_rgb, "random_xi_diffuse_albedo": zeros_rgb, "pe_rgb": zeros_rgb,
    })
    return out


def render_neus_stage1(rays, model, cos_anneal_ratio, n_samples=64, n_importance=64, n_outside=32, up_sample_steps=4,
                       white_bkgd=True, lindisp=False, perturb=1.0, is_eval=False, *, t_rand=None):
    """neus/volume_render/sdf_render.py:238-365 (stage-1 `render_neus`; n_outside must be passed as 0 like every stage-2
    caller does -- the NeRF++ background is out of scope)."""
    return render_neus(rays, model, cos_anneal_ratio, n_samples, n_importance, n_outside, up_sample_steps, white_bkgd,
                       lindisp, perturb, is_eval, stage1_alpha=True, t_rand=t_rand)


def get_neus_surface(implicit_network, points, view_dirs, pred_normals, n_samp=32, dist=0.05):
    """NormalTrainRunner.get_neus_surface: -> (final_x [m,3], final_normal [m,3], gradient_error)."""
    p, v, pn = _f(points), _f(view_dirs), _f(pred_normals)
    m, dev = p.shape[0], p.device
    tk = torch.linspace(0, dist, n_samp).to(dev)
    xs = torch.empty(m * n_samp, 3, device=dev)
    call("rb_surface_points", ptr(p), ptr(v), ptr(tk), c_long(m), c_int(n_samp), ptr(xs), stream_ptr())
    sdf, grad = implicit_network.neus_model.sdf_network.eval_points(xs, 2.0, 0.5, full=False, grad=True)
    s = implicit_network.neus_model.inv_s_unclipped()     # cached on the parameter's version: no device read per call
    x_out, n_out = torch.empty(m, 3, device=dev), torch.empty(m, 3, device=dev)
    gerr = torch.zeros(2, device=dev)
    call("rb_surface_finish", ptr(sdf), ptr(grad), ptr(xs), ptr(p), ptr(pn), c_long(m), c_int(n_samp), c_float(s),
         ptr(x_out), ptr(n_out), ptr(gerr), stream_ptr())
    return x_out, n_out, gerr[0] / (gerr[1] + 1e-5)
