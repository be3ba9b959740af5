"""Drop-in for the reference's model/sdf_render.py (NeuS ray-march with hierarchical sampling, :263-374) and for
NormalTrainRunner.get_neus_surface (training/train_normal.py:239-286) on the HIP kernels.

render_neus(rays, model, cos_anneal_ratio, n_samples=64, n_importance=64, n_outside=0, up_sample_steps=4, ...)
supports n_outside = 0 (every stage-2 caller), lindisp = False, and both sampling modes: is_eval / perturb = 0 (deterministic) and
perturb > 0 (the default of the only stage-2 caller, wrap_renderer, :397-399: one torch.rand([R,1]) shift per ray; `t_rand=` pins it).
`model` is robir_amd.nets.NeuSModel (sdf / sdf+feat+gradient / colour run on the MFMA kernels)."""
import collections
import ctypes

import torch

from . import ops
from ._lib import call, ptr, stream_ptr

c_long, c_int, c_float = ctypes.c_long, ctypes.c_int, ctypes.c_float

Rays = collections.namedtuple("Rays", ("origins", "directions", "viewdirs", "radii", "lossmult", "near", "far"))


def _f(t):
    return t.float().contiguous()


def render_neus(rays, model, cos_anneal_ratio=1.0, n_samples=64, n_importance=64, n_outside=0, up_sample_steps=4,
                white_bkgd=True, lindisp=False, perturb=1.0, is_eval=False, stage1_alpha=False, need_grad_error=True, *, t_rand=None):
    """model/sdf_render.py:263-374 (stage 2; `cos_anneal_ratio` is ignored there).  stage1_alpha=True renders with the
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
        raise NotImplementedError("HIP render_neus: n_outside=0 (no NeRF++ background), lindisp=False")
    o, d = _f(rays.origins), _f(rays.directions)
    near, far = _f(rays.near).reshape(-1), _f(rays.far).reshape(-1)
    R, dev = o.shape[0], o.device
    net = model.sdf_network
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

    def points(zz, want_dirs=False):
        n = zz.shape[1]
        pts = torch.empty(R * n, 3, device=dev)
        dd = torch.empty(R * n, 3, device=dev) if want_dirs else None
        call("rb_ray_points", ptr(o), ptr(d), ptr(zz), c_long(R), c_int(n), ptr(pts), ptr(dd), S())
        return pts, dd

    if n_importance > 0:
        per = n_importance // up_sample_steps
        u = torch.linspace(0.0 + 0.5 / per, 1.0 - 0.5 / per, steps=per).to(dev)
        sdf = net.eval_points(points(z)[0], full=False)[0].reshape(R, n_samples)
        for i in range(up_sample_steps):
            n = z.shape[1]
            wtmp = torch.empty(R, n, device=dev)
            zn = torch.empty(R, per, device=dev)
            call("rb_neus_upsample", ptr(o), ptr(d), ptr(z), ptr(sdf), c_long(R), c_int(n), c_int(per),
                 c_float(64 * 2 ** i), c_float(radius), ptr(u), ptr(wtmp), ptr(zn), S())
            last = i + 1 == up_sample_steps
            sn = None if last else net.eval_points(points(zn)[0], full=False)[0].reshape(R, per).contiguous()
            z2 = torch.empty(R, n + per, device=dev)
            s2 = None if last else torch.empty(R, n + per, device=dev)
            call("rb_neus_merge", ptr(z), ptr(sdf), c_int(n), ptr(zn), ptr(sn), c_int(per), c_long(R), ptr(z2), ptr(s2), S())
            z = z2
            sdf = s2 if not last else sdf
    n = z.shape[1]
    zmid = torch.empty(R, n, device=dev)
    call("rb_neus_mid_z", ptr(z), c_long(R), c_int(n), c_float(sample_dist), ptr(zmid), S())
    pts, dirs = points(zmid, want_dirs=True)
    pruned = not need_grad_error and not stage1_alpha
    if pruned:
        sdf_mid = net.eval_points(pts, full=False)[0]                    # [M]: the value the full pass would produce, bit for bit
        w0 = torch.empty(R, n, device=dev)
        keep = torch.empty(R * n, dtype=torch.uint8, device=dev)
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        call("rb_neus_weights", ptr(sdf_mid), ptr(pts), c_long(R), c_int(n), c_float(model.inv_s()), c_float(radius), ptr(w0),
             ptr(keep), ptr(cnt), S())
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
    else:
        out, grad = net.eval_points(pts, full=True, grad=True)           # [M,257], [M,3]
        col = model.color_network(pts, grad, dirs, out[:, 1:])
        sdf_stride = 257
    rgb, dist, acc = torch.empty(R, 3, device=dev), torch.empty(R, device=dev), torch.empty(R, device=dev)
    nrm, w = torch.empty(R, 3, device=dev), torch.empty(R, n, device=dev)
    gerr = torch.zeros(2, device=dev)
    z = z.contiguous()
    call("rb_neus_finish", ptr(out), c_long(sdf_stride), ptr(col), ptr(grad), ptr(pts), ptr(zmid), ptr(near), ptr(far), c_long(R),
         c_int(n), c_float(model.inv_s()), c_float(radius), c_int(1 if white_bkgd else 0),
         ptr(z if stage1_alpha else None), ptr(d if stage1_alpha else None), c_float(sample_dist),
         c_float(float(cos_anneal_ratio)), ptr(rgb), ptr(dist), ptr(acc), ptr(nrm), ptr(w), ptr(gerr), S())
    ge = gerr[0] / (gerr[1] + 1e-5)
    if pruned:
        ge = torch.full((), float("nan"), device=dev)        # not evaluated on the skipped samples
    if stage1_alpha:        # the stage-1 function's own result dict (neus/volume_render/sdf_render.py:358-365)
        return {"rgb": rgb, "dist": dist, "acc": acc, "sim_or_grad": ge, "weights": w, "means": zmid}
    return {"rgb": rgb, "dist": dist, "acc": acc, "grad_error": ge, "grad": nrm, "weights": w}


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
