"""NeuS volume rendering: ray-march with hierarchical sampling (model/sdf_render.py:37-374),
the 16-sample 'borrow_color' render used for secondary hits (model/neus_model.py:828-884) and the
Norm-stage 32-sample surface refinement (training/train_normal.py:239-286)."""
import torch

from . import nets


def alpha_from_sdf(sdf, inv_s):
    """alpha_i from consecutive-sample SDFs: prev = [s0..s_{n-2}, s_{n-1}], next = [s1..s_{n-1}, s_{n-1}]
    (sdf_render.py:209-218, neus_model.py:838-847).  sdf [R,n] -> unclipped alpha [R,n]."""
    nxt = torch.cat([sdf[:, 1:], sdf[:, -1:]], 1)
    prv = torch.cat([sdf[:, :-1], sdf[:, -1:]], 1)
    c0, c1 = torch.sigmoid(prv * inv_s), torch.sigmoid(nxt * inv_s)
    return (c0 - c1 + 1e-5) / (c0 + 1e-5)


def transmittance_weights(alpha, eps=1e-7):
    """w_i = alpha_i * prod_{j<i}(1 - alpha_j + eps)  (sdf_render.py:236-237)."""
    ones = torch.ones(alpha.shape[0], 1, dtype=alpha.dtype)
    return alpha * torch.cumprod(torch.cat([ones, 1.0 - alpha + eps], -1), -1)[:, :-1]


def neus_point_color(sd, x, dirs):
    """NeuSModel.forward (neus_model.py:749-752) on NeuS-space points: colour(x, grad(x), dirs, feat), sdf."""
    out = nets.sdf_raw(sd, x)
    g = nets.sdf_raw_gradient(sd, x)
    return nets.color_raw(sd, x, g, dirs, out[:, 1:]), out[:, :1]


def borrow_color(sd, points, view_dirs, batch=8192):
    """batch_borrow_color/borrow_color/volume_render (neus_model.py:828-884): 16 samples along
    -view/|view| over t in linspace(-0.01, 0.05, 16) in NeuS units, colour-weighted by NeuS alpha
    (no inside-sphere mask, no background)."""
    if points.shape[0] == 0:
        return torch.zeros_like(points)
    res = []
    t = torch.linspace(-0.01, 0.05, 16)[:, None]
    s = nets.inv_s(sd)
    for b in range(0, points.shape[0], batch):
        p, v = points[b:b + batch], view_dirs[b:b + batch]
        d = (-v / v.norm(dim=-1, keepdim=True))[:, None, :]
        x = p[:, None, :] * 2 + d * t                                     # [m,16,3]
        dd = d.expand(-1, 16, -1)
        col, sdf = neus_point_color(sd, x.reshape(-1, 3), dd.reshape(-1, 3))
        sdf = sdf.reshape(-1, 16)
        a = alpha_from_sdf(sdf, s).clip(0.0, 1.0)
        w = transmittance_weights(a)
        res.append((col.reshape(-1, 16, 3) * w[:, :, None]).sum(1))
    return torch.cat(res, 0)


# ----------------------------------------------------------------------------- hierarchical march
def inverse_cdf_samples(bins, weights, n_new):
    """sample_pdf(det=True) (sdf_render.py:37-67): bins [R,n], weights [R,n-1] -> [R,n_new]."""
    w = weights + 1e-5
    pdf = w / w.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], -1)       # [R,n]
    u = torch.linspace(0.5 / n_new, 1.0 - 0.5 / n_new, n_new).expand(cdf.shape[0], n_new).contiguous()
    idx = torch.searchsorted(cdf, u, right=True)
    lo = (idx - 1).clamp(min=0)
    hi = idx.clamp(max=cdf.shape[-1] - 1)
    c_lo, c_hi = cdf.gather(1, lo), cdf.gather(1, hi)
    b_lo, b_hi = bins.gather(1, lo), bins.gather(1, hi)
    den = c_hi - c_lo
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    return b_lo + (u - c_lo) / den * (b_hi - b_lo)


def importance_z(rays_o, rays_d, z, sdf, n_new, s, radius):
    """up_sample (sdf_render.py:70-114) with fixed inv_s = s."""
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z[..., None]
    r = torch.linalg.norm(pts, dim=-1)
    inside = (r[:, :-1] < radius) | (r[:, 1:] < radius)
    s0, s1, z0, z1 = sdf[:, :-1], sdf[:, 1:], z[:, :-1], z[:, 1:]
    mid = (s0 + s1) * 0.5
    cos = (s1 - s0) / (z1 - z0 + 1e-5)
    prev = torch.cat([torch.zeros(z.shape[0], 1), cos[:, :-1]], -1)
    cos = torch.minimum(prev, cos).clip(-1e3, 0.0) * inside
    dz = z1 - z0
    c0 = torch.sigmoid((mid - cos * dz * 0.5) * s)
    c1 = torch.sigmoid((mid + cos * dz * 0.5) * s)
    alpha = (c0 - c1 + 1e-5) / (c0 + 1e-5)
    w = transmittance_weights(alpha)
    return inverse_cdf_samples(z, w, n_new)


def render_neus(sd, rays_o, rays_d, near, far, n_samples=64, n_importance=64, up_sample_steps=4,
                white_bkgd=True, cos_anneal_ratio=None, t_rand=None):
    """render_neus with n_outside=0 (sdf_render.py:263-374) in NeuS space; is_eval=True / perturb=0 unless `t_rand` [R,1] is given:
    the uniform draw of perturb > 0 (:293-295), every coarse sample of a ray shifted by (t_rand - 0.5) * 2.0 / n_samples.
    rays_o/d [R,3]; near/far [R,1].  -> dict(rgb, dist, acc, grad, weights, grad_error, z_vals).
    cos_anneal_ratio not None: the stage-1 render_core (neus/volume_render/sdf_render.py:172-190), whose alpha comes from
    the SDF extrapolated half a section along the ray with the annealed cosine instead of from the neighbouring samples."""
    R = rays_o.shape[0]
    sample_dist = 2.0 / n_samples
    radius = 2.0                                                        # NeuSModel.radius() (neus_model.py:743)
    z = near + (far - near) * torch.linspace(0.0, 1.0, n_samples)[None, :]
    if t_rand is not None:
        z = z + (t_rand - 0.5) * 2.0 / n_samples
    sdf_only = lambda p: nets.sdf_raw(sd, p)[:, :1]
    if n_importance > 0:
        sdf = sdf_only((rays_o[:, None, :] + rays_d[:, None, :] * z[..., None]).reshape(-1, 3)).reshape(R, -1)
        per = n_importance // up_sample_steps
        for i in range(up_sample_steps):
            zn = importance_z(rays_o, rays_d, z, sdf, per, 64 * 2 ** i, radius)
            zcat, order = torch.sort(torch.cat([z, zn], -1), -1)
            if i + 1 < up_sample_steps:                                  # cat_z_vals (sdf_render.py:117-132)
                sn = sdf_only((rays_o[:, None, :] + rays_d[:, None, :] * zn[..., None]).reshape(-1, 3)).reshape(R, -1)
                sdf = torch.cat([sdf, sn], -1).gather(1, order)
            z = zcat
    n = z.shape[1]
    # render_core (sdf_render.py:175-260)
    dz = torch.cat([z[:, 1:] - z[:, :-1], torch.full((R, 1), sample_dist)], -1)
    zmid = z + dz * 0.5
    pts = (rays_o[:, None, :] + rays_d[:, None, :] * zmid[..., None]).reshape(-1, 3)
    dirs = rays_d[:, None, :].expand(R, n, 3).reshape(-1, 3)
    out = nets.sdf_raw(sd, pts)
    grads = nets.sdf_raw_gradient(sd, pts)
    col = nets.color_raw(sd, pts, grads, dirs, out[:, 1:]).reshape(R, n, 3)
    if cos_anneal_ratio is None:
        alpha = alpha_from_sdf(out[:, :1].reshape(R, n), nets.inv_s(sd)).clip(0.0, 1.0)
    else:
        true_cos = (dirs * grads).sum(-1, keepdim=True)
        iter_cos = -(torch.relu(-true_cos * 0.5 + 0.5) * (1.0 - cos_anneal_ratio) + torch.relu(-true_cos) * cos_anneal_ratio)
        s0 = out[:, :1]
        nxt, prv = s0 + iter_cos * dz.reshape(-1, 1) * 0.5, s0 - iter_cos * dz.reshape(-1, 1) * 0.5
        c0, c1 = torch.sigmoid(prv * nets.inv_s(sd)), torch.sigmoid(nxt * nets.inv_s(sd))
        alpha = ((c0 - c1 + 1e-5) / (c0 + 1e-5)).reshape(R, n).clip(0.0, 1.0)
    pnorm = torch.linalg.norm(pts, dim=-1).reshape(R, n)
    alpha = alpha * (pnorm < radius).float()
    w = transmittance_weights(alpha)
    acc = w.sum(-1)
    rgb = (col * w[:, :, None]).sum(1)
    if white_bkgd:
        rgb = rgb + 1.0 * (1.0 - acc[:, None])
    g3 = grads.reshape(R, n, 3)
    relax = (pnorm < radius * 1.2).float()
    gerr = (relax * (torch.linalg.norm(g3, dim=-1) - 1.0) ** 2).sum() / (relax.sum() + 1e-5)
    nrm = (w[..., None] * g3).sum(-2)
    nrm = nrm / (torch.linalg.norm(nrm, dim=-1, keepdim=True) + 0.0001)
    nrm[acc > 0.8] = 1.0                                                 # sic (sdf_render.py:361)
    dist = (w * zmid).sum(-1) / acc
    dist = torch.clip(torch.nan_to_num(dist, torch.inf), near.squeeze(-1), far.squeeze(-1))
    return {"rgb": rgb, "dist": dist, "acc": acc, "grad": nrm, "weights": w, "grad_error": gerr, "z_vals": z}


def neus_surface(sd, points, view_dirs, pred_normals, n_samp=32, dist=0.05):
    """NormalTrainRunner.get_neus_surface (train_normal.py:239-286): refine traced hit points with a
    32-sample NeuS render back along the ray; alpha clipped to [0.01,0.99]; residual to the input."""
    t = torch.linspace(0, dist, n_samp)[:, None]
    xs = (points[:, None, :] - t * view_dirs[:, None, :]).reshape(-1, 3)
    sdf = nets.implicit_forward(sd, xs)[:, :1].reshape(-1, n_samp)
    nrm = nets.implicit_gradient(sd, xs).reshape(-1, n_samp, 3)
    s = torch.exp(sd["implicit_network.neus_model.deviation_network.variance"] * 10.0)   # unclipped here
    a = alpha_from_sdf(sdf, s).clip(0.01, 0.99)
    w = transmittance_weights(a, eps=1e-10)[..., None]
    res = 1 - w.sum(-2)
    x = (xs.reshape(-1, n_samp, 3) * w).sum(-2) + res * points
    n = (nrm * w).sum(-2) + res * pred_normals
    relax = (torch.linalg.norm(xs, dim=-1).reshape(-1, n_samp) < 1.2).float()
    gerr = (relax * (torch.linalg.norm(nrm, dim=-1) - 1.0) ** 2).sum() / (relax.sum() + 1e-5)
    return x, n, gerr
