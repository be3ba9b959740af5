"""IDR sphere tracer with sampler + secant root finding -- the `use_octree=False` ray tracer
(model/ray_tracing.py:26-326; utils/rend_util.py:141-163).  Every ray is independent; the SDF network is called inside the
loops.  training=True adds the module's training-mode behaviour: the secant runs only on rays the object mask agrees with (:256) and
the rays without a surface get the tail of :73-100 (projection of the camera onto rays that miss the bounding sphere, minimal_sdf_points
with its n_steps uniform draws, :299-326)."""
import torch


def sphere_intersection(cam_loc, dirs, r=1.0):
    """get_sphere_intersection (rend_util.py:141-163): cam_loc [3], dirs [N,3] -> t [N,2] (clamped >= 0.01), mask [N]."""
    dot = dirs @ cam_loc
    under = dot ** 2 - (cam_loc.norm() ** 2 - r ** 2)
    mask = under > 0
    t = torch.zeros(dirs.shape[0], 2)
    root = torch.sqrt(under[mask])
    t[mask] = torch.stack([-root, root], -1) - dot[mask][:, None]
    return t.clamp_min(0.01), mask


def trace(sdf_fn, cam_loc, dirs, object_mask, r=1.0, thr=5.0e-5, line_step=0.5, line_iters=3, trace_iters=10,
          n_steps=100, n_secant=32, training=False, steps_u=None):
    """RayTracing.forward.  cam_loc [3]; dirs [N,3]; object_mask [N] bool.  -> points [N,3], hit [N], dist [N].
    training=True: the module in training mode; steps_u [n_steps] = the uniform draws of minimal_sdf_points (:305)."""
    N = dirs.shape[0]
    t01, inter = sphere_intersection(cam_loc, dirs, r)
    at = lambda t: cam_loc[None, :] + t[:, None] * dirs
    un_s, un_e = inter.clone(), inter.clone()
    acc_s = torch.where(inter, t01[:, 0], torch.zeros(N))
    acc_e = torch.where(inter, t01[:, 1], torch.zeros(N))
    min_dis, max_dis = acc_s.clone(), acc_e.clone()                     # :124-126
    p_s = torch.where(inter[:, None], at(t01[:, 0]), torch.zeros(N, 3))
    p_e = torch.where(inter[:, None], at(t01[:, 1]), torch.zeros(N, 3))

    def masked_sdf(p, m):
        out = torch.zeros(N)
        if m.any():
            out[m] = sdf_fn(p[m])
        return out

    nxt_s, nxt_e = masked_sdf(p_s, un_s), masked_sdf(p_e, un_e)
    it = 0
    while True:
        cur_s = torch.where(un_s, nxt_s, torch.zeros(N))
        cur_s[cur_s <= thr] = 0
        cur_e = torch.where(un_e, nxt_e, torch.zeros(N))
        cur_e[cur_e <= thr] = 0
        un_s, un_e = un_s & (cur_s > thr), un_e & (cur_e > thr)
        if (not un_s.any() and not un_e.any()) or it == trace_iters:
            break
        it += 1
        acc_s, acc_e = acc_s + cur_s, acc_e - cur_e
        p_s, p_e = at(acc_s), at(acc_e)
        nxt_s, nxt_e = masked_sdf(p_s, un_s), masked_sdf(p_e, un_e)
        bad_s, bad_e = nxt_s < 0, nxt_e < 0
        k = 0
        while (bad_s.any() or bad_e.any()) and k < line_iters:      # back-step rays that crossed the surface
            f = (1 - line_step) / (2 ** k)
            acc_s = torch.where(bad_s, acc_s - f * cur_s, acc_s)
            acc_e = torch.where(bad_e, acc_e + f * cur_e, acc_e)
            p_s = torch.where(bad_s[:, None], at(acc_s), p_s)
            p_e = torch.where(bad_e[:, None], at(acc_e), p_e)
            if bad_s.any():
                nxt_s[bad_s] = sdf_fn(p_s[bad_s])
            if bad_e.any():
                nxt_e[bad_e] = sdf_fn(p_e[bad_e])
            bad_s, bad_e = nxt_s < 0, nxt_e < 0
            k += 1
        un_s, un_e = un_s & (acc_s < acc_e), un_e & (acc_s < acc_e)
    hit = acc_s < acc_e
    pts, dist = p_s.clone(), acc_s.clone()
    idx = un_s.nonzero()[:, 0]                                           # rays the sphere tracing did not converge on
    if idx.numel() > 0:
        lo, hi = acc_s[idx], acc_e[idx]
        z = lo[:, None] + torch.linspace(0, 1, n_steps)[None, :] * (hi - lo)[:, None]        # [m, n_steps]
        P = cam_loc[None, None, :] + z[..., None] * dirs[idx][:, None, :]
        s = sdf_fn(P.reshape(-1, 3)).reshape(-1, n_steps)
        first = torch.argmin(torch.sign(s) * torch.arange(n_steps, 0, -1).float()[None, :], -1)   # first negative sample
        rows = torch.arange(idx.numel())
        sp, sd = P[rows, first], z[rows, first]
        neg = s[rows, first] < 0
        out = ~(object_mask[idx] & neg)
        if out.any():                                                    # minimal-SDF point for rays that found no surface
            j = torch.argmin(s[out], -1)
            sp[out], sd[out] = P[out][torch.arange(int(out.sum())), j], z[out][torch.arange(int(out.sum())), j]
        hit_s = neg.clone()
        if training:                                                     # :256: only rays the object mask agrees with are refined
            neg = neg & object_mask[idx]
        if neg.any():                                                    # secant refinement between the bracketing samples
            r_ = rows[neg]
            z_hi, s_hi = z[r_, first[neg]].clone(), s[r_, first[neg]].clone()
            z_lo, s_lo = z[r_, first[neg] - 1].clone(), s[r_, first[neg] - 1].clone()
            d_ = dirs[idx][neg]
            zp = (-s_lo * (z_hi - z_lo) / (s_hi - s_lo + 1e-8) + z_lo).clamp(0.0, 2e1)
            for _ in range(n_secant):
                sm = sdf_fn(cam_loc[None, :] + zp[:, None] * d_)
                up, dn = sm > 0, sm < 0
                z_lo, s_lo = torch.where(up, zp, z_lo), torch.where(up, sm, s_lo)
                z_hi, s_hi = torch.where(dn, zp, z_hi), torch.where(dn, sm, s_hi)
                zp = (-s_lo * (z_hi - z_lo) / (s_hi - s_lo + 1e-8) + z_lo).clamp(0.0, 2e1)
            sp[neg], sd[neg] = cam_loc[None, :] + zp[:, None] * d_, zp
        pts[idx], dist[idx], hit[idx] = sp, sd, hit_s
    if not training:
        return pts, hit, dist
    # ---- training-mode tail (:73-100)
    in_mask = ~hit & object_mask & ~un_s
    out_mask = ~object_mask & ~un_s
    left = (in_mask | out_mask) & ~inter                                  # rays that miss the bounding sphere: foot of the camera on the ray
    if left.any():
        dist[left] = -(dirs[left] * cam_loc[None, :]).sum(-1)
        pts[left] = cam_loc[None, :] + dist[left][:, None] * dirs[left]
    m = (in_mask | out_mask) & inter
    if m.any():
        sel = hit & out_mask
        min_dis[sel] = dist[sel]
        lo, hi = min_dis[m][:, None], max_dis[m][:, None]
        zt = steps_u[None, :].repeat(int(m.sum()), 1) * (hi - lo) + lo                    # minimal_sdf_points (:299-326)
        Pa = cam_loc[None, None, :].repeat(int(m.sum()), n_steps, 1) + zt[..., None] * dirs[m][:, None, :].repeat(1, n_steps, 1)
        sv = sdf_fn(Pa.reshape(-1, 3)).reshape(-1, n_steps)
        j = sv.argmin(-1)
        rr = torch.arange(int(m.sum()))
        pts[m], dist[m] = Pa[rr, j], zt[rr, j]
    return pts, hit, dist
