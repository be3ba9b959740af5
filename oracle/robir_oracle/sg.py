"""Spherical-Gaussian shading and visibility sampling.  model/sg_render.py:62-565."""
import math

import torch

TINY = 1e-6
# clamped-cosine SG fit used by the reference (sg_render.py:381-383)
MU_COS, LAMBDA_COS, ALPHA_COS = 32.7080, 0.0315, 31.7003


def unit_eps(v):
    """sg_render.py:107-108"""
    return v / (v.norm(dim=-1, keepdim=True) + TINY)


def hemisphere_int(lam, cos_beta):
    """Closed-form hemisphere integral of an SG (sg_render.py:62-81)."""
    lam = lam + TINY
    il = 1.0 / lam
    t = torch.sqrt(lam) * (1.6988 + 10.8438 * il) / (1.0 + 6.2201 * il + 10.2415 * il * il)
    ea = torch.exp(-t)
    pos = (cos_beta >= 0).float()
    eb = torch.exp(-t * cos_beta.clamp(min=0.0))
    s_up = (1.0 - ea * eb) / (1.0 - ea + eb - ea * eb)
    b = torch.exp(t * cos_beta.clamp(max=0.0))
    s_dn = (b - ea) / ((1.0 - ea) * (b + 1.0))
    s = pos * s_up + (1.0 - pos) * s_dn
    a_b = 2.0 * math.pi / lam * (torch.exp(-lam) - torch.exp(-2.0 * lam))
    a_u = 2.0 * math.pi / lam * (1.0 - torch.exp(-lam))
    return a_b * (1.0 - s) + a_u * s


def sg_product(lobe1, lam1, mu1, lobe2, lam2, mu2):
    """'lambda trick' approximate product of two SGs, lam1 << lam2 (sg_render.py:84-104)."""
    ratio = lam1 / lam2
    lobe1, lobe2 = unit_eps(lobe1), unit_eps(lobe2)
    dot = (lobe1 * lobe2).sum(-1, keepdim=True)
    tmp = torch.sqrt(ratio * ratio + 1.0 + 2.0 * ratio * dot)
    tmp = torch.min(tmp, ratio + 1.0)
    lam3 = lam2 * tmp
    lobe3 = (ratio / tmp) * lobe1 + (1.0 / tmp) * lobe2
    mu3 = mu1 * mu2 * torch.exp(lam2 * (tmp - ratio - 1.0))
    return lobe3, lam3, mu3


def _cone_dirs(axis, u_theta, u_phi, phi_range):
    """Directions in a cone around `axis` [..,1,3] from uniform draws (sg_render.py:123-146, 213-240)."""
    z = torch.zeros_like(axis)
    z[..., 2] = 1
    U = unit_eps(torch.cross(z, axis, dim=-1))
    V = unit_eps(torch.cross(axis, U, dim=-1))
    th = (u_theta * 2 * math.pi).unsqueeze(-1)
    ph = (u_phi * phi_range).unsqueeze(-1)
    return U * torch.cos(th) * torch.sin(ph) + V * torch.sin(th) * torch.sin(ph) + axis * torch.cos(ph)


def _run_vis(vis_fn, p, d, batch):
    out = torch.zeros(p.shape[0], 2)
    for s in range(0, p.shape[0], batch):
        out[s:s + batch] = vis_fn(p[s:s + batch], d[s:s + batch])
    return out


def diffuse_visibility(points, normals, vis_fn, lobes, lambdas, u_theta, u_phi, thr=1.0, argmax_vis=False,
                       return_dirs=False, batch=2000000, bounding=False):
    """get_diffuse_visibility (sg_render.py:111-195).  points/normals [n,3]; lobes [L,3]; lambdas [L,1];
    u_theta/u_phi [L,nsamp] uniform draws.  -> vis [L,n]; bounding=True (sg_render.py:185-186) -> per-sample vis [L,nsamp,n]."""
    L, nsamp = u_theta.shape
    n = points.shape[0]
    axis = unit_eps(lobes.unsqueeze(-2))                                  # [L,1,3]
    lam = lambdas.unsqueeze(-2)                                           # [L,1,1]
    sharp = lam[:, :, 0].clamp(min=1e-4)                                  # [L,1]
    rng = sharp.min().clamp(max=thr)
    phi_range = torch.arccos((-0.95 * rng) / sharp + 1)                   # [L,1]
    dirs = _cone_dirs(axis, u_theta, u_phi, phi_range)                    # [L,nsamp,3]
    flat = dirs.reshape(-1, 3)
    front = ((normals.unsqueeze(1) * flat.unsqueeze(0)).sum(-1)) > TINY
    pi_, di_ = front.nonzero(as_tuple=True)
    logits = _run_vis(vis_fn, points[pi_], flat[di_], batch)          # sg_render.py:158: batch_size = 2000000
    pv = logits.argmax(-1).float() if argmax_vis else torch.softmax(logits, -1)[..., 1]
    vis = torch.zeros(n, L * nsamp)
    vis[front] = pv
    vis = vis.reshape(n, L, nsamp).permute(1, 2, 0)                       # [L,nsamp,n]
    if bounding:
        return vis
    w = torch.exp(lam * ((dirs * axis).sum(-1, keepdim=True) - 1.0))      # [L,nsamp,1]
    out = (vis * w).sum(1) / (w.sum(1) + TINY)                            # [L,n]
    if return_dirs:
        return out, dirs, int(front.sum())
    return out


def specular_visibility(points, normals, viewdirs, vis_fn, lobes, lambdas, u_theta, u_phi,
                        testing=False, inv=False, argmax_vis=False):
    """get_specular_visibility, single-view branch (sg_render.py:198-301).
    lobes [n,3] / lambdas [n,1] = warped BRDF lobe of each point (passed but only lambdas + reflection are
    used for sampling); u_theta/u_phi [n,nsamp].  -> vis [n]."""
    n, nsamp = u_theta.shape
    light_dirs = lobes.unsqueeze(-2)
    ndv = (normals * viewdirs).sum(-1, keepdim=True).clamp(min=0.0)
    refl = (-viewdirs + 2 * ndv * normals).unsqueeze(-2)                  # [n,1,3]
    sharp = lambdas.unsqueeze(-2)[..., 0].clip(min=0.1, max=50)           # [n,1]
    rng = sharp.min().clamp(max=1)                                        # batch-global (sg_render.py:222)
    phi_range = torch.arccos((-0.95 * rng) / sharp + 1)
    dirs = _cone_dirs(refl, u_theta, u_phi, phi_range)                    # [n,nsamp,3]
    front = (normals.unsqueeze(1) * dirs).sum(-1) > TINY                  # [n,nsamp]
    pi_, si_ = front.nonzero(as_tuple=True)
    logits = _run_vis(vis_fn, points[pi_], dirs[pi_, si_], 10000000)
    if argmax_vis:
        pv = (logits.argmin(-1) if inv else logits.argmax(-1)).float()
    else:
        pv = torch.softmax(logits, -1)[..., 0 if inv else 1]
    vis = torch.zeros(n, nsamp)
    vis[front] = pv
    w = torch.exp(sharp * ((dirs * light_dirs).sum(-1) - 1.0))            # [n,nsamp]
    if testing:
        bad_row = torch.isinf(w.sum(-1))
        if bad_row.any():
            sub = w[bad_row]
            w[bad_row] = torch.isinf(sub).float()
    return (vis * w).sum(-1) / (w.sum(-1) + TINY)


def kl_divergence(x, mu=0.05):
    """utils/utils.py:14-17."""
    rho_hat = torch.mean(x, 0)
    rho = torch.full_like(rho_hat, mu)
    return torch.mean(rho * torch.log(rho / (rho_hat + 1e-4)) + (1 - rho) * torch.log((1 - rho) / (1 - rho_hat + 1e-4)))


def render_with_sg(points, normal, viewdirs, lgt_sgs, specular_reflectance, roughness, diffuse_albedo, draws,
                   comp_vis=True, vis_fn=None, lin_diff=False, testing=False, indir_integral=None, metallic=None,
                   argmax_vis=False, stats=None, diffuse_vis=None, prefit=False, fun_spec=False):
    """render_with_sg, single-view (sg_render.py:343-565); fun_spec=True returns the specular term as a function of
    (roughness, draws) and sg_rgb = the diffuse term (sg_render.py:413,544-551).
    lgt_sgs [n,M,7].  draws: dict with 'dvis_theta','dvis_phi' [M,32] (comp_vis only) and
    'svis_theta','svis_phi' [n,8]."""
    if viewdirs.dim() == 3:
        return _render_with_sg_multi_view(points, normal, viewdirs, lgt_sgs, specular_reflectance, roughness, diffuse_albedo, draws,
                                          comp_vis, vis_fn, lin_diff, testing, indir_integral, metallic, argmax_vis, stats, diffuse_vis,
                                          prefit, fun_spec)
    n, M = lgt_sgs.shape[0], lgt_sgs.shape[1]
    supervise = torch.tensor(0.0)
    l_lobe = lgt_sgs[..., :3] / (lgt_sgs[..., :3].norm(dim=-1, keepdim=True) + TINY)
    l_lam = lgt_sgs[..., 3:4].abs()
    l_mu0 = lgt_sgs[..., -3:].abs()
    nrm = normal.unsqueeze(-2).expand(n, M, 3)
    view = viewdirs.unsqueeze(-2).expand(n, M, 3)
    f0 = specular_reflectance.unsqueeze(1).expand(n, M, 3)

    vis_shadow = torch.zeros(n, 3)
    if comp_vis:
        # first row's light is used for every point (sg_render.py:388-390)
        lv = diffuse_visibility(points, normal, vis_fn, l_lobe[0], l_lam[0], draws["dvis_theta"], draws["dvis_phi"],
                                argmax_vis=argmax_vis, return_dirs=stats is not None)
        if stats is not None:
            lv, _, cnt = lv
            stats["diffuse_vis_evals"] = stats.get("diffuse_vis_evals", 0) + cnt
        light_vis = lv.permute(1, 0).unsqueeze(-1).expand(n, M, 3)
        if diffuse_vis is not None:                                          # CESR (sg_render.py:393-403)
            pred = diffuse_vis.reshape(-1, M, 1).expand(n, M, 3)
            factor = {"warmup": 0.1, "project": 0.2}.get(prefit, 1.0)
            supervise = kl_divergence((light_vis - pred).abs()[..., 0], 0.01) * factor
            if prefit != "warmup":
                light_vis = pred
        vis_shadow = (light_vis * l_mu0).sum(1) / torch.clamp(l_mu0.sum(1), 1e-4)

    # ---------------- specular (sg_render.py:414-500)
    def specular_rgb_fn(roughness, draws=draws):
        return _specular(points, nrm, view, f0, roughness, diffuse_albedo, metallic, l_lobe, l_lam, l_mu0, vis_fn, draws, testing,
                         comp_vis, argmax_vis)

    spec = None if fun_spec else specular_rgb_fn(roughness)

    # ---------------- diffuse (sg_render.py:505-536)
    l_mu_d = l_mu0 * light_vis if comp_vis else l_mu0
    dmu = l_mu_d if lin_diff else l_mu_d * (diffuse_albedo / math.pi).unsqueeze(-2)
    p_lobe, p_lam, p_mu = sg_product(nrm, LAMBDA_COS, MU_COS, l_lobe, l_lam, dmu)
    diff = p_mu * hemisphere_int(p_lam, (p_lobe * nrm).sum(-1, keepdim=True)) \
        - dmu * ALPHA_COS * hemisphere_int(l_lam, (l_lobe * nrm).sum(-1, keepdim=True))
    diff = diff.sum(-2).clamp(min=0.0)
    if indir_integral is not None:
        diff = indir_integral if lin_diff else indir_integral * (diffuse_albedo / math.pi)
    if fun_spec:
        return {"sg_rgb": diff, "sg_specular_rgb": specular_rgb_fn, "sg_diffuse_rgb": diff, "vis_shadow": vis_shadow,
                "supervise": supervise}
    return {"sg_rgb": spec + diff, "sg_specular_rgb": spec, "sg_diffuse_rgb": diff, "vis_shadow": vis_shadow,
            "supervise": supervise}


def _warped_lobe(nrm, view, roughness):
    """The BRDF lobe of `roughness` around the normal, warped to the view (sg_render.py:416-428) -> lobe [n,M,3], lambda [n,M,1]."""
    n, M = nrm.shape[0], nrm.shape[1]
    r4 = 2.0 / (roughness * roughness * roughness * roughness)
    vdl = (nrm * view).sum(-1, keepdim=True).clamp(min=0.0)
    w_lobe = 2 * vdl * nrm - view
    return w_lobe / (w_lobe.norm(dim=-1, keepdim=True) + TINY), r4.unsqueeze(1).expand(n, M, 1) / (4 * vdl + TINY)


def _render_with_sg_multi_view(points, normal, viewdirs, lgt_sgs, specular_reflectance, roughness, diffuse_albedo, draws, comp_vis, vis_fn,
                               lin_diff, testing, indir_integral, metallic, argmax_vis, stats, diffuse_vis, prefit, fun_spec):
    """MULTI_VIEW form of render_with_sg: viewdirs [V,n,3] (sg_render.py:356, 375-378, 465-470; get_specular_visibility's multi_view
    branches, :227-231 and :247-258).  Light visibility, diffuse term, shadow and supervision do not see the view: computed once, [n,.].
    The specular term is the single-view formula of every view, except for its sampled visibility: ONE set of draws [n,nsamp] serves all
    views, the cone range comes from the smallest sharpness over ALL views and points (:222), and the branch always takes the arg-max of
    the two logits -- never inverted, never the soft-max, whatever inv / argmax_vis say (:257).  -> sg_specular_rgb, sg_rgb [V,n,3].
    In the reference's broadcast form the rows of view v are rows v n .. v n + n - 1 of one batch: restated that way."""
    V, n, M = viewdirs.shape[0], lgt_sgs.shape[0], lgt_sgs.shape[1]
    base = render_with_sg(points, normal, viewdirs[0], lgt_sgs, specular_reflectance, roughness, diffuse_albedo, draws, comp_vis=comp_vis,
                          vis_fn=vis_fn, lin_diff=lin_diff, testing=testing, indir_integral=indir_integral, metallic=metallic,
                          argmax_vis=argmax_vis, stats=stats, diffuse_vis=diffuse_vis, prefit=prefit, fun_spec=True)
    l_lobe = lgt_sgs[..., :3] / (lgt_sgs[..., :3].norm(dim=-1, keepdim=True) + TINY)
    l_lam, l_mu0 = lgt_sgs[..., 3:4].abs(), lgt_sgs[..., -3:].abs()
    nrm = normal.unsqueeze(-2).expand(n, M, 3)
    f0 = specular_reflectance.unsqueeze(1).expand(n, M, 3)

    def specular_rgb_fn(roughness, draws=draws):
        views = [viewdirs[v].unsqueeze(-2).expand(n, M, 3) for v in range(V)]
        warped = [_warped_lobe(nrm, vw, roughness) for vw in views]
        bvis = specular_visibility(points.repeat(V, 1), normal.repeat(V, 1), viewdirs.reshape(V * n, 3), vis_fn,
                                   torch.cat([w[0][:, 0] for w in warped]), torch.cat([w[1][:, 0] for w in warped]),
                                   draws["svis_theta"].repeat(V, 1), draws["svis_phi"].repeat(V, 1), testing=testing, inv=False,
                                   argmax_vis=True).reshape(V, n)
        return torch.stack([_specular(points, nrm, views[v], f0, roughness, diffuse_albedo, metallic, l_lobe, l_lam, l_mu0, vis_fn, draws,
                                      testing, comp_vis, argmax_vis, bvis=bvis[v]) for v in range(V)])

    if fun_spec:
        return dict(base, sg_specular_rgb=specular_rgb_fn)
    spec = specular_rgb_fn(roughness)
    return dict(base, sg_rgb=spec + base["sg_diffuse_rgb"], sg_specular_rgb=spec)


def _specular(points, nrm, view, f0, roughness, diffuse_albedo, metallic, l_lobe, l_lam, l_mu0, vis_fn, draws, testing, comp_vis,
              argmax_vis, bvis=None):
    """specular_rgb_fn (sg_render.py:413-500): warped BRDF lobe of `roughness`, its sampled visibility, product with the light SGs and
    the clamped cosine, hemisphere integral."""
    n, M = nrm.shape[0], nrm.shape[1]
    r4 = 2.0 / (roughness * roughness * roughness * roughness)             # [n,1]
    b_lam = r4.unsqueeze(1).expand(n, M, 1)
    b_mu = (r4 / math.pi).expand(n, 3).unsqueeze(1).expand(n, M, 3)
    vdl = (nrm * view).sum(-1, keepdim=True).clamp(min=0.0)
    w_lobe = 2 * vdl * nrm - view
    w_lobe = w_lobe / (w_lobe.norm(dim=-1, keepdim=True) + TINY)
    w_lam = b_lam / (4 * vdl + TINY)
    half = w_lobe + view
    half = half / (half.norm(dim=-1, keepdim=True) + TINY)
    vdh = (view * half).sum(-1, keepdim=True).clamp(min=0.0)
    fres_w = torch.pow(2.0, -(5.55473 * vdh + 6.8316) * vdh)
    if metallic is None:
        Fr = f0 + (1.0 - f0) * fres_w
    else:
        sc = (1.0 - metallic[:, None, :]) * f0 + diffuse_albedo[:, None, :] * metallic[:, None, :]
        Fr = sc + (1.0 - sc) * fres_w
    d1 = (w_lobe * nrm).sum(-1, keepdim=True).clamp(min=0.0)
    d2 = (view * nrm).sum(-1, keepdim=True).clamp(min=0.0)
    k = ((roughness + 1.0) * (roughness + 1.0) / 8.0).unsqueeze(1).expand(n, M, 1)
    G = (d1 / (d1 * (1 - k) + k + TINY)) * (d2 / (d2 * (1 - k) + k + TINY))
    w_mu = b_mu * (Fr * G / (4 * d1 * d2 + TINY))
    if bvis is None:
        bvis = specular_visibility(points, nrm[:, 0, :], view[:, 0, :], vis_fn, w_lobe[:, 0], w_lam[:, 0],
                                   draws["svis_theta"], draws["svis_phi"], testing=testing, inv=not comp_vis,
                                   argmax_vis=argmax_vis)
    l_mu_s = l_mu0 * bvis[:, None, None]
    f_lobe, f_lam, f_mu = sg_product(l_lobe, l_lam, l_mu_s, w_lobe, w_lam, w_mu)
    p_lobe, p_lam, p_mu = sg_product(nrm, LAMBDA_COS, MU_COS, f_lobe, f_lam, f_mu)
    spec = p_mu * hemisphere_int(p_lam, (p_lobe * nrm).sum(-1, keepdim=True)) \
        - f_mu * ALPHA_COS * hemisphere_int(f_lam, (f_lobe * nrm).sum(-1, keepdim=True))
    return spec.sum(-2).clamp(min=0.0)


def render_with_all_sg(points, normal, viewdirs, lgt_sgs, specular_reflectance, roughness, diffuse_albedo, draws,
                       indir_integral=None, indir_lgt_sgs=None, vis_fn=None, lin_diff=False, testing=False,
                       metallic=None, argmax_vis=False, stats=None, diffuse_vis=None, prefit=False):
    """render_with_all_sg (sg_render.py:304-337): direct pass (128 lobes, visibility) + indirect pass
    (per-point 24 lobes, no light visibility, inverted specular visibility, diffuse := integral)."""
    n = normal.shape[0]
    if lgt_sgs.dim() == 2:
        lgt_sgs = lgt_sgs.unsqueeze(0).expand(n, lgt_sgs.shape[0], 7)
    d_direct = {"dvis_theta": draws["dvis_theta"], "dvis_phi": draws["dvis_phi"],
                "svis_theta": draws["svis_theta_dir"], "svis_phi": draws["svis_phi_dir"]}
    ret = render_with_sg(points, normal, viewdirs, lgt_sgs, specular_reflectance, roughness, diffuse_albedo,
                         d_direct, comp_vis=True, vis_fn=vis_fn, lin_diff=lin_diff, testing=testing,
                         metallic=metallic, argmax_vis=argmax_vis, stats=stats, diffuse_vis=diffuse_vis, prefit=prefit)
    z = torch.zeros_like(points)
    ind = {"sg_rgb": z, "sg_diffuse_rgb": z, "sg_specular_rgb": z}
    if indir_lgt_sgs is not None:
        d_ind = {"svis_theta": draws["svis_theta_ind"], "svis_phi": draws["svis_phi_ind"]}
        ind = render_with_sg(points, normal, viewdirs, indir_lgt_sgs, specular_reflectance, roughness,
                             diffuse_albedo, d_ind, comp_vis=False, vis_fn=vis_fn, lin_diff=lin_diff,
                             testing=testing, indir_integral=indir_integral, metallic=metallic,
                             argmax_vis=argmax_vis)
    ret.update({"indir_rgb": ind["sg_rgb"], "indir_diffuse_rgb": ind["sg_diffuse_rgb"],
                "indir_specular_rgb": ind["sg_specular_rgb"]})
    return ret


def envmap_sg(lgt_sgs, dirs):
    """render_envmap_sg (sg_render.py:26-42): sum_k mu_k exp(lambda_k (d.lobe_k - 1)); no epsilon in the norm."""
    lobe = lgt_sgs[:, :3] / lgt_sgs[:, :3].norm(dim=-1, keepdim=True)
    lam, mu = lgt_sgs[:, 3:4].abs(), lgt_sgs[:, -3:].abs()
    d = dirs.unsqueeze(-2)
    return (mu * torch.exp(lam * ((d * lobe).sum(-1, keepdim=True) - 1.0))).sum(-2)


def envmap_grid(lgt_sgs, H, W, upper_hemi=False):
    """compute_envmap (sg_render.py:9-23)."""
    phi = torch.linspace(0.0, math.pi / 2 if upper_hemi else math.pi, H)
    theta = torch.linspace(math.pi, -math.pi, W)
    phi, theta = torch.meshgrid(phi, theta, indexing="ij")
    d = torch.stack([torch.cos(theta) * torch.sin(phi), torch.sin(theta) * torch.sin(phi), torch.cos(phi)], -1)
    return envmap_sg(lgt_sgs, d)


def envmap_lookup(envmap, dirs):
    """render_envmap (sg_render.py:45-59): bilinear lat-long lookup of envmap [H,W,3] along dirs [n,3] -> [n,3].  The reference goes through
    F.grid_sample(align_corners=True, zero padding); restated here as the explicit four-texel blend: query_x = -atan2(y, x) / pi,
    query_y = (arccos(z) - 1e-6) / pi * 2 - 1 map [-1, 1] onto the texel CENTRES 0 .. W-1 / 0 .. H-1, texels outside the map count as 0."""
    H, W = envmap.shape[:2]
    phi = torch.arccos(dirs[:, 2]) - TINY
    theta = torch.atan2(dirs[:, 1], dirs[:, 0])
    qx, qy = -theta / math.pi, (phi / math.pi) * 2 - 1
    x, y = (qx + 1) / 2 * (W - 1), (qy + 1) / 2 * (H - 1)
    x0, y0 = torch.floor(x), torch.floor(y)
    fx, fy = x - x0, y - y0
    out = torch.zeros(dirs.shape[0], envmap.shape[2], dtype=envmap.dtype)
    for dy, wy in ((0, 1 - fy), (1, fy)):
        for dx, wx in ((0, 1 - fx), (1, fx)):
            xi, yi = (x0 + dx).long(), (y0 + dy).long()
            ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
            tex = envmap[yi.clamp(0, H - 1), xi.clamp(0, W - 1)]
            out = out + torch.where(ok[:, None], tex, torch.zeros_like(tex)) * (wx * wy)[:, None]
    return out
