"""IDR sphere tracer -- drop-in for model/ray_tracing.py `RayTracing` (the `use_octree=False` ray tracer).

Same constructor arguments, `forward(sdf, cam_loc, object_mask, ray_directions)` and the four public stages with the reference's
signatures (model/ray_tracing.py:6-326): `sphere_tracing`, `ray_sampler`, `secant`, `minimal_sdf_points`; forward is written in terms
of them.  `sdf` is any callable [M,3] -> [M] on device tensors (IDRNetwork passes the MFMA SDF kernel); every per-ray update in
between runs in the HIP kernels of csrc/raytrace.hip.  Rays are independent, so instead of the reference's boolean-mask gathers the SDF
is evaluated on all 2N start/end points each step and masked rows are ignored -- same per-ray arithmetic, no host synchronisation
inside the sphere-tracing loop.

With the module in training mode the call behaves like the reference's (:68-100, 256, 299-326; still without gradients, as under the
reference's no_grad): the secant runs only where the object mask agrees, rays without a surface get the foot of the camera centre
(rays that miss the bounding sphere) or the minimal-SDF point among n_steps uniform depths (`min_sdf_steps` pins the draws).
"""
import torch
import torch.nn as nn

from . import ops


class RayTracing(nn.Module):
    def __init__(self, object_bounding_sphere=1.0, sdf_threshold=5.0e-5, line_search_step=0.5, line_step_iters=1,
                 sphere_tracing_iters=10, n_steps=100, n_rootfind_steps=8):
        super().__init__()
        self.object_bounding_sphere = object_bounding_sphere
        self.sdf_threshold = sdf_threshold
        self.sphere_tracing_iters = sphere_tracing_iters
        self.line_step_iters = line_step_iters
        self.line_search_step = line_search_step
        self.n_steps = n_steps
        self.n_secant_steps = n_rootfind_steps
        self._bound = None
        self.min_sdf_steps = None      # tests: the n_steps uniform draws of minimal_sdf_points (ray_tracing.py:305) instead of fresh ones

    def bind(self, implicit_network):
        """Default SDF when forward() is called with sdf=None (mirrors OctreeTracing.bind)."""
        object.__setattr__(self, "_bound", implicit_network)       # not a submodule: no duplicate state-dict keys

    def generate(self, *a, **k):          # OctreeTracing API no-op so callers can treat both tracers alike
        return None

    @staticmethod
    def _rays(cam_loc, ray_directions):
        batch, npix, _ = ray_directions.shape
        dirs = ray_directions.reshape(-1, 3).float().contiguous()
        cam = cam_loc.reshape(-1, 3).float().contiguous()
        if batch > 1:
            assert npix == 1 and cam.shape[0] == batch, "per-ray origins come as [N,3] origins with [N,1,3] directions"
        return cam, dirs

    @torch.no_grad()
    def forward(self, sdf, cam_loc, object_mask, ray_directions):
        training = self.training       # the module's training-mode behaviour (ray_tracing.py:68-100, 256): still no gradients, like the reference's call under no_grad
        if sdf is None:
            sdf = self._bound.sdf_only
        batch, npix, _ = ray_directions.shape
        cam, dirs = self._rays(cam_loc, ray_directions)
        N = dirs.shape[0]
        dev = dirs.device
        if N == 0:
            return (torch.zeros(0, 3, device=dev), torch.zeros(0, dtype=torch.bool, device=dev), torch.zeros(0, device=dev))
        # get_sphere_intersection (utils/rend_util.py:141-163) is evaluated by the state's init kernel: mask_intersect / sphere_intersections = None
        points, sampler_mask, dist, acc_e, min_dis, max_dis, inter = self._sphere_tracing(sdf, cam, dirs, None, None)
        hit = dist < acc_e
        if bool(sampler_mask.any()):                        # not converged: sampler + secant (ray_tracing.py:208-297)
            mm = torch.zeros(N, 2, device=dev)
            mm[:, 0], mm[:, 1] = dist, acc_e
            sp, s_hit, sd = self.ray_sampler(sdf, cam_loc, object_mask, ray_directions, mm.reshape(batch, npix, 2), sampler_mask)
            points[sampler_mask], dist[sampler_mask], hit[sampler_mask] = sp[sampler_mask], sd[sampler_mask], s_hit[sampler_mask]
        if not training:
            return points, hit, dist
        # ---- training-mode tail (ray_tracing.py:73-100): rays without a surface get a point for the mask loss
        obj = object_mask.reshape(-1).bool()
        in_mask, out_mask = ~hit & obj & ~sampler_mask, ~obj & ~sampler_mask
        cam_all = cam.expand(N, 3)
        left = (in_mask | out_mask) & ~inter                # miss the bounding sphere: the foot of the camera centre on the ray
        if bool(left.any()):
            dist[left] = -(dirs[left] * cam_all[left]).sum(-1)
            points[left] = cam_all[left] + dist[left].unsqueeze(1) * dirs[left]
        m = (in_mask | out_mask) & inter
        if bool(m.any()):                                   # minimal_sdf_points (:299-326): n_steps uniform depths between entry and exit
            sel = hit & out_mask
            min_dis[sel] = dist[sel]
            points[m], dist[m] = self.minimal_sdf_points(npix, sdf, cam_loc, dirs, m, min_dis, max_dis)
        return points, hit, dist

    def _sphere_tracing(self, sdf, cam, dirs, mask_intersect, sphere_intersections):
        """Both-sided sphere tracing on the device state (csrc/raytrace.hip ops 0-6)."""
        N = dirs.shape[0]
        st = ops.RayTraceState(cam, dirs)
        if mask_intersect is None:
            st.step(0, float(self.object_bounding_sphere) ** 2)
        else:       # a caller's own bounding-sphere intersections (the public sphere_tracing signature, ray_tracing.py:105-126)
            mi = mask_intersect.reshape(-1).bool()
            si = sphere_intersections.reshape(-1, 2).float()
            z = torch.zeros(N, device=dirs.device)
            st.f[0], st.f[1] = torch.where(mi, si[:, 0], z), torch.where(mi, si[:, 1], z)
            st.b[0], st.b[1] = mi.to(torch.uint8), mi.to(torch.uint8)
            cam_all = cam.expand(N, 3)
            st.pts[:N] = torch.where(mi[:, None], cam_all + st.f[0][:, None] * dirs, torch.zeros_like(dirs))
            st.pts[N:] = torch.where(mi[:, None], cam_all + st.f[1][:, None] * dirs, torch.zeros_like(dirs))
        min_dis, max_dis, inter = st.f[0].clone(), st.f[1].clone(), st.b[0].bool()    # entry / exit distances before the loop moves them
        st.step(1, sdf2=sdf(st.pts))
        for it in range(self.sphere_tracing_iters + 1):
            st.step(3, self.sdf_threshold)
            if it == self.sphere_tracing_iters:
                break
            st.step(4)
            st.step(1, sdf2=sdf(st.pts))
            for k in range(self.line_step_iters):
                st.step(5, (1 - self.line_search_step) / (2 ** k))
                st.step(2, sdf2=sdf(st.pts))
            st.step(6)
        return st.pts[:N].clone(), st.b[0].bool(), st.f[0].clone(), st.f[1].clone(), min_dis, max_dis, inter

    @torch.no_grad()
    def sphere_tracing(self, batch_size, num_pixels, sdf, cam_loc, ray_directions, mask_intersect, sphere_intersections):
        """model/ray_tracing.py:102-206 -> (curr_start_points [N,3], unfinished_mask_start [N], acc_start_dis [N], acc_end_dis [N],
        min_dis [N], max_dis [N])."""
        cam, dirs = self._rays(cam_loc, ray_directions)
        return self._sphere_tracing(sdf, cam, dirs, mask_intersect, sphere_intersections)[:6]

    @torch.no_grad()
    def ray_sampler(self, sdf, cam_loc, object_mask, ray_directions, sampler_min_max, sampler_mask):
        """model/ray_tracing.py:208-274: n_steps samples between the two distances of every ray in sampler_mask, the first sign change,
        then the secant -> (sampler_pts [N,3], sampler_net_obj_mask [N], sampler_dists [N]); rows outside sampler_mask are zero / False."""
        cam, dirs = self._rays(cam_loc, ray_directions)
        N, dev = dirs.shape[0], dirs.device
        sampler_mask = sampler_mask.reshape(-1).bool()
        obj = object_mask.reshape(-1).bool()
        pts_out = torch.zeros(N, 3, device=dev)
        dist_out = torch.zeros(N, device=dev)
        hit_out = torch.zeros(N, dtype=torch.bool, device=dev)
        idx = sampler_mask.nonzero()[:, 0]
        m = idx.numel()
        if m == 0:
            return pts_out, hit_out, dist_out
        mm = sampler_min_max.reshape(-1, 2).float()
        cam_m = cam if cam.shape[0] == 1 else cam[idx].contiguous()
        d_m = dirs[idx].contiguous()
        lin = torch.linspace(0, 1, steps=self.n_steps, device=dev)
        z, P = ops.raytrace_samples(cam_m, d_m, mm[idx, 0].contiguous(), mm[idx, 1].contiguous(), lin)
        s = torch.cat([sdf(p) for p in torch.split(P, 1 << 20, dim=0)])
        sp, sd, neg, bracket = ops.raytrace_pick(s, z, P, obj[idx])
        net_surface = neg.bool()
        if self.training:                                  # :256: the secant only where the object mask agrees
            neg = neg & obj[idx].to(neg.dtype)
        if self.n_secant_steps >= 0 and bool(neg.any()):
            zp, pmid = self._secant(cam_m, d_m, neg, bracket, sdf)
            on = neg.bool()
            sp = torch.where(on[:, None], pmid, sp)
            sd = torch.where(on, zp, sd)
        pts_out[idx], dist_out[idx], hit_out[idx] = sp, sd, net_surface
        return pts_out, hit_out, dist_out

    def _secant(self, cam_m, d_m, on, bracket, sdf):
        """n_secant_steps refinements of the bracket [4,m] = (z_low, z_high, sdf_low, sdf_high) on the rows with `on` set."""
        m, dev = d_m.shape[0], d_m.device
        zp = torch.zeros(m, dtype=torch.float32, device=dev)
        pmid = torch.zeros(m, 3, dtype=torch.float32, device=dev)
        ops.raytrace_secant(cam_m, d_m, on, None, 0, bracket, zp, pmid)
        for _ in range(self.n_secant_steps):
            ops.raytrace_secant(cam_m, d_m, on, sdf(pmid), 1, bracket, zp, pmid)
        return zp, pmid

    @torch.no_grad()
    def secant(self, sdf_low, sdf_high, z_low, z_high, cam_loc, ray_directions, sdf):
        """model/ray_tracing.py:276-297: the secant method on [z_low, z_high] for n_secant_steps -> z_pred [m].  Like the reference the four
        bracket tensors are updated in place."""
        d_m = ray_directions.reshape(-1, 3).float().contiguous()
        cam_m = cam_loc.reshape(-1, 3).float().contiguous()
        m = d_m.shape[0]
        bracket = torch.stack([z_low.float(), z_high.float(), sdf_low.float(), sdf_high.float()]).contiguous()
        on = torch.ones(m, dtype=torch.uint8, device=d_m.device)
        zp, _ = self._secant(cam_m, d_m, on, bracket, sdf)
        for t, row in ((z_low, 0), (z_high, 1), (sdf_low, 2), (sdf_high, 3)):
            t.copy_(bracket[row])
        return zp

    @torch.no_grad()
    def minimal_sdf_points(self, num_pixels, sdf, cam_loc, ray_directions, mask, min_dis, max_dis):
        """model/ray_tracing.py:299-326: among n_steps uniform depths in [min_dis, max_dis] of every ray in `mask` the point with the
        smallest SDF -> (points [k,3], dists [k]).  ray_directions [N,3] flat, as the reference's forward passes it."""
        dirs = ray_directions.reshape(-1, 3).float()
        N, dev = dirs.shape[0], dirs.device
        mask = mask.reshape(-1).bool()
        k = int(mask.sum())
        cam_all = cam_loc.reshape(-1, 3).float()
        cam_all = cam_all.unsqueeze(1).repeat(1, num_pixels, 1).reshape(-1, 3) if cam_all.shape[0] * num_pixels == N else cam_all.expand(N, 3)
        steps = self.min_sdf_steps if self.min_sdf_steps is not None else torch.empty(self.n_steps).uniform_(0.0, 1.0)
        steps = steps.to(dev).float()
        lo, hi = min_dis[mask].unsqueeze(-1), max_dis[mask].unsqueeze(-1)
        zt = steps.unsqueeze(0).repeat(k, 1) * (hi - lo) + lo
        Pa = cam_all[mask].unsqueeze(1).repeat(1, self.n_steps, 1) + zt.unsqueeze(-1) * dirs[mask].unsqueeze(1).repeat(1, self.n_steps, 1)
        sv = torch.cat([sdf(p) for p in torch.split(Pa.reshape(-1, 3), 1 << 20, dim=0)]).reshape(k, self.n_steps)
        j = sv.argmin(-1)
        rows = torch.arange(k, device=dev)
        return Pa[rows, j], zt[rows, j]
