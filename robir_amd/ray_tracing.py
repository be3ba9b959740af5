"""IDR sphere tracer -- drop-in for model/ray_tracing.py `RayTracing` (the `use_octree=False` ray tracer).

Same constructor arguments and `forward(sdf, cam_loc, object_mask, ray_directions)` signature as the reference
(model/ray_tracing.py:6-72).  `sdf` is any callable [M,3] -> [M] on device tensors (IDRNetwork passes the MFMA SDF kernel);
every per-ray update in between runs in the HIP kernels of csrc/raytrace.hip.  Rays are independent, so instead of the
reference's boolean-mask gathers the SDF is evaluated on all 2N start/end points each step and masked rows are ignored --
same per-ray arithmetic, no host synchronisation inside the sphere-tracing loop.

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

    @torch.no_grad()
    def forward(self, sdf, cam_loc, object_mask, ray_directions):
        training = self.training       # the module's training-mode behaviour (ray_tracing.py:68-100, 256): still no gradients, like the reference's call under no_grad
        if sdf is None:
            sdf = self._bound.sdf_only
        batch, npix, _ = ray_directions.shape
        dirs = ray_directions.reshape(-1, 3).float().contiguous()
        cam = cam_loc.reshape(-1, 3).float().contiguous()
        if batch > 1:
            assert npix == 1 and cam.shape[0] == batch, "per-ray origins come as [N,3] origins with [N,1,3] directions"
        N = dirs.shape[0]
        dev = dirs.device
        if N == 0:
            return (torch.zeros(0, 3, device=dev), torch.zeros(0, dtype=torch.bool, device=dev), torch.zeros(0, device=dev))
        obj = object_mask.reshape(-1).bool()
        st = ops.RayTraceState(cam, dirs)
        st.step(0, float(self.object_bounding_sphere) ** 2)
        if training:       # sphere entry / exit distances and the rays that meet the sphere (ray_tracing.py:124-126), before the loop moves them
            min_dis, max_dis, inter = st.f[0].clone(), st.f[1].clone(), st.b[0].bool()
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
        acc_s, acc_e = st.f[0], st.f[1]
        hit = acc_s < acc_e
        points, dist = st.pts[:N].clone(), acc_s.clone()
        idx = st.b[0].nonzero()[:, 0]                       # not converged: sampler + secant (ray_tracing.py:208-297)
        m = idx.numel()
        if m > 0:
            cam_m = cam if cam.shape[0] == 1 else cam[idx].contiguous()
            d_m = dirs[idx].contiguous()
            lin = torch.linspace(0, 1, steps=self.n_steps, device=dev)
            z, P = ops.raytrace_samples(cam_m, d_m, acc_s[idx], acc_e[idx], lin)
            s = torch.cat([sdf(p) for p in torch.split(P, 1 << 20, dim=0)])
            sp, sd, neg, bracket = ops.raytrace_pick(s, z, P, obj[idx])
            net_surface = neg.bool()
            if training:                                   # :256: the secant only where the object mask agrees
                neg = neg & obj[idx].to(neg.dtype)
            if self.n_secant_steps >= 0 and bool(neg.any()):
                zp = torch.zeros(m, dtype=torch.float32, device=dev)
                pmid = torch.zeros(m, 3, dtype=torch.float32, device=dev)
                ops.raytrace_secant(cam_m, d_m, neg, None, 0, bracket, zp, pmid)
                for _ in range(self.n_secant_steps):
                    ops.raytrace_secant(cam_m, d_m, neg, sdf(pmid), 1, bracket, zp, pmid)
                on = neg.bool()
                sp = torch.where(on[:, None], pmid, sp)
                sd = torch.where(on, zp, sd)
            points[idx], dist[idx], hit[idx] = sp, sd, net_surface
        if not training:
            return points, hit, dist
        # ---- training-mode tail (ray_tracing.py:73-100): rays without a surface get a point for the mask loss
        sampler = st.b[0].bool()
        in_mask, out_mask = ~hit & obj & ~sampler, ~obj & ~sampler
        cam_all = cam.expand(N, 3)
        left = (in_mask | out_mask) & ~inter                # miss the bounding sphere: the foot of the camera centre on the ray
        if bool(left.any()):
            dist[left] = -(dirs[left] * cam_all[left]).sum(-1)
            points[left] = cam_all[left] + dist[left].unsqueeze(1) * dirs[left]
        m = (in_mask | out_mask) & inter
        k = int(m.sum())
        if k > 0:                                           # minimal_sdf_points (:299-326): n_steps uniform depths between entry and exit
            sel = hit & out_mask
            min_dis[sel] = dist[sel]
            steps = self.min_sdf_steps if self.min_sdf_steps is not None else torch.empty(self.n_steps).uniform_(0.0, 1.0)
            steps = steps.to(dev).float()
            lo, hi = min_dis[m].unsqueeze(-1), max_dis[m].unsqueeze(-1)
            zt = steps.unsqueeze(0).repeat(k, 1) * (hi - lo) + lo
            Pa = cam_all[m].unsqueeze(1).repeat(1, self.n_steps, 1) + zt.unsqueeze(-1) * dirs[m].unsqueeze(1).repeat(1, self.n_steps, 1)
            sv = torch.cat([sdf(p) for p in torch.split(Pa.reshape(-1, 3), 1 << 20, dim=0)]).reshape(k, self.n_steps)
            j = sv.argmin(-1)
            rows = torch.arange(k, device=dev)
            points[m], dist[m] = Pa[rows, j], zt[rows, j]
        return points, hit, dist
