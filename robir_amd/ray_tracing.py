"""IDR sphere tracer -- drop-in for model/ray_tracing.py `RayTracing` (the `use_octree=False` ray tracer), eval mode.

Same constructor arguments and `forward(sdf, cam_loc, object_mask, ray_directions)` signature as the reference
(model/ray_tracing.py:6-72).  `sdf` is any callable [M,3] -> [M] on device tensors (IDRNetwork passes the MFMA SDF kernel);
every per-ray update in between runs in the HIP kernels of csrc/raytrace.hip.  Rays are independent, so instead of the
reference's boolean-mask gathers the SDF is evaluated on all 2N start/end points each step and masked rows are ignored --
same per-ray arithmetic, no host synchronisation inside the sphere-tracing loop.

The training-mode tail (minimal_sdf_points with uniform_ draws, :73-100) is not part of the forward renderer and raises.
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

    def bind(self, implicit_network):
        """Default SDF when forward() is called with sdf=None (mirrors OctreeTracing.bind)."""
        object.__setattr__(self, "_bound", implicit_network)       # not a submodule: no duplicate state-dict keys

    def generate(self, *a, **k):          # OctreeTracing API no-op so callers can treat both tracers alike
        return None

    @torch.no_grad()
    def forward(self, sdf, cam_loc, object_mask, ray_directions):
        if self.training:
            raise NotImplementedError("RayTracing training mode (minimal_sdf_points, ray_tracing.py:73-100) is out of scope")
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
            if self.n_secant_steps >= 0 and bool(neg.any()):
                zp = torch.zeros(m, dtype=torch.float32, device=dev)
                pmid = torch.zeros(m, 3, dtype=torch.float32, device=dev)
                ops.raytrace_secant(cam_m, d_m, neg, None, 0, bracket, zp, pmid)
                for _ in range(self.n_secant_steps):
                    ops.raytrace_secant(cam_m, d_m, neg, sdf(pmid), 1, bracket, zp, pmid)
                on = neg.bool()
                sp = torch.where(on[:, None], pmid, sp)
                sd = torch.where(on, zp, sd)
            points[idx], dist[idx], hit[idx] = sp, sd, neg.bool()
        return points, hit, dist
