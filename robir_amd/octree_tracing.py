"""Drop-in for the reference's model/octree_tracing.py (+ the live parts of utils/octree.py) on HIP kernels.

OctreeTracing (octree_tracing.py:8-60): same constructor, generate(sdf_fn, tex_sampler=None),
__call__(sdf=, cam_loc=, object_mask=, ray_directions=) -> (x [M,3], hit [M] bool, t [M]);  `.sdf_octree.max_iter`
is read/written by callers (OctreeVisModel, octree_tracing.py:68), so the built tree is exposed as `.sdf_octree`.
OctreeVisModel (octree_tracing.py:63-85): traced visibility as a VisModel callable.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import ops, _lib
from ._lib import ptr, call, stream_ptr
import ctypes

CELL = 0.05
LEVELS = 4


class OctreeSDF:
    """Device-resident counterpart of utils/octree.py:375-438 (build + cast)."""

    def __init__(self, tables, max_iter=-1):
        self.tables = tables
        self.max_iter = max_iter
        self.min_step = tables.min_step

    # ------------------------------------------------------------------ build (octree.py:377-409, 124-181)
    @classmethod
    def build(cls, sdf_network, bounds, max_iter=-1, in_scale=2.0, out_scale=0.5, thr=0.5, eval_chunk=1 << 18):
        """sdf_network: robir_amd.nets.SDFNetwork (NeuS-space net; stage-2 sdf(x) = out_scale*net(in_scale*x))."""
        dev = next(sdf_network.parameters()).device
        bmin = np.array(bounds[0], dtype=np.float32)
        size = np.array([bounds[1][i] - bounds[0][i] for i in range(3)], dtype=np.float64).astype(np.float32)
        ncell = np.ceil(size / np.float32(CELL)).astype(np.int64)
        root_size = (ncell.astype(np.float32) * np.float32(CELL)).astype(np.float32)
        res = ncell.astype(np.int32)
        n0 = int(np.prod(ncell))
        cap = max(2 * n0, n0 + (1 << 20))
        node = torch.empty(cap, 8, dtype=torch.float32, device=dev)
        centre = torch.empty(cap, 3, dtype=torch.float32, device=dev)
        rm, rs, rr = ops._host3(bmin, ctypes.c_float), ops._host3(root_size, ctypes.c_float), ops._host3(res, ctypes.c_int)
        call("rb_octree_base_grid", rm, rs, rr, ptr(node), ptr(centre), stream_ptr())

        def sdf_at(c):
            outs = [sdf_network.eval_points(c[i:i + eval_chunk], in_scale, out_scale, full=False, precise=True)[0]
                    for i in range(0, c.shape[0], eval_chunk)]
            return torch.cat(outs) if len(outs) > 1 else outs[0]

        first, count, total = 0, n0, n0
        for _ in range(LEVELS):
            sdf = sdf_at(centre[first:first + count])
            flag = torch.empty(count, dtype=torch.int32, device=dev)
            rank = torch.empty(count, dtype=torch.int32, device=dev)
            tmp = torch.empty((count + 1023) // 1024 + 1, dtype=torch.int32, device=dev)
            tot = torch.zeros(1, dtype=torch.int32, device=dev)
            call("rb_octree_mark_split", ptr(node), ctypes.c_long(first), ctypes.c_long(count), ptr(sdf),
                 ctypes.c_float(thr), ptr(flag), ptr(rank), ptr(tmp), ptr(tot), stream_ptr())
            ns = int(tot.item())
            if ns == 0:
                break
            need = total + 8 * ns
            if need > node.shape[0]:
                grow = max(need, 2 * node.shape[0])
                node = torch.cat([node, torch.empty(grow - node.shape[0], 8, dtype=torch.float32, device=dev)])
                centre = torch.cat([centre, torch.empty(grow - centre.shape[0], 3, dtype=torch.float32, device=dev)])
            call("rb_octree_subdivide", ptr(node), ptr(centre), ctypes.c_long(first), ctypes.c_long(count), ptr(flag),
                 ptr(rank), ctypes.c_long(total), stream_ptr())
            first, count, total = total, 8 * ns, need
        node = node[:total].contiguous()
        centre = centre[:total].contiguous()
        nrm = torch.empty(total, 3, dtype=torch.float32, device=dev)
        for i in range(0, total, eval_chunk):
            c = centre[i:i + eval_chunk]
            sdf, grad = sdf_network.eval_points(c, in_scale, out_scale, full=False, grad=True, precise=True)
            call("rb_octree_store_cells", ptr(node), ptr(nrm), ctypes.c_long(i), ctypes.c_long(c.shape[0]), ptr(sdf),
                 ptr(grad), stream_ptr())
        torch.cuda.current_stream().synchronize()
        leaf = float(np.float32(CELL) / np.float32(2 ** LEVELS))
        tables = ops.OctreeTablesDev(node, nrm, total, bmin, root_size, res, leaf + 1e-4)
        print(total, "boxes", total * 44 // 1024 // 1024, "MB")
        return cls(tables, max_iter)

    @classmethod
    def from_host_tables(cls, T, device, max_iter=-1):
        """Upload octree tables held on the host: any object with box_min/box_size [B,3], child [B,8], is_split [B],
        sdf_val [B], sdf_nrm [B,3], root_min/root_size [3], base_index [nx,ny,nz], min_step (e.g. a tree built elsewhere)."""
        B = T.box_min.shape[0]
        node = torch.zeros(B, 8, dtype=torch.float32)
        node[:, 0:3] = T.box_min
        fc = torch.where(T.is_split, T.child[:, 0], torch.full((B,), -1, dtype=torch.long)).to(torch.int32)
        node[:, 3] = fc.view(torch.float32)
        node[:, 4:7] = T.box_size
        node[:, 7] = T.sdf_val
        tables = ops.OctreeTablesDev(node.to(device), T.sdf_nrm.float().contiguous().to(device), B,
                                     T.root_min.numpy().astype(np.float32), T.root_size.numpy().astype(np.float32),
                                     np.array(T.base_index.shape, dtype=np.int32), T.min_step)
        return cls(tables, max_iter)

    # ------------------------------------------------------------------ cast (octree.py:421-438, 493-585)
    def step_size(self, R):
        if self.max_iter > 0:
            return 0.01 if R > 100000 else 0.005
        return 0.001

    def cast(self, rays_o, rays_d, return_is_hit=False):
        """One lock-step batch: rays_o, rays_d [R,3] -> t [R,1] (, hit [R])."""
        x, hit, t = self.cast_full(rays_o, rays_d)
        if return_is_hit:
            return t[..., None], hit
        return t[..., None]

    def cast_full(self, rays_o, rays_d, sched_cap=0):
        R = rays_d.shape[0]
        if R == 0:
            z = torch.zeros(0, device=rays_d.device)
            return torch.zeros(0, 3, device=rays_d.device), z.bool(), z
        step = self.step_size(R)
        if R <= 1024:
            x, hit, t, sched = ops.octree_cast_batched(self.tables, rays_o.contiguous(), True, rays_d.contiguous(), R,
                                                       self.max_iter, step, sched_cap)
            self.last_sched = sched
            return x, hit, t
        x, hit, t, counters = ops.octree_cast_general(self.tables, rays_o.contiguous(), rays_d.contiguous(), self.max_iter,
                                                      step)
        self.last_counters = counters
        return x, hit, t

    def cast_chunks(self, cam_loc, rays_d, chunk=1024, sched_cap=0):
        """Many independent lock-step batches (consecutive chunks of `chunk` <= 1024 rays sharing one origin each):
        exactly what the reference computes when it renders the chunks one after another.  cam_loc [3] or [nb,3]."""
        R = rays_d.shape[0]
        nb = (R + chunk - 1) // chunk
        o = cam_loc.reshape(-1, 3).float()
        if o.shape[0] == 1:
            o = o.expand(nb, 3)
        x, hit, t, sched = ops.octree_cast_batched(self.tables, o.contiguous(), False, rays_d.contiguous(), chunk,
                                                   self.max_iter, self.step_size(chunk), sched_cap)
        self.last_sched = sched
        return x, hit, t


class OctreeTracing(nn.Module):
    def __init__(self, object_bounding_sphere=1.0, sdf_threshold=5.0e-5, line_search_step=0.5, line_step_iters=1,
                 sphere_tracing_iters=10, n_steps=100, n_rootfind_steps=8, max_iter=-1):
        super().__init__()
        self.object_bounding_sphere = object_bounding_sphere
        self.sdf_threshold = sdf_threshold
        self.sphere_tracing_iters = sphere_tracing_iters
        self.line_step_iters = line_step_iters
        self.line_search_step = line_search_step
        self.n_steps = n_steps
        self.n_secant_steps = n_rootfind_steps
        self.sdf_octree = None
        self.max_iter = max_iter
        self._implicit = None

    def bind(self, implicit_network):
        """Give the tracer the network whose SDF it caches (IDRNetwork does this); generate() then evaluates the SDF
        and its gradient with the HIP kernels instead of calling the opaque sdf_fn."""
        object.__setattr__(self, "_implicit", implicit_network)

    def generate(self, sdf_fn, tex_sampler=None):
        """octree_tracing.py:31-41.  The tables are built from the BOUND network (values with the library-grade softplus, gradients by the
        reverse pass -- an opaque callable cannot be differentiated on the device), so `sdf_fn` is CHECKED against it instead of being
        evaluated cell by cell: on a fixed probe lattice inside the box it must reproduce the bound network's signed distance to 1e-5, else
        this raises (a caller handing over a different field would otherwise get the wrong octree silently).  sdf_fn=None: no check."""
        box_min, box_max = [-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]
        if tex_sampler is not None:      # mesh bounding box +- 1e-3 in halved coordinates (octree_tracing.py:33-37)
            v = tex_sampler.tex_sampler.vert.view(3, -1).permute(1, 0) * 0.5
            m = tex_sampler.tex_sampler.mask.view(-1) > 0.9
            box_max = [c.item() + 1e-3 for c in v[m].max(0)[0]]
            box_min = [c.item() - 1e-3 for c in v[m].min(0)[0]]
        print("[BOX]", box_min, box_max)
        if self._implicit is None:
            raise RuntimeError("OctreeTracing.generate: bind(implicit_network) first (the HIP build evaluates the SDF "
                               "network directly; an opaque sdf_fn callable cannot be differentiated on the device)")
        if sdf_fn is not None:
            self._check_sdf_fn(sdf_fn, box_min, box_max)
        self.sdf_octree = OctreeSDF.build(self._implicit.neus_model.sdf_network, [box_min, box_max], max_iter=self.max_iter)

    def _check_sdf_fn(self, sdf_fn, box_min, box_max, n=4096, tol=1e-5):
        dev = next(self._implicit.parameters()).device
        # a fixed low-discrepancy lattice (no draw from torch's generators: the renderer's random streams must not move)
        i = np.arange(n, dtype=np.float64) + 0.5
        frac = np.stack([(i * a) % 1.0 for a in (0.8191725133961645, 0.6710436067037893, 0.5497004779019703)], -1)
        lo, hi = np.array(box_min, dtype=np.float64), np.array(box_max, dtype=np.float64)
        probe = torch.from_numpy((lo + frac * (hi - lo)).astype(np.float32)).to(dev)
        with torch.no_grad():
            got = sdf_fn(probe)
            want = self._implicit.sdf_only(probe)
        from . import deferred
        got = deferred.plain(got)
        got = got.reshape(-1).float()
        if got.shape[0] != n:
            raise ValueError(f"OctreeTracing.generate: sdf_fn returned {tuple(got.shape)} for {n} probe points")
        err = float((got - want.reshape(-1)).abs().max())
        if not err <= tol:
            raise ValueError(f"OctreeTracing.generate: sdf_fn differs from the bound implicit network by {err:.3g} on the probe lattice "
                             f"(> {tol:g}).  The HIP octree is built from the bound network (bind(implicit_network)); pass that network's own "
                             "signed distance (what every runner does: `lambda x: model.implicit_network(x)[:, 0]`) or bind the other network")

    def forward(self, sdf, cam_loc, object_mask, ray_directions):
        """cam_loc [K,3], ray_directions [K,P,3] -> x [K*P,3], hit [K*P] bool, t [K*P]; one lock-step batch."""
        K, P, _ = ray_directions.shape
        rays_d = ray_directions.reshape(-1, 3).float()
        if K == 1 and K * P <= 1024:
            return self.sdf_octree.cast_chunks(cam_loc.reshape(1, 3), rays_d, chunk=K * P)
        rays_o = cam_loc[:, None, :].expand(K, P, 3).reshape(-1, 3).float()
        return self.sdf_octree.cast_full(rays_o, rays_d)


class OctreeVisModel(nn.Module):
    """octree_tracing.py:63-85: traced visibility as a VisModel -- [is_hit, ~is_hit] as float 'logits' of ONE lock-step
    secondary cast of the rays it is given.  robir_amd.sg_render recognises this class: the light-SG visibility then runs the
    fused cull + grouped cast of csrc/octree_vis.hip (no (point, direction) pairs materialised, any number of chunks per call,
    the reference's 2 M-pair batches), the BRDF-lobe visibility the grouped cast per chunk (`forward_groups`)."""

    def __init__(self, ray_tracer):
        super().__init__()
        self.ray_tracer = ray_tracer
        self.ray_tracer.sdf_octree.max_iter = 32

    def intersect_sphere(self, points, view_dirs, radius=1.0):
        """octree_tracing.py:70-76: where the ray from a point INSIDE the sphere of `radius` leaves it (NaN outside, like the reference)."""
        shape = points.shape
        return ops.intersect_sphere(points.reshape(-1, 3).float().contiguous(), view_dirs.reshape(-1, 3).float().contiguous(),
                                    float(radius)).reshape(shape)

    def forward(self, points, view_dirs):
        with torch.no_grad():
            _, is_hit = self.ray_tracer.sdf_octree.cast(points, view_dirs, return_is_hit=True)
            return torch.stack([is_hit, ~is_hit], dim=-1).float()

    def forward_groups(self, points, view_dirs, group_start):
        """The same for rays partitioned into independent lock-step groups (one per chunk of a multi-chunk render): equals
        calling forward() once per group."""
        with torch.no_grad():
            tree = self.ray_tracer.sdf_octree
            _, is_hit, _ = ops.octree_cast_grouped(tree.tables, points.float().contiguous(), view_dirs.float().contiguous(),
                                                   group_start, tree.max_iter)
            return torch.stack([is_hit, ~is_hit], dim=-1).float()
