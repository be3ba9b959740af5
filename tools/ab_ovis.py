#!/usr/bin/env python
"""Traced light visibility of a whole 800x800 view under the four settings of (compaction, chunk groups):
`python tools/ab_ovis.py [chunks]` -> time of ops.dvis_octree per setting (HIP events) and equality of the results."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robir_amd import ops, renderer, synth  # noqa: E402

n_chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 625
dev = torch.device("cuda:0")
os.environ.setdefault("ROBIR_PRECISION", "split")
model = renderer.build_synthetic_model(dev)
uv, pose, K = synth.synth_camera(800, 800)
uv_d = torch.from_numpy(uv[: n_chunks * 1024]).to(dev)
dirs = ops.camera_rays(pose, K, uv_d)
cam = torch.from_numpy(pose[:3, 3]).to(dev).reshape(1, 3)
tree = model.ray_tracer.sdf_octree
_, hit, dist = tree.cast_chunks(cam, dirs, chunk=1024)
pts = ops.points_along(cam.expand(dirs.shape[0], 3).contiguous(), dirs, dist)
idx = hit.nonzero()[:, 0]
hp = pts[idx].contiguous()
cid = (idx // 1024).to(torch.int32).contiguous()
nrm = ops.normalize3(model.implicit_network.gradient(hp)[:, 0, :].contiguous(), 1e-4, 1)
lgt = model.envmap_material_network.lgtSGs.detach()
g = torch.Generator(device=dev).manual_seed(1)
u = torch.rand(2, n_chunks, 128, 32, device=dev, generator=g)
d_, w_, ws_ = ops.dvis_dirs(lgt, u[0], u[1], 1.0)
T = model.octree_ray_tracer.sdf_octree.tables
ref = None
for compact, per_call in ((True, 48), (True, 16), (True, 160), (False, 48), (False, 100000)):
    ops.OVIS_COMPACT, ops.OVIS_CHUNKS_PER_CALL = compact, per_call
    try:
        times = []
        for rep in range(2):
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = ops.dvis_octree(T, hp, nrm, cid, n_chunks, d_, w_, ws_, 128, 32, max_iter=32)
            e.record()
            torch.cuda.synchronize()
            times.append(s.elapsed_time(e))
        lay = [int(v) for v in ops.LAST_OCTREE_VIS_LAYOUT.cpu()]
        same = None if ref is None else bool(torch.equal(out, ref))
        ref = out if ref is None else ref
        print(f"compact={compact} chunks/call={per_call}: {min(times):.1f} ms (runs {times[0]:.0f} {times[1]:.0f}), pairs {lay[0]}, "
              f"mem peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB, equal to first: {same}", flush=True)
    except Exception as ex:      # e.g. out of memory for the ungrouped whole view
        print(f"compact={compact} chunks/call={per_call}: {type(ex).__name__}: {str(ex)[:100]}", flush=True)
    torch.cuda.reset_peak_memory_stats()
