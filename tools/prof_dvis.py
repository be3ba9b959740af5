#!/usr/bin/env python
"""Stand-alone driver of the fused light-visibility kernel for rocprofv3 (kernel trace or --pmc passes):
`python tools/prof_dvis.py [fp32|f16x3] [n_chunks]` renders n_chunks 1024-px chunks of the 800x800 synthetic view
up to the visibility stage only."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robir_amd import ops, renderer, sg_render, synth  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
n_chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
model = renderer.build_synthetic_model(dev)
uv, pose, K = synth.synth_camera(800, 800)
first = 300 * 800 // 1024
sl = slice(first * 1024, (first + n_chunks) * 1024)
uv_d = torch.from_numpy(uv[sl]).to(dev)
dirs = ops.camera_rays(pose, K, uv_d)
cam = torch.from_numpy(pose[:3, 3]).to(dev).reshape(1, 3)
_, hit, dist = model.ray_tracer.sdf_octree.cast_chunks(cam, dirs, chunk=1024)
pts = ops.points_along(cam.expand(dirs.shape[0], 3).contiguous(), dirs, dist)
idx = hit.nonzero()[:, 0]
hp = pts[idx].contiguous()
cid = (idx // 1024).to(torch.int32).contiguous()
nrm = ops.normalize3(model.implicit_network.gradient(hp)[:, 0, :].contiguous(), 1e-4, 1)
lgt = model.envmap_material_network.lgtSGs.detach()
g = torch.Generator(device=dev).manual_seed(1)
u = torch.rand(2, n_chunks, 128, 32, device=dev, generator=g)
if prec == "variants":
    import itertools
    names = ["f16x3", "f16x3-v2"]
    if len(sys.argv) > 3:
        names = sys.argv[3].split(",")
    times = {k: [] for k in names}
    for rep in range(4):
        for k in names:
            sg_render.VIS_PRECISION = k
            sp = model.visibility_network.packed_split()
            A = ops.linear_64_256(ops.feat_pe10(hp), sp["point"])
            d_, w_, ws_ = ops.dvis_dirs(lgt, u[0], u[1], 1.0)
            Bd = ops.linear_64_256(ops.feat_pe10(d_), sp["dir"])
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            o = ops.dvis_fused(nrm, cid, A, Bd, d_, w_, ws_, sp, 128, 32, False, None, precision=k)
            e.record()
            torch.cuda.synchronize()
            times[k].append(s.elapsed_time(e))
    for k in names:
        print(k, " ".join(f"{t:.2f}" for t in times[k]), "ms  (min %.2f)" % min(times[k]))
    if os.environ.get("RB_V2_TIMED") == "1":
        import ctypes
        from robir_amd import _lib
        buf = (ctypes.c_ulonglong * 8)()
        _lib.lib().rb_dvis_v2_debug(buf)
        tot = sum(buf[:6]) or 1
        print("v2 phases (wave-0 clocks, share):", ", ".join(f"{n}={100 * buf[i] / tot:.1f}%" for i, n in
              enumerate(["prologue", "ring-start", "gather", "layers", "head", "final"])), "total", tot)
    sys.exit(0)
sg_render.VIS_PRECISION = prec
stats = {}
for it in range(3):
    out = sg_render._diffuse_vis_core(hp, nrm, model.visibility_network, lgt, u[0], u[1], 1.0, False, cid, n_chunks, stats)
torch.cuda.synchronize()
ev = int(stats["diffuse_vis_evals"]) // 3
print(f"{prec}: {hp.shape[0]} points, {ev} evals per launch, out mean {float(out.mean()):.6f}")
