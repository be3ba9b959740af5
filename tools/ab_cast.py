#!/usr/bin/env python
"""Timing of the general lock-step cast (secondary rays, max_iter = 32): per-iteration launches vs the one-launch persistent kernel.
`ROBIR_CAST_LPR=1|4|16 python tools/ab_cast.py [rays]`"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robir_amd import ops, renderer
dev = torch.device("cuda:0")
model = renderer.build_synthetic_model(dev)
oct_ = model.ray_tracer.sdf_octree
from robir_amd.octree_tracing import OctreeSDF
od = OctreeSDF(oct_.tables, 32)
for n in [int(a) for a in sys.argv[1:]] or [7500, 60000, 500000]:
    g = np.random.Generator(np.random.PCG64(11))
    o = g.standard_normal((n, 3)).astype(np.float32); o = 0.5 * o / np.linalg.norm(o, axis=1, keepdims=True)
    d = g.standard_normal((n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    o_t, d_t = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    step = od.step_size(n)
    for one in (False, True):
        f = lambda: ops.octree_cast_general(od.tables, o_t, d_t, 32, step, one_launch=one)
        r = f(); torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(10): f()
        torch.cuda.synchronize()
        c = r[3].cpu()
        print(f"rays {n:7d} one_launch={one}: {(time.time() - t0) / 10 * 1e3:.3f} ms   hit {float(r[1].float().mean()):.3f}  active per iteration {c[:6].tolist()} ... iterations {int((c > 0).sum())}", flush=True)
