#!/usr/bin/env python
"""Timing of the exact-operand SDF kernels of one library build: `python tools/ab_sdf_x6.py <tag> <lib.so>` (A/B and ablation builds
made by tools/build_variant.sh; -DSXA_* builds give wrong results on purpose)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag, lib = sys.argv[1], sys.argv[2]
from robir_amd import _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(ROOT, lib)
from robir_amd import ops, packing, synth  # noqa: E402

dev = torch.device("cuda:0")
w = synth.synth_state_dict(0, variance=0.3)
g = torch.Generator().manual_seed(1)


def timed(fn, reps=7):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


n = 1 << 20
p = ((torch.rand(n, 3, generator=g) - 0.5) * 0.6).to(dev)
mac0 = 64 * 256 + 2 * 256 * 256 + 256 * 208 + 288 * 256 + 3 * 256 * 256
out = []
b0, b1 = packing.pack_sdf_x6(w, dev, full=False), packing.pack_sdf_x6(w, dev, full=True)
t = timed(lambda: ops.sdf_points_x6(p, n, b0, False))
out.append(f"dist {t:.3f} ms ({2 * (mac0 + 256 * 16) * n / t / 1e9 / 416.7:.3f})")
t = timed(lambda: ops.sdf_points_x6(p, n, b1, True))
out.append(f"full {t:.3f} ms ({2 * (mac0 + 256 * 272) * n / t / 1e9 / 416.7:.3f})")
if "grad" in sys.argv[3:]:
    back = packing.pack_sdf_back_x6(w, dev) + (packing.pack_sdf_back_x6(w, dev, two_tile=True)[0],)
    t = timed(lambda: ops.sdf_value_grad_x6(p, n, b1, back))
    out.append(f"value+grad {t:.3f} ms")
    v, gr = ops.sdf_value_grad_x6(p, n, b1, back)
    out.append(f"sum {float(v.double().sum()):.9e} {float(gr.double().sum()):.9e}")
else:
    v = ops.sdf_points_x6(p, n, b1, True)
    out.append(f"sum {float(v.double().sum()):.9e}")
print(tag, " | ".join(out), flush=True)
