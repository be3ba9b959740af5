#!/usr/bin/env python
"""Timing of the SDF value kernels of one library build: `python tools/ab_sdf.py <tag> <lib.so> [points]` (A/B and ablation builds
made by tools/build_variant.sh; modes 0 = distance only, 1 = all 257 outputs, 5 via sdf_value_grad)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag, lib = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20
from robir_amd import _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(ROOT, lib)
from robir_amd import ops, packing, synth  # noqa: E402

dev = torch.device("cuda:0")
sd = synth.synth_state_dict(0, variance=0.3)
full, dist = packing.pack_sdf_h3(sd, dev, full=True), packing.pack_sdf_h3(sd, dev, full=False)
back = packing.pack_sdf_back_h3(sd, dev)
g = torch.Generator().manual_seed(1)
x = ((torch.rand(n, 3, generator=g) - 0.5) * 1.2).to(dev)
X = ops.feat_pe10(x, scale=2.0)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e))
    return best


t0 = timed(lambda: ops.sdf_mlp_h3(X, n, dist, 0, packing.H3_SCALE_LOG2, 0.5, 1.0))
t1 = timed(lambda: ops.sdf_mlp_h3(X, n, full, 1, packing.H3_SCALE_LOG2, 0.5, 1.0))
# (the backward kernel detects sigmoid rows by their content: never run it behind an ablated value pass)
t5 = timed(lambda: ops.sdf_value_grad(x, n, full, back, packing.H3_SCALE_LOG2, in_scale=2.0, out_scale=0.5)) if (tag == "base" or tag.startswith("ok")) else float("nan")
o0 = ops.sdf_mlp_h3(X, n, dist, 0, packing.H3_SCALE_LOG2, 0.5, 1.0)[0]
o1 = ops.sdf_mlp_h3(X, n, full, 1, packing.H3_SCALE_LOG2, 0.5, 1.0)[0]
print(f"{tag}: checksums {float(o0.double().sum()):.9e} {float(o1.double().abs().sum()):.9e}")
print(f"{tag}: {n} points: mode 0 {t0:.3f} ms, mode 1 {t1:.3f} ms, value+gradient op {t5:.3f} ms", flush=True)
