#!/usr/bin/env python
"""Timing and agreement of the two forms of the exact-operand colour kernel (csrc/color_x6.hip: one tile per wave, color_x6t.hip: two):
`python tools/ab_color_x6.py [rows ...]`."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("ROBIR_AB_LIB"):
    from robir_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["ROBIR_AB_LIB"])
from robir_amd import ops, packing, synth  # noqa: E402

dev = torch.device("cuda:0")
w = synth.synth_state_dict(0, variance=0.3)
g = torch.Generator().manual_seed(1)
blob = packing.pack_color_x6(w, dev)
MAC = 320 * 256 + 3 * 256 * 256 + 256 * 16          # padded multiply-adds per row, as the kernels compute them


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


for n in [int(a) for a in sys.argv[1:]] or [1 << 15, 1 << 17, 1 << 20, 3 << 20]:
    x = ((torch.rand(n, 3, generator=g) - 0.5)).to(dev)
    v = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
    nr = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
    out = torch.randn(n, 257, generator=g).to(dev)
    line = [f"rows {n}"]
    res = []
    for two in (False, True):
        t = timed(lambda: ops.color_x6_points(x, v, nr, out[:, 1:], blob, two_tile=two))
        res.append(ops.color_x6_points(x, v, nr, out[:, 1:], blob, two_tile=two))
        line.append(f"{'two' if two else 'one'}-tile {t:.3f} ms ({2 * MAC * n / t / 1e9 / 416.7:.3f} of 417 TFLOP/s)")
    line.append(f"max |diff| {float((res[0] - res[1]).abs().max()):.2e}")
    print(" | ".join(line), flush=True)
ops.range_check(sync=True)
