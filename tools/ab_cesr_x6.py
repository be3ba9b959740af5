#!/usr/bin/env python
"""Timing of the exact-operand CESR nets of one library build: `python tools/ab_cesr_x6.py <tag> <lib.so>` (A/B and ablation builds made
by tools/build_variant.sh; -DQX_ABL_* builds give wrong results on purpose).  shadow_net: 8192 points x 128 labels = 2^20 rows."""
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
c = synth.synth_cesr_nets(0)
g = torch.Generator().manual_seed(1)


def timed(fn, reps=5):
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


npts, nl = 8192, 128
n = npts * nl
p = ((torch.rand(npts, 3, generator=g) - 0.5) * 0.6).to(dev)
sh = packing.pack_softplus512_x6({"net." + k: v for k, v in c["shadow_net"].items()}, "net.", 191, dev)
no = packing.pack_softplus512_x6({"net." + k: v for k, v in c["normal_net"].items()}, "net.", 63, dev)
if "fp8" in sys.argv[3:]:      # a library built with -DQX_FP8=1 (experiment): the K = 512 layers in the bf8 layout
    sh, no = packing.repack_softplus512_x6_fp8(sh, dev, 191), packing.repack_softplus512_x6_fp8(no, dev, 63)
mac_sh, mac_no = 1836032, 1836544
t = timed(lambda: ops.cesr_net_x6_points(p, n, 2, sh, nl))
out = [f"shadow {t:.3f} ms ({2 * mac_sh * n / t / 1e9 / 416.7:.3f})"]
pn = ((torch.rand(n, 3, generator=g) - 0.5) * 0.6).to(dev)
t = timed(lambda: ops.cesr_net_x6_points(pn, n, 0, no, 1))
out.append(f"normal {t:.3f} ms ({2 * mac_no * n / t / 1e9 / 416.7:.3f})")
y = ops.cesr_net_x6_points(p, n, 2, sh, nl)
out.append(f"sum {float(y.double().sum()):.9e}")
print(tag, " | ".join(out), flush=True)

if "f16" in sys.argv[3:]:
    shf = packing.pack_softplus512_f16({"net." + k: v for k, v in c["shadow_net"].items()}, "net.", 191, dev)
    nof = packing.pack_softplus512_f16({"net." + k: v for k, v in c["normal_net"].items()}, "net.", 63, dev)
    t = timed(lambda: ops.cesr_net_f16_points(p, n, 2, shf, nl))
    o2 = [f"f16 shadow {t:.3f} ms ({2 * mac_sh * n / t / 1e9 / 2500:.3f} of 2500)"]
    t = timed(lambda: ops.cesr_net_f16_points(pn, n, 0, nof, 1))
    o2.append(f"f16 normal {t:.3f} ms ({2 * mac_no * n / t / 1e9 / 2500:.3f})")
    yf = ops.cesr_net_f16_points(p, n, 2, shf, nl)
    e = (yf - y).abs() / (y.abs() + y.abs().mean())
    o2.append(f"shadow f16 vs exact: median {float(e.median()):.2e} p99 {float(e.flatten().kthvalue(int(0.99 * e.numel())).values):.2e} max {float(e.max()):.2e} finite {bool(torch.isfinite(yf).all())}")
    yn, ynf = ops.cesr_net_x6_points(pn[:65536], 65536, 0, no, 1), ops.cesr_net_f16_points(pn[:65536], 65536, 0, nof, 1)
    e = (ynf - yn).abs() / (yn.abs() + yn.abs().mean())
    o2.append(f"normal f16 vs exact: median {float(e.median()):.2e} max {float(e.max()):.2e}")
    # ragged sizes: rows not a multiple of the round
    for m in (1, 47, 64 * 3 + 5, 1000):
        a, b = ops.cesr_net_f16_points(pn[:m], m, 0, nof, 1), ynf[:m]
        assert torch.equal(a, b), m
    ops.range_check(sync=True)
    print(tag, " | ".join(o2), flush=True)
