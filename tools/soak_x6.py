#!/usr/bin/env python
"""Soak of the exact-operand chunk-stream kernels: every kernel run `reps` times on fixed inputs, every output compared bit for bit with
the first run (a missed counted wait or a ring-slot race shows up as a changed bit).  `python tools/soak_x6.py [reps] [points]`"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robir_amd import ops, packing, synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300001
dev = torch.device("cuda:0")
w = synth.synth_state_dict(0, variance=0.3)
c = synth.synth_cesr_nets(0)
g = torch.Generator().manual_seed(5)
x = ((torch.rand(n, 3, generator=g) - 0.5) * 1.2).to(dev)
v = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
feat = torch.randn(n, 257, generator=g).to(dev)
hdr = torch.rand(n, 1, generator=g).to(dev)
x6f, back6 = packing.pack_sdf_x6(w, dev, full=True), packing.pack_sdf_back_x6(w, dev) + (packing.pack_sdf_back_x6(w, dev, two_tile=True)[0],)
col6, vis6, ill6 = packing.pack_color_x6(w, dev), packing.pack_vis_x6(w, dev), packing.pack_illum_x6(w, dev)
sh6 = packing.pack_softplus512_x6({"net." + k: t for k, t in c["shadow_net"].items()}, "net.", 191, dev)
no6 = packing.pack_softplus512_x6({"net." + k: t for k, t in c["normal_net"].items()}, "net.", 63, dev)
npt = max(1, n // 128)
cases = {
    "sdf value+grad": lambda: torch.cat([t.reshape(-1) for t in ops.sdf_value_grad_x6(x, n, x6f, back6, 2.0, 0.5)]),
    "colour": lambda: ops.color_x6_points(x, v, v, feat[:, 1:], col6),
    "visibility": lambda: ops.vis_x6_points(x, v, vis6, 1),
    "illum decoder": lambda: ops.wide_x6_points(x, hdr, ill6, False),
    "shadow_net": lambda: ops.cesr_net_x6_points(x[:npt], npt * 128, 2, sh6, 128),
    "normal_net": lambda: ops.cesr_net_x6_points(x, n, 0, no6),
}
bad = 0
for name, fn in cases.items():
    ref = fn().clone()
    t0 = time.time()
    diff = 0
    for _ in range(reps):
        if not torch.equal(fn(), ref):
            diff += 1
    torch.cuda.synchronize()
    print(f"{name:16s} {reps} runs of {n} rows: {diff} differ from the first ({time.time() - t0:.1f} s)", flush=True)
    bad += diff
ops.range_check(sync=True)
print("SOAK", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
