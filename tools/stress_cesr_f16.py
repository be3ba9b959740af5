"""Stress of the f16 CESR kernel's ragged launches (round 6: a launch whose last tiles lie beyond M faulted intermittently before the rows
beyond M were made to compute on the last valid row's inputs): many launches per row count, every one bit-equal to the first and to the
leading rows of a full launch.   python tools/stress_cesr_f16.py [iters]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robir_amd import ops, packing, synth  # noqa: E402

dev = torch.device("cuda:0")
c = synth.synth_cesr_nets(0)
g = np.random.Generator(np.random.PCG64(11))
pts = torch.from_numpy((g.standard_normal((203, 3)) * 0.25).astype(np.float32)).to(dev)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for which, kind, nl, k_in in (("shadow_net", 2, 128, 191), ("normal_net", 0, 1, 63)):
    blob = packing.pack_softplus512_f16({"net." + k: v for k, v in c[which].items()}, "net.", k_in, dev)
    exact = ops.cesr_net_x6_points(pts, pts.shape[0] * nl, kind, packing.pack_softplus512_x6({"net." + k: v for k, v in c[which].items()}, "net.", k_in, dev), nl)
    full = ops.cesr_net_f16_points(pts, pts.shape[0] * nl, kind, blob, nl)
    e = ((full - exact).abs() / (exact.abs() + exact.abs().mean())).flatten()
    print(which, "full launch vs exact: median %.2e max %.2e" % (float(e.median()), float(e.max())), flush=True)
    for M in ((1, 48, 100, 128, 129, 150, 176, 191, 192, 193, 385) if nl == 128 else (1, 47, 130, 191, 193)):
        npts = (M + nl - 1) // nl
        bad = 0
        for _ in range(iters):
            a = ops.cesr_net_f16_points(pts[:npts].contiguous(), M, kind, blob, nl)
            torch.cuda.synchronize()
            bad += int(not torch.equal(a, full[:M]))
        print(which, "M", M, "launches", iters, "not equal to the full launch's rows:", bad, flush=True)
