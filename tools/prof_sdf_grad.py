#!/usr/bin/env python
"""SDF value + gradient (all 257 outputs, d sdf/dx) of the NeuS net on random points: the forward-mode rows of rb_sdf_mlp_ring
mode 3 next to the reverse-mode pass rb_sdf_value_grad (ops.SDF_GRAD), with their agreement.
`python tools/prof_sdf_grad.py [points]` (default 2^20); under rocprofv3 --kernel-trace this is the workload of
profiles/r02_sdf_grad_kernel_stats.md."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robir_amd import ops, renderer  # noqa: E402

dev = torch.device("cuda:0")


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    with torch.no_grad():
        m = renderer.build_synthetic_model(dev, build_octrees=False)
        net = m.implicit_network.neus_model.sdf_network
        torch.manual_seed(0)
        x = (torch.rand(M, 3, device=dev) * 2 - 1) * 0.9
        res = {}
        for mode in ("forward", "reverse"):
            ops.SDF_GRAD = mode
            res[mode] = net.eval_points(x, 2.0, 0.5, full=True, grad=True)
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(3):
                net.eval_points(x, 2.0, 0.5, full=True, grad=True)
            torch.cuda.synchronize()
            t = (time.time() - t0) / 3
            print(f"{mode:8s} mode: {t * 1e3:.2f} ms per {M} points = {M / t:.3g} points/s")
        d = (res["forward"][1] - res["reverse"][1]).abs().max(1)[0]
        print("values identical:", bool(torch.equal(res["forward"][0], res["reverse"][0])), " gradient: max |difference|",
              float(d.max()), "median", float(d.median()), "(|gradient| ~ 1)")
        ops.range_check(sync=True)


if __name__ == "__main__":
    main()
