#!/usr/bin/env python
"""Soak test of the reverse-mode SDF gradient (rb_sdf_value_grad): many repetitions on 2^20 points, every result compared bit
for bit with the first (the value pass's counted waits and the backward pass's content-detected sigmoid rows are the parts of
this library whose failure mode would be a rare, run-dependent difference).  `python tools/soak_sdf_grad.py [repetitions]`"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robir_amd import ops, renderer  # noqa: E402

dev = torch.device("cuda:0")


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    with torch.no_grad():
        m = renderer.build_synthetic_model(dev, build_octrees=False)
        net = m.implicit_network.neus_model.sdf_network
        torch.manual_seed(0)
        M = 1 << 20
        x = (torch.rand(M, 3, device=dev) * 2 - 1) * 0.9
        ops.SDF_GRAD = "reverse"
        o0, g0 = net.eval_points(x, 2.0, 0.5, full=True, grad=True)
        ho, hg = o0.sum().double(), g0.sum().double()
        bad = 0
        t0 = time.time()
        for r in range(reps):
            o, g = net.eval_points(x, 2.0, 0.5, full=True, grad=True)
            if r % 50 == 49 or r == reps - 1:          # full comparison every 50th run, checksums in between
                same = bool(torch.equal(o, o0)) and bool(torch.equal(g, g0))
            else:
                same = bool((o.sum().double() == ho) & (g.sum().double() == hg))
            if not same:
                bad += 1
                d = (g - g0).abs().max(1)[0]
                print(f"run {r}: {int((d > 0).sum())} rows differ, max {float(d.max()):.3g}")
        ops.range_check(sync=True)
        dt = time.time() - t0
        print(f"{reps} runs of 2^20 points in {dt:.1f} s: {bad} runs differ from the first "
              f"({reps * (M // 128) * 117 * 8:.3g} sigmoid-row arrivals, {reps * (M // 128) * 259:.3g} counted waits per wave)")
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
