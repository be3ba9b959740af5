#!/usr/bin/env python
"""The runners' per-chunk evaluation loop (training/train_pbr.py:248-281) on an 800x800 view with deferred chunk forwards
(robir_amd/deferred.py): host time of the recording loop, time until the numbers are on the host, for several pass sizes
(model.deferred_chunks), next to the immediate per-chunk loop and one render_chunks pass.
`python tools/prof_deferred.py [limit ...]`"""
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from robir_amd import renderer  # noqa: E402
from test_deferred_gpu import _view, split_input, plot_loop  # noqa: E402
from test_runner_hooks_gpu import make_pbr_runner_hook  # noqa: E402

dev = torch.device("cuda:0")


def main():
    limits = [int(a) for a in sys.argv[1:]] or [1024, 256, 128, 64, 32]
    with torch.no_grad():
        model = renderer.build_synthetic_model(dev)
        sys.path.insert(0, os.path.join(ROOT, "overlay"))
        model.get_sg_render = make_pbr_runner_hook(types.SimpleNamespace(model=model, train_spec=True, no_normal=False,
                                                                         is_training=False))
        mi, total = _view(dev, 800, 800)
        from robir_amd import deferred
        ramps = [int(a) for a in os.environ.get("PROF_RAMPS", str(deferred.RAMP_START)).split(",")]
        for limit, ramp in [(l, r) for l in limits for r in ramps]:
            model.deferred_chunks = limit
            deferred.RAMP_START = ramp
            model.__dict__.pop("_defer_ramp", None)
            best = (1e9, 0, 0)
            for _ in range(3):
                split = split_input(mi, total)
                torch.cuda.synchronize()
                t0 = time.time()
                merged = plot_loop(model, split, total)
                t1 = time.time()
                got = {k: v.cpu() for k, v in merged.items()}
                torch.cuda.synchronize()
                t2 = time.time()
                best = min(best, (t2 - t0, t1 - t0, t2 - t1))
            print(f"deferred_chunks={limit:5d} ramp start {ramp:3d} ({len(deferred.pass_sizes(625, limit, ramp))} passes): {best[0]:.3f} s = {total / best[0]:.3g} rays/s  (recording loop {best[1]:.3f} s, "
                  f"reading the merged image {best[2]:.3f} s)")
        model.deferred_chunks = 0
        split = split_input(mi, total)[280:344]
        plot_loop(model, split[:4], 4096)["pred_rgb"].cpu()
        torch.cuda.synchronize()
        t0 = time.time()
        plot_loop(model, split, 64 * 1024)["pred_rgb"].cpu()
        t = time.time() - t0
        print(f"immediate forward(), 64 central chunks: {t / 64 * 1e3:.2f} ms per chunk = {1024 * 64 / t:.3g} rays/s")
        uv, hdr = mi["uv"][0], model.gamma.hdr_shift.as_input().expand(total, 1).contiguous()
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.time()
            model.render_chunks(uv, mi["pose"][0], mi["intrinsics"][0], hdr)["sg_rgb"].cpu()
            t = time.time() - t0
        print(f"one render_chunks pass over the 625 chunks: {t:.3f} s = {total / t:.3g} rays/s")


if __name__ == "__main__":
    main()
