#!/usr/bin/env python
"""Cost of the per-rank views of bench.py on one GPU: hit fraction, visibility evaluations and time per view."""
import contextlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from robir_amd import renderer, synth  # noqa: E402

dev = torch.device("cuda:0")
with contextlib.redirect_stdout(sys.stderr):
    model = renderer.build_synthetic_model(dev, seed=0, variance=0.3)
uv, _, K = synth.synth_camera(bench.H, bench.W)
uv_d, K_d = torch.from_numpy(uv).to(dev), torch.from_numpy(K).to(dev)
hdr = torch.full((bench.H * bench.W, 1), 0.5, device=dev)
for r in range(8):
    pose_d = torch.from_numpy(bench.view_pose(r)).to(dev)
    stats = {}
    bench.render_image(model, uv_d, pose_d, K_d, hdr, 625, stats)
    torch.cuda.synchronize()
    stats.clear()
    t0 = time.perf_counter()
    out = bench.render_image(model, uv_d, pose_d, K_d, hdr, 625, stats)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("view %d: hit %.4f  evals %.1f M  %.1f ms" % (r, float(out[:, 16].mean()), int(stats["diffuse_vis_evals"]) / 1e6, dt * 1e3), flush=True)
