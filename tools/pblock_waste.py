#!/usr/bin/env python
"""How many MFMA lanes a (16 points x 1 direction) tiling of the light-visibility pairs would waste on the synthetic view: tiles kept when
ANY of the 16 points faces the direction, against the per-point compaction of today's tile list (padding of the last tile only).
`python tools/pblock_waste.py [n_chunks]`"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robir_amd import ops, renderer, synth  # noqa: E402

n_chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
model = renderer.build_synthetic_model(dev)
uv, pose, K = synth.synth_camera(800, 800)
for first_row in (100, 300, 500):
    first = first_row * 800 // 1024
    sl = slice(first * 1024, (first + n_chunks) * 1024)
    uv_d = torch.from_numpy(uv[sl]).to(dev)
    dirs = ops.camera_rays(pose, K, uv_d)
    cam = torch.from_numpy(pose[:3, 3]).to(dev).reshape(1, 3)
    _, hit, dist = model.ray_tracer.sdf_octree.cast_chunks(cam, dirs, chunk=1024)
    pts = ops.points_along(cam.expand(dirs.shape[0], 3).contiguous(), dirs, dist)
    idx = hit.nonzero()[:, 0]
    hp = pts[idx].contiguous()
    cid = (idx // 1024)
    nrm = ops.normalize3(model.implicit_network.gradient(hp)[:, 0, :].contiguous(), 1e-4, 1)
    lgt = model.envmap_material_network.lgtSGs.detach()
    g = torch.Generator(device=dev).manual_seed(1)
    u = torch.rand(2, n_chunks, 128, 32, device=dev, generator=g)
    d_, w_, ws_ = ops.dvis_dirs(lgt, u[0], u[1], 1.0)
    d_ = d_.view(n_chunks, 4096, 3)
    useful = tiles_now = tiles_pb = 0
    for c in range(n_chunks):
        nc = nrm[cid == c]
        if nc.shape[0] == 0:
            continue
        front = (nc @ d_[c].T) > 1e-6                      # [points, 4096]
        useful += int(front.sum())
        tiles_now += int(((front.sum(1) + 15) // 16).sum())
        pad = (-nc.shape[0]) % 16
        fb = torch.cat([front, torch.zeros(pad, 4096, dtype=torch.bool, device=dev)]).view(-1, 16, 4096)
        tiles_pb += int(fb.any(1).sum())
    print(f"rows {first_row}: points {idx.numel()}  front pairs {useful}  tiles today {tiles_now} (lanes used {useful / tiles_now / 16:.3f})  "
          f"point-block tiles {tiles_pb} (lanes used {useful / tiles_pb / 16:.3f})  ratio {tiles_pb / tiles_now:.3f}")
