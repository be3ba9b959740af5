#!/usr/bin/env python
"""The two forms of the exact-operand light-visibility kernel (one workgroup per point | persistent grid over the global tile list) by
launch size and samples per lobe: `python tools/ab_dvis_forms.py` -> ms per launch and pairs per microsecond for
(n_chunks, nsamp) in {8, 32, 128} x {32, 8}.  Where the tile-list form wins decides ops.DVIS_STREAM_MAX_POINTS / the auto rule."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robir_amd import ops, renderer, synth  # noqa: E402

dev = torch.device("cuda:0")
model = renderer.build_synthetic_model(dev)
uv, pose, K = synth.synth_camera(800, 800)
lgt = model.envmap_material_network.lgtSGs.detach()
sp = model.visibility_network.packed_split()
for n_chunks in (8, 32, 128):
    first = 300 * 800 // 1024 - n_chunks // 2
    sl = slice(first * 1024, (first + n_chunks) * 1024)
    uv_d = torch.from_numpy(uv[sl]).to(dev)
    dirs = ops.camera_rays(pose, K, uv_d)
    cam = torch.from_numpy(pose[:3, 3]).to(dev).reshape(1, 3)
    _, hit, dist = model.ray_tracer.sdf_octree.cast_chunks(cam, dirs, chunk=1024)
    pts = ops.points_along(cam.expand(dirs.shape[0], 3).contiguous(), dirs, dist)
    idx = hit.nonzero()[:, 0]
    hp = pts[idx].contiguous()
    cid = (idx // 1024).to(torch.int32).contiguous()
    nrm = ops.normalize3(model.implicit_network.gradient(hp)[:, 0, :].contiguous(), 1e-4, 1)
    A = ops.linear_pe10_256(hp, sp["point"])
    for nsamp in (32, 8):
        g = torch.Generator(device=dev).manual_seed(1)
        u = torch.rand(2, n_chunks, 128, nsamp, device=dev, generator=g)
        d_, w_, ws_ = ops.dvis_dirs(lgt, u[0], u[1], 1.0)
        Bd = ops.linear_pe10_256(d_, sp["dir"])
        res = {}
        for form in ("f16x6-pt", "f16x6-stream"):
            cnt = torch.zeros(1, dtype=torch.int64, device=dev)
            best = 1e9
            for rep in range(4):
                cnt.zero_()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                o = ops.dvis_fused(nrm, cid, A, Bd, d_, w_, ws_, sp, 128, nsamp, False, cnt, precision=form)
                e.record()
                torch.cuda.synchronize()
                best = min(best, s.elapsed_time(e))
            res[form] = (best, int(cnt), o)
        same = torch.equal(res["f16x6-pt"][2], res["f16x6-stream"][2])
        pt, st = res["f16x6-pt"], res["f16x6-stream"]
        print(f"{n_chunks:4d} chunks ({hp.shape[0]:6d} points) nsamp {nsamp:2d}: per-point {pt[0]:8.3f} ms ({pt[1] / pt[0] / 1e3:7.1f} pairs/us) | "
              f"tile list {st[0]:8.3f} ms ({st[1] / st[0] / 1e3:7.1f} pairs/us)  ratio {st[0] / pt[0]:.3f}  bit-identical {same}", flush=True)
