#!/usr/bin/env python
"""Soak test of the light-visibility kernels: many launches on the central chunks of the bench view, every result compared bit
for bit with the first (k_dvis_v2 -- the bench kernel -- keeps weight copies and row loads in flight under counted waits; a row
consumed before it landed would show up as a run-dependent difference).  `python tools/soak_dvis.py [chunks] [repetitions]
[precision]`  (precision: f16x3-v2 | f16x3-v3 | f16x3-v4)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robir_amd import ops, renderer, synth  # noqa: E402

dev = torch.device("cuda:0")


def main():
    n_chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    prec = sys.argv[3] if len(sys.argv) > 3 else "f16x3-v2"
    with torch.no_grad():
        model = renderer.build_synthetic_model(dev)
        uv, pose, K = synth.synth_camera(800, 800)
        first = (625 - n_chunks) // 2
        dirs = ops.camera_rays(pose, K, torch.from_numpy(uv[first * 1024:(first + n_chunks) * 1024]).to(dev))
        cam = (torch.from_numpy(pose[:3, 3]).to(dev) * 2.0).reshape(1, 3).contiguous()
        _, hit, dist = model.ray_tracer.sdf_octree.cast_chunks(cam, dirs, chunk=1024)
        pts = ops.points_along(cam.expand(dirs.shape[0], 3).contiguous(), dirs, dist)
        idx = hit.nonzero()[:, 0]
        hp = pts[idx].contiguous()
        cid = (idx // 1024).to(torch.int32).contiguous()
        nrm = ops.normalize3(model.implicit_network.gradient(hp)[:, 0, :].contiguous(), 1e-4, 1)
        lgt = model.envmap_material_network.lgtSGs.detach()
        u = torch.rand(2, n_chunks, 128, 32, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
        sp = model.visibility_network.packed_split()
        A = ops.linear_64_256(ops.feat_pe10(hp), sp["point"])
        d_, w_, ws_ = ops.dvis_dirs(lgt, u[0], u[1], 1.0)
        Bd = ops.linear_64_256(ops.feat_pe10(d_), sp["dir"])
        run = lambda: ops.dvis_fused(nrm, cid, A, Bd, d_, w_, ws_, sp, 128, 32, False, None, precision=prec)
        ref = run()
        bad = 0
        t0 = time.time()
        for r in range(reps):
            o = run()
            if not bool(torch.equal(o, ref)):
                bad += 1
                d = (o - ref).abs().max(1)[0]
                print(f"run {r}: {int((d > 0).sum())} of {o.shape[0]} points differ, max {float(d.max()):.3g}")
        ops.range_check(sync=True)
        print(f"{prec}: {reps} launches on {hp.shape[0]} points ({n_chunks} chunks) in {time.time() - t0:.1f} s: {bad} differ from the first")
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
