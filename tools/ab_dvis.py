#!/usr/bin/env python
"""A/B of two builds of the visibility kernel: `python tools/ab_dvis.py <tag> <lib.so> [n_chunks]` runs the default kernel of
that library on a fixed workload, prints its timing and stores the output under gpurun_out/ab_<tag>.pt;
`python tools/ab_dvis.py compare a b` compares two stored outputs bit by bit."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "gpurun_out")
if sys.argv[1] == "compare":
    a, b = (torch.load(os.path.join(OUT, "ab_%s.pt" % t)) for t in sys.argv[2:4])
    for k in a:
        print(k, "bitwise equal:", bool(torch.equal(a[k], b[k])), "max diff %.3g" % float((a[k] - b[k]).abs().max()))
    sys.exit(0)
tag, lib = sys.argv[1], sys.argv[2]
n_chunks = int(sys.argv[3]) if len(sys.argv) > 3 else 32
from robir_amd import _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(ROOT, lib)
from robir_amd import ops, renderer, synth  # noqa: E402

dev = torch.device("cuda:0")
model = renderer.build_synthetic_model(dev)
uv, pose, K = synth.synth_camera(800, 800)
first = 300 * 800 // 1024
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
lgt = model.envmap_material_network.lgtSGs.detach()
g = torch.Generator(device=dev).manual_seed(1)
u = torch.rand(2, n_chunks, 128, 32, device=dev, generator=g)
sp = model.visibility_network.packed_split()
A = ops.linear_64_256(ops.feat_pe10(hp), sp["point"])
d_, w_, ws_ = ops.dvis_dirs(lgt, u[0], u[1], 1.0)
Bd = ops.linear_64_256(ops.feat_pe10(d_), sp["dir"])
out, times = {}, []
for rep in range(5):
    for argmax in (False, True):
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        o = ops.dvis_fused(nrm, cid, A, Bd, d_, w_, ws_, sp, 128, 32, argmax, None, precision=os.environ.get("AB_PRECISION", "f16x6"))
        e.record()
        torch.cuda.synchronize()
        if not argmax:
            times.append(s.elapsed_time(e))
        out["argmax" if argmax else "softmax"] = o[::16].cpu()       # every 16th point: the files travel back from the GPU box
os.makedirs(OUT, exist_ok=True)
torch.save(out, os.path.join(OUT, "ab_%s.pt" % tag))
print(tag, " ".join(f"{t:.2f}" for t in times), "ms (min %.2f)" % min(times), flush=True)
