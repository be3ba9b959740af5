#!/usr/bin/env python
"""Times the non-visibility MLP kernels of the PBR forward on a synthetic batch:
`python tools/prof_mlp.py [n_points]` -> ms per call of SDF value / SDF gradient (jvp) / illum / material encoders."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robir_amd import renderer  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 84000
dev = torch.device("cuda:0")
model = renderer.build_synthetic_model(dev, build_octrees=False)
g = torch.Generator(device=dev).manual_seed(0)
x = (torch.rand(n, 3, device=dev, generator=g) - 0.5) * 0.6
hdr = torch.full((n, 1), 0.5, device=dev)


def timeit(name, fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e))
    print(f"{name:28s} {best:8.3f} ms  ({n} points)")


imp = model.implicit_network
timeit("sdf_only", lambda: imp.sdf_only(x))
timeit("sdf full (257 outputs)", lambda: imp(x))
timeit("sdf gradient (jvp)", lambda: imp.gradient(x))
timeit("indirect illum net", lambda: model.indirect_illum_network(x, hdr))
timeit("material net", lambda: model.envmap_material_network(x, train_spec=True))
d = torch.nn.functional.normalize(torch.rand(n * 16, 3, device=dev, generator=g) - 0.5, dim=-1)
X = __import__("robir_amd").ops.feat_vis(x, d, rep=16)
from robir_amd import ops, packing  # noqa: E402
vn = model.visibility_network
timeit("vis MLP fp32 (16 dirs/pt)", lambda: ops.vis_mlp(X, vn.packed_full()))
timeit("vis MLP f16x3 (16 dirs/pt)", lambda: ops.vis_mlp_h3(X, vn.packed_full_h3(), packing.H3_SCALE_LOG2))
