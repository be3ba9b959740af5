#!/usr/bin/env python
"""EXPERIMENT: the stand-alone visibility MLP kernel k_vis_x6 with h.xl and l.xh as bf8 products on
v_mfma_f32_16x16x128_f8f6f4 (a library built by `bash tools/build_variant.sh vfp8 vis_x6.hip -DVX_FP8=1 ...`), against the shipped
exact-operand kernel and a float64 evaluation of the network (plain torch on the CPU, written out below).
`python tools/ab_vis_fp8.py <lib.so> [fp8]`: time per 2^20 rows, error of the logits against float64 (median / 99th percentile / maximum of
|a - b| / (|b| + mean|b|)) for this library's kernel and for PyTorch-CPU fp32."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib, fp8 = sys.argv[1], "fp8" in sys.argv[2:]
from robir_amd import _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(ROOT, lib)
from robir_amd import ops, packing, synth  # noqa: E402


def vis_logits(sd, p, d):
    """[PE10(p) | PE10(d)] -> 256 x 4 ReLU -> 2 (model/implicit_differentiable_renderer.py:250-258) in the dtype of its arguments"""
    def pe(v):
        return torch.cat([v] + [f(v * 2.0 ** k) for k in range(10) for f in (torch.sin, torch.cos)], -1)
    h = torch.cat([pe(p), pe(d)], -1)
    for i in range(5):
        h = F.linear(h, sd["visibility_network.vis_layer.%d.weight" % (2 * i)], sd["visibility_network.vis_layer.%d.bias" % (2 * i)])
        if i < 4:
            h = torch.relu(h)
    return h


dev = torch.device("cuda:0")
sd_np = synth.synth_state_dict(0)
blob = packing.pack_vis_x6(sd_np, dev)
if fp8:
    blob = packing.repack_vis_x6_fp8(blob, dev)
g = torch.Generator().manual_seed(5)
n = 8192
x = (torch.rand(n, 3, generator=g) - 0.5) * 0.8
view = F.normalize(torch.randn(n, 3, generator=g), dim=-1)
sd64 = {k: torch.from_numpy(v).double() for k, v in sd_np.items()}
sd32 = {k: torch.from_numpy(v).float() for k, v in sd_np.items()}
r64 = vis_logits(sd64, x.double(), view.double())
o32 = vis_logits(sd32, x, view).double()
got = ops.vis_x6_points(x.to(dev), view.to(dev), blob, 1).cpu().double()
scale = r64.abs() + r64.abs().mean()


def err(a):
    e = ((a - r64).abs() / scale).flatten()
    return "%.2e / %.2e / %.2e" % (float(e.median()), float(e.kthvalue(int(0.99 * e.numel())).values), float(e.max()))


print("%s%s: kernel vs float64 %s | PyTorch-CPU fp32 vs float64 %s" % (lib, " (bf8 low-weight products)" if fp8 else "", err(got), err(o32)), flush=True)
m = 1 << 20
xb = ((torch.rand(m, 3, generator=g) - 0.5) * 0.8).to(dev)
vb = F.normalize(torch.randn(m, 3, generator=g), dim=-1).to(dev)
best = 1e9
for _ in range(6):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    ops.vis_x6_points(xb, vb, blob, 1)
    b.record()
    torch.cuda.synchronize()
    best = min(best, a.elapsed_time(b))
mac = 126 * 256 + 3 * 256 * 256 + 256 * 2
print("   %.3f ms per 2^20 rows = %.3f of 417 TFLOP/s (algorithmic MACs)" % (best, 2.0 * mac * m / best / 1e9 / 416.7), flush=True)
