#!/usr/bin/env python
"""Timing of the chunk-stream colour / 512-wide kernels of one library build: `python tools/ab_wide.py <tag> <lib.so>` (A/B and
ablation builds made by tools/build_variant.sh)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag, lib = sys.argv[1], sys.argv[2]
which = sys.argv[3] if len(sys.argv) > 3 else "color,decoder,encoder,normal,shadow"
from robir_amd import _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(ROOT, lib)
from robir_amd import ops, packing, synth  # noqa: E402

dev = torch.device("cuda:0")
w = synth.synth_state_dict(0, variance=0.3)
c = synth.synth_cesr_nets(0)
s = packing.H3_SCALE_LOG2
g = torch.Generator().manual_seed(1)


def timed(fn, reps=7):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


n = 1 << 20
p = ((torch.rand(n, 3, generator=g) - 0.5) * 0.6).to(dev)
hdr = torch.rand(n, 1, generator=g).to(dev)
out = []
if "color" in which:
    blob = packing.pack_color_h3(w, dev)
    v = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
    nr = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
    feat = torch.randn(n, 257, generator=g).to(dev)
    t = timed(lambda: ops.color_mlp_h3_points(p, v, nr, feat[:, 1:], blob, s, ring=True))
    out.append(f"color {t:.3f} ms ({6 * (320 * 256 + 3 * 256 * 256 + 256 * 16) * n / t / 1e9 / 2500:.3f})")
if "decoder" in which:
    ill16 = packing.pack_illum_h3(w, dev)
    t = timed(lambda: ops.wide_mlp_points(p, hdr, ill16, False, s, ring=True))
    out.append(f"decoder {t:.3f} ms ({6 * (64 * 512 + 3 * 512 * 512 + 512 * 144) * n / t / 1e9 / 2500:.3f})")
if "encoder" in which:
    enc16 = packing.pack_sparse_ae_encoder_h3(w, "envmap_material_network.spec_brdf_encoder_layer", dev)
    t = timed(lambda: ops.wide_mlp_points(p, None, enc16, True, s, ring=True))
    out.append(f"encoder {t:.3f} ms ({6 * (64 * 512 + 3 * 512 * 512 + 512 * 32) * n / t / 1e9 / 2500:.3f})")
if "normal" in which:
    no16 = packing.pack_softplus512_h3({"net." + k: v for k, v in c["normal_net"].items()}, "net.", 63, dev)
    t = timed(lambda: ops.cesr_net_points(p, n, 0, no16, 1, s, ring=True))
    out.append(f"normal {t:.3f} ms ({6 * (64 * 512 + 2 * 512 * 512 + 512 * 464 + 544 * 512 + 3 * 512 * 512 + 512 * 16) * n / t / 1e9 / 2500:.3f})")
if "shadow" in which:
    sh16 = packing.pack_softplus512_h3({"net." + k: v for k, v in c["shadow_net"].items()}, "net.", 191, dev)
    m = 4096 * 128
    t = timed(lambda: ops.cesr_net_points(p[:4096], m, 2, sh16, 128, s, ring=True))
    out.append(f"shadow {t:.3f} ms ({6 * (192 * 512 + 2 * 512 * 512 + 512 * 336 + 544 * 512 + 3 * 512 * 512 + 512 * 16) * m / t / 1e9 / 2500:.3f})")
print(f"{tag:14s} " + "  ".join(out), flush=True)
