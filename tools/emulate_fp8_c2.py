#!/usr/bin/env python
"""What would it cost in ACCURACY to run the two cheapest of the six exact-operand products on the fp8 MFMA?  (CPU only.)

The exact-operand kernels carry an fp32 value as v = h + m 2^-11 + l 2^-22 (three f16 pieces) and form w x from six products in three
weight classes: c0 = h.xh, c1 = h.xm + m.xh, c2 = h.xl + m.xm + l.xh.  The two outer products of c2 have one operand with at most three
significant bits (an l piece); this script rounds BOTH operands of those two products to bf8 (e5m2: the top byte of an f16) and
measures, on the light-visibility MLP (126 -> 256 -> 256 -> 256 -> 2, ReLU; synthetic weights, 4096 random encoded rows), the distance of the
logits from a float64 evaluation for
    fp32     numpy float32 matmuls (the reference's arithmetic)
    x6       all six products exact, every class accumulated in fp32 (what the kernels do)
    x6-fp8   the same with h.xl and l.xh formed from bf8 operands (register- and stream-neutral: xl and the l weights are only ever used there,
             so their f16 copies go; bf8 copies of xh and of the h weights take their place -- 4 f16 + 2 fp8 MFMAs per k-block instead of 6 f16)
    x6-fp8t  x6-fp8 with the ACTIVATIONS' bf8 copies taken by truncation (the top bytes of the f16 pieces: one v_perm_b32 per four values)
    x6-fp8c2 all three c2 products from bf8 operands
python tools/emulate_fp8_c2.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robir_amd import synth  # noqa: E402


def f16_rtz(v):
    h = v.astype(np.float16)
    over = np.abs(h.astype(np.float64)) > np.abs(v)
    h[over] = np.nextafter(h[over], np.float16(0))
    return h.astype(np.float64)


def split3(v):
    """v (float64 holding fp32 values) -> h, m, l as float64 with v = h + m 2^-11 + l 2^-22 up to the l rounding (sx_split_pair)"""
    h = f16_rtz(v)
    r1 = (v - h) * 2048.0
    m = f16_rtz(r1)
    r2 = (r1 - m) * 2048.0
    l = r2.astype(np.float16).astype(np.float64)
    return h, m, l


def bf8(v):
    """round to e5m2 = the top byte of the f16 pattern (round to nearest even)"""
    b = v.astype(np.float16).view(np.uint16).astype(np.uint32)
    b = (b + 0x7F + ((b >> 8) & 1)) & 0xFF00
    return b.astype(np.uint16).view(np.float16).astype(np.float64)


def bf8t(v):
    """e5m2 by TRUNCATION: the top byte of the f16 pattern as it stands (what one v_perm_b32 per four values extracts)"""
    b = v.astype(np.float16).view(np.uint16) & np.uint16(0xFF00)
    return b.view(np.float16).astype(np.float64)


def acc32(a, b):
    """sum_k a[r,k] b[n,k] accumulated in fp32 k-block (32) by k-block, products exact (as the MFMA forms them)"""
    out = np.zeros((a.shape[0], b.shape[0]), dtype=np.float32)
    for k0 in range(0, a.shape[1], 32):
        out = (out.astype(np.float64) + a[:, k0:k0 + 32] @ b[:, k0:k0 + 32].T).astype(np.float32)
    return out


def layer(x, W, bias, mode):
    if mode == "f64":
        return x @ W.T + bias
    if mode == "fp32":
        return (x.astype(np.float32) @ W.astype(np.float32).T + bias.astype(np.float32)).astype(np.float64)
    xh, xm, xl = split3(x)
    wh, wm, wl = split3(W)
    c0 = (acc32(xh, wh).astype(np.float64) + bias).astype(np.float32)          # the bias starts the c0 accumulator
    c1 = (acc32(xm, wh).astype(np.float64) + acc32(xh, wm)).astype(np.float32)
    if mode == "x6":
        c2 = acc32(xl, wh).astype(np.float64) + acc32(xm, wm) + acc32(xh, wl)
    elif mode == "x6-fp8":
        c2 = acc32(bf8(xl), bf8(wh)).astype(np.float64) + acc32(xm, wm) + acc32(bf8(xh), bf8(wl))
    elif mode == "x6-fp8t":      # activations truncated (v_perm_b32 of the f16 pieces), weights rounded (packed on the host)
        c2 = acc32(bf8t(xl), bf8(wh)).astype(np.float64) + acc32(xm, wm) + acc32(bf8t(xh), bf8(wl))
    else:
        c2 = acc32(bf8(xl), bf8(wh)).astype(np.float64) + acc32(bf8(xm), bf8(wm)) + acc32(bf8(xh), bf8(wl))
    c2 = c2.astype(np.float32)
    C = np.float32(1.0 / 2048.0)
    y = np.float32(c2 * C + c1)            # combine: fma(fma(c2, C, c1), C, c0)
    return np.float32(y * C + c0).astype(np.float64)


def run(x, layers, mode):
    for i, (W, b) in enumerate(layers):
        x = layer(x, W, b, mode)
        if i + 1 < len(layers):
            x = np.maximum(x, 0.0)
        if mode != "f64":
            x = x.astype(np.float32).astype(np.float64)
    return x


def main():
    sd = synth.synth_state_dict(0)
    layers = [(sd["visibility_network.vis_layer.%d.weight" % i].astype(np.float64), sd["visibility_network.vis_layer.%d.bias" % i].astype(np.float64))
              for i in (0, 2, 4, 6, 8)]
    rng = np.random.default_rng(3)
    n = 4096
    p, d = (rng.random((n, 3)) - 0.5) * 0.8, rng.standard_normal((n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    enc = lambda v, L: np.concatenate([v] + [f(v * 2.0 ** k) for k in range(L) for f in (np.sin, np.cos)], axis=1)  # noqa: E731
    x = np.concatenate([enc(p, 10), enc(d, 10)], axis=1).astype(np.float32).astype(np.float64)      # 63 + 63 = 126 columns
    ref = run(x, layers, "f64")
    scale = np.abs(ref) + np.abs(ref).mean()
    print("light-visibility MLP logits, %d rows: error against float64, |a - b| / (|b| + mean|b|): median / 99th percentile / maximum" % n)
    for mode in ("fp32", "x6", "x6-fp8", "x6-fp8t", "x6-fp8c2"):
        e = (np.abs(run(x, layers, mode) - ref) / scale).ravel()
        print("  %-9s %.2e / %.2e / %.2e" % (mode, np.median(e), np.quantile(e, 0.99), e.max()))


if __name__ == "__main__":
    main()
