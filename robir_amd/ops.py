"""Thin Python bindings over the C-ABI (include/robir_hip.h): allocate outputs with torch, pass raw pointers,
sizes and the current HIP stream.  No arithmetic happens here."""
import ctypes
import os

import torch

from . import _lib
from ._lib import ptr, stream_ptr, call

c_long, c_int, c_float = ctypes.c_long, ctypes.c_int, ctypes.c_float


def _f32(t):
    assert t.dtype == torch.float32, t.dtype
    return t.contiguous()


RANGE_FAMILIES = ("light-visibility kernel (rb_dvis_fused*)", "visibility MLP (rb_vis_mlp_h3 / rb_vis_x6_points)",
                  "SDF MLP (rb_sdf_mlp_h3 / rb_sdf_x6_points / rb_sdf_value_grad*)", "colour MLP (rb_color_mlp_h3 / rb_color_x6_points)",
                  "512-wide nets (rb_wide_mlp_h3 / rb_wide_x6*)", "CESR nets (rb_cesr_net_h3 / rb_cesr_net_x6_points)")


def range_check(sync=False):
    """Raise RobirHipError if a split-precision or exact-operand kernel saw an activation beyond the f16 range of its leading operand piece
    (include/robir_hip.h: rb_range_check).  sync=False costs nothing and reports kernels that have completed; sync=True waits
    for the current stream first.  The legacy library keeps its own sentinel block: read too once it has been loaded."""
    mask = ctypes.c_uint(0)
    call("rb_range_check", c_int(1 if sync else 0), stream_ptr(), ctypes.byref(mask))
    bits = mask.value
    if _lib.legacy_loaded():
        rc = _lib.legacy().rb_range_check(c_int(0), stream_ptr(), ctypes.byref(mask))      # the stream has been waited for above
        if rc != 0:
            raise _lib.RobirHipError(f"rb_range_check (legacy library) failed ({rc})")
        bits |= mask.value
    if bits:
        fams = [RANGE_FAMILIES[i] for i in range(len(RANGE_FAMILIES)) if bits >> i & 1]
        raise _lib.RobirHipError(
            "f16-piece arithmetic (split precision f16x3 / exact operands f16x6) overflowed its activation range in: " + "; ".join(fams) + " -- the outputs of "
            "the call(s) since the last check are not fp32-accurate.  Select the exact f32-input MFMA kernels for this "
            "checkpoint: ROBIR_MLP_PRECISION=fp32 and ROBIR_VIS_PRECISION=fp32 (INTEGRATION.md, 'range limits')")


def feat_vis(p, d, rep=1):
    """p [M/rep,3] points, d [M,3] directions (rep consecutive directions per point)."""
    p, d = _f32(p), _f32(d)
    M = d.shape[0]
    assert p.shape[0] * rep == M
    X = torch.empty(M, 128, dtype=torch.float32, device=p.device)
    call("rb_feat_vis", ptr(p), ptr(d), c_long(M), c_int(rep), ptr(X), stream_ptr())
    return X


def feat_pe10(x, scale=1.0, extra=None, jvp=False):
    x = _f32(x)
    M = x.shape[0]
    X = torch.empty(M * (4 if jvp else 1), 64, dtype=torch.float32, device=x.device)
    e = _f32(extra).reshape(-1) if extra is not None else None
    call("rb_feat_pe10", ptr(x), c_long(M), c_float(scale), ptr(e), c_int(1 if jvp else 0), ptr(X), stream_ptr())
    return X


def feat_ipe(x, var=1e-5, noise=None, noise_scale=0.0):
    x = _f32(x)
    M = x.shape[0]
    X = torch.empty(M, 64, dtype=torch.float32, device=x.device)
    nz = _f32(noise) if noise is not None else None
    call("rb_feat_ipe", ptr(x), c_long(M), c_float(var), ptr(nz), c_float(noise_scale), ptr(X), stream_ptr())
    return X


def feat_color(x, view, normal, feat, x_scale=1.0, feat_scale=1.0):
    x, view, normal = _f32(x), _f32(view), _f32(normal)
    assert feat.dtype == torch.float32 and feat.stride(-1) == 1
    M = x.shape[0]
    X = torch.empty(M, 304, dtype=torch.float32, device=x.device)
    call("rb_feat_color", ptr(x), c_float(x_scale), ptr(view), ptr(normal), ctypes.c_void_p(feat.data_ptr()),
         c_long(feat.stride(0)), c_float(feat_scale), c_long(M), ptr(X), stream_ptr())
    return X


def color_mlp_h3_two(x, view, normal, feat, blob, scale_log2, x_scale=1.0, feat_scale=1.0):
    """feat_color + color_mlp_h3 without the assembled [M,304] rows: the 256 feature columns are read where they are."""
    x, view, normal = _f32(x), _f32(view), _f32(normal)
    assert feat.dtype == torch.float32 and feat.stride(-1) == 1
    M = x.shape[0]
    tail = torch.empty(M, 48, dtype=torch.float32, device=x.device)
    rgb = torch.empty(M, 3, dtype=torch.float32, device=x.device)
    call("rb_feat_color_tail", ptr(x), c_float(x_scale), ptr(view), ptr(normal), c_long(M), ptr(tail), stream_ptr())
    call("rb_color_mlp_h3_two", ctypes.c_void_p(feat.data_ptr()), c_long(feat.stride(0)), c_float(feat_scale), ptr(tail), c_long(M),
         ptr(blob), c_int(scale_log2), ptr(rgb), stream_ptr())
    return rgb


COLOR_RING_MIN_ROWS = 1           # the chunk-stream kernel wins at every size (one round: 0.040 vs 0.061 ms); 0 rows never launch


def color_mlp_h3_points(x, view, normal, feat, blob, scale_log2, x_scale=1.0, feat_scale=1.0, ring=None):
    """The colour net with its encoding fused: feature columns read in place, [x | PE4(view) | normal] encoded inside the kernel.
    ring (default: by batch size): the eight-wave chunk-stream kernel (csrc/color_ring8.hip) -- bit-identical rgb."""
    x, view, normal = _f32(x), _f32(view), _f32(normal)
    assert feat.dtype == torch.float32 and feat.stride(-1) == 1
    M = x.shape[0]
    rgb = torch.empty(M, 3, dtype=torch.float32, device=x.device)
    if ring is None:
        ring = M >= COLOR_RING_MIN_ROWS
    if ring:
        call("rb_color_ring_points", ctypes.c_void_p(feat.data_ptr()), c_long(feat.stride(0)), c_float(feat_scale), ptr(x),
             c_float(x_scale), ptr(view), ptr(normal), c_long(M), ptr(blob), c_int(scale_log2), ptr(rgb), c_int(0), stream_ptr())
        return rgb
    call("rb_color_mlp_h3_points", ctypes.c_void_p(feat.data_ptr()), c_long(feat.stride(0)), c_float(feat_scale), ptr(x), c_float(x_scale),
         ptr(view), ptr(normal), c_long(M), ptr(blob), c_int(scale_log2), ptr(rgb), stream_ptr())
    return rgb


def vis_mlp(X, blob):
    M = X.shape[0]
    Y = torch.empty(M, 2, dtype=torch.float32, device=X.device)
    call("rb_vis_mlp", ptr(X), c_long(M), ptr(blob), ptr(Y), stream_ptr())
    return Y


def vis_mlp_h3(X, blob, scale_log2):
    M = X.shape[0]
    Y = torch.empty(M, 2, dtype=torch.float32, device=X.device)
    call("rb_vis_mlp_h3", ptr(X), c_long(M), ptr(blob), c_int(scale_log2), ptr(Y), stream_ptr())
    return Y


def vis_mlp_points(p, d, blob, rep=1, scale_log2=None):
    """Visibility MLP straight from points p [M/rep,3] and directions d [M,3] (encoding fused): scale_log2 None = f32-input MFMA
    kernel (blob from pack_vis), else the split-precision kernel (blob from pack_vis_h3)."""
    p, d = _f32(p), _f32(d)
    M = d.shape[0]
    assert p.shape[0] * rep == M
    Y = torch.empty(M, 2, dtype=torch.float32, device=d.device)
    if scale_log2 is None:
        call("rb_vis_mlp_points", ptr(p), ptr(d), c_long(M), c_int(rep), ptr(blob), ptr(Y), stream_ptr())
    else:
        call("rb_vis_mlp_h3_points", ptr(p), ptr(d), c_long(M), c_int(rep), ptr(blob), c_int(scale_log2), ptr(Y), stream_ptr())
    return Y


def vis_x6_points(p, d, blob, rep=1):
    """vis_mlp_points on exact three-piece operands (csrc/vis_x6.hip; blob = packing.pack_vis_x6)."""
    p, d = _f32(p), _f32(d)
    M = d.shape[0]
    assert p.shape[0] * rep == M
    Y = torch.empty(M, 2, dtype=torch.float32, device=d.device)
    call("rb_vis_x6_points", ptr(p), ptr(d), c_long(M), c_int(rep), ptr(blob), ptr(Y), c_int(0), stream_ptr())
    return Y


def wide_x6_points(x, extra, blob, encoder):
    """wide_mlp_points on exact three-piece operands (csrc/wide_x6.hip; blob = packing.pack_wide_x6 / pack_illum_x6 / ..._encoder_x6)."""
    x = _f32(x)
    M = x.shape[0]
    e = _f32(extra).reshape(-1) if extra is not None else None
    Y = torch.empty(M, 32 if encoder else 144, dtype=torch.float32, device=x.device)
    if M > 0:
        call("rb_wide_x6_points", ptr(x), ptr(e), c_long(M), ptr(blob), c_int(1 if encoder else 0), ptr(Y), c_int(0), stream_ptr())
    return Y


def wide_x6(X, blob, encoder):
    """wide_x6_points on feature rows X [M,64]."""
    M = X.shape[0]
    Y = torch.empty(M, 32 if encoder else 144, dtype=torch.float32, device=X.device)
    if M > 0:
        call("rb_wide_x6", ptr(X), c_long(M), ptr(blob), c_int(1 if encoder else 0), ptr(Y), c_int(0), stream_ptr())
    return Y


def linear_pe10_256(x, blob):
    """linear_64_256(feat_pe10(x)) with the encoding fused."""
    x = _f32(x)
    M = x.shape[0]
    Y = torch.empty(M, 256, dtype=torch.float32, device=x.device)
    call("rb_linear_pe10_256", ptr(x), c_long(M), ptr(blob), ptr(Y), stream_ptr())
    return Y


WIDE_RING_MIN_ROWS = 1            # the chunk-stream kernels win at every size (one round of 64 rows: 0.067 vs 0.127 ms)


def wide_mlp_points(x, extra, blob, encoder, scale_log2=None, ring=None):
    """512-wide nets on [PE10(x) | extra] with the encoding fused: encoder=False -> raw SG outputs [M,144], True -> raw latent [M,32];
    scale_log2 None = f32-input MFMA kernel, else split precision; ring (split precision; default: by batch size): the chunk-stream
    kernel (csrc/wide_ring.h) -- bit-identical outputs."""
    x = _f32(x)
    M = x.shape[0]
    e = _f32(extra).reshape(-1) if extra is not None else None
    Y = torch.empty(M, 32 if encoder else 144, dtype=torch.float32, device=x.device)
    if ring is None:
        ring = M >= WIDE_RING_MIN_ROWS
    if scale_log2 is None:
        call("rb_wide_mlp_points", ptr(x), ptr(e), c_long(M), ptr(blob), c_int(1 if encoder else 0), ptr(Y), stream_ptr())
    elif ring:
        call("rb_wide_mlp_ring_points", ptr(x), ptr(e), c_long(M), ptr(blob), c_int(1 if encoder else 0), c_int(scale_log2), ptr(Y),
             c_int(0), stream_ptr())
    else:
        call("rb_wide_mlp_h3_points", ptr(x), ptr(e), c_long(M), ptr(blob), c_int(1 if encoder else 0), c_int(scale_log2), ptr(Y),
             stream_ptr())
    return Y


def linear_64_256(X, blob):
    M = X.shape[0]
    Y = torch.empty(M, 256, dtype=torch.float32, device=X.device)
    call("rb_linear_64_256", ptr(X), c_long(M), ptr(blob), ptr(Y), stream_ptr())
    return Y


def sdf_mlp(X, M, blob, mode, out_scale=1.0, grad_scale=1.0):
    full = mode in (1, 3)
    out0 = torch.empty((M, 257) if full else (M,), dtype=torch.float32, device=X.device)
    grad = torch.empty(M, 3, dtype=torch.float32, device=X.device) if (mode & 3) >= 2 else None
    call("rb_sdf_mlp", ptr(X), c_long(M), ptr(blob), c_int(mode), c_float(out_scale), c_float(grad_scale), ptr(out0),
         ptr(grad), stream_ptr())
    return out0, grad


def sdf_mlp_points(x, M, blob, mode, in_scale=1.0, out_scale=1.0, grad_scale=1.0):
    """rb_sdf_mlp (f32-input MFMA) straight from the points: the encoding, tangent rows included, is evaluated inside the kernel."""
    x = _f32(x)
    full = mode in (1, 3)
    out0 = torch.empty((M, 257) if full else (M,), dtype=torch.float32, device=x.device)
    grad = torch.empty(M, 3, dtype=torch.float32, device=x.device) if (mode & 3) >= 2 else None
    call("rb_sdf_mlp_points", ptr(x), c_long(M), c_float(in_scale), ptr(blob), c_int(mode), c_float(out_scale), c_float(grad_scale),
         ptr(out0), ptr(grad), stream_ptr())
    return out0, grad


def color_mlp_points(x, view, normal, feat, blob, x_scale=1.0, feat_scale=1.0):
    """rb_color_mlp (f32-input MFMA) with the feature columns read in place and [x | PE4(view) | normal] encoded in the kernel."""
    x, view, normal = _f32(x), _f32(view), _f32(normal)
    assert feat.dtype == torch.float32 and feat.stride(-1) == 1
    M = x.shape[0]
    rgb = torch.empty(M, 3, dtype=torch.float32, device=x.device)
    call("rb_color_mlp_points", ctypes.c_void_p(feat.data_ptr()), c_long(feat.stride(0)), c_float(feat_scale), ptr(x), c_float(x_scale),
         ptr(view), ptr(normal), c_long(M), ptr(blob), ptr(rgb), stream_ptr())
    return rgb


def color_x6_points(x, view, normal, feat, blob, x_scale=1.0, feat_scale=1.0, two_tile=None):
    """color_mlp_points on exact three-piece operands (blob = packing.pack_color_x6): csrc/color_x6t.hip (two tiles per wave, rounds of 128
    rows) from SDF_TWO_TILE_MIN_ROWS rows on, csrc/color_x6.hip (one tile, rounds of 64) below; two_tile forces either."""
    x, view, normal = _f32(x), _f32(view), _f32(normal)
    assert feat.dtype == torch.float32 and feat.stride(-1) == 1
    M = x.shape[0]
    rgb = torch.empty(M, 3, dtype=torch.float32, device=x.device)
    if two_tile is None:
        two_tile = sdf_two_tile(M)
    call("rb_color_x6_points", ctypes.c_void_p(feat.data_ptr()), c_long(feat.stride(0)), c_float(feat_scale), ptr(x), c_float(x_scale),
         ptr(view), ptr(normal), c_long(M), ptr(blob), ptr(rgb), c_int(1 if two_tile else 0), c_int(0), stream_ptr())
    return rgb


import os as _os
SDF_KERNEL = _os.environ.get("ROBIR_SDF_KERNEL", "ring")     # "ring" | "v1" (first-generation k_sdf_mlp_h3)


def sdf_mlp_h3(X, M, blob, mode, scale_log2, out_scale=1.0, grad_scale=1.0):
    full = mode in (1, 3)
    out0 = torch.empty((M, 257) if full else (M,), dtype=torch.float32, device=X.device)
    grad = torch.empty(M, 3, dtype=torch.float32, device=X.device) if mode >= 2 else None
    if SDF_KERNEL == "ring":        # second generation: weight ring + activations between the MFMAs (csrc/sdf_ring.hip)
        call("rb_sdf_mlp_ring", ptr(X), c_long(M), ptr(blob), c_int(mode), c_int(scale_log2), c_float(out_scale),
             c_float(grad_scale), ptr(out0), ptr(grad), c_int(0), stream_ptr())
    else:
        call("rb_sdf_mlp_h3", ptr(X), c_long(M), ptr(blob), c_int(mode), c_int(scale_log2), c_float(out_scale),
             c_float(grad_scale), ptr(out0), ptr(grad), stream_ptr())
    return out0, grad


SDF_FUSED_PE = _os.environ.get("ROBIR_SDF_FUSED_PE", "1") == "1"   # value rows / value+gradient straight from the points (csrc/sdf_ring8.hip)


def sdf_points_h3(x, M, blob, full, scale_log2, in_scale=1.0, out_scale=1.0):
    """SDF network on points x [M,3] (evaluated at x * in_scale) with the positional encoding fused into the kernel:
    -> [M,257] (full) or [M].  Bit-identical to feat_pe10 + sdf_mlp_h3 (modes 1 / 0)."""
    x = _f32(x)
    out0 = torch.empty((M, 257) if full else (M,), dtype=torch.float32, device=x.device)
    call("rb_sdf_points_ring", ptr(x), c_long(M), c_float(in_scale), ptr(blob), c_int(1 if full else 0), c_int(scale_log2),
         c_float(out_scale), ptr(out0), c_int(0), stream_ptr())
    return out0


def sdf_points_jvp_h3(x, M, blob, full, scale_log2, in_scale=1.0, out_scale=1.0, grad_scale=1.0):
    """Forward-mode value + gradient rows straight from the points (k_sdf_ring<2|3, FUSED>): -> (out0, grad [M,3])."""
    x = _f32(x)
    out0 = torch.empty((M, 257) if full else (M,), dtype=torch.float32, device=x.device)
    grad = torch.empty(M, 3, dtype=torch.float32, device=x.device)
    call("rb_sdf_points_ring_jvp", ptr(x), c_long(M), c_float(in_scale), ptr(blob), c_int(3 if full else 2), c_int(scale_log2),
         c_float(out_scale), c_float(grad_scale), ptr(out0), ptr(grad), c_int(0), stream_ptr())
    return out0, grad


SDF_GRAD = _os.environ.get("ROBIR_SDF_GRAD", "reverse")     # "reverse" (csrc/sdf_back.hip) | "forward" (mode 3 rows)
SDF_GRAD_MIN_POINTS = 16384       # below this the three-launch reverse form does not pay (0.25 ms floor; measured crossover)
SDF_GRAD_F32_MIN_POINTS = 16384   # ... of the f32-input-MFMA form (three launches against one of the four-row forward-mode kernel)
SDF_GRAD_SLAB = 1 << 20           # points per slab of the reverse form (8.5 KB of scratch per point)
_sdf_grad_scratch = {}


def release_scratch():
    """Drop the cached per-(device, stream) scratch of sdf_value_grad (8.9 GB per stream at the default slab)."""
    _sdf_grad_scratch.clear()


def sdf_ring_waves():
    """Current setting of rb_sdf_ring_waves (8 = csrc/sdf_ring8.hip, 4 = csrc/sdf_ring.hip) without changing it."""
    return int(_lib.legacy().rb_sdf_ring_waves(0))


def sdf_value_grad(x, M, blob, back, scale_log2, in_scale=1.0, out_scale=1.0):
    """All 257 outputs + d sdf / dx by the reverse-mode pass: x [M,3] -> out [M,257], grad [M,3]."""
    wb, w8 = back
    out0 = torch.empty(M, 257, dtype=torch.float32, device=x.device)
    grad = torch.empty(M, 3, dtype=torch.float32, device=x.device)
    if M == 0:
        return out0, grad
    slab = min(M, SDF_GRAD_SLAB)
    need = int(_lib.legacy().rb_sdf_value_grad_scratch_floats(c_long(slab)))
    key = (x.device, torch.cuda.current_stream().cuda_stream)
    scratch = _sdf_grad_scratch.get(key)
    if scratch is None or scratch.numel() < need:
        scratch = _sdf_grad_scratch[key] = torch.empty(need, dtype=torch.float32, device=x.device)
    x = _f32(x)
    fused = SDF_FUSED_PE and sdf_ring_waves() == 8      # the encoding is fused into the eight-wave kernel
    for a in range(0, M, slab):
        n = min(slab, M - a)
        if fused:
            call("rb_sdf_value_grad_points", ptr(x[a:a + n]), c_long(n), c_float(in_scale), ptr(blob), ptr(wb), ptr(w8),
                 c_int(scale_log2), c_float(out_scale), c_float(out_scale * in_scale), ptr(out0[a:a + n]), ptr(grad[a:a + n]),
                 ptr(scratch), c_int(0), stream_ptr())
        else:
            X = feat_pe10(x[a:a + n], scale=in_scale)
            call("rb_sdf_value_grad", ptr(X), c_long(n), ptr(blob), ptr(wb), ptr(w8), c_int(scale_log2), c_float(out_scale),
                 c_float(out_scale * in_scale), ptr(out0[a:a + n]), ptr(grad[a:a + n]), ptr(scratch), c_int(0), stream_ptr())
    return out0, grad


def sdf_value_grad_f32(x, M, blob, back, in_scale=1.0, out_scale=1.0):
    """sdf_value_grad at the reference's precision (f32-input MFMA; csrc/mlp_kernels.hip k_sdf_mlp<5> + k_sdf_back_f32): x [M,3] ->
    out [M,257], grad [M,3].  blob = packing.pack_sdf(full=True), back = packing.pack_sdf_back."""
    wt, w8 = back
    out0 = torch.empty(M, 257, dtype=torch.float32, device=x.device)
    grad = torch.empty(M, 3, dtype=torch.float32, device=x.device)
    if M == 0:
        return out0, grad
    slab = min(M, SDF_GRAD_SLAB)
    need = int(_lib.lib().rb_sdf_value_grad_f32_scratch_floats(c_long(slab)))
    key = (x.device, torch.cuda.current_stream().cuda_stream, "f32")
    scratch = _sdf_grad_scratch.get(key)
    if scratch is None or scratch.numel() < need:
        scratch = _sdf_grad_scratch[key] = torch.empty(need, dtype=torch.float32, device=x.device)
    x = _f32(x)
    for a in range(0, M, slab):
        n = min(slab, M - a)
        call("rb_sdf_value_grad_f32_points", ptr(x[a:a + n]), c_long(n), c_float(in_scale), ptr(blob), ptr(wt), ptr(w8),
             c_float(out_scale), c_float(out_scale * in_scale), ptr(out0[a:a + n]), ptr(grad[a:a + n]), ptr(scratch), stream_ptr())
    return out0, grad


SDF_TWO_TILE_MIN_ROWS = int(os.environ.get("ROBIR_SDF_TWO_TILE_MIN_ROWS", "16385"))
_CU_COUNT = []


def _compute_units():
    if not _CU_COUNT:
        _CU_COUNT.append(torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count if torch.cuda.is_available() else 256)
    return _CU_COUNT[0]


def sdf_two_tile(M):
    """The exact-operand SDF / colour kernels come in two forms: two 16-row tiles per wave (rounds of 128 rows, half the LDS traffic per
    MFMA: csrc/sdf_x6t.hip, sdf_back_x6t.hip, color_x6t.hip) and one tile per wave (rounds of 64 rows).  Both run one persistent
    workgroup per compute unit, so a launch costs (rounds per workgroup) x (time of a round), and a two-tile round takes 1.42-1.5 x a
    one-tile round (tools/sweep_two_tile.py: 0.135 / 0.095 ms distance, 0.31 / 0.21 value + gradient, 0.083 / 0.055 colour): the
    two-tile form wins where it needs fewer than two thirds of the passes -- 16385..32768 rows, and everything beyond 49152 on 256
    compute units.  ROBIR_SDF_TWO_TILE_MIN_ROWS: rows below it always take the one-tile form (0: always two tiles).  The two forms
    agree to fp32 summation order."""
    if SDF_TWO_TILE_MIN_ROWS <= 0:
        return True
    if M < SDF_TWO_TILE_MIN_ROWS:
        return False
    cus = _compute_units()
    r1, r2 = -(-M // (64 * cus)), -(-M // (128 * cus))
    return 2 * r1 > 3 * r2


def sdf_points_x6(x, M, blob, full, in_scale=1.0, out_scale=1.0):
    """SDF value rows on exact three-piece operands (csrc/sdf_x6.hip, sdf_x6t.hip): x [M,3] -> out [M,257] | [M]; blob = packing.pack_sdf_x6(full)."""
    x = _f32(x)
    out0 = torch.empty((M, 257) if full else (M,), dtype=torch.float32, device=x.device)
    if M > 0:
        call("rb_sdf_x6_points", ptr(x), c_long(M), c_float(in_scale), ptr(blob), c_int(1 if full else 0), c_float(out_scale), ptr(out0),
             c_int(1 if sdf_two_tile(M) else 0), c_int(0), stream_ptr())
    return out0


def sdf_value_grad_x6(x, M, blob, back, in_scale=1.0, out_scale=1.0):
    """sdf_value_grad_f32 with both passes on exact three-piece operands (blob = packing.pack_sdf_x6(full=True), back =
    (packing.pack_sdf_back_x6(...), packing.pack_sdf_back_x6(..., two_tile=True)[0]) -- the one-tile kernel's transposed layers, the
    head row, the two-tile kernel's transposed layers; a 2-tuple (round 3's form) runs the one-tile kernels at every size)."""
    wt, w8 = back[0], back[1]
    wt2 = back[2] if len(back) > 2 else None
    out0 = torch.empty(M, 257, dtype=torch.float32, device=x.device)
    grad = torch.empty(M, 3, dtype=torch.float32, device=x.device)
    if M == 0:
        return out0, grad
    slab = min(M, SDF_GRAD_SLAB)
    need = int(_lib.lib().rb_sdf_value_grad_f32_scratch_floats(c_long(slab)))
    key = (x.device, torch.cuda.current_stream().cuda_stream, "f32")
    scratch = _sdf_grad_scratch.get(key)
    if scratch is None or scratch.numel() < need:
        scratch = _sdf_grad_scratch[key] = torch.empty(need, dtype=torch.float32, device=x.device)
    x = _f32(x)
    for a in range(0, M, slab):
        n = min(slab, M - a)
        two = wt2 is not None and sdf_two_tile(n)
        call("rb_sdf_value_grad_x6_points", ptr(x[a:a + n]), c_long(n), c_float(in_scale), ptr(blob), ptr(wt2 if two else wt), ptr(w8),
             c_float(out_scale), c_float(out_scale * in_scale), ptr(out0[a:a + n]), ptr(grad[a:a + n]), ptr(scratch), c_int(1 if two else 0),
             stream_ptr())
    return out0, grad


def color_mlp(X, blob):
    M = X.shape[0]
    Y = torch.empty(M, 3, dtype=torch.float32, device=X.device)
    call("rb_color_mlp", ptr(X), c_long(M), ptr(blob), ptr(Y), stream_ptr())
    return Y


def color_mlp_h3(X, blob, scale_log2):
    M = X.shape[0]
    rgb = torch.empty(M, 3, dtype=torch.float32, device=X.device)
    call("rb_color_mlp_h3", ptr(X), c_long(M), ptr(blob), c_int(scale_log2), ptr(rgb), stream_ptr())
    return rgb


def illum_mlp(X, blob):
    M = X.shape[0]
    raw = torch.empty(M, 144, dtype=torch.float32, device=X.device)
    call("rb_illum_mlp", ptr(X), c_long(M), ptr(blob), ptr(raw), stream_ptr())
    return raw


def wide_mlp_h3(X, blob, encoder, scale_log2, ring=None):
    """Split-precision 512-wide nets on feature rows: encoder=False -> raw SG outputs [M,144]; encoder=True -> raw latent [M,32].
    ring (default: by batch size): the chunk-stream kernel (csrc/wide_ring.h) -- bit-identical outputs."""
    M = X.shape[0]
    Y = torch.empty(M, 32 if encoder else 144, dtype=torch.float32, device=X.device)
    if ring is None:
        ring = M >= WIDE_RING_MIN_ROWS
    if ring:
        call("rb_wide_mlp_ring", ptr(X), c_long(M), ptr(blob), c_int(1 if encoder else 0), c_int(scale_log2), ptr(Y), c_int(0), stream_ptr())
    else:
        call("rb_wide_mlp_h3", ptr(X), c_long(M), ptr(blob), c_int(1 if encoder else 0), c_int(scale_log2), ptr(Y), stream_ptr())
    return Y


def illum_decode(raw):
    M = raw.shape[0]
    sgs = torch.empty(M, 24, 7, dtype=torch.float32, device=raw.device)
    call("rb_illum_decode", ptr(raw), c_long(M), ptr(sgs), stream_ptr())
    return sgs


def ae_encode(X, blob):
    M = X.shape[0]
    raw = torch.empty(M, 32, dtype=torch.float32, device=X.device)
    call("rb_ae_encode", ptr(X), c_long(M), ptr(blob), ptr(raw), stream_ptr())
    return raw


def ae_latent(raw, var=None, act=0, noise=None, noise_scale=0.0):
    M = raw.shape[0]
    lat = torch.empty_like(raw)
    lat2 = torch.empty_like(raw) if noise is not None else None
    call("rb_ae_latent", ptr(raw), c_long(M), ptr(var), c_int(act), ptr(_f32(noise) if noise is not None else None),
         c_float(noise_scale), ptr(lat), ptr(lat2), stream_ptr())
    return lat, lat2


def ae_decode(lat, blob, n_out, sigmoid_out):
    M = lat.shape[0]
    Y = torch.empty(M, n_out, dtype=torch.float32, device=lat.device)
    call("rb_ae_decode", ptr(lat), c_long(M), ptr(blob), c_int(n_out), c_int(1 if sigmoid_out else 0), ptr(Y),
         stream_ptr())
    return Y


def axpy(a, b, s):
    a, b = _f32(a), _f32(b)
    y = torch.empty_like(a)
    call("rb_axpy", ptr(a), ptr(b), c_float(s), c_long(a.numel()), ptr(y), stream_ptr())
    return y


def normalize3(x, eps, mode):
    x = _f32(x)
    y = torch.empty_like(x)
    call("rb_normalize3", ptr(x), c_long(x.shape[0]), c_float(eps), c_int(mode), ptr(y), stream_ptr())
    return y


def dvis_dirs(lgt, u_theta, u_phi, thr=1.0, direct=False):
    """lgt [L,7]; u_theta/u_phi [C,L,nsamp] (or [L,nsamp]).  -> dirs [C*L*nsamp,3], wdir [C*L*nsamp], wsum [C*L]."""
    lgt, u_theta, u_phi = _f32(lgt), _f32(u_theta), _f32(u_phi)
    if u_theta.dim() == 2:
        u_theta, u_phi = u_theta[None], u_phi[None]
    C, L, ns = u_theta.shape
    dev = lgt.device
    dirs = torch.empty(C * L * ns, 3, dtype=torch.float32, device=dev)
    wdir = torch.empty(C * L * ns, dtype=torch.float32, device=dev)
    wsum = torch.empty(C * L, dtype=torch.float32, device=dev)
    call("rb_dvis_dirs", ptr(lgt), c_int(L), c_int(ns), c_int(C), c_int(1 if direct else 0), ptr(u_theta.contiguous()), ptr(u_phi.contiguous()),
         c_float(thr), ptr(dirs), ptr(wdir), ptr(wsum), stream_ptr())
    return dirs, wdir, wsum


DVIS_KERNEL_NAMES = {"fp32": "k_dvis_fused<fp32>", "f16x6": "k_dvis_x6t", "f16x6-pt": "k_dvis_x6t", "f16x6-stream": "k_dvis_x6t<stream>",
                     "f16x6-1t": "k_dvis_x6", "f16x1": "k_dvis_f16p", "f16x3-auto": "k_dvis_v2", "f16x3-v3": "k_dvis3_stream",
                     "f16x3-v2": "k_dvis_v2", "f16x3": "k_dvis_fused<H3>"}


DVIS_STREAM_WORKGROUPS = 0        # persistent workgroups of the streaming visibility kernel; 0 = one per compute unit
DVIS_STREAM_MAX_POINTS = 8192     # launches up to this many surface points take the tile-list (streaming) form
DVIS_STREAM_SHORT_LIST = 1024     # ... and any launch whose points have at most this many sampled directions (L * nsamp)
# the f16 throughput mode's kernel: 3 = the point-block form (csrc/vis_diffuse_f16p.hip: sixteen points x one direction per tile; needs
# ascending chunk ids, else 2), 2 = the per-point tile list, two-chunk steps, 1 = round 4's kernel -- the same bits, 1 / 2 kept for A/B
DVIS_F16_GEN = int(os.environ.get("ROBIR_DVIS_F16_GEN", "3"))


def chunk_ids_ascending(chunk_id):
    """True if chunk_id (int32 [n] or None) never decreases.  The answer is remembered ON the tensor (`_robir_ascending`): constructors that
    know (the renderer's hit lists: pixel order) set it and save the device -> host read this costs otherwise."""
    if chunk_id is None or chunk_id.numel() < 2:
        return True
    known = getattr(chunk_id, "_robir_ascending", None)
    if known is None:
        known = bool((chunk_id[1:] >= chunk_id[:-1]).all())
        chunk_id._robir_ascending = known
    return known
DVIS_X6_FP8 = os.environ.get("ROBIR_DVIS_X6_FP8", "1") == "1"      # the two-tile light-visibility kernel as built (csrc/vis_diffuse_x6t.hip XT_FP8 = 1: two of its six products on the bf8 MFMA) takes the weight layout of packing.repack_x6_chunks_fp8; 0 for a library built with -DXT_FP8=0 (a mismatch is refused by the entry point)
DVIS_X6_FORM = os.environ.get("ROBIR_DVIS_X6_FORM", "auto")          # what "f16x6" runs: auto | f16x6-pt | f16x6-stream | f16x6-1t


def dvis_fused(normals, chunk_id, A, Bd, dirs, wdir, wsum, split, L, nsamp, argmax_vis=False, eval_count=None,
               precision="fp32"):
    """precision: 'fp32' (f32-input MFMA, exact fp32 fma chain), 'f16x6' (exact fp32 operands as three halves each, six f16
    MFMA products per multiply-add: not narrower than fp32) or 'f16x3*' (split precision, 22-bit operands, ~2^-22 relative)."""
    normals = _f32(normals)
    if precision == "f16x3-auto":
        # same arithmetic, bit-identical results: the streaming family balances small launches (a single 1024-pixel chunk) over
        # the CUs; at whole-view sizes the one-point-per-workgroup kernel is as fast and needs no scratch
        # (the tile-list forms cut a point's L*nsamp directions into whole 16-sample tiles: other shapes take the per-point kernel)
        precision = "f16x3-v3" if normals.shape[0] <= DVIS_STREAM_MAX_POINTS and (L * nsamp) % 16 == 0 else "f16x3-v2"
    h3 = precision.startswith("f16x3")
    X6 = ("f16x6", "f16x6-1t", "f16x6-pt", "f16x6-stream")
    assert h3 or precision in ("fp32", "f16x1") or precision in X6, precision
    if precision == "f16x6":
        # "auto": the persistent tile-list form for launches up to DVIS_STREAM_MAX_POINTS points (a single 1024-pixel chunk: balanced
        # over the CUs, 0.4 % tile padding), one workgroup per point beyond (no scratch); the two are bit-identical
        # (same bits); a light whose L*nsamp is not a multiple of 16 (odd lobe counts with nsamp = 8) cannot be cut into whole tiles of
        # the global list and takes the per-point form at every size
        # ... and at EVERY size for short direction lists (L * nsamp <= 1024: the CESR hook's nsamp = 8): a point then has four rounds of
        # 128 pairs, the per-point prologue / half-empty last round weigh 5-6 % (tools/ab_dvis_forms.py, 8 / 32 / 128 chunks: 0.94-0.95 of
        # the per-point form's time; with 4096 directions 0.99-1.00 -- there the per-point form stays: one launch, no 10 GB of scratch)
        stream_ok = (L * nsamp) % 16 == 0 and (normals.shape[0] <= DVIS_STREAM_MAX_POINTS or L * nsamp <= DVIS_STREAM_SHORT_LIST)
        precision = DVIS_X6_FORM if DVIS_X6_FORM != "auto" else ("f16x6-stream" if stream_ok else "f16x6-pt")
    if precision == "f16x1" and (L * nsamp) % 16 != 0:
        raise ValueError(f"the f16 throughput kernel (ROBIR_PRECISION=f16) exists in the tile-list form only: L*nsamp = {L}*{nsamp} must be a "
                         "multiple of 16 -- use ROBIR_PRECISION=exact for this light")
    # "f16x3-v2" = second-generation split-precision kernel: two tiles per wave, one workgroup per CU, head on the matrix
    # pipe (csrc/vis_diffuse_v2.hip); "f16x3" = first generation: one 16-sample tile per wave, two workgroups per CU,
    # weights staged by LDS-DMA; "f16x6*" = exact three-piece operands: "-1t" round 3's one tile per wave, one workgroup per point
    # (csrc/vis_diffuse_x6.hip); "-pt" two tiles per wave, one workgroup per point; "-stream" two tiles per wave, persistent grid
    # over the global tile list (both csrc/vis_diffuse_x6t.hip, bit-identical to each other)
    n = normals.shape[0]
    out = torch.empty(n, L, dtype=torch.float32, device=normals.device)
    if chunk_id is not None:
        assert chunk_id.dtype == torch.int32
    if precision in ("f16x3-v3", "f16x6-stream", "f16x1"):
        # streaming form: global tile list + persistent grid (csrc/vis_diffuse_v3.hip, vis_diffuse_x6t.hip); scratch sized for the
        # worst case (every direction front-facing) so that nothing has to be read back to the host
        LS = L * nsamp
        dev = normals.device
        if precision == "f16x1" and DVIS_F16_GEN == 3 and chunk_ids_ascending(chunk_id):
            # point-block form: scratch by items (a block of 16 points, two items where a block spans a chunk boundary)
            n_chunks = max(1, dirs.shape[0] // LS)
            items_max = (n + 15) // 16 + n_chunks - 1
            entries = torch.empty(items_max * LS, dtype=torch.int32, device=dev)
            pair_vis = torch.empty(items_max * LS * 16, dtype=torch.float32, device=dev)
            round_info = torch.empty(items_max * LS // 16, 4, dtype=torch.int32, device=dev)
            item_info = torch.empty(items_max, 4, dtype=torch.int32, device=dev)
            counters = torch.empty(4, dtype=torch.int64, device=dev)
            call("rb_dvis_pblock_f16", ptr(normals), ptr(chunk_id), c_long(n), ptr(A), ptr(Bd), ptr(dirs), ptr(wdir), ptr(wsum),
                 ptr(split["hidden_f16_head"]), c_int(L), c_int(nsamp), c_int(1 if argmax_vis else 0), c_int(items_max), ptr(entries),
                 ptr(pair_vis), ptr(round_info), ptr(item_info), ptr(counters), c_int(DVIS_STREAM_WORKGROUPS), ptr(out), ptr(eval_count),
                 stream_ptr())
            if os.environ.get("ROBIR_RANGE_CHECK", "") == "sync" and int(counters[3].item()) != 0:
                # counters[3]: chunk ids not ascending (a list whose order was only ASSERTED through the `_robir_ascending` attribute) or more
                # items than the scratch holds -- the kernel then leaves its NaN fill in `out`.  Read in the debug mode only: a host sync.
                raise _lib.RobirHipError("rb_dvis_pblock_f16: chunk ids are not ascending or the item scratch overflowed -- the outputs "
                                         "of this call are NaN (ADVICE r5; run without the `_robir_ascending` tag to let ops check the order)")
            return out
        pair_j = torch.empty(n * LS, dtype=torch.int16, device=dev)
        pair_vis = torch.empty(n * LS, dtype=torch.float32, device=dev)
        tile_info = torch.empty(n * LS // 16, 2, dtype=torch.int32, device=dev)
        point_info = torch.empty(n, 2, dtype=torch.int32, device=dev)
        counters = torch.empty(2, dtype=torch.int64, device=dev)
        x6 = precision in ("f16x6-stream", "f16x1")      # "f16x1": plain f16, one product (csrc/vis_diffuse_f16t.hip): NARROWER than fp32
        blob, fmt = split["hidden_x6_head" if x6 else "hidden_h3_head"], split["x6_head_scale_log2" if x6 else "h3_head_scale_log2"]
        if precision == "f16x6-stream" and DVIS_X6_FP8:
            blob, fmt = split["hidden_x6_head_fp8"], 8
        if precision == "f16x1" and DVIS_F16_GEN >= 2:
            blob, fmt = split["hidden_f16_head"], 1       # rb_dvis_stream_f16: format 1 = the h-only blob -> second-generation kernel
        call("rb_dvis_stream_f16" if precision == "f16x1" else ("rb_dvis_stream_x6" if x6 else "rb_dvis_stream"), ptr(normals), ptr(chunk_id), c_long(n), ptr(A), ptr(Bd), ptr(dirs), ptr(wdir), ptr(wsum),
             ptr(blob), c_int(L), c_int(nsamp), c_int(1 if argmax_vis else 0),
             c_int(fmt), ptr(pair_j), ptr(pair_vis), ptr(tile_info), ptr(point_info), ptr(counters),
             c_int(DVIS_STREAM_WORKGROUPS), ptr(out), ptr(eval_count), stream_ptr())
        return out
    if precision in X6:
        fp8 = DVIS_X6_FP8 and precision != "f16x6-1t"
        call("rb_dvis_fused_x6" if precision == "f16x6-1t" else "rb_dvis_fused_x6t", ptr(normals), ptr(chunk_id), c_long(n), ptr(A), ptr(Bd), ptr(dirs), ptr(wdir), ptr(wsum),
             ptr(split["hidden_x6_head_fp8" if fp8 else "hidden_x6_head"]), c_int(L), c_int(nsamp), c_int(1 if argmax_vis else 0),
             c_int(8 if fp8 else split["x6_head_scale_log2"]), ptr(out), ptr(eval_count), stream_ptr())
        return out
    if precision == "f16x3-v2":
        call("rb_dvis_fused_v2", ptr(normals), ptr(chunk_id), c_long(n), ptr(A), ptr(Bd), ptr(dirs), ptr(wdir), ptr(wsum),
             ptr(split["hidden_h3_head"]), c_int(L), c_int(nsamp), c_int(1 if argmax_vis else 0),
             c_int(split["h3_head_scale_log2"]), ptr(out), ptr(eval_count), stream_ptr())
        return out
    # precision 5 = the first-generation split-precision kernel: compiled into the legacy library only
    (_lib.call_legacy if h3 else call)("rb_dvis_fused", ptr(normals), ptr(chunk_id), c_long(n), ptr(A), ptr(Bd), ptr(dirs), ptr(wdir), ptr(wsum),
         ptr(split["hidden_h3"] if h3 else split["hidden"]), ptr(split["w_last"]), ptr(split["b_last"]), c_int(L),
         c_int(nsamp), c_int(1 if argmax_vis else 0), c_int(5 if h3 else 0), c_int(split["h3_scale_log2"] if h3 else 0),
         ptr(out), ptr(eval_count), stream_ptr())
    return out


def spec_vis_sample(normal, view, rough, chunk_id, n_chunks, u_theta, u_phi):
    normal, view, rough = _f32(normal), _f32(view), _f32(rough).reshape(-1)
    u_theta, u_phi = _f32(u_theta), _f32(u_phi)
    n, ns = u_theta.shape
    dev = normal.device
    sharp = torch.empty(n, dtype=torch.float32, device=dev)
    cmin = torch.empty(n_chunks, dtype=torch.int32, device=dev)
    dirs = torch.empty(n * ns, 3, dtype=torch.float32, device=dev)
    wts = torch.empty(n * ns, dtype=torch.float32, device=dev)
    front = torch.empty(n * ns, dtype=torch.uint8, device=dev)
    call("rb_spec_vis_sample", ptr(normal), ptr(view), ptr(rough), ptr(chunk_id), c_long(n), c_int(n_chunks), c_int(ns),
         ptr(u_theta), ptr(u_phi), ptr(sharp), ptr(cmin), ptr(dirs), ptr(wts), ptr(front), stream_ptr())
    return dirs, wts, front


def spec_vis_sample_lobes(normal, view, lobes, lambdas, chunk_id, n_chunks, u_theta, u_phi):
    """spec_vis_sample for a caller's own lobes [n,3] / lambdas [n] (the reference's get_specular_visibility signature)."""
    normal, view, lobes, lambdas = _f32(normal), _f32(view), _f32(lobes), _f32(lambdas).reshape(-1)
    u_theta, u_phi = _f32(u_theta), _f32(u_phi)
    n, ns = u_theta.shape
    assert lobes.shape == (n, 3) and lambdas.shape == (n,), (lobes.shape, lambdas.shape)
    dev = normal.device
    sharp = torch.empty(n, dtype=torch.float32, device=dev)
    cmin = torch.empty(n_chunks, dtype=torch.int32, device=dev)
    dirs = torch.empty(n * ns, 3, dtype=torch.float32, device=dev)
    wts = torch.empty(n * ns, dtype=torch.float32, device=dev)
    front = torch.empty(n * ns, dtype=torch.uint8, device=dev)
    call("rb_spec_vis_sample_lobes", ptr(normal), ptr(view), ptr(lobes), ptr(lambdas), ptr(chunk_id), c_long(n), c_int(n_chunks), c_int(ns),
         ptr(u_theta), ptr(u_phi), ptr(sharp), ptr(cmin), ptr(dirs), ptr(wts), ptr(front), stream_ptr())
    return dirs, wts, front


def spec_vis_reduce(logits, front, wts, n, nsamp, inv, argmax_vis, testing):
    bvis = torch.empty(n, dtype=torch.float32, device=logits.device)
    call("rb_spec_vis_reduce", ptr(logits), ptr(front), ptr(wts), c_long(n), c_int(nsamp), c_int(int(inv)),
         c_int(int(argmax_vis)), c_int(int(testing)), ptr(bvis), stream_ptr())
    return bvis


def sg_shade(normal, view, lgt, f0, rough, albedo, bvis, light_vis=None, metallic=None, indir_integral=None,
             lin_diff=False, want_shadow=False):
    normal, view, lgt, albedo = _f32(normal), _f32(view), _f32(lgt), _f32(albedo)
    rough = _f32(rough).reshape(-1)
    n = normal.shape[0]
    per_point = lgt.dim() == 3
    M = lgt.shape[-2]
    dev = normal.device
    rgb, spec, diff = (torch.empty(n, 3, dtype=torch.float32, device=dev) for _ in range(3))
    shadow = torch.empty(n, 3, dtype=torch.float32, device=dev) if want_shadow else None
    # f0: device tensor holding the scalar |specular_reflectance| (read by the kernel -- no host copy, no cache to go stale)
    f0 = (f0.detach().to(device=dev, dtype=torch.float32).reshape(-1)[:1].contiguous() if isinstance(f0, torch.Tensor)
          else torch.full((1,), float(f0), device=dev))
    call("rb_sg_shade", ptr(normal), ptr(view), ptr(lgt), c_int(1 if per_point else 0), c_int(M), ptr(f0), ptr(rough),
         ptr(albedo), ptr(_f32(metallic).reshape(-1) if metallic is not None else None),
         ptr(_f32(light_vis) if light_vis is not None else None), ptr(bvis),
         ptr(_f32(indir_integral) if indir_integral is not None else None), c_int(1 if lin_diff else 0), c_long(n),
         ptr(rgb), ptr(spec), ptr(diff), ptr(shadow), stream_ptr())
    return rgb, spec, diff, shadow


# ------------------------------------------------------------------------------------------------ octree
def _host3(a, ctype):
    import numpy as np
    arr = np.ascontiguousarray(a, dtype=np.float32 if ctype is ctypes.c_float else np.int32)
    return (ctype * 3)(*[arr[i].item() for i in range(3)])


class OctreeTablesDev:
    """Device octree tables in the layout of include/robir_hip.h + host root description."""

    def __init__(self, node, nrm, B, root_min, root_size, res, min_step):
        self.node, self.nrm, self.B = node, nrm, int(B)
        self.root_min, self.root_size, self.res = root_min, root_size, res       # numpy float32[3], float32[3], int32[3]
        self.min_step = float(min_step)

    def args(self):
        return (ptr(self.node), ptr(self.nrm), c_long(self.B), _host3(self.root_min, ctypes.c_float),
                _host3(self.root_size, ctypes.c_float), _host3(self.res, ctypes.c_int))

    @property
    def clamp_dt(self):
        return self.min_step * 10


def octree_cast_batched(T, origins, per_ray_origin, dirs, batch, max_iter, step, sched_cap=0):
    dirs = _f32(dirs)
    origins = _f32(origins)
    R = dirs.shape[0]
    dev = dirs.device
    x = torch.empty(R, 3, dtype=torch.float32, device=dev)
    hit = torch.empty(R, dtype=torch.uint8, device=dev)
    t = torch.empty(R, dtype=torch.float32, device=dev)
    nb = (R + batch - 1) // batch
    sched = torch.zeros(nb, sched_cap, 2, dtype=torch.int32, device=dev) if sched_cap > 0 else None
    call("rb_octree_cast_batched", *T.args(), ptr(origins), c_int(1 if per_ray_origin else 0), ptr(dirs), c_long(R),
         c_int(batch), c_int(max_iter), ctypes.c_double(step), c_float(T.clamp_dt), ptr(x), ptr(hit), ptr(t), ptr(sched),
         c_int(sched_cap), stream_ptr())
    return x, hit.bool(), t, sched


CAST_ONE_LAUNCH = _os.environ.get("ROBIR_CAST_ONE_LAUNCH", "1") == "1"    # the persistent-grid form of the general lock-step cast
CAST_COOP_REFUSED = 0              # how often the cooperative launch was refused and the per-iteration launches took over
CAST_ONE_LAUNCH_MAX_RAYS = 16384   # above this the per-iteration launches fill the chip and win (tools/ab_cast.py: 60 k rays 1.3 vs 2.1 ms)


def octree_cast_general(T, origins, dirs, max_iter, step, check_every=16, max_total=4096, one_launch=None):
    """One lock-step batch of any size: origins/dirs [R,3].  one_launch (default CAST_ONE_LAUNCH): init, every iteration and finish
    in one persistent launch (k_cast_coop) instead of one launch per iteration -- the same results bit for bit, no host sync."""
    dirs, origins = _f32(dirs), _f32(origins)
    R = dirs.shape[0]
    dev = dirs.device
    t = torch.empty(R, dtype=torch.float32, device=dev)
    leaf = torch.empty(R, dtype=torch.int32, device=dev)
    active = torch.empty(R, dtype=torch.uint8, device=dev)
    a = T.args()
    if (CAST_ONE_LAUNCH and R <= CAST_ONE_LAUNCH_MAX_RAYS) if one_launch is None else one_launch:
        counters = torch.zeros(max_total + 2, dtype=torch.int32, device=dev)
        arrive = torch.zeros(1024, dtype=torch.int64, device=dev)                # grid barrier: two 64-bit slots per workgroup (2 x 512, by epoch parity)
        x = torch.empty(R, 3, dtype=torch.float32, device=dev)
        hit = torch.empty(R, dtype=torch.uint8, device=dev)
        t_out = torch.empty(R, dtype=torch.float32, device=dev)
        rc = 0
        if R > 0:
            # status 2: the cooperative launch was refused by the runtime's STATIC occupancy check (or the device lacks cooperative
            # launches) and nothing ran -- the per-iteration launches below give the same bits.  The grid is at most one workgroup per
            # compute unit; residency beside concurrent work on other streams rests on the runtime's cooperative queue and is not
            # stress-tested here: ROBIR_CAST_ONE_LAUNCH=0 where in doubt (csrc/octree.hip)
            rc = _lib.lib().rb_octree_cast_coop(*a, ptr(origins), ptr(dirs), c_long(R), c_int(max_iter), ctypes.c_double(step),
                                                c_int(max_total), c_float(T.clamp_dt), ptr(t), ptr(leaf), ptr(active), ptr(counters),
                                                ptr(arrive), ptr(x), ptr(hit), ptr(t_out), stream_ptr())
            if rc not in (0, 2):
                raise _lib.RobirHipError(f"rb_octree_cast_coop failed ({rc}): {_lib.lib().rb_last_error().decode()}")
        if rc == 0:
            return x, hit.bool(), t_out, counters
        global CAST_COOP_REFUSED
        CAST_COOP_REFUSED += 1
    counters = torch.zeros(max_total + 2, dtype=torch.int32, device=dev)
    call("rb_octree_cast_init", *a, ptr(origins), ptr(dirs), c_long(R), c_int(max_iter), ptr(t), ptr(leaf), ptr(active),
         ptr(counters), stream_ptr())
    if max_iter > 0:
        call("rb_octree_cast_iter", *a, ptr(origins), ptr(dirs), c_long(R), c_int(max_iter), ctypes.c_double(step),
             c_int(0), c_int(max_iter + 1), ptr(t), ptr(leaf), ptr(active), ptr(counters), stream_ptr())
    else:
        it = 0
        while it < max_total:
            call("rb_octree_cast_iter", *a, ptr(origins), ptr(dirs), c_long(R), c_int(max_iter), ctypes.c_double(step),
                 c_int(it), c_int(check_every), ptr(t), ptr(leaf), ptr(active), ptr(counters), stream_ptr())
            it += check_every
            if int(counters[it].item()) == 0:      # host sync once per `check_every` iterations
                break
    x = torch.empty(R, 3, dtype=torch.float32, device=dev)
    hit = torch.empty(R, dtype=torch.uint8, device=dev)
    t_out = torch.empty(R, dtype=torch.float32, device=dev)
    call("rb_octree_cast_finish", *a, ptr(origins), ptr(dirs), c_long(R), c_int(max_iter), c_float(T.clamp_dt), ptr(t),
         ptr(leaf), ptr(x), ptr(hit), ptr(t_out), stream_ptr())
    return x, hit.bool(), t_out, counters


LAST_OCTREE_VIS_LAYOUT = None
OVIS_COMPACT = _os.environ.get("ROBIR_OVIS_COMPACT", "1") == "1"   # stable compaction of the active rays between iterations
OVIS_CHUNKS_PER_CALL = 48       # chunks per traced-visibility launch group: bounds the scratch (28 B per pair slot) to ~5.6 GB


def dvis_octree(T, points, normals, chunk_id, n_chunks, dirs, wdir, wsum, L, nsamp, argmax_vis=False, eval_count=None,
                batch_pairs=2000000, max_iter=32, max_points_per_chunk=1024):
    """Traced light visibility (OctreeVisModel as the VisModel, csrc/octree_vis.hip) -> vis [n, L].  chunk_id ascending.
    Many chunks are walked in groups of OVIS_CHUNKS_PER_CALL (every chunk is its own set of lock-step groups, so the split
    changes nothing): the scratch is sized for the largest group instead of the whole view."""
    global LAST_OCTREE_VIS_LAYOUT
    points, normals = _f32(points), _f32(normals)
    n = points.shape[0]
    dev = points.device
    LS = L * nsamp
    out = torch.empty(n, L, dtype=torch.float32, device=dev)
    if n == 0:
        return out
    if chunk_id is not None and n_chunks > OVIS_CHUNKS_PER_CALL:
        # first row of every chunk (chunk_id ascending): one small device -> host read per view
        bounds = torch.searchsorted(chunk_id, torch.arange(0, n_chunks + 1, OVIS_CHUNKS_PER_CALL, device=dev, dtype=chunk_id.dtype)
                                    ).tolist() + [n]
        total = None
        for k, c0 in enumerate(range(0, n_chunks, OVIS_CHUNKS_PER_CALL)):
            a, b = bounds[k], bounds[k + 1] if c0 + OVIS_CHUNKS_PER_CALL < n_chunks else n
            nc = min(OVIS_CHUNKS_PER_CALL, n_chunks - c0)
            if b > a:
                out[a:b] = _dvis_octree_impl(T, points[a:b], normals[a:b], (chunk_id[a:b] - c0).contiguous(), nc, dirs[c0 * LS:(c0 + nc) * LS],
                                       wdir[c0 * LS:(c0 + nc) * LS], wsum[c0 * L:(c0 + nc) * L], L, nsamp, argmax_vis, eval_count,
                                       batch_pairs, max_iter, max_points_per_chunk)
                total = LAST_OCTREE_VIS_LAYOUT.clone() if total is None else total + LAST_OCTREE_VIS_LAYOUT
        LAST_OCTREE_VIS_LAYOUT = total
        return out
    cap = n * LS
    # group table size: groups per chunk from the largest chunk population (k_ovis_layout would truncate the table otherwise).
    # Callers with chunks beyond 1024 points pass max_points_per_chunk; a count that cannot fit is caught here (host read
    # only in that case)
    per_chunk = n if chunk_id is None else max_points_per_chunk
    if chunk_id is not None and n > n_chunks * per_chunk:
        per_chunk = int(torch.bincount(chunk_id.long()).max())
    max_groups = n_chunks * ((per_chunk * LS + batch_pairs - 1) // batch_pairs) + 1
    i32 = lambda m: torch.empty(m, dtype=torch.int32, device=dev)
    i64 = lambda m: torch.empty(m, dtype=torch.int64, device=dev)
    pcount, prank, counters = i32(n), i32(n), i32(34 * max_groups)
    # layout: 4 scalars + 4096 per-workgroup (records read, ray steps) statistics slots (no device-wide atomics)
    chunk_tab, group_tab, point_span, layout = i64(4 * n_chunks + 4), i64(2 * max_groups), i64(2 * n), i64(4 + 2 * 4096)
    pair_p, leaf_st, grp = i32(cap), i32(cap), i32(cap)
    pair_j = torch.empty(cap, dtype=torch.int16, device=dev)
    t_st = torch.empty(cap, dtype=torch.float32, device=dev)
    act_st = torch.empty(cap, dtype=torch.uint8, device=dev)
    common = (*T.args(), ptr(points), ptr(normals), ptr(chunk_id), c_long(n), c_int(n_chunks), ptr(dirs), ptr(wdir),
              ptr(wsum), c_int(L), c_int(nsamp), c_int(1 if argmax_vis else 0), c_long(batch_pairs), c_int(max_iter), ptr(pcount),
              ptr(prank), ptr(chunk_tab), ptr(group_tab), c_int(max_groups), ptr(counters), ptr(pair_p), ptr(pair_j), ptr(t_st),
              ptr(leaf_st), ptr(act_st), ptr(grp), ptr(point_span), ptr(layout))
    if OVIS_COMPACT and cap < 2 ** 31 - 1:
        nblk = cap // 2048 + 2
        alive_a, alive_b, blk_cnt = i32(cap), i32(cap), i32(nblk)
        flags = torch.empty(cap + 8, dtype=torch.uint8, device=dev)
        blk_off, n_alive = i64(nblk), i64(2)
        call("rb_dvis_octree", *common, ptr(alive_a), ptr(alive_b), ptr(flags), ptr(blk_cnt), ptr(blk_off), ptr(n_alive),
             ptr(out), ptr(eval_count), stream_ptr())
    else:           # the plain walk: no compaction scratch
        call("rb_dvis_octree", *common, *([ptr(None)] * 6), ptr(out), ptr(eval_count), stream_ptr())
    lay4 = layout[:4].clone()            # device tensor [pairs, groups, node records read, ray-iterations] of the last call
    lay4[2:4] = layout[4:].view(4096, 2).sum(0)
    LAST_OCTREE_VIS_LAYOUT = lay4
    return out


_dvis_octree_impl = dvis_octree      # the chunk groups recurse through this name: a wrapper of ops.dvis_octree sees one call per view


def octree_cast_grouped(T, origins, dirs, group_start, max_iter=32):
    """Secondary lock-step cast of explicit rays in independent groups (group g = rays group_start[g]:group_start[g+1], a device
    int64 tensor of G+1 ascending offsets): what the reference computes calling the tracer once per group."""
    origins, dirs = _f32(origins), _f32(dirs)
    R = dirs.shape[0]
    dev = dirs.device
    G = group_start.numel() - 1
    assert group_start.dtype == torch.int64 and G >= 1
    x = torch.empty(R, 3, dtype=torch.float32, device=dev)
    hit = torch.empty(R, dtype=torch.uint8, device=dev)
    t = torch.empty(R, dtype=torch.float32, device=dev)
    if R == 0:
        return x, hit.bool(), t
    gsize = torch.empty(G, dtype=torch.int64, device=dev)
    grp, leaf_st = (torch.empty(R, dtype=torch.int32, device=dev) for _ in range(2))
    t_st = torch.empty(R, dtype=torch.float32, device=dev)
    act_st = torch.empty(R, dtype=torch.uint8, device=dev)
    counters = torch.empty(34 * G, dtype=torch.int32, device=dev)
    call("rb_octree_cast_grouped", *T.args(), ptr(origins), ptr(dirs), c_long(R), ptr(group_start.contiguous()), c_int(G),
         c_int(max_iter), c_float(T.clamp_dt), ptr(gsize), ptr(grp), ptr(t_st), ptr(leaf_st), ptr(act_st), ptr(counters), ptr(x),
         ptr(hit), ptr(t), stream_ptr())
    return x, hit.bool(), t


def pose_matrix(pose):
    """The 4x4 camera-to-world matrix of a pose given as 4x4 (returned as is) or as the 7-vector (qr, qi, qj, qk | cam_loc) of
    get_camera_params' quaternion branch (utils/rend_util.py:52-57; quat_to_rot, :107-124, normalises the quaternion first).  A host
    array stays a host array, a device tensor a device tensor (eleven scalar products of one pose: parameter conversion, no kernel)."""
    import numpy as np
    is_t = isinstance(pose, torch.Tensor)
    if (pose.numel() if is_t else np.asarray(pose).size) != 7:
        return pose
    q = (pose.detach().float().reshape(7) if is_t else torch.from_numpy(np.asarray(pose, dtype=np.float32).reshape(7)))
    loc = q[4:]
    qr, qi, qj, qk = (q[:4] / torch.clamp(q[:4].norm(), min=1e-12)).unbind(0)
    rows = [torch.stack([1 - 2 * (qj ** 2 + qk ** 2), 2 * (qj * qi - qk * qr), 2 * (qi * qk + qr * qj), loc[0]]),
            torch.stack([2 * (qj * qi + qk * qr), 1 - 2 * (qi ** 2 + qk ** 2), 2 * (qj * qk - qi * qr), loc[1]]),
            torch.stack([2 * (qk * qi - qj * qr), 2 * (qj * qk + qi * qr), 1 - 2 * (qi ** 2 + qj ** 2), loc[2]]),
            torch.tensor([0.0, 0.0, 0.0, 1.0], device=q.device)]
    m = torch.stack(rows)
    return m if is_t else m.numpy()


def camera_rays(pose, K, uv):
    """pose [4,4] (or the 7-vector quaternion form: pose_matrix), K [3,3]: host arrays, or device tensors (then they are read on the
    device: no blocking copy); uv [N,2] device."""
    import numpy as np
    pose = pose_matrix(pose)
    if isinstance(pose, torch.Tensor) and pose.is_cuda and isinstance(K, torch.Tensor) and K.is_cuda:
        uv = _f32(uv)
        N = uv.shape[0]
        dirs = torch.empty(N, 3, dtype=torch.float32, device=uv.device)
        call("rb_camera_rays_dev", ptr(_f32(pose.detach()).reshape(16)), ptr(_f32(K.detach()).reshape(9)), ptr(uv), c_long(N),
             ptr(dirs), stream_ptr())
        return dirs
    # host arrays: uploaded (one small blocking copy; the per-chunk path hands device tensors over)
    to_np = lambda a: a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    uv = _f32(uv)
    p = torch.from_numpy(np.ascontiguousarray(to_np(pose), dtype=np.float32).reshape(16).copy()).to(uv.device)
    k = torch.from_numpy(np.ascontiguousarray(to_np(K), dtype=np.float32).reshape(9).copy()).to(uv.device)
    N = uv.shape[0]
    dirs = torch.empty(N, 3, dtype=torch.float32, device=uv.device)
    call("rb_camera_rays_dev", ptr(p), ptr(k), ptr(uv), c_long(N), ptr(dirs), stream_ptr())
    return dirs


def points_along(origins, dirs, t, batch=None):
    """origins [N,3] (per ray) or [N/batch,3]."""
    origins, dirs, t = _f32(origins), _f32(dirs), _f32(t)
    N = dirs.shape[0]
    pts = torch.empty(N, 3, dtype=torch.float32, device=dirs.device)
    per_ray = batch is None
    call("rb_points_along", ptr(origins), c_int(1 if per_ray else 0), c_long(1 if per_ray else batch), ptr(dirs), ptr(t),
         c_long(N), ptr(pts), stream_ptr())
    return pts


def tonemap(x, shift, mode):
    x = _f32(x)
    shift = _f32(shift).reshape(-1)
    n = x.shape[0]
    y = torch.empty_like(x)
    stride = 0 if shift.numel() == 1 else 1
    assert stride == 0 or shift.numel() == n
    call("rb_tonemap", ptr(x), c_long(n), ptr(shift), c_int(stride), c_int(mode), ptr(y), stream_ptr())
    return y


# ------------------------------------------------------------------------------------------------ secondary rays / NeuS
def sphere_dirs(u1, u2, normals, points, nsamp):
    u1, u2, normals, points = _f32(u1).reshape(-1), _f32(u2).reshape(-1), _f32(normals), _f32(points)
    n = points.shape[0]
    dev = points.device
    dirs = torch.empty(n * nsamp, 3, dtype=torch.float32, device=dev)
    back = torch.empty(n * nsamp, dtype=torch.uint8, device=dev)
    cosw = torch.empty(n * nsamp, dtype=torch.float32, device=dev)
    origins = torch.empty(n, 3, dtype=torch.float32, device=dev)
    call("rb_sphere_dirs", ptr(u1), ptr(u2), ptr(normals), ptr(points), c_long(n), c_int(nsamp), ptr(dirs), ptr(back),
         ptr(cosw), ptr(origins), stream_ptr())
    return dirs, back, cosw, origins


def borrow_points(points, view, tk):
    points, view, tk = _f32(points), _f32(view), _f32(tk)
    m, ns = points.shape[0], tk.numel()
    x = torch.empty(m * ns, 3, dtype=torch.float32, device=points.device)
    d = torch.empty(m * ns, 3, dtype=torch.float32, device=points.device)
    call("rb_borrow_points", ptr(points), ptr(view), ptr(tk), c_long(m), c_int(ns), ptr(x), ptr(d), stream_ptr())
    return x, d


def neus_composite(sdf, color, inv_s, lo=0.0, hi=1.0, eps=1e-7, mask=None, want_weights=False):
    """sdf [m,ns]; color [m,ns,3] or None."""
    sdf = _f32(sdf)
    m, ns = sdf.shape
    dev = sdf.device
    rgb = torch.empty(m, 3, dtype=torch.float32, device=dev) if color is not None else None
    w = torch.empty(m, ns, dtype=torch.float32, device=dev) if want_weights else None
    call("rb_neus_composite", ptr(sdf), ptr(_f32(color) if color is not None else None),
         ptr(_f32(mask) if mask is not None else None), c_long(m), c_int(ns), c_float(inv_s), c_float(lo), c_float(hi),
         c_float(eps), ptr(rgb), ptr(w), stream_ptr())
    return rgb, w


def trace_integrate(rad, cosw, back, n, nsamp):
    out = torch.empty(n, 3, dtype=torch.float32, device=rad.device)
    call("rb_trace_integrate", ptr(_f32(rad)), ptr(cosw), ptr(back), c_long(n), c_int(nsamp), ptr(out), stream_ptr())
    return out


def envmap_sg(lgt, dirs):
    lgt, dirs = _f32(lgt), _f32(dirs)
    n = dirs.shape[0]
    rgb = torch.empty(n, 3, dtype=torch.float32, device=dirs.device)
    call("rb_envmap_sg", ptr(lgt), c_int(lgt.shape[0]), ptr(dirs), c_long(n), ptr(rgb), stream_ptr())
    return rgb


def envmap_lookup(env, dirs):
    env, dirs = _f32(env), _f32(dirs)
    H, W = env.shape[0], env.shape[1]
    n = dirs.shape[0]
    rgb = torch.empty(n, 3, dtype=torch.float32, device=dirs.device)
    call("rb_envmap_lookup", ptr(env), c_int(H), c_int(W), ptr(dirs), c_long(n), ptr(rgb), stream_ptr())
    return rgb


def cesr_net(X, M, kind, blob, n_label=1):
    """kind 0 normal_net (X[M,64] -> [M,3]); 1 shadow_net dense rows (X[M,192] -> [M,2]);
    2 shadow_net on (point, label) pairs (X = point features [M/n_label,64], M rows -> [M,2])."""
    n_out = 3 if kind == 0 else 2
    Y = torch.empty(M, n_out, dtype=torch.float32, device=X.device)
    call("rb_cesr_net", ptr(_f32(X)), c_long(M), c_int(kind), c_int(n_label), ptr(blob), ptr(Y), stream_ptr())
    return Y


def cesr_net_h3(X, M, kind, blob, scale_log2, n_label=1):
    n_out = 3 if kind == 0 else 2
    Y = torch.empty(M, n_out, dtype=torch.float32, device=X.device)
    call("rb_cesr_net_h3", ptr(_f32(X)), c_long(M), c_int(kind), c_int(n_label), ptr(blob), c_int(scale_log2), ptr(Y),
         stream_ptr())
    return Y


def cesr_net_points(x, M, kind, blob, n_label=1, scale_log2=None, ring=None):
    """cesr_net / cesr_net_h3 on PE10(x) with the encoding fused (kind 0: M = points; kind 2: M = points * n_label rows).
    ring (split precision; default: by batch size): the chunk-stream kernel (csrc/wide_ring.h) -- bit-identical outputs."""
    x = _f32(x)
    Y = torch.empty(M, 3 if kind == 0 else 2, dtype=torch.float32, device=x.device)
    if ring is None:
        ring = M >= WIDE_RING_MIN_ROWS
    if scale_log2 is None:
        call("rb_cesr_net_points", ptr(x), c_long(M), c_int(kind), c_int(n_label), ptr(blob), ptr(Y), stream_ptr())
    elif ring:
        call("rb_cesr_net_ring_points", ptr(x), c_long(M), c_int(kind), c_int(n_label), ptr(blob), c_int(scale_log2), ptr(Y), c_int(0),
             stream_ptr())
    else:
        call("rb_cesr_net_h3_points", ptr(x), c_long(M), c_int(kind), c_int(n_label), ptr(blob), c_int(scale_log2), ptr(Y), stream_ptr())
    return Y


def scatter_rows(srcs, dst_widths, idx, N, fill=1.0):
    """Per-pixel outputs in one launch: srcs[k] [n, 1 or w_k] rows of the pixels idx -> list of K tensors [N, w_k] (other rows =
    fill); entries of srcs may be None (a block that only holds the default).  One allocation, one fill, one scatter."""
    dev = idx.device
    total = sum(dst_widths)
    flat = torch.full((N * total,), fill, dtype=torch.float32, device=dev)
    outs, off = [], 0
    for w in dst_widths:
        outs.append(flat[off:off + N * w].view(N, w))
        off += N * w
    n = idx.shape[0]
    live = [k for k, s in enumerate(srcs) if s is not None]
    if n > 0 and live:
        # blocks without a source must come last in the scatter's view of the buffer: keep the order, skip by offset instead
        assert live == list(range(len(live))), "sources first, default-only blocks last"
        keep = [_f32(srcs[k]) for k in live]
        K = len(keep)
        src_arr = (ctypes.c_void_p * K)(*[t.data_ptr() for t in keep])
        sw = (ctypes.c_int * K)(*[int(t.shape[-1]) for t in keep])
        dw = (ctypes.c_int * K)(*[int(dst_widths[k]) for k in live])
        call("rb_scatter_rows", src_arr, sw, dw, c_int(K), ptr(idx), c_long(n), c_long(N), ptr(flat), stream_ptr())
    return outs


def cesr_net_x6_points(x, M, kind, blob, n_label=1):
    """cesr_net_points on exact three-piece operands (csrc/cesr_x6.hip; blob = packing.pack_softplus512_x6)."""
    x = _f32(x)
    Y = torch.empty(M, 3 if kind == 0 else 2, dtype=torch.float32, device=x.device)
    if M > 0:
        call("rb_cesr_net_x6_points", ptr(x), c_long(M), c_int(kind), c_int(n_label), ptr(blob), ptr(Y), c_int(0), stream_ptr())
    return Y


CESR_F16_TILES = int(os.environ.get("ROBIR_CESR_F16_TILES", "3"))        # 16-row tiles per wave of the f16 CESR kernel (what the library is built with: csrc/cesr_f16.hip FX_TILES)


def cesr_net_f16_points(x, M, kind, blob, n_label=1):
    """The CESR nets in PLAIN f16 (csrc/cesr_f16.hip; blob = packing.pack_softplus512_f16): the f16 throughput mode -- NARROWER than fp32."""
    x = _f32(x)
    Y = torch.empty(M, 3 if kind == 0 else 2, dtype=torch.float32, device=x.device)
    if M > 0:
        call("rb_cesr_net_f16_points", ptr(x), c_long(M), c_int(kind), c_int(n_label), ptr(blob), ptr(Y), c_int(CESR_F16_TILES), c_int(0),
             stream_ptr())
    return Y


def material_decode(brdf, brdf_r):
    brdf, brdf_r = _f32(brdf), _f32(brdf_r)
    n, dev = brdf.shape[0], brdf.device
    mk = lambda c: torch.empty(n, c, dtype=torch.float32, device=dev)
    alb, rough, metal, alb_r, rough_r, metal_r = mk(3), mk(1), mk(1), mk(3), mk(1), mk(1)
    call("rb_material_decode", ptr(brdf), ptr(brdf_r), c_long(n), ptr(alb), ptr(rough), ptr(metal), ptr(alb_r), ptr(rough_r),
         ptr(metal_r), stream_ptr())
    return alb, rough, metal, alb_r, rough_r, metal_r


def abs_scale(x, s=1.0, take_abs=True):
    x = _f32(x)
    y = torch.empty_like(x)
    call("rb_abs_scale", ptr(x), c_long(x.numel()), c_float(s), c_int(1 if take_abs else 0), ptr(y), stream_ptr())
    return y


def softmax2(logits, which=1):
    logits = _f32(logits)
    n = logits.shape[0]
    p = torch.empty(n, dtype=torch.float32, device=logits.device)
    call("rb_softmax2", ptr(logits), c_long(n), c_int(which), ptr(p), stream_ptr())
    return p


def lin_diff_combine(diffuse, albedo, spec):
    diffuse, albedo, spec = _f32(diffuse), _f32(albedo), _f32(spec)
    rgb = torch.empty_like(diffuse)
    call("rb_lin_diff_combine", ptr(diffuse), ptr(albedo), ptr(spec), c_long(diffuse.shape[0]), ptr(rgb), stream_ptr())
    return rgb


# ---- IDR sphere tracer (use_octree=False) -----------------------------------------------------------------------------
class RayTraceState:
    """Device state of rb_raytrace_step for N rays (layout in include/robir_hip.h)."""

    def __init__(self, cam, dirs):
        self.cam, self.dirs = _f32(cam).reshape(-1, 3), _f32(dirs)
        self.N = N = self.dirs.shape[0]
        assert self.cam.shape[0] in (1, N)
        self.cs = 0 if self.cam.shape[0] == 1 else 3
        dev = dirs.device
        self.f = torch.zeros(6, N, dtype=torch.float32, device=dev)
        self.b = torch.zeros(4, N, dtype=torch.uint8, device=dev)
        self.pts = torch.zeros(2 * N, 3, dtype=torch.float32, device=dev)
        self.ctrl = torch.zeros(2, dtype=torch.int32, device=dev)

    def step(self, op, param=0.0, sdf2=None):
        s = _f32(sdf2) if sdf2 is not None else None
        call("rb_raytrace_step", c_int(op), ptr(self.cam), c_int(self.cs), ptr(self.dirs), c_long(self.N), c_float(param),
             ptr(s), ptr(self.f), ptr(self.b), ptr(self.pts), ptr(self.ctrl), stream_ptr())


def raytrace_samples(cam, dirs, lo, hi, lin):
    cam, dirs, lo, hi, lin = _f32(cam).reshape(-1, 3), _f32(dirs), _f32(lo), _f32(hi), _f32(lin)
    m, n = dirs.shape[0], lin.shape[0]
    z = torch.empty(m, n, dtype=torch.float32, device=dirs.device)
    P = torch.empty(m * n, 3, dtype=torch.float32, device=dirs.device)
    call("rb_raytrace_samples", ptr(cam), c_int(0 if cam.shape[0] == 1 else 3), ptr(dirs), ptr(lo), ptr(hi), ptr(lin),
         c_long(m), c_int(n), ptr(z), ptr(P), stream_ptr())
    return z, P


def raytrace_pick(sdf, z, P, obj):
    m, n = z.shape
    dev = z.device
    obj = obj.to(torch.uint8).contiguous()
    pts = torch.empty(m, 3, dtype=torch.float32, device=dev)
    dist = torch.empty(m, dtype=torch.float32, device=dev)
    hit = torch.empty(m, dtype=torch.uint8, device=dev)
    bracket = torch.empty(4, m, dtype=torch.float32, device=dev)
    call("rb_raytrace_pick", ptr(_f32(sdf)), ptr(z), ptr(P), ptr(obj), c_long(m), c_int(n), ptr(pts), ptr(dist), ptr(hit),
         ptr(bracket), stream_ptr())
    return pts, dist, hit, bracket


def raytrace_secant(cam, dirs, on, smid, phase, bracket, zp, pmid):
    cam = _f32(cam).reshape(-1, 3)
    s = _f32(smid) if smid is not None else None
    call("rb_raytrace_secant", ptr(cam), c_int(0 if cam.shape[0] == 1 else 3), ptr(dirs), ptr(on), ptr(s),
         c_long(dirs.shape[0]), c_int(phase), ptr(bracket), ptr(zp), ptr(pmid), stream_ptr())


# ------------------------------------------------------------------------------------------------ public helper names (csrc/surface.hip)
def pe_encode(x, freq, include_input=True):
    """x [n,d], freq [n_freq] device band values -> [n, (d if include_input) + 2 d n_freq]."""
    x, freq = _f32(x), _f32(freq)
    n, d = x.shape
    out = torch.empty(n, (d if include_input else 0) + 2 * d * freq.shape[0], dtype=torch.float32, device=x.device)
    call("rb_pe_encode", ptr(x), c_long(n), c_int(d), ptr(freq), c_int(freq.shape[0]), c_int(1 if include_input else 0), ptr(out),
         stream_ptr())
    return out


def expected_sin(x, var, want_var=True):
    x, var = _f32(x), _f32(var)
    assert x.shape == var.shape
    y = torch.empty_like(x)
    yv = torch.empty_like(x) if want_var else None
    call("rb_expected_sin", ptr(x), ptr(var), c_long(x.numel()), ptr(y), ptr(yv), stream_ptr())
    return y, yv


def tonemap_curve(x, shift, curve):
    """Free functions of model/color_correction.py:31-73 (no clamp).  x any shape; shift None, one value, [..., 1] against x [..., w], or
    anything that broadcasts to x's shape."""
    x = _f32(x)
    y = torch.empty_like(x)
    width, stride, sh = 1, 0, None
    if shift is not None:
        sh = shift.to(device=x.device, dtype=torch.float32)
        if sh.numel() == 1:
            sh = sh.reshape(1).contiguous()
        elif sh.dim() == x.dim() and sh.shape[-1] == 1 and sh.shape[:-1] == x.shape[:-1]:
            width, stride, sh = x.shape[-1], 1, sh.contiguous()
        else:
            width, stride, sh = 1, 1, sh.expand(x.shape).contiguous()
    call("rb_tonemap_curve", ptr(x), c_long(x.numel()), c_int(width), ptr(sh), c_int(stride), c_int(curve), ptr(y), stream_ptr())
    return y


def sample_pdf(bins, weights, u):
    """bins [R,n], weights [R,n-1], u [n_s] (shared) or [R,n_s] -> (samples [R,n_s], cdf [R,n])."""
    bins, weights, u = _f32(bins), _f32(weights), _f32(u)
    R, n = bins.shape
    assert weights.shape == (R, n - 1)
    n_s = u.shape[-1]
    assert u.dim() == 1 or u.shape == (R, n_s)
    cdf = torch.empty(R, n, dtype=torch.float32, device=bins.device)
    out = torch.empty(R, n_s, dtype=torch.float32, device=bins.device)
    call("rb_sample_pdf", ptr(bins), ptr(weights), c_long(R), c_int(n), ptr(u), c_long(0 if u.dim() == 1 else n_s), c_int(n_s), ptr(cdf),
         ptr(out), stream_ptr())
    return out, cdf


def neus_core_aux(sdf, sdf_stride, pts, z, R, n, inv_s, radius, sample_dist):
    dev = pts.device
    dists, cdf, inside = (torch.empty(R, n, dtype=torch.float32, device=dev) for _ in range(3))
    call("rb_neus_core_aux", ptr(sdf), c_long(sdf_stride), ptr(pts), ptr(z), c_long(R), c_int(n), c_float(inv_s), c_float(radius),
         c_float(sample_dist), ptr(dists), ptr(cdf), ptr(inside), stream_ptr())
    return dists, cdf, inside


def sample_dirs(normals, theta, phi):
    normals, theta, phi = _f32(normals), _f32(theta), _f32(phi)
    n = theta.numel()
    assert normals.numel() == 3 * n and phi.numel() == n
    out = torch.empty(n, 3, dtype=torch.float32, device=normals.device)
    call("rb_sample_dirs", ptr(normals), ptr(theta), ptr(phi), c_long(n), ptr(out), stream_ptr())
    return out


def intersect_sphere(origins, dirs, radius):
    origins, dirs = _f32(origins), _f32(dirs)
    n = origins.shape[0]
    out = torch.empty(n, 3, dtype=torch.float32, device=origins.device)
    call("rb_intersect_sphere", ptr(origins), ptr(dirs), c_long(n), c_float(radius), ptr(out), stream_ptr())
    return out
