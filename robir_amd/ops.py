"""Thin Python bindings over the C-ABI (include/robir_hip.h): allocate outputs with torch, pass raw pointers,
sizes and the current HIP stream.  No arithmetic happens here."""
import ctypes

import torch

from . import _lib
from ._lib import ptr, stream_ptr, call

c_long, c_int, c_float = ctypes.c_long, ctypes.c_int, ctypes.c_float


def _f32(t):
    assert t.dtype == torch.float32, t.dtype
    return t.contiguous()


def feat_vis(p, d):
    p, d = _f32(p), _f32(d)
    M = p.shape[0]
    X = torch.empty(M, 128, dtype=torch.float32, device=p.device)
    call("rb_feat_vis", ptr(p), ptr(d), c_long(M), ptr(X), stream_ptr())
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


def vis_mlp(X, blob):
    M = X.shape[0]
    Y = torch.empty(M, 2, dtype=torch.float32, device=X.device)
    call("rb_vis_mlp", ptr(X), c_long(M), ptr(blob), ptr(Y), stream_ptr())
    return Y


def linear_64_256(X, blob):
    M = X.shape[0]
    Y = torch.empty(M, 256, dtype=torch.float32, device=X.device)
    call("rb_linear_64_256", ptr(X), c_long(M), ptr(blob), ptr(Y), stream_ptr())
    return Y


def sdf_mlp(X, M, blob, mode, out_scale=1.0, grad_scale=1.0):
    full = mode in (1, 3)
    out0 = torch.empty((M, 257) if full else (M,), dtype=torch.float32, device=X.device)
    grad = torch.empty(M, 3, dtype=torch.float32, device=X.device) if mode >= 2 else None
    call("rb_sdf_mlp", ptr(X), c_long(M), ptr(blob), c_int(mode), c_float(out_scale), c_float(grad_scale), ptr(out0),
         ptr(grad), stream_ptr())
    return out0, grad


def color_mlp(X, blob):
    M = X.shape[0]
    Y = torch.empty(M, 3, dtype=torch.float32, device=X.device)
    call("rb_color_mlp", ptr(X), c_long(M), ptr(blob), ptr(Y), stream_ptr())
    return Y


def illum_mlp(X, blob):
    M = X.shape[0]
    raw = torch.empty(M, 144, dtype=torch.float32, device=X.device)
    call("rb_illum_mlp", ptr(X), c_long(M), ptr(blob), ptr(raw), stream_ptr())
    return raw


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
