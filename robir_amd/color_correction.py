"""Tone-mapping curves of model/color_correction.py:31-73 as free functions on the HIP kernel rb_tonemap_curve (csrc/surface.hip).
ACESToneMapping / GammaCorrect (the objects the forward uses, with the clamp of make_shift and the fused per-row kernel) live in
robir_amd/nets.py and are re-exported here so that this module mirrors the reference's export list one to one."""
import torch

from . import ops
from .nets import ACESToneMapping, GammaCorrect  # noqa: F401


def _t(x, like=None):
    if not isinstance(x, torch.Tensor):
        x = torch.tensor(x, dtype=torch.float32, device=like.device if isinstance(like, torch.Tensor) else "cuda")
    return x.float()


def _curve(code, x, t=None):
    x = _t(x, t)
    return ops.tonemap_curve(x.contiguous(), None if t is None else _t(t, x), code)


def aces_fn(x):
    """x (2.51 x + 0.03) / (x (2.43 x + 0.59) + 0.14)          (:31-34)"""
    return _curve(0, x)


def aces_inv(x):
    """the positive root of aces_fn's quadratic                  (:37-41)"""
    return _curve(1, x)


def warp_aces_inv(x, t):  # hdr with energy
    """0.73 aces_inv(x t) / aces_inv(0.73 t)                     (:44-45)"""
    return _curve(3, x, t)


def warp_aces_fn(x, t):
    """aces_fn(aces_inv(0.73 t) / 0.73 x) / t                    (:48-49)"""
    return _curve(2, x, t)


def scale_aces_inv(x, t):
    """aces_inv(x t^0.2)  -- hdr_mode 0, every shipped conf      (:52-54)"""
    return _curve(5, x, t)


def scale_aces_fn(x, t):
    """aces_fn(x) / t^0.2                                        (:57-59)"""
    return _curve(4, x, t)


def identity_fn(x, t):
    return x


def ln_space_fn(x, shift):
    """u = x (0.5 + shift) / 0.5;  u / (1 + shift u)             (:66-68)"""
    return _curve(7, x, shift)


def ln_space_inv(x, shift):
    """y = x / (1 - shift x);  y 0.5 / (0.5 + shift)             (:71-73)"""
    return _curve(8, x, shift)
