"""Positional encodings.  model/embedder.py:7-55, model/neus_model.py:136-184 (PE),
model/neus_model.py:14-57,71-91 + model/embedder.py:58-61 (full-covariance IPE)."""
import math

import torch


def pe(x, n_freq):
    """[..., d] -> [..., d*(1+2L)] laid out [x | sin(2^0 x) | cos(2^0 x) | sin(2^1 x) | ...]
    (embedder.py:17-38; bands 2**linspace(0, L-1, L) are exact powers of two)."""
    freqs = torch.tensor([2.0 ** k for k in range(n_freq)], dtype=x.dtype, device=x.device)
    ang = x.unsqueeze(-2) * freqs[:, None]                      # [..., L, d]
    waves = torch.stack([torch.sin(ang), torch.cos(ang)], -2)   # [..., L, 2, d]
    return torch.cat([x, waves.flatten(-3)], -1)


def pe_dim(n_freq, d=3):
    return d * (1 + 2 * n_freq)


def ipe_isotropic(x, var, n_deg=10):
    """Integrated PE with isotropic covariance var*I, full-covariance code path of the reference
    (neus_model.py:45-57 with diag=False; embedder.py:58-61).  [n,3] -> [n, 6*n_deg]:
    [exp(-v_k/2) sin(2^k x) for all k,c | exp(-v_k/2) sin(2^k x + pi/2) for all k,c],
    v_k = var*4^k, arguments wrapped mod 100*pi once |arg| >= 100*pi (neus_model.py:14-22)."""
    scales = torch.tensor([2.0 ** k for k in range(n_deg)], dtype=x.dtype, device=x.device)
    y = (x.unsqueeze(-2) * scales[:, None]).flatten(-2)         # [n, 3*n_deg], index 3k+c
    v = torch.tensor(var, dtype=x.dtype) * (scales * scales)    # fp32(var) * 4^k  (exact scaling)
    v = v.repeat_interleave(x.shape[-1])
    arg = torch.cat([y, y + 0.5 * math.pi], -1)
    big = 100.0 * math.pi
    arg = torch.where(arg.abs() < big, arg, torch.remainder(arg, big))
    damp = torch.exp(-0.5 * torch.cat([v, v], -1))
    return damp * torch.sin(arg)
