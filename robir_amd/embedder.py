"""Positional encodings with the reference's call surface (model/embedder.py:7-61, model/neus_model.py:71-94,136-184),
evaluated by the HIP feature kernels (accurate sinf/cosf)."""
import torch

from . import ops


class Embedder:
    """get_embedder's object: .embed(x) and .out_dim (include_input, log-sampled bands, sin/cos)."""

    def __init__(self, multires, input_dims=3):
        if input_dims != 3 or multires != 10:
            raise NotImplementedError("HIP PE kernel: 3-D inputs, L = 10 (the 4-band view encoding is fused into "
                                      "rb_feat_color)")
        self.multires, self.out_dim = multires, 3 * (1 + 2 * multires)

    def embed(self, x):
        shape = list(x.shape[:-1]) + [self.out_dim]
        return ops.feat_pe10(x.reshape(-1, 3).float().contiguous())[:, :self.out_dim].reshape(shape)


def get_embedder(multires, input_dims=3):
    e = Embedder(multires, input_dims)
    return (lambda x, eo=e: eo.embed(x)), e.out_dim


PE = Embedder


def isotropic_cov(mean, var, d_in=3):
    return torch.eye(d_in, device=mean.device).expand(list(mean.shape[:-1]) + [d_in, d_in]) * var


class IPE:
    def __init__(self, min_deg=0, max_deg=16, in_dim=3, diag=True):
        if min_deg != 0 or max_deg != 10 or in_dim != 3:
            raise NotImplementedError("HIP IPE kernel: degrees 0..9, 3-D inputs")
        self.max_deg = max_deg

    def feature_dim(self):
        return 60

    def __call__(self, mean, cov):
        var = float(cov.reshape(-1, 3, 3)[0, 0, 0]) if cov.numel() else 0.0
        shape = list(mean.shape[:-1]) + [60]
        return ops.feat_ipe(mean.reshape(-1, 3).float().contiguous(), var)[:, :60].reshape(shape)


def ipe_embedder(multires, var=0.005):
    ipe = IPE(max_deg=multires)
    return (lambda x: ops.feat_ipe(x.reshape(-1, 3).float().contiguous(), var)[:, :60]), ipe.feature_dim()
