"""Positional encodings with the reference's call surface, evaluated by HIP kernels (accurate sinf / cosf).

Two reference modules export encodings and both are mirrored here name for name:
  model/embedder.py      Embedder(**kwargs) / get_embedder(multires, input_dims=3) / ipe_embedder(multires, var=0.005)     (:7-61)
  model/neus_model.py    expected_sin / integrated_pos_enc / isotropic_cov / IPE / PE / get_embedder(..., Embedder=PE, windowed=False)
                         (:14-94, 136-224, 303-309) -- exported by overlay/model/neus_model.py under those names (the `Neus*` spellings
                         below only keep the two get_embedder / Embedder families apart inside this file).
The fused network kernels encode their inputs themselves (csrc/mlp_engine.h); these objects serve callers that want the code
rows: L = 10 on 3-D points takes the 64-column feature kernel the nets were validated with, every other (input_dims, bands)
combination the general kernel rb_pe_encode (csrc/surface.hip).  tinycudann hash grids and the windowed (scheduled) encodings
are training-time machinery of stage 1: present, raising NotImplementedError."""
import torch
import torch.nn as nn

from . import ops

_OUT_OF_SCOPE_TCNN = ("tinycudann hash-grid / fully-fused-MLP path: never instantiated by a shipped configuration (hashing=False, "
                      "model/neus_model.py:774) and tinycudann is a CUDA-only dependency -- OUT OF SCOPE (SURVEY.md section 2 row 2)")


def _bands(max_freq_log2, num_freqs, log_sampling):
    if log_sampling:
        return 2. ** torch.linspace(0., max_freq_log2, steps=num_freqs)
    return torch.linspace(2. ** 0., 2. ** max_freq_log2, steps=num_freqs)


class _Encoder:
    """Shared body of Embedder / PE: kwargs = the reference's embed_kwargs dict."""

    def create_embedding_fn(self):
        kw = self.kwargs
        fns = kw.get("periodic_fns", [torch.sin, torch.cos])
        if [getattr(f, "__name__", "") for f in fns] != ["sin", "cos"]:
            raise NotImplementedError("HIP positional encoding: periodic_fns = [torch.sin, torch.cos]")
        d, n = int(kw["input_dims"]), int(kw["num_freqs"])
        self._freq_host = _bands(kw["max_freq_log2"], n, kw["log_sampling"]).float() if n > 0 else torch.zeros(0)
        self._freq_dev = {}
        self.out_dim = (d if kw["include_input"] else 0) + 2 * d * n
        # the reference keeps one closure per output block; here the whole row comes from one kernel launch
        self.embed_fns = [self._encode]

    def _encode(self, inputs):
        kw = self.kwargs
        d = int(kw["input_dims"])
        shape = list(inputs.shape[:-1]) + [self.out_dim]
        x = inputs.reshape(-1, d).float().contiguous()
        if d == 3 and kw["num_freqs"] == 10 and kw["include_input"] and kw["log_sampling"] and kw["max_freq_log2"] == 9:
            return ops.feat_pe10(x)[:, :63].reshape(shape)          # the feature kernel of the 10-band nets
        f = self._freq_dev.get(x.device)
        if f is None:
            f = self._freq_dev[x.device] = self._freq_host.to(x.device)
        return ops.pe_encode(x, f, bool(kw["include_input"])).reshape(shape)

    def embed(self, inputs):
        return torch.cat([fn(inputs) for fn in self.embed_fns], -1)


# ------------------------------------------------------------------------------------------------ model/embedder.py
class Embedder(_Encoder):
    """model/embedder.py:7-38."""

    def __init__(self, **kwargs):
        self.kwargs = kwargs
        self.create_embedding_fn()


def get_embedder(multires, input_dims=3):
    """model/embedder.py:41-55 -> (embed function, out_dim)."""
    e = Embedder(include_input=True, input_dims=input_dims, max_freq_log2=multires - 1, num_freqs=multires, log_sampling=True,
                 periodic_fns=[torch.sin, torch.cos])
    return (lambda x, eo=e: eo.embed(x)), e.out_dim


def ipe_embedder(multires, var=0.005):
    """model/embedder.py:58-61."""
    ipe = IPE(max_deg=multires)
    return (lambda x: ipe(x, isotropic_cov(x, var))), ipe.feature_dim()


# ------------------------------------------------------------------------------------------------ model/neus_model.py
def expected_sin(x, x_var):
    """model/neus_model.py:14-24: mean and variance of sin(z), z ~ N(x, x_var)."""
    shape = x.shape
    x_var = x_var.expand(shape) if isinstance(x_var, torch.Tensor) else torch.full(shape, float(x_var), device=x.device)
    y, yv = ops.expected_sin(x.float().contiguous().reshape(-1), x_var.float().contiguous().reshape(-1))
    return y.reshape(shape), yv.reshape(shape)


def integrated_pos_enc(x_coord, min_deg, max_deg, diag=False):
    """model/neus_model.py:27-57.  (IPE.forward reaches the diag=False branch with per-axis variances -- a reference quirk kept as is.)"""
    if diag:
        x, var_diag = x_coord
    else:
        # full covariances: the reference multiplies by a block matrix of scaled identities (x @ basis, diag(basis^T cov basis)); every
        # column holds ONE non-zero product, so the values are exactly the per-axis scalings below -- no GEMM needed
        x, x_cov = x_coord
        var_diag = torch.diagonal(x_cov, 0, -2, -1)
    scales = torch.tensor([2 ** i for i in range(min_deg, max_deg)], device=x.device)
    shape = list(x.shape[:-1]) + [-1]
    y = torch.reshape(x[..., None, :] * scales[:, None], shape)
    y_var = torch.reshape(var_diag[..., None, :] * scales[:, None] ** 2, shape)
    return expected_sin(torch.cat([y, y + 0.5 * torch.pi], dim=-1), torch.cat([y_var] * 2, dim=-1))[0]


def isotropic_cov(mean, var, d_in=3):
    """model/neus_model.py:60-68."""
    if isinstance(var, torch.Tensor):
        var = var.view(-1, 1, 1)
    init_shape = list(mean.shape[:-1])
    cov = torch.eye(d_in, device=mean.device) * var
    if cov.dim() == 2:
        cov = cov[None]
    return cov.expand(mean.reshape(-1, d_in).shape[0], -1, -1).reshape(init_shape + [d_in, d_in])


class IPE(nn.Module):
    """model/neus_model.py:71-94."""

    def __init__(self, min_deg=0, max_deg=16, in_dim=3, diag=True):
        super().__init__()
        self.min_deg, self.max_deg, self.diag, self.in_dim = min_deg, max_deg, diag, in_dim

    def forward(self, mean, cov):
        init_shape = list(mean.shape[:-1]) + [-1]
        mean = mean.reshape(-1, mean.shape[-1])
        cov = cov.reshape(-1, *cov.shape[-2:])
        if not self.diag:
            cov = torch.diagonal(cov, 0, 1, 2)
        return integrated_pos_enc((mean, cov), self.min_deg, self.max_deg).reshape(init_shape)

    def feature_dim(self) -> int:
        return (self.max_deg - self.min_deg) * 2 * self.in_dim


class TCNNLinear(nn.Module):
    def __init__(self, in_dim, out_dim, **kwargs):
        raise NotImplementedError("TCNNLinear: " + _OUT_OF_SCOPE_TCNN)

    def forward(self, x):
        raise NotImplementedError("TCNNLinear: " + _OUT_OF_SCOPE_TCNN)


def tcnn_encoding(max_level, n_feature, in_dim, hashmap_size=18):
    raise NotImplementedError("tcnn_encoding: " + _OUT_OF_SCOPE_TCNN)


class PE(_Encoder, nn.Module):
    """model/neus_model.py:136-224 (gin-registered there)."""

    def __init__(self, input_dims=3, num_freq=10, include_input=True, log_sampling=True, schedule=None):
        nn.Module.__init__(self)
        if schedule is not None:
            raise NotImplementedError("PE(schedule=...): the windowed (coarse-to-fine) encoding is driven by stage-1's training-step "
                                      "Curve (utils/schedule.py) -- training machinery, OUT OF SCOPE (SURVEY.md section 2 row 22)")
        self.kwargs = {"input_dims": input_dims, "include_input": include_input, "max_freq_log2": num_freq - 1, "num_freqs": num_freq,
                       "log_sampling": log_sampling, "periodic_fns": [torch.sin, torch.cos]}
        self.create_embedding_fn()
        self.window_curve = None

    def forward(self, inputs):
        return self.embed(inputs)

    def feature_dim(self) -> int:
        return self.out_dim

    def windowed_embed(self, x):
        code = self.embed(x)
        if self.window_curve is None:
            return code
        raise NotImplementedError("windowed PE: needs a stage-1 training schedule (OUT OF SCOPE, SURVEY.md section 2 row 22)")

    def get_cosine_easing_window(self):
        if self.window_curve is None:
            raise NotImplementedError("get_cosine_easing_window: no schedule (stage-1 training machinery, OUT OF SCOPE, SURVEY.md section 2 row 22)")
        return self.cosine_easing_window(0, self.kwargs["max_freq_log2"], self.kwargs["num_freqs"], self.window_curve())

    @classmethod
    def cosine_easing_window(cls, min_freq_log2, max_freq_log2, num_bands, alpha):
        """A Tukey window sliding over the bands (host-side schedule arithmetic on num_bands values, model/neus_model.py:203-224)."""
        if max_freq_log2 is None:
            max_freq_log2 = num_bands - 1.0
        x = torch.clip(alpha - torch.linspace(min_freq_log2, max_freq_log2, num_bands), 0.0, 1.0)
        return 0.5 * (1 + torch.cos(torch.pi * x + torch.pi))


class Hash(nn.Module):
    def __init__(self, n_levels=16, n_features=2, in_dim=3, schedule=None, bbox=None):
        raise NotImplementedError("Hash: " + _OUT_OF_SCOPE_TCNN)

    def forward(self, x):
        raise NotImplementedError("Hash: " + _OUT_OF_SCOPE_TCNN)

    def feature_dim(self) -> int:
        raise NotImplementedError("Hash: " + _OUT_OF_SCOPE_TCNN)

    def windowed_embed(self, x):
        raise NotImplementedError("Hash: " + _OUT_OF_SCOPE_TCNN)

    def get_cosine_easing_window(self):
        raise NotImplementedError("Hash: " + _OUT_OF_SCOPE_TCNN)

    @classmethod
    def cosine_easing_window(cls, min_freq_log2, max_freq_log2, num_bands, alpha):
        return PE.cosine_easing_window(min_freq_log2, max_freq_log2, num_bands, alpha)


def get_embedder_neus(multires, input_dims=3, Embedder=PE, windowed=False):
    """model/neus_model.py:303-309 (exported as model.neus_model.get_embedder)."""
    embedder_obj = Embedder(input_dims=input_dims, num_freq=multires)

    def embed(x, eo=embedder_obj):
        return eo.embed(x) if not windowed else eo.windowed_embed(x)

    return embed, embedder_obj.out_dim
