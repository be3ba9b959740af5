"""The small MLPs of the hot path, as pure functions of a weight dict (reference state-dict keys).

SDF / colour: model/neus_model.py:312-438, 489-560, 755-818.
Visibility / indirect illumination: model/implicit_differentiable_renderer.py:170-258.
Sparse auto-encoders / materials: model/sg_envmap_material.py:40-99, 188-247.
"""
import math

import torch
import torch.nn.functional as F

from .encoding import pe, ipe_isotropic

SDF = "implicit_network.neus_model.sdf_network."
COL = "implicit_network.neus_model.color_network."
VIS = "visibility_network.vis_layer."
ILL = "indirect_illum_network."
MAT = "envmap_material_network."


def as_torch(sd):
    return {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(v)) for k, v in sd.items()}


def wn_weight(sd, prefix):
    """weight-norm fold W = g * v / |v|_row  (torch.nn.utils.weight_norm, dim=0)."""
    v, g = sd[prefix + "weight_v"], sd[prefix + "weight_g"]
    return v * (g / v.norm(dim=1, keepdim=True))


# ------------------------------------------------------------------ SDF network
def sdf_raw(sd, x):
    """SDFNetwork.forward (neus_model.py:385-417) on NeuS-space points x [M,3] -> [M,257].
    PE(L=10) -> 9 weight-normed layers, Softplus(beta=100) between, skip concat /sqrt(2) into layer 4."""
    if x.numel() == 0:
        return torch.ones_like(x)
    enc = pe(x, 10)
    h = enc
    for l in range(9):
        if l == 4:
            h = torch.cat([h, enc], 1) / math.sqrt(2.0)
        h = F.linear(h, wn_weight(sd, SDF + "lin%d." % l), sd[SDF + "lin%d.bias" % l])
        if l < 8:
            h = F.softplus(h, beta=100)
    return h


def implicit_forward(sd, pts):
    """ImplicitNetworkMy.forward (neus_model.py:788-792): stage-2 points -> x2 -> net -> ALL 257 outputs /2."""
    return sdf_raw(sd, pts * 2.0) / 2.0


def implicit_gradient(sd, pts):
    """ImplicitNetworkMy.gradient (neus_model.py:803-818): d(sdf(2x)/2)/dx, unnormalised, [M,3]."""
    if pts.numel() == 0:
        return torch.ones_like(pts)
    with torch.enable_grad():
        x = pts.detach().clone().requires_grad_(True)
        y = implicit_forward(sd, x)[:, :1]
        (g,) = torch.autograd.grad(y, x, torch.ones_like(y))
    return g.detach()


def sdf_raw_gradient(sd, x):
    """SDFNetwork.gradient (neus_model.py:425-438) in NeuS space."""
    with torch.enable_grad():
        x = x.detach().clone().requires_grad_(True)
        y = sdf_raw(sd, x)[:, :1]
        (g,) = torch.autograd.grad(y, x, torch.ones_like(y))
    return g.detach()


def inv_s(sd):
    """SingleVarianceNetwork (neus_model.py:644-650) clipped as its callers do (:832, sdf_render.py:203)."""
    return torch.exp(sd["implicit_network.neus_model.deviation_network.variance"] * 10.0).clip(1e-6, 1e6)


# ------------------------------------------------------------------ colour network
def color_raw(sd, x, normals, view, feat):
    """RenderingNetwork.forward, mode 'idr' (neus_model.py:535-560):
    cat[x, PE4(view), normal, feat] (289) -> 256x4 ReLU -> 3 -> sigmoid."""
    h = torch.cat([x, pe(view, 4), normals, feat], -1)
    for l in range(5):
        h = F.linear(h, wn_weight(sd, COL + "lin%d." % l), sd[COL + "lin%d.bias" % l])
        if l < 4:
            h = torch.relu(h)
    return torch.sigmoid(h)


# ------------------------------------------------------------------ visibility network
def vis_logits(sd, p, d):
    """VisNetwork.forward (implicit_differentiable_renderer.py:250-258): PE10(p)|PE10(d) -> 256x4 ReLU -> 2."""
    h = torch.cat([pe(p, 10), pe(d, 10)], -1)
    for i in range(5):
        h = F.linear(h, sd[VIS + "%d.weight" % (2 * i)], sd[VIS + "%d.bias" % (2 * i)])
        if i < 4:
            h = torch.relu(h)
    return h


# ------------------------------------------------------------------ sparse auto-encoder
def _seq(sd, prefix, n, h, act):
    for i in range(n):
        h = F.linear(h, sd[prefix + "%d.weight" % (2 * i)], sd[prefix + "%d.bias" % (2 * i)])
        if i < n - 1:
            h = act(h)
    return h


def sparse_ae(sd, prefix, x, noise, smooth_on_latent, latent_act, out_act, var=None):
    """SparseAE.forward (sg_envmap_material.py:74-99).  `noise` replaces the torch.randn draw:
    [n,32] latent noise (x0.01) if smooth_on_latent else [n,in_dim] input noise (x0.02)."""
    lrelu = lambda t: F.leaky_relu(t, 0.2)

    def encode(v):
        z = _seq(sd, prefix + ".brdf_encoder_layer.", 5, v, lrelu)
        return z * (1.0 - (var if var is not None else torch.zeros(32)))

    lat = latent_act(encode(x))
    out = _seq(sd, prefix + ".brdf_decoder_layer.", 3, lat, lrelu)
    if smooth_on_latent:
        lat2 = lat + noise * 0.01
    else:
        lat2 = latent_act(encode(x + noise * 0.02))
    out2 = _seq(sd, prefix + ".brdf_decoder_layer.", 3, lat2, lrelu)
    if out_act is not None:
        out, out2 = out_act(out), out_act(out2)
    return out, out2


# ------------------------------------------------------------------ indirect illumination
def indirect_illum(sd, pts, hdr_shift, noise):
    """IndirctIllumNetwork.forward (implicit_differentiable_renderer.py:199-222).
    -> lgt_sgs [n,24,7] (unit lobe from two sigmoids, lambda = sigmoid*30+0.1, mu = relu), env_int [n,3]."""
    feat = pe(pts, 10) if hdr_shift is None else torch.cat([pe(pts, 10), hdr_shift], -1)   # None: no_hdr (hdr_mode -1)
    out = _seq(sd, ILL + "lobe_layer.", 5, feat, torch.relu).reshape(-1, 24, 6)
    ab = torch.sigmoid(out[..., :2])
    theta, phi = ab[..., :1] * 2 * math.pi, ab[..., 1:2] * math.pi
    lobes = torch.cat([torch.cos(theta) * torch.sin(phi), torch.sin(theta) * torch.sin(phi), torch.cos(phi)], -1)
    lam = torch.sigmoid(out[..., 2:3]) * 30 + 0.1
    mu = torch.relu(out[..., 3:])
    sgs = torch.cat([lobes, lam, mu], -1)
    _, rnd = sparse_ae(sd, ILL + "integral_layer", feat, noise, False, F.softplus, None)
    return sgs, rnd.abs()


# ------------------------------------------------------------------ materials
def materials(sd, pts, spec_noise, normal_noise):
    """EnvmapMaterialNetwork.forward with train_spec=True (sg_envmap_material.py:188-247)."""
    brdf, brdf_r = sparse_ae(sd, MAT + "spec_brdf_encoder_layer", pe(pts, 10), spec_noise, True,
                             torch.sigmoid, torch.sigmoid)
    nm, nm_r = sparse_ae(sd, MAT + "normal_decoder_layer", ipe_isotropic(pts, 1e-5, 10), normal_noise, False,
                         torch.sigmoid, None)
    unit = lambda v: v / torch.clamp(v.norm(dim=-1, keepdim=True), 1e-4)
    return {
        "sg_lgtSGs": sd[MAT + "lgtSGs"],
        "sg_specular_reflectance": sd[MAT + "specular_reflectance"],
        "sg_roughness": brdf[..., 3:4] * 0.9 + 0.09,
        "sg_metallic": brdf[..., 4:5] * 0.99 + 0.01,
        "sg_normal_map": unit(nm),
        "sg_diffuse_albedo": brdf[..., :3],
        "random_xi_roughness": brdf_r[..., 3:4] * 0.9 + 0.09,
        "random_xi_metallic": brdf_r[..., 4:5],
        "random_xi_diffuse_albedo": brdf_r[..., :3],
        "random_xi_normal": unit(nm_r),
    }


def normal_map_only(sd, pts, normal_noise):
    """train_norm=True branch (sg_envmap_material.py:203,214-234) used by forward('Illum')."""
    nm, nm_r = sparse_ae(sd, MAT + "normal_decoder_layer", ipe_isotropic(pts, 1e-5, 10), normal_noise, False,
                         torch.sigmoid, None)
    unit = lambda v: v / torch.clamp(v.norm(dim=-1, keepdim=True), 1e-4)
    return unit(nm), unit(nm_r)


# ------------------------------------------------------------------ CESR-stage nets
def softplus_net512(sd, x):
    """SDFNetwork(d_in, d_out, 512, 8, skip [4], multires=0) (neus_model.py:312-417 without positional encoding),
    as the CESR runner builds shadow_net / normal_net (training/train_cesr.py:106-110).  sd: keys lin{l}.weight_v/_g/bias."""
    h = x
    for l in range(9):
        if l == 4:
            h = torch.cat([h, x], 1) / math.sqrt(2.0)
        h = F.linear(h, wn_weight(sd, "lin%d." % l), sd["lin%d.bias" % l])
        if l < 8:
            h = F.softplus(h, beta=100)
    return h
