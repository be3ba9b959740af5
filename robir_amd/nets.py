"""nn.Module mirrors of the reference's networks: identical constructor arguments and state-dict keys
(so Norm/Vis/PBR checkpoints load unchanged), forward passes run on the HIP kernels.

Mirrors: model/implicit_differentiable_renderer.py:170-258 (IndirctIllumNetwork, VisNetwork),
model/sg_envmap_material.py:40-275 (SparseAE, EnvmapMaterialNetwork),
model/neus_model.py:312-438,489-560,644-650,682-884 (SDFNetwork, RenderingNetwork, SingleVarianceNetwork,
NeuSModel, ImplicitNetworkMy).

Forward-only: the kernels carry no autograd.  A forward pass of a module that is in training mode, with grad enabled and
a parameter that requires grad, raises ForwardOnlyError (`forward_only_guard`; the drop-in is for inference /
--plot_only rendering) instead of handing detached outputs to a loss.
Every forward that the reference randomises takes the draws as an optional explicit tensor (`noise=`); when omitted
they are drawn with torch.randn on the device, in the reference's order.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from . import ops, packing


class _Packed:
    """Lazy, parameter-version-keyed cache of packed weight blobs.  The Parameter objects of a module are looked up once
    (`.to()`, `load_state_dict` and optimiser steps change them in place: new storage / bumped version, same object);
    walking the module tree on every call cost ~10 % of a per-chunk forward()."""

    def __init__(self):
        self._cache = {}
        self._params = {}

    def get(self, key, module, builder):
        params = self._params.get(id(module))
        if params is None:
            params = self._params[id(module)] = list(module.parameters())
        sig = (tuple((p.data_ptr(), p._version) for p in params), params[0].device)
        ent = self._cache.get(key)
        if ent is None or ent[0] != sig:
            sd = {k: v.detach() for k, v in module.state_dict(prefix="").items()}
            ent = (sig, builder(sd))
            self._cache[key] = ent
        return ent[1]


class ForwardOnlyError(RuntimeError):
    pass


PRECISE_GRAD_SPLIT = os.environ.get("ROBIR_PRECISE_GRAD", "split") == "split"      # "jvp": value and gradient in one f32-input pass (round 3)


def forward_only_guard(module):
    """The HIP kernels have no backward: a training-mode call that autograd would have to differentiate must not silently
    return detached tensors (loss.backward() would then train only whatever still carries a graph)."""
    if module.training and torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters()):
        raise ForwardOnlyError(f"{type(module).__name__}: robir_amd kernels are forward-only -- call .eval(), wrap the call in "
                               "torch.no_grad(), or freeze the parameters (training stays on the reference's modules)")


# ----------------------------------------------------------------------------------------- small-batch concurrency
# A single 1024-pixel chunk gives the 512-wide nets ~660 rows = 11 workgroups per launch: each launch is one workgroup's
# latency (0.12 ms) on a machine with 256 CUs, and a chunk runs five of them back to back.  Independent nets of one forward
# are therefore issued on side streams when the batch is small (joined before anything consumes them).  Whole-view batches
# fill the GPU by themselves and stay on the caller's stream.
import os as _os
SIDE_STREAMS = _os.environ.get("ROBIR_SIDE_STREAMS", "1") != "0"
SIDE_STREAM_MAX_ROWS = 8192
BORROW_SLAB_ROWS = int(_os.environ.get("ROBIR_BORROW_SLAB_ROWS", "65536"))     # rays per borrow_color slab (NeuSRenderer.batch_borrow_color)
_SIDE_POOL = {}


def run_concurrently(thunks, rows):
    """[t() for t in thunks]; for small batches on a GPU each thunk after the first runs on its own side stream.  Safe without
    record_stream bookkeeping: a side section starts only after an event recorded on the caller's stream (everything enqueued
    before -- including the consumers of tensors a previous side section allocated -- precedes it), and the caller's stream
    waits for every side stream before this function returns."""
    if len(thunks) < 2 or not SIDE_STREAMS or rows > SIDE_STREAM_MAX_ROWS or not torch.cuda.is_available():
        return [t() for t in thunks]
    cur = torch.cuda.current_stream()
    pool = _SIDE_POOL.get(cur.device)
    if pool is None:
        pool = _SIDE_POOL[cur.device] = [torch.cuda.Stream(device=cur.device) for _ in range(3)]
    ev = cur.record_event()
    outs = [None] * len(thunks)
    used = []
    for i in range(1, len(thunks)):
        st = pool[(i - 1) % len(pool)]
        st.wait_event(ev)
        with torch.cuda.stream(st):
            outs[i] = thunks[i]()
        used.append(st)
    outs[0] = thunks[0]()
    for st in used:
        cur.wait_stream(st)
    return outs


def _require_dims(name, got, want):
    if list(got) != list(want):
        raise NotImplementedError(f"{name}: the HIP kernels are built for dims {want}, got {list(got)}")


def _dev(module):
    return next(module.parameters()).device


# ----------------------------------------------------------------------------------------- visibility
from .precision import mlp_precision  # noqa: E402,F401  ('f16x6' under the default policy, 'f16x3' under ROBIR_PRECISION=split)


class VisNetwork(nn.Module):
    """implicit_differentiable_renderer.py:225-258."""

    def __init__(self, points_multires=10, dirs_multires=4, dims=[128, 128, 128, 128]):
        super().__init__()
        if points_multires != 10 or dirs_multires != 10:
            raise NotImplementedError("HIP visibility kernels are built for points_multires = dirs_multires = 10")
        _require_dims("visibility_network", dims, [256] * 4)
        layers, dim = [], 126
        for d in dims:
            layers += [nn.Linear(dim, d), nn.ReLU()]
            dim = d
        layers.append(nn.Linear(dim, 2))
        self.vis_layer = nn.Sequential(*layers)
        self._packed = _Packed()

    def _rename(self, sd):
        return {"visibility_network." + k: v for k, v in sd.items()}

    def packed_full(self):
        return self._packed.get("full", self, lambda sd: packing.pack_vis(self._rename(sd), _dev(self)))

    def packed_full_h3(self):
        return self._packed.get("full_h3", self, lambda sd: packing.pack_vis_h3(self._rename(sd), _dev(self)))

    def packed_full_x6(self):
        return self._packed.get("full_x6", self, lambda sd: packing.pack_vis_x6(self._rename(sd), _dev(self)))

    def packed_split(self):
        return self._packed.get("split", self, lambda sd: packing.pack_vis_split(self._rename(sd), _dev(self)))

    def logits_from_features(self, X):
        """X [M,128] = [PE10(p) | PE10(d)] (ops.feat_vis) -> logits [M,2]; arithmetic per robir_amd.MLP_PRECISION."""
        forward_only_guard(self)
        if mlp_precision() != "f16x3":          # feature rows: the f32-input MFMA kernel under 'fp32' and 'f16x6' alike
            return ops.vis_mlp(X, self.packed_full())
        return ops.vis_mlp_h3(X, self.packed_full_h3(), packing.H3_SCALE_LOG2)

    def logits_from_points(self, points, dirs, rep=1):
        """points [M/rep,3], dirs [M,3] (rep consecutive directions per point) -> logits [M,2]: [PE10(p) | PE10(d)] is encoded
        inside the MLP kernel (no feature rows)."""
        forward_only_guard(self)
        if not ops.SDF_FUSED_PE:
            return self.logits_from_features(ops.feat_vis(points.float().contiguous(), dirs.float().contiguous(), rep=rep))
        if mlp_precision() == "f16x6":      # exact three-piece operands (csrc/vis_x6.hip)
            return ops.vis_x6_points(points, dirs, self.packed_full_x6(), rep)
        if mlp_precision() != "f16x3":
            return ops.vis_mlp_points(points, dirs, self.packed_full(), rep)
        return ops.vis_mlp_points(points, dirs, self.packed_full_h3(), rep, packing.H3_SCALE_LOG2)

    def forward(self, points, view_dirs):
        forward_only_guard(self)
        if points.shape[0] == 0:
            return torch.zeros(0, 2, device=points.device)
        return self.logits_from_points(points.float().contiguous(), view_dirs.float().contiguous())


# ----------------------------------------------------------------------------------------- sparse auto-encoder
class SparseAE(nn.Module):
    """sg_envmap_material.py:40-118 (forward/encode)."""

    def __init__(self, in_dim, out_dim, smooth_on_latent=True, out_act=torch.sigmoid, latent_dim=32, high_lr=False):
        super().__init__()
        if latent_dim != 32 or in_dim > 64 or out_dim > 16:
            raise NotImplementedError("HIP SparseAE kernels: in_dim <= 64, latent 32, out_dim <= 16")
        enc, dim = [], in_dim
        for d in [512, 512, 512, 512]:
            enc += [nn.Linear(dim, d), nn.LeakyReLU(0.2)]
            dim = d
        enc.append(nn.Linear(dim, latent_dim))
        self.brdf_encoder_layer = nn.Sequential(*enc)
        dec, dim = [], latent_dim
        for d in [128, 128]:
            dec += [nn.Linear(dim, d), nn.LeakyReLU(0.2)]
            dim = d
        dec.append(nn.Linear(dim, out_dim))
        self.brdf_decoder_layer = nn.Sequential(*dec)
        self.in_dim, self.out_dim, self.latent_dim = in_dim, out_dim, latent_dim
        self.smooth_on_latent = smooth_on_latent
        self.out_act = out_act
        self.high_lr = high_lr
        self.lc_act = torch.sigmoid            # runners replace this with F.softplus for the integral layer
        self.var = torch.zeros(latent_dim)     # plain attribute, like the reference (not in the state dict)
        self._packed = _Packed()

    def _blobs(self):
        return self._packed.get("ae", self, lambda sd: packing.pack_sparse_ae(
            {"ae." + k: v for k, v in sd.items()}, "ae", _dev(self)))

    def _encode(self, X):
        if mlp_precision() == "f16x3":
            blob = self._packed.get("enc_h3", self, lambda sd: packing.pack_sparse_ae_encoder_h3(
                {"ae." + k: v for k, v in sd.items()}, "ae", _dev(self)))
            return ops.wide_mlp_h3(X, blob, True, packing.H3_SCALE_LOG2)
        if mlp_precision() == "f16x6":      # exact three-piece operands (csrc/wide_x6.hip)
            blob = self._packed.get("enc_x6", self, lambda sd: packing.pack_sparse_ae_encoder_x6(
                {"ae." + k: v for k, v in sd.items()}, "ae", _dev(self)))
            return ops.wide_x6(X, blob, True)
        return ops.ae_encode(X, self._blobs()[0])

    def _encode_points(self, pts):
        """_encode(feat_pe10(pts)) with the encoding fused into the encoder kernel."""
        if not ops.SDF_FUSED_PE:
            return self._encode(ops.feat_pe10(pts))
        if mlp_precision() == "f16x3":
            blob = self._packed.get("enc_h3", self, lambda sd: packing.pack_sparse_ae_encoder_h3(
                {"ae." + k: v for k, v in sd.items()}, "ae", _dev(self)))
            return ops.wide_mlp_points(pts, None, blob, True, packing.H3_SCALE_LOG2)
        if mlp_precision() == "f16x6":      # exact three-piece operands (csrc/wide_x6.hip)
            blob = self._packed.get("enc_x6", self, lambda sd: packing.pack_sparse_ae_encoder_x6(
                {"ae." + k: v for k, v in sd.items()}, "ae", _dev(self)))
            return ops.wide_x6_points(pts, None, blob, True)
        return ops.wide_mlp_points(pts, None, self._blobs()[0], True)

    def run_points(self, pts, noise):
        """run(feat_pe10(pts), noise=noise) for latent-smoothed auto-encoders, straight from the points."""
        forward_only_guard(self)
        assert self.smooth_on_latent
        enc, dec = self._blobs()
        sig_out = self.out_act is not None
        if sig_out and getattr(self.out_act, "__name__", "") != "sigmoid":
            raise NotImplementedError("out_act must be torch.sigmoid or None")
        lat, lat2 = ops.ae_latent(self._encode_points(pts), self._var(pts.device), self._latent_act_code(), noise, 0.01)
        return ops.ae_decode(lat, dec, self.out_dim, sig_out), ops.ae_decode(lat2, dec, self.out_dim, sig_out)

    def _latent_act_code(self):
        name = getattr(self.lc_act, "__name__", "")
        if name == "softplus":
            return 1
        if name == "sigmoid":
            return 0
        raise NotImplementedError(f"latent activation {self.lc_act}")

    def _var(self, dev):
        v = self.var
        if isinstance(v, torch.Tensor) and bool((v != 0).any()):
            return v.to(device=dev, dtype=torch.float32).contiguous()
        return None

    def run(self, X, noise=None, X_noisy=None, need_first=True):
        """X [n,64] padded features.  smooth_on_latent: noise [n,32]; else X_noisy [n,64] = features of the perturbed input.
        need_first=False skips the un-perturbed pass (the indirect-illumination integral only uses the second output)."""
        forward_only_guard(self)
        enc, dec = self._blobs()
        dev = X.device
        sig_out = self.out_act is not None
        if sig_out and getattr(self.out_act, "__name__", "") != "sigmoid":
            raise NotImplementedError("out_act must be torch.sigmoid or None")
        if self.smooth_on_latent:
            lat, lat2 = ops.ae_latent(self._encode(X), self._var(dev), self._latent_act_code(), noise, 0.01)
        else:
            lat2, _ = ops.ae_latent(self._encode(X_noisy), self._var(dev), self._latent_act_code())
            if not need_first:
                return None, ops.ae_decode(lat2, dec, self.out_dim, sig_out)
            lat, _ = ops.ae_latent(self._encode(X), self._var(dev), self._latent_act_code())
        return ops.ae_decode(lat, dec, self.out_dim, sig_out), ops.ae_decode(lat2, dec, self.out_dim, sig_out)

    def run_pass(self, X):
        """One encode -> latent activation -> decode pass on features X [n,64] (no latent noise)."""
        forward_only_guard(self)
        enc, dec = self._blobs()
        lat, _ = ops.ae_latent(self._encode(X), self._var(X.device), self._latent_act_code())
        return ops.ae_decode(lat, dec, self.out_dim, self.out_act is not None)

    def encode(self, values):
        """sg_envmap_material.py:96-99: encoder output * (1 - var) on already-embedded rows [n, in_dim] (the pre-activation latent)."""
        forward_only_guard(self)
        flat = values.reshape(-1, values.shape[-1]).float()
        X = torch.zeros(flat.shape[0], 64, device=flat.device)
        X[:, :self.in_dim] = flat
        raw, _ = ops.ae_latent(self._encode(X), self._var(flat.device), 2)
        return raw.reshape(list(values.shape[:-1]) + [self.latent_dim])

    def kl_divergence(self, rho, rho_hat):
        raise NotImplementedError("SparseAE.kl_divergence: a training loss term (model/sg_envmap_material.py:101-105; callers model/loss.py) -- "
                                  "losses are OUT OF SCOPE for the forward renderer (SURVEY.md section 2 row 9)")

    def kl_smooth_loss(self, points, kl_w, smooth_w):
        raise NotImplementedError("SparseAE.kl_smooth_loss: a training loss term (model/sg_envmap_material.py:107-118) -- losses are OUT OF "
                                  "SCOPE for the forward renderer (SURVEY.md section 2 row 9)")

    def forward(self, points, noise=None):
        """points [n,in_dim]: already-embedded inputs (the reference's parameter name, sg_envmap_material.py:74)."""
        values = points
        n = values.shape[0]
        X = torch.zeros(n, 64, device=values.device)
        X[:, :self.in_dim] = values
        if self.smooth_on_latent:
            if noise is None:
                noise = torch.randn(n, 32, device=values.device)
            return self.run(X, noise=noise)
        if noise is None:
            noise = torch.randn(n, self.in_dim, device=values.device)
        Xn = torch.zeros(n, 64, device=values.device)
        Xn[:, :self.in_dim] = ops.axpy(values.contiguous(), noise.contiguous(), 0.02)
        return self.run(X, X_noisy=Xn)


# ----------------------------------------------------------------------------------------- indirect illumination
class IndirctIllumNetwork(nn.Module):
    """implicit_differentiable_renderer.py:170-222."""

    def __init__(self, multires=0, dims=[128, 128, 128, 128], num_lgt_sgs=24, no_hdr=False):
        super().__init__()
        if multires != 10 or num_lgt_sgs != 24:
            raise NotImplementedError("HIP indirect-illumination kernels: multires=10, 24 lobes")
        _require_dims("indirect_illum_network", dims, [512] * 4)
        self.num_lgt_sgs = num_lgt_sgs
        self.use_hdr = not no_hdr              # hdr_mode == -1: no hdr-shift input column (feature 63 stays zero)
        in_dim = 64 if self.use_hdr else 63
        layers, dim = [], in_dim
        for d in dims:
            layers += [nn.Linear(dim, d), nn.ReLU()]
            dim = d
        layers.append(nn.Linear(dim, num_lgt_sgs * 6))
        self.lobe_layer = nn.Sequential(*layers)
        self.integral_layer = SparseAE(in_dim, 3, out_act=None, smooth_on_latent=False)
        self.integral_layer.lc_act = torch.nn.functional.softplus
        self._packed = _Packed()

    def forward(self, points, hdr_shift, noise=None):
        forward_only_guard(self)
        n = points.shape[0]
        dev = points.device
        blob = self._packed.get("lobe", self.lobe_layer, lambda sd: packing.pack_illum(
            {"indirect_illum_network.lobe_layer." + k: v for k, v in sd.items()}, dev))
        X = ops.feat_pe10(points.float().contiguous(), extra=hdr_shift.float().contiguous() if self.use_hdr else None)
        if noise is None:
            noise = torch.randn(n, 64, device=dev)
        elif noise.shape[1] < 64:                # no_hdr: the reference draws randn_like of the 63 embedded columns
            noise = torch.nn.functional.pad(noise, (0, 64 - noise.shape[1]))
        noise = noise.float().contiguous()

        def lobes():
            hdr = hdr_shift.float().contiguous() if self.use_hdr else None
            if mlp_precision() == "f16x3":
                blob3 = self._packed.get("lobe_h3", self.lobe_layer, lambda sd: packing.pack_illum_h3(
                    {"indirect_illum_network.lobe_layer." + k: v for k, v in sd.items()}, dev))
                if ops.SDF_FUSED_PE:        # [PE10(x) | hdr_shift] encoded inside the lobe net's kernel
                    return ops.illum_decode(ops.wide_mlp_points(points, hdr, blob3, False, packing.H3_SCALE_LOG2))
                return ops.illum_decode(ops.wide_mlp_h3(X, blob3, False, packing.H3_SCALE_LOG2))
            if ops.SDF_FUSED_PE and mlp_precision() == "f16x6":      # exact three-piece operands (csrc/wide_x6.hip)
                blob6 = self._packed.get("lobe_x6", self.lobe_layer, lambda sd: packing.pack_illum_x6(
                    {"indirect_illum_network.lobe_layer." + k: v for k, v in sd.items()}, dev))
                return ops.illum_decode(ops.wide_x6_points(points, hdr, blob6, False))
            if ops.SDF_FUSED_PE:
                return ops.illum_decode(ops.wide_mlp_points(points, hdr, blob, False))
            return ops.illum_decode(ops.illum_mlp(X, blob))

        def integral():       # only the perturbed pass is used (implicit_differentiable_renderer.py:220)
            return ops.abs_scale(self.integral_layer.run_pass(ops.axpy(X, noise, 0.02)), 1.0)

        sgs, integ = run_concurrently([lobes, integral], n)      # two independent nets
        return sgs, integ


# ----------------------------------------------------------------------------------------- materials + light
def fibonacci_sphere(samples=1):
    from .synth import fibonacci_lobes
    return fibonacci_lobes(samples).astype(np.float64)


def compute_energy(lgtSGs):
    lam = torch.abs(lgtSGs[:, 3:4])
    mu = torch.abs(lgtSGs[:, 4:])
    return mu * 2.0 * np.pi / lam * (1.0 - torch.exp(-2.0 * lam))


class EnvmapMaterialNetwork(nn.Module):
    """sg_envmap_material.py:121-275."""

    def __init__(self, multires=0, brdf_encoder_dims=[512, 512, 512, 512], brdf_decoder_dims=[128, 128],
                 num_lgt_sgs=32, upper_hemi=False, specular_albedo=0.02, latent_dim=32):
        super().__init__()
        if multires != 10:
            raise NotImplementedError("HIP material kernels: multires=10")
        self.numLgtSGs = num_lgt_sgs
        self.envmap = None
        self.latent_dim = latent_dim
        self.brdf_encoder_layer = SparseAE(63, 5, out_act=None)
        self.spec_brdf_encoder_layer = SparseAE(63, 5, high_lr=True)
        self.normal_decoder_layer = SparseAE(60, 3, out_act=None, smooth_on_latent=False)
        self.specular_reflectance = nn.Parameter(torch.full((1, 1), float(specular_albedo)))
        from .synth import synth_light_sgs
        self.lgtSGs = nn.Parameter(torch.from_numpy(synth_light_sgs(0, num_lgt_sgs)))
        self.upper_hemi = upper_hemi

    def restrict_lobes_upper(self, lgtSGs):
        return torch.cat((lgtSGs[..., :1], torch.abs(lgtSGs[..., 1:2]), lgtSGs[..., 2:]), dim=-1)

    def forward(self, points, train_spec=False, train_norm=False, noise=None):
        """noise: dict with optional 'spec' [n,32] and 'normal' [n,60]."""
        forward_only_guard(self)
        noise = noise or {}
        n, dev = points.shape[0], points.device
        pts = points.float().contiguous()
        nz_n = noise.get("normal")
        want_spec = train_norm is False or train_spec
        nz_s = None
        if want_spec:
            nz_s = noise.get("spec")
            if nz_s is None:
                nz_s = torch.randn(n, 32, device=dev)
            nz_s = nz_s.float().contiguous()
        if nz_n is None:
            nz_n = torch.randn(n, 60, device=dev)
        nz_n = nz_n.float().contiguous()
        # the spec auto-encoder and the two passes of the normal auto-encoder (clean / perturbed input) are independent
        thunks = [lambda: ops.normalize3(self.normal_decoder_layer.run_pass(ops.feat_ipe(pts, 1e-5)), 1e-4, 1),
                  lambda: ops.normalize3(self.normal_decoder_layer.run_pass(ops.feat_ipe(pts, 1e-5, nz_n, 0.02)), 1e-4, 1)]
        if want_spec:
            thunks.append(lambda: self.spec_brdf_encoder_layer.run_points(pts, nz_s))
        res = run_concurrently(thunks, n)
        normal_map, random_xi_normal = res[0], res[1]
        if want_spec:
            brdf, brdf_r = res[2]
        if train_norm:
            return {"sg_normal_map": normal_map, "random_xi_normal": random_xi_normal}
        lgtSGs = self.restrict_lobes_upper(self.lgtSGs) if self.upper_hemi else self.lgtSGs
        alb, rough, metal, alb_r, rough_r, metal_r = ops.material_decode(brdf, brdf_r)
        return {
            "sg_lgtSGs": lgtSGs,
            "sg_specular_reflectance": self.specular_reflectance,
            "sg_roughness": rough,
            "sg_metallic": metal,
            "sg_normal_map": normal_map,
            "sg_diffuse_albedo": alb,
            "random_xi_roughness": rough_r,
            "random_xi_metallic": metal_r,
            "random_xi_diffuse_albedo": alb_r,
            "random_xi_normal": random_xi_normal,
        }

    def get_light(self):
        lgtSGs = self.lgtSGs.clone().detach()
        return self.restrict_lobes_upper(lgtSGs) if self.upper_hemi else lgtSGs

    def load_light(self, path):
        """sg_envmap_material.py:257-268: `<path>/sg_128.npy` light SGs + `<path>.exr` background map (read by robir_amd.exr:
        the reference goes through imageio)."""
        from . import deferred
        deferred.flush_all()               # recorded chunk forwards were meant for the light that is loaded now
        sg = torch.from_numpy(np.load(os.path.join(path, "sg_128.npy"))).to(self.lgtSGs.data.device)
        self.lgtSGs.data = sg
        from .exr import read_exr
        self.envmap = torch.from_numpy(np.ascontiguousarray(read_exr(path + ".exr")[:, :, :3])).to(sg.device)

    def parameter_groups(self, lr=0.0005):
        return [{"lr": lr, "params": self.brdf_encoder_layer.parameters()},
                {"lr": lr, "params": self.spec_brdf_encoder_layer.parameters()},
                {"lr": lr, "params": self.normal_decoder_layer.parameters()},
                {"lr": lr, "params": self.specular_reflectance},
                {"lr": lr, "params": self.lgtSGs}]


# ----------------------------------------------------------------------------------------- NeuS networks
def _wn_linear(k_in, n_out):
    return nn.utils.weight_norm(nn.Linear(k_in, n_out))


class SDFNetwork(nn.Module):
    """neus_model.py:312-438.  Three shapes are compiled:
      (d_in 3,   d_out 257, 256 x 8, skip [4], multires 10)  -- the NeuS SDF network (default arguments)
      (d_in 63,  d_out 3,   512 x 8, skip [4], multires 0)   -- CESR normal_net  (training/train_cesr.py:109)
      (d_in 191, d_out 2,   512 x 8, skip [4], multires 0)   -- CESR shadow_net  (training/train_cesr.py:107)"""

    def __init__(self, d_in, d_out, d_hidden, n_layers, skip_in=(4,), multires=10, bias=0.5, scale=1,
                 geometric_init=True, weight_norm=True, inside_outside=False, embed="Default"):
        super().__init__()
        if multires > 0 and embed == "IPE":
            raise NotImplementedError("SDFNetwork(embed='IPE'): the HIP SDF kernels encode with PE (every shipped configuration sets "
                                      "ENCODING = 'PE', confs_sg/env_path.py:5); the IPE-SDF variant is OUT OF SCOPE (SURVEY.md section 2 row 2)")
        key = (d_in, d_out, d_hidden, n_layers, tuple(skip_in), multires)
        kinds = {(3, 257, 256, 8, (4,), 10): "neus", (63, 3, 512, 8, (4,), 0): "normal", (191, 2, 512, 8, (4,), 0): "shadow"}
        if key not in kinds or not weight_norm or scale != 1:
            raise NotImplementedError(f"SDFNetwork{key}: the HIP kernels are compiled for {list(kinds)}")
        self.kind = kinds[key]
        self.d_in = d_in
        first = 63 if self.kind == "neus" else d_in
        dims = [first] + [d_hidden] * n_layers + [d_out]
        for l in range(n_layers + 1):
            out = dims[l + 1] - dims[0] if l + 1 in skip_in else dims[l + 1]
            setattr(self, "lin%d" % l, _wn_linear(dims[l], out))
        self.scale = 1
        self._packed = _Packed()

    def _sd(self, sd):
        return {"implicit_network.neus_model.sdf_network." + k: v for k, v in sd.items()}

    def packed(self, full=True):
        if self.kind == "neus":
            return self._packed.get("full" if full else "sdf", self,
                                    lambda sd: packing.pack_sdf(self._sd(sd), _dev(self), full=full))
        return self._packed.get("w512", self, lambda sd: packing.pack_softplus512(
            {"net." + k: v for k, v in sd.items()}, "net.", self.d_in, _dev(self)))

    def packed_h3(self, full=True):
        assert self.kind == "neus"
        return self._packed.get("full_h3" if full else "sdf_h3", self,
                                lambda sd: packing.pack_sdf_h3(self._sd(sd), _dev(self), full=full))

    def packed_back_h3(self):
        assert self.kind == "neus"
        return self._packed.get("back_h3", self, lambda sd: packing.pack_sdf_back_h3(self._sd(sd), _dev(self)))

    def packed_x6(self, full=True):
        assert self.kind == "neus"
        return self._packed.get("x6_full" if full else "x6_dist", self, lambda sd: packing.pack_sdf_x6(self._sd(sd), _dev(self), full=full))

    def packed_back_x6(self):
        assert self.kind == "neus"
        def both(sd):
            wt, w8 = packing.pack_sdf_back_x6(self._sd(sd), _dev(self))
            return wt, w8, packing.pack_sdf_back_x6(self._sd(sd), _dev(self), two_tile=True)[0]
        return self._packed.get("back_x6", self, both)

    def packed_back(self):
        assert self.kind == "neus"
        return self._packed.get("back", self, lambda sd: packing.pack_sdf_back(self._sd(sd), _dev(self)))

    def eval_points(self, x, in_scale=1.0, out_scale=1.0, full=True, grad=False, precise=False):
        """NeuS shape only.  x [M,3] -> (out [M,257] | [M], grad [M,3] | None); grad = d(out_scale*sdf(in_scale*x))/dx.
        precise: library-grade softplus (expf / log1pf on the f32-input MFMA, sdf-only modes) for VALUES that feed exact threshold
        decisions (the octree build).  With grad=True under the default policy the flag covers the value only: the gradient comes from
        the exact-operand reverse pass (ROBIR_PRECISE_GRAD=split, the default), whose softplus is the hardware-transcendental form."""
        assert self.kind == "neus"
        forward_only_guard(self)
        x = x.float().contiguous()
        M = x.shape[0]
        assert not (precise and full)
        mode = (1 if full else 0) + (2 if grad else 0) + (4 if precise else 0)
        if (mode == 3 and mlp_precision() == "f16x3" and ops.SDF_KERNEL == "ring" and ops.SDF_GRAD == "reverse"
                and M >= ops.SDF_GRAD_MIN_POINTS):
            # values once + one row vector back through the transposed layers, instead of three tangent rows per point
            return ops.sdf_value_grad(x, M, self.packed_h3(True), self.packed_back_h3(), packing.H3_SCALE_LOG2, in_scale,
                                      out_scale)
        x6 = mlp_precision() == "f16x6"
        if grad and precise and not full and x6 and ops.SDF_FUSED_PE and ops.SDF_GRAD == "reverse" and PRECISE_GRAD_SPLIT:
            # the octree's cell table (octree_tracing.build): the VALUE with the library-grade softplus (one row per point on the f32-input
            # MFMA), the GRADIENT by the policy's reverse pass on exact operands -- not three more tangent rows per point on the slow pipe
            val = ops.sdf_mlp_points(x, M, self.packed(False), 4, in_scale, out_scale, out_scale * in_scale)[0]
            return val, ops.sdf_value_grad_x6(x, M, self.packed_x6(True), self.packed_back_x6(), in_scale, out_scale)[1]
        if (grad and not precise and mlp_precision() in ("fp32", "f16x6") and ops.SDF_FUSED_PE and ops.SDF_GRAD == "reverse"
                and M >= (1 if x6 else ops.SDF_GRAD_F32_MIN_POINTS)):      # exact operands: three launches of 0.07 ms beat 0.31
            # the same at the reference's precision: value pass on exact three-piece operands (or the f32-input MFMA) + one pass over
            # the transposed layers on the f32-input MFMA
            if x6:
                out, g = ops.sdf_value_grad_x6(x, M, self.packed_x6(True), self.packed_back_x6(), in_scale, out_scale)
            else:
                out, g = ops.sdf_value_grad_f32(x, M, self.packed(True), self.packed_back(), in_scale, out_scale)
            return (out if full else out[:, 0].contiguous()), g
        if not grad and not precise and x6 and ops.SDF_FUSED_PE:
            return ops.sdf_points_x6(x, M, self.packed_x6(full), full, in_scale, out_scale), None
        if (not grad and not precise and mlp_precision() == "f16x3" and ops.SDF_KERNEL == "ring" and ops.SDF_FUSED_PE
                and ops.sdf_ring_waves() == 8):
            # value rows straight from the points: positional encoding fused into the network kernel (csrc/sdf_ring8.hip)
            return ops.sdf_points_h3(x, M, self.packed_h3(full), full, packing.H3_SCALE_LOG2, in_scale, out_scale), None
        if grad and not precise and mlp_precision() == "f16x3" and ops.SDF_KERNEL == "ring" and ops.SDF_FUSED_PE:
            return ops.sdf_points_jvp_h3(x, M, self.packed_h3(full), full, packing.H3_SCALE_LOG2, in_scale, out_scale,
                                         out_scale * in_scale)
        if ops.SDF_FUSED_PE and (precise or mlp_precision() in ("fp32", "f16x6")):
            # f32-input MFMA kernel with the encoding (tangent rows included) evaluated inside it
            return ops.sdf_mlp_points(x, M, self.packed(full), mode, in_scale, out_scale, out_scale * in_scale)
        X = ops.feat_pe10(x, scale=in_scale, jvp=grad)
        if not precise and mlp_precision() == "f16x3":
            return ops.sdf_mlp_h3(X, M, self.packed_h3(full), mode, packing.H3_SCALE_LOG2, out_scale, out_scale * in_scale)
        return ops.sdf_mlp(X, M, self.packed(full), mode, out_scale, out_scale * in_scale)

    def packed_w512_h3(self):
        return self._packed.get("w512_h3", self, lambda sd: packing.pack_softplus512_h3(
            {"net." + k: v for k, v in sd.items()}, "net.", self.d_in, _dev(self)))

    def _cesr(self, X, M, kind, n_label=1):
        forward_only_guard(self)
        if mlp_precision() == "f16x3":
            return ops.cesr_net_h3(X, M, kind, self.packed_w512_h3(), packing.H3_SCALE_LOG2, n_label)
        return ops.cesr_net(X, M, kind, self.packed(), n_label)

    def _cesr_points(self, pts, M, kind, n_label=1):
        """_cesr on PE10(pts) with the encoding evaluated inside the kernel (kind 0 normal_net, 2 shadow_net x labels)."""
        forward_only_guard(self)
        from .precision import cesr_precision
        if cesr_precision() == "f16x1":     # plain f16, ONE product per multiply-add (csrc/cesr_f16.hip): the labelled throughput mode, NARROWER than fp32
            blob = self._packed.get("w512_f16", self, lambda sd: packing.pack_softplus512_f16(
                {"net." + k: v for k, v in sd.items()}, "net.", self.d_in, _dev(self)))
            return ops.cesr_net_f16_points(pts, M, kind, blob, n_label)
        if mlp_precision() == "f16x3":
            return ops.cesr_net_points(pts, M, kind, self.packed_w512_h3(), n_label, packing.H3_SCALE_LOG2)
        if mlp_precision() == "f16x6":      # exact three-piece operands (csrc/cesr_x6.hip)
            blob = self._packed.get("w512_x6", self, lambda sd: packing.pack_softplus512_x6(
                {"net." + k: v for k, v in sd.items()}, "net.", self.d_in, _dev(self)))
            return ops.cesr_net_x6_points(pts, M, kind, blob, n_label)
        return ops.cesr_net_points(pts, M, kind, self.packed(), n_label)

    def eval_point_labels(self, Xp, n_label=128):
        """shadow_net on every (point, one-hot label) pair: Xp [n,64] PE10 features -> logits [n*n_label, 2]."""
        assert self.kind == "shadow"
        if Xp.shape[1] == 3:            # points [n,3]: encoded inside the kernel
            return self._cesr_points(Xp, Xp.shape[0] * n_label, 2, n_label)
        return self._cesr(Xp, Xp.shape[0] * n_label, 2, n_label)

    def forward(self, inputs, var=0.0001, chunk=1024):
        if inputs.numel() == 0:
            return torch.ones_like(inputs)
        shape = list(inputs.shape[:-1]) + [-1]
        if self.kind == "neus":
            out, _ = self.eval_points(inputs.reshape(-1, 3))
            return out.reshape(shape)
        flat = inputs.reshape(-1, self.d_in).float()
        M = flat.shape[0]
        kp = 64 if self.kind == "normal" else 192
        X = torch.zeros(M, kp, device=flat.device)
        X[:, :self.d_in] = flat
        return self._cesr(X, M, 0 if self.kind == "normal" else 1).reshape(shape)

    def sdf(self, x):
        assert self.kind == "neus"
        out, _ = self.eval_points(x.reshape(-1, 3), full=False)
        return out[:, None]

    def sdf_hidden_appearance(self, x):
        return self.forward(x)

    def gradient(self, x):
        assert self.kind == "neus"
        _, g = self.eval_points(x.reshape(-1, 3), full=False, grad=True)
        return g.unsqueeze(1)


class RenderingNetwork(nn.Module):
    """neus_model.py:489-560, mode 'idr' (d_feature 256, d_hidden 256, 4 layers, multires_view 4)."""

    def __init__(self, d_feature, mode, d_in, d_out, d_hidden, n_layers, weight_norm=True, multires_view=4, squeeze_out=True):
        super().__init__()
        got = (d_feature, mode, d_in, d_out, d_hidden, n_layers, bool(weight_norm), multires_view, bool(squeeze_out))
        want = (256, "idr", 9, 3, 256, 4, True, 4, True)
        if got != want:
            raise NotImplementedError(f"RenderingNetwork{got}: the HIP colour kernels are compiled for {want} (what NeuSModel builds, "
                                      "model/neus_model.py:717-718)")
        self.mode, self.squeeze_out = mode, True
        dims = [289, 256, 256, 256, 256, 3]
        for l in range(5):
            setattr(self, "lin%d" % l, _wn_linear(dims[l], dims[l + 1]))
        self._packed = _Packed()

    def packed(self):
        return self._packed.get("c", self, lambda sd: packing.pack_color(
            {"implicit_network.neus_model.color_network." + k: v for k, v in sd.items()}, _dev(self)))

    def packed_x6(self):
        return self._packed.get("c_x6", self, lambda sd: packing.pack_color_x6(
            {"implicit_network.neus_model.color_network." + k: v for k, v in sd.items()}, _dev(self)))

    def packed_h3(self):
        return self._packed.get("c_h3", self, lambda sd: packing.pack_color_h3(
            {"implicit_network.neus_model.color_network." + k: v for k, v in sd.items()}, _dev(self)))

    def forward(self, points, normals, view_dirs, feature_vectors, x_scale=1.0, feat_scale=1.0):
        forward_only_guard(self)
        if mlp_precision() == "f16x3":
            fn = ops.color_mlp_h3_points if ops.SDF_FUSED_PE else ops.color_mlp_h3_two      # encoding inside the kernel | tail rows
            return fn(points, view_dirs, normals, feature_vectors, self.packed_h3(), packing.H3_SCALE_LOG2,
                      x_scale=x_scale, feat_scale=feat_scale)
        if ops.SDF_FUSED_PE and mlp_precision() == "f16x6":
            return ops.color_x6_points(points, view_dirs, normals, feature_vectors, self.packed_x6(), x_scale=x_scale,
                                       feat_scale=feat_scale)
        if ops.SDF_FUSED_PE:
            return ops.color_mlp_points(points, view_dirs, normals, feature_vectors, self.packed(), x_scale=x_scale,
                                        feat_scale=feat_scale)
        X = ops.feat_color(points.float().contiguous(), view_dirs.float().contiguous(), normals.float().contiguous(),
                           feature_vectors, x_scale=x_scale, feat_scale=feat_scale)
        if mlp_precision() == "f16x3":
            return ops.color_mlp_h3(X, self.packed_h3(), packing.H3_SCALE_LOG2)
        return ops.color_mlp(X, self.packed())


class SingleVarianceNetwork(nn.Module):
    def __init__(self, init_val):
        super().__init__()
        self.register_parameter("variance", nn.Parameter(torch.tensor(init_val)))

    def forward(self, x):
        return torch.ones([len(x), 1], device=x.device) * torch.exp(self.variance * 10.0)


class HashSDFNetwork(nn.Module):
    def __init__(self, d_in, d_out, multires=12, dx_curve=0.1, separated=False):
        raise NotImplementedError("HashSDFNetwork: tinycudann hash-grid SDF, never instantiated (hashing=False, model/neus_model.py:774) -- "
                                  "OUT OF SCOPE (SURVEY.md section 2 row 2)")

    def forward(self, inputs):
        raise NotImplementedError("HashSDFNetwork is OUT OF SCOPE (SURVEY.md section 2 row 2)")

    def sdf(self, x):
        raise NotImplementedError("HashSDFNetwork is OUT OF SCOPE (SURVEY.md section 2 row 2)")

    def sdf_hidden_appearance(self, x):
        raise NotImplementedError("HashSDFNetwork is OUT OF SCOPE (SURVEY.md section 2 row 2)")

    def gradient(self, x, dx=None):
        raise NotImplementedError("HashSDFNetwork is OUT OF SCOPE (SURVEY.md section 2 row 2)")


class NeRF(nn.Module):
    def __init__(self, D=8, W=256, d_in=3, d_in_view=3, multires=10, multires_view=4, output_ch=4, skips=[4], use_viewdirs=True):
        raise NotImplementedError("NeRF: the NeRF++ outside-the-sphere background model, built only when NeuSModel(outside=True) -- no shipped "
                                  "configuration does (n_outside=0, neus/config/render.gin:12) -- OUT OF SCOPE (SURVEY.md section 2 rows 2, 6)")

    def forward(self, input_pts, input_views):
        raise NotImplementedError("NeRF is OUT OF SCOPE (SURVEY.md section 2 rows 2, 6)")


def auto_flatten(f):
    """neus_model.py:653-662: call `f(self, x [M,3], ...)` on the flattened points and give the result x's leading shape."""
    import functools

    @functools.wraps(f)
    def wrapper(self, x, *args, **kwargs):
        lead = list(x.shape[:-1])
        return f(self, x.reshape(-1, 3), *args, **kwargs).reshape(lead + [-1])
    return wrapper


def auto_flatten2(f):
    """neus_model.py:665-678: the two-output form for `f(self, x, dirs, ...) -> (rgb, a)`; dirs [R,3] broadcast over the samples of x [R,S,3]."""
    import functools

    @functools.wraps(f)
    def wrapper(self, x, dirs, *args, **kwargs):
        lead = list(x.shape[:-1])
        if len(lead) + 1 > dirs.dim():
            dirs = dirs[:, None, :].expand(x.shape)
        rgb, a = f(self, x.reshape(-1, 3), dirs.reshape(-1, 3), *args, **kwargs)
        return rgb.reshape(lead + [-1]), a.reshape(lead + [-1])
    return wrapper


class NeuSModel(nn.Module):
    """neus_model.py:682-752 (mode 'idr', hashing False, no outside NeRF)."""

    def __init__(self, mode="idr", hashing=False, outside=False, embed="IPE"):
        super().__init__()
        if mode != "idr" or hashing or outside or embed != "PE":
            raise NotImplementedError(f"NeuSModel(mode={mode!r}, hashing={hashing}, outside={outside}, embed={embed!r}): only the shipped NeuS "
                                      "configuration is built -- mode 'idr', no hash grid, no outside NeRF, embed='PE' (ImplicitNetworkMy passes "
                                      "confs_sg.env_path.ENCODING = 'PE', model/neus_model.py:771-776; the constructor's own default 'IPE' is not "
                                      "what any shipped checkpoint uses).  OUT OF SCOPE: SURVEY.md section 2 row 2")
        self.color_network = RenderingNetwork(d_feature=256, mode=mode, d_in=9, d_out=3, d_hidden=256, n_layers=4)
        self.sdf_network = SDFNetwork(d_in=3, d_out=257, d_hidden=256, n_layers=8, embed=embed)
        self.deviation_network = SingleVarianceNetwork(init_val=0.3)

    def sdf(self, x):
        return self.sdf_network.sdf(x)

    def sdf_and_feat(self, x):
        out = self.sdf_network(x)
        return out[..., :1], out[..., 1:]

    def color(self, x, gradients, dirs, feature_vector):
        return self.color_network(x, gradients, dirs, feature_vector)

    def grad(self, x):
        shape = list(x.shape[:-1]) + [-1]
        return self.sdf_network.gradient(x.reshape(-1, 3)).reshape(shape)

    def dev(self, x):
        return self.deviation_network(x)

    def radius(self):
        return 2.0

    def background(self, x, dirs):
        raise NotImplementedError("NeuSModel.background: the NeRF++ outside model is never built (outside=False, n_outside=0 in every "
                                  "configuration) -- OUT OF SCOPE (SURVEY.md section 2 rows 2, 6)")

    def inv_s(self):
        """exp(10 * variance) clipped to [1e-6, 1e6] as a host float.  Reading it is a device synchronisation, and borrow_color
        needs it for every 8192-point batch: cached on the parameter itself (same object, storage and version -- a real
        nn.Parameter, not a temporary).  Optimizer steps, load_state_dict and any in-place op on the parameter bump its version;
        a write through `variance.data` does not (autograd's escape hatch) -- call refresh_inv_s() after one."""
        v = self.deviation_network.variance
        key = (v.data_ptr(), v._version, v.device)
        if getattr(self, "_inv_s_key", None) != key:
            raw = float(torch.exp(v.detach() * 10.0))
            self._inv_s_raw = raw
            self._inv_s_val = min(max(raw, 1e-6), 1e6)
            self._inv_s_key = key
        return self._inv_s_val

    def inv_s_unclipped(self):
        """exp(10 * variance) as NormalTrainRunner.get_neus_surface uses it (no clip), from the same cache (no host read per call)."""
        self.inv_s()
        return self._inv_s_raw

    def refresh_inv_s(self):
        self._inv_s_key = None

    def forward(self, pnts, dirs, **kwargs):
        shape = list(pnts.shape[:-1]) + [-1]
        if len(shape) > dirs.dim():
            dirs = dirs[:, None, :].expand(pnts.shape)
        x, d = pnts.reshape(-1, 3), dirs.reshape(-1, 3)
        out, g = self.sdf_network.eval_points(x, full=True, grad=True)
        rgb = self.color_network(x, g, d, out[:, 1:])
        return rgb.view(shape), out[:, :1].reshape(shape)


def load_neus_checkpoint(neus_model, path):
    """Stage-1 checkpoint `{step:06d}.tar` = {'global_step', 'resume_time', 'model': state_dict, ...} written by
    neus/optimization/log.py:75-88, read like neus_model.py:779-781 (strict=False).  Keys of the three networks this
    build evaluates (sdf_network, color_network, deviation_network) must all be present; anything else the checkpoint
    carries (e.g. an outside-NeRF) is reported, not loaded."""
    state = torch.load(path, map_location="cpu", weights_only=False)
    if "model" not in state:
        raise KeyError(f"{path}: no 'model' entry (found {sorted(state)}) -- not a NeuS stage-1 checkpoint")
    res = neus_model.load_state_dict(state["model"], strict=False)
    if res.missing_keys:
        raise KeyError(f"{path}: NeuS checkpoint lacks {len(res.missing_keys)} tensors of the SDF / colour / variance "
                       f"networks, e.g. {res.missing_keys[:4]}")
    if res.unexpected_keys:
        import warnings
        warnings.warn(f"{path}: {len(res.unexpected_keys)} tensors not used by this build "
                      f"(e.g. {res.unexpected_keys[:4]})", RuntimeWarning, stacklevel=2)
    return int(state.get("global_step", 0))


class ImplicitNetworkMy(nn.Module):
    """neus_model.py:755-884: stage-2 wrapper around the NeuS model (points x2 in, all 257 outputs /2 out).
    The constructor arguments of the conf (dims, multires, ...) are ignored exactly like the reference does; the NeuS
    checkpoint named by confs_sg.env_path is loaded when that module is importable and the file exists."""

    def __init__(self, feature_vector_size=None, d_in=None, d_out=None, dims=None, geometric_init=True, bias=1.0,
                 skip_in=(), weight_norm=True, multires=0, bgr=False):
        super().__init__()
        try:
            from confs_sg.env_path import NEUS_LOG_DIR, NEUS_ITER      # reference-side global (confs_sg/env_path.py)
            from confs_sg import env_path as _env
            encoding = getattr(_env, "ENCODING", "PE")
        except ImportError:
            NEUS_LOG_DIR = None
            encoding = "PE"
        self.neus_model = NeuSModel(mode="idr", hashing=False, embed=encoding)        # raises for anything but 'PE'
        self.bgr = bgr
        if NEUS_LOG_DIR is None:
            # stand-alone use (tests, bench, robir_amd.render): the caller loads the NeuS weights itself
            import warnings
            warnings.warn("ImplicitNetworkMy: confs_sg.env_path is not set up -- the NeuS SDF / colour networks keep their "
                          "random initialisation until a state dict is loaded", RuntimeWarning, stacklevel=2)
        else:
            # the reference raises here when the checkpoint is missing (neus_model.py:779); so do we -- a wrong
            # NEUS_LOG_DIR / NEUS_ITER must not silently render a randomly initialised surface
            path = os.path.join(NEUS_LOG_DIR, "{:06d}.tar".format(NEUS_ITER))
            if not os.path.exists(path):
                raise FileNotFoundError(f"NeuS checkpoint {path} (confs_sg.env_path) does not exist")
            load_neus_checkpoint(self.neus_model, path)

    def normalize(self, x):
        return x * 2.0

    def forward(self, points, compute_grad=False):
        if points.numel() == 0:
            return torch.ones_like(points)
        out, _ = self.neus_model.sdf_network.eval_points(points.reshape(-1, 3), 2.0, 0.5, full=True)
        return out

    def sdf_only(self, points):
        """[M] stage-2 signed distance (last layer restricted to its first row)."""
        return self.neus_model.sdf_network.eval_points(points.reshape(-1, 3), 2.0, 0.5, full=False)[0]

    def color(self, points, normals, view_dirs, feature_vectors):
        c = self.neus_model.color_network(points, normals, view_dirs, feature_vectors, x_scale=2.0)
        return c.flip(-1) if self.bgr else c

    def gradient(self, x):
        if x.numel() == 0:
            return torch.ones_like(x)
        _, g = self.neus_model.sdf_network.eval_points(x.reshape(-1, 3), 2.0, 0.5, full=False, grad=True)
        return g.unsqueeze(1)

    def get_parameter_groups(self, **lr_dict):
        res = []
        if "color" in lr_dict:
            res += [{"lr": lr_dict["color"], "params": self.neus_model.sdf_network.parameters()}]
        if "sdf" in lr_dict:
            res += [{"lr": lr_dict["sdf"], "params": self.neus_model.color_network.parameters()}]
        return res

    def volume_render(self, sdf, color):
        """sdf [m,ns,1], color [m,ns,3] -> [m,3] (neus_model.py:828-854)."""
        rgb, _ = ops.neus_composite(sdf.reshape(sdf.shape[0], -1).contiguous(), color.contiguous(),
                                    self.neus_model.inv_s())
        return rgb

    def borrow_color(self, points, view_dirs):
        tk = torch.linspace(-0.01, 0.05, 16).to(points.device)
        x, d = ops.borrow_points(points, view_dirs, tk)
        net = self.neus_model.sdf_network
        out, g = net.eval_points(x, 1.0, 1.0, full=True, grad=True)
        col = self.neus_model.color_network(x, g, d, out[:, 1:])
        m = points.shape[0]
        rgb, _ = ops.neus_composite(out[:, 0].reshape(m, 16).contiguous(), col.reshape(m, 16, 3), self.neus_model.inv_s())
        return rgb

    def batch_borrow_color(self, points, view_dirs, batch_size=None):
        """neus_model.py:873-884.  The reference walks the rays in batches of 8192 (its default) to bound ITS memory; the rows are
        independent (nothing batch-global), so with `batch_size` left at its default a batch here is a slab of BORROW_SLAB_ROWS rays (x 16
        samples: 1 M network evaluations, 1 GB of outputs + 9 GB of gradient scratch): a view's 3.4 M secondary hits in 53 slabs instead of
        420 batches x 3 kernels of 0.4-0.8 ms.  An explicit `batch_size` is honoured (a caller bounding memory).  Results are equal to fp32
        summation order whatever the batch size (the SDF / colour kernels pick their one- or two-tile form by launch size)."""
        if points.shape[0] == 0:
            return torch.zeros_like(points)
        step = BORROW_SLAB_ROWS if batch_size is None else max(1, int(batch_size))
        with torch.no_grad():
            outs = [self.borrow_color(points[i:i + step].contiguous(), view_dirs[i:i + step].contiguous())
                    for i in range(0, points.shape[0], step)]
        return outs[0] if len(outs) == 1 else torch.cat(outs, 0)


# ----------------------------------------------------------------------------------------- tone mapping
class ACESToneMapping(nn.Module):
    """color_correction.py:76-137: hdr_mode 0 scale_aces (every shipped conf), 1 warp_aces, 2 ln_space, else identity."""

    def __init__(self, hdr_mode=0):
        super().__init__()
        self.adapt_illum = nn.Parameter(torch.tensor(0.0))
        self.hdr_mode = hdr_mode
        self._curve = hdr_mode if hdr_mode in (0, 1, 2) else 3

    def as_input(self):
        return torch.clamp(self.adapt_illum * 10 + 0.5, 0, 1).view(1, 1)

    def make_shift(self, shift):
        if shift is None:
            shift = self.as_input()
        if not isinstance(shift, torch.Tensor):
            shift = torch.tensor(shift, device=self.adapt_illum.device)
        if shift.dim() == 0:
            shift = shift[None]
        return shift.detach().float()

    def _apply_tm(self, x, raw_shift, mode):
        shape = x.shape
        rows = x.numel() // 3
        p = self.adapt_illum
        key = (p.data_ptr(), p._version, x.device)
        if raw_shift is None and getattr(self, "_shift_key", None) == key:
            sh = self._shift_val               # the model's own shift: ten tiny launches per call otherwise
        else:
            sh = self.make_shift(raw_shift).reshape(-1).to(x.device)
            if sh.numel() != 1 and sh.numel() != rows:
                sh = sh.expand(rows) if sh.numel() == 1 else sh.reshape(-1)
            sh = sh.contiguous()
            if raw_shift is None:
                self._shift_key, self._shift_val = key, sh
        from . import deferred

        def run(v):
            xs = v.detach().float().reshape(-1, 3).contiguous()
            return ops.tonemap(xs, sh, mode + 16 * self._curve).reshape(shape)
        if deferred.is_deferred(x):        # a recorded chunk forward (deferred.py): tone-map when its pass has run
            return deferred.lazy_like(x, run)
        return run(x)

    def hdr2ldr(self, x, raw_shift=None):
        return self._apply_tm(x, raw_shift, 0)

    def ldr2hdr(self, x, raw_shift=None):
        return self._apply_tm(x, raw_shift, 1)

    def fit_data(self, data):
        raise NotImplementedError("Energy.gen_cache is a training-time pre-fit (out of scope, SURVEY.md section 2 #9)")

    def plot(self, shift=1.0):
        raise NotImplementedError("ACESToneMapping.plot: a matplotlib debugging plot (model/color_correction.py:103-109) -- plots are OUT OF "
                                  "SCOPE (SURVEY.md section 2)")

    def scalar(self, shift):
        raise NotImplementedError("ACESToneMapping.scalar: reads the Energy cache that fit_data pre-fits at training time "
                                  "(model/color_correction.py:111-113, model/energy_integral.py) -- OUT OF SCOPE (SURVEY.md section 2 #9)")


class GammaCorrect(nn.Module):
    """color_correction.py:7-28."""

    def __init__(self, gamma=2.2, hdr_mode=0):
        super().__init__()
        self.gamma = nn.Parameter(torch.tensor(float(gamma)))
        self.indir_coef = nn.Parameter(torch.tensor(1.0))
        self.dir_coef = nn.Parameter(torch.tensor(2.0))
        self.coef = nn.Parameter(torch.tensor(1.0))
        self.hdr_shift = ACESToneMapping(hdr_mode=hdr_mode)

    def forward(self, x):
        return torch.pow(x, 1 / self.gamma)

    def inv(self, x):
        return torch.pow(x, self.gamma)
