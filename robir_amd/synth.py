"""Deterministic synthetic weights / cameras / RNG draws for parity tests and bench.py.

There is no trained checkpoint and no dataset on the GPU box (and none ships with the
reference), so every test and the benchmark run on synthetic parameters.  The generator is
our own (counter-style: one independent PCG64 stream per tensor name), *not* the reference's
torch initialisers; golden vectors are produced by loading exactly these tensors into the
reference's modules (oracle/gen_golden.py).

The SDF network follows the NeuS "geometric initialisation" idea the reference uses
(model/neus_model.py:358-376) so that the zero level set is a sphere of radius ~0.5 in NeuS
units (0.25 in stage-2 units): last layer N(sqrt(pi)/sqrt(256), 1e-4) with bias -0.5, first
layer only sees the raw xyz, skip layer ignores the re-injected encoding.
State-dict key names are the reference's nn.Module tree so that the
same dict loads into the reference, into the oracle and into robir_amd.
"""
import hashlib
import math

import numpy as np

SDF_DIMS = [63, 256, 256, 256, 256, 256, 256, 256, 256, 257]   # layer l: in -> out (skip at l=4)
COLOR_DIMS = [289, 256, 256, 256, 256, 3]
VIS_DIMS = [126, 256, 256, 256, 256, 2]
ILLUM_DIMS = [64, 512, 512, 512, 512, 144]
AE_ENC = [512, 512, 512, 512, 32]
AE_DEC = [128, 128]


def _rng(seed, name):
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return np.random.Generator(np.random.PCG64(int.from_bytes(h[:8], "little")))


def _uniform(seed, name, shape, bound):
    return (_rng(seed, name).random(shape, dtype=np.float64) * 2.0 - 1.0).astype(np.float32) * np.float32(bound)


def _normal(seed, name, shape, mean, std):
    return (_rng(seed, name).standard_normal(shape) * std + mean).astype(np.float32)


def _relu_linear(sd, seed, prefix, fan_in, fan_out, bias_bound=0.1, gain=1.0):
    bound = gain * math.sqrt(6.0 / fan_in)          # variance preserving for ReLU-like nets
    sd[prefix + ".weight"] = _uniform(seed, prefix + ".weight", (fan_out, fan_in), bound)
    sd[prefix + ".bias"] = _uniform(seed, prefix + ".bias", (fan_out,), bias_bound)


def fibonacci_lobes(n):
    """n points spiralling from +y to -y with the golden angle (cf. sg_envmap_material.py:12-30)."""
    i = np.arange(n, dtype=np.float64)
    y = 1.0 - 2.0 * i / (n - 1)
    r = np.sqrt(np.maximum(1.0 - y * y, 0.0))
    ang = i * (math.pi * (3.0 - math.sqrt(5.0)))
    return np.stack([np.cos(ang) * r, y, np.sin(ang) * r], -1).astype(np.float32)


def synth_light_sgs(seed, n_lobes=128, sharp=False):
    """[n,7] light SGs: (lobe xyz, lambda, mu rgb).  sharp=True mimics the value ranges of the
    shipped fits envmaps/envmap*/sg_128.npy (|lambda| up to ~500, un-normalised lobes, coloured)."""
    g = _rng(seed, "lgtSGs")
    sg = np.zeros((n_lobes, 7), np.float64)
    lob = fibonacci_lobes(n_lobes // 2)
    sg[: n_lobes // 2, :3] = lob
    sg[n_lobes // 2:, :3] = lob
    if not sharp:
        sg[:, 3] = 10.0 + np.abs(g.standard_normal(n_lobes)) * 20.0
        mu = np.abs(g.standard_normal(n_lobes))
        energy = mu * 2.0 * math.pi / sg[:, 3] * (1.0 - np.exp(-2.0 * sg[:, 3]))
        mu = mu / energy.sum() * 2.0 * math.pi * 0.8
        sg[:, 4:] = mu[:, None]
    else:
        sg[:, :3] *= g.uniform(0.3, 3.5, (n_lobes, 1))
        sg[:, 3] = np.exp(g.uniform(math.log(1.2), math.log(500.0), n_lobes)) * np.where(g.random(n_lobes) < 0.2, -1, 1)
        sg[:, 4:] = g.uniform(0.0, 1.0, (n_lobes, 3)) ** 3 * 6.8
    return sg.astype(np.float32)


def synth_state_dict(seed=0, variance=0.3, sharp_light=False, dtype=np.float32, scene="sphere"):
    """Full IDRNetwork state dict (reference key names) as numpy float32 arrays.
    scene = "sphere": the geometric initialisation (zero level set = a sphere of radius 0.5 NeuS units, convex);
    scene = "nonconvex": the SDF network fitted to two overlapping spheres and a torus around them (robir_amd/data/nonconvex_sdf.npz,
    recipe oracle/fit_nonconvex.py): concavities, a hole, secondary rays that re-hit, encoding columns that carry weight."""
    if scene not in ("sphere", "nonconvex"):
        raise ValueError("scene must be sphere or nonconvex")
    sd = {}
    # ---- NeuS SDF network (weight-normed; geometric init) ----
    p = "implicit_network.neus_model.sdf_network.lin%d"
    for l in range(9):
        k_in, n_out = SDF_DIMS[l], SDF_DIMS[l + 1]
        if l == 3:
            n_out = 256 - 63                        # layer feeding the skip concat
        if l == 4:
            k_in = 256
        name = p % l
        if l == 8:
            v = _normal(seed, name + ".v", (n_out, k_in), math.sqrt(math.pi) / math.sqrt(k_in), 1e-4)
            b = np.full((n_out,), -0.5, np.float32)
        else:
            v = _normal(seed, name + ".v", (n_out, k_in), 0.0, math.sqrt(2.0) / math.sqrt(n_out))
            b = np.zeros((n_out,), np.float32)
            if l == 0:
                v[:, 3:] = 0.0
            if l == 4:
                v[:, -60:] = 0.0
        sd[name + ".weight_v"] = v
        sd[name + ".weight_g"] = np.linalg.norm(v.astype(np.float64), axis=1, keepdims=True).astype(np.float32)
        sd[name + ".bias"] = b
    # ---- NeuS colour network (weight-normed, g != |v| so the fold is exercised) ----
    p = "implicit_network.neus_model.color_network.lin%d"
    for l in range(5):
        k_in, n_out = COLOR_DIMS[l], COLOR_DIMS[l + 1]
        name = p % l
        v = _uniform(seed, name + ".v", (n_out, k_in), math.sqrt(6.0 / k_in))
        nrm = np.linalg.norm(v.astype(np.float64), axis=1, keepdims=True)
        sd[name + ".weight_v"] = v
        sd[name + ".weight_g"] = (nrm * _rng(seed, name + ".g").uniform(0.8, 1.2, (n_out, 1))).astype(np.float32)
        sd[name + ".bias"] = _uniform(seed, name + ".b", (n_out,), 0.1)
    sd["implicit_network.neus_model.deviation_network.variance"] = np.array(variance, np.float32)
    # ---- indirect illumination ----
    for i in range(5):
        _relu_linear(sd, seed, "indirect_illum_network.lobe_layer.%d" % (2 * i), ILLUM_DIMS[i], ILLUM_DIMS[i + 1])
    _sparse_ae(sd, seed, "indirect_illum_network.integral_layer", 64, 3)
    # ---- visibility ----
    for i in range(5):
        _relu_linear(sd, seed, "visibility_network.vis_layer.%d" % (2 * i), VIS_DIMS[i], VIS_DIMS[i + 1])
    # ---- materials / light ----
    sd["envmap_material_network.specular_reflectance"] = np.full((1, 1), 0.05, np.float32)
    sd["envmap_material_network.lgtSGs"] = synth_light_sgs(seed, 128, sharp_light)
    _sparse_ae(sd, seed, "envmap_material_network.brdf_encoder_layer", 63, 5)
    _sparse_ae(sd, seed, "envmap_material_network.spec_brdf_encoder_layer", 63, 5)
    _sparse_ae(sd, seed, "envmap_material_network.normal_decoder_layer", 60, 3)
    sd["gamma.gamma"] = np.array(1.0, np.float32)
    sd["gamma.indir_coef"] = np.array(1.0, np.float32)
    sd["gamma.dir_coef"] = np.array(2.0, np.float32)
    sd["gamma.coef"] = np.array(1.0, np.float32)
    sd["gamma.hdr_shift.adapt_illum"] = np.array(0.0, np.float32)
    if scene == "nonconvex":
        import os
        fit = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "nonconvex_sdf.npz"))
        assert all(k in sd and sd[k].shape == fit[k].shape for k in fit.files), "nonconvex_sdf.npz does not match the SDF network"
        sd.update({k: fit[k] for k in fit.files})
    return {k: np.ascontiguousarray(v.astype(dtype)) for k, v in sd.items()}


def _sparse_ae(sd, seed, prefix, in_dim, out_dim):
    dims = [in_dim] + AE_ENC
    for i in range(5):
        _relu_linear(sd, seed, prefix + ".brdf_encoder_layer.%d" % (2 * i), dims[i], dims[i + 1])
    dims = [32] + AE_DEC + [out_dim]
    for i in range(3):
        _relu_linear(sd, seed, prefix + ".brdf_decoder_layer.%d" % (2 * i), dims[i], dims[i + 1])


def synth_cesr_nets(seed=0):
    """State dicts of the CESR runner's shadow_net (191 -> 2) and normal_net (63 -> 3): SDFNetwork(.., 512, 8, [4], 0)
    (training/train_cesr.py:106-110).  He-style normal weights with g = 0.9..1.1 |v| (weight norm exercised)."""
    out = {}
    for name, k_in, n_out in (("shadow_net", 191, 2), ("normal_net", 63, 3)):
        sd = {}
        dims = [k_in] + [512] * 8 + [n_out]
        for l in range(9):
            o = dims[l + 1] - dims[0] if l + 1 == 4 else dims[l + 1]
            key = f"{name}.lin{l}"
            v = _normal(seed, key + ".v", (o, dims[l]), 0.0, math.sqrt(2.0) / math.sqrt(dims[l]))
            nrm = np.linalg.norm(v.astype(np.float64), axis=1, keepdims=True)
            sd[f"lin{l}.weight_v"] = v
            sd[f"lin{l}.weight_g"] = (nrm * _rng(seed, key + ".g").uniform(0.9, 1.1, (o, 1))).astype(np.float32)
            sd[f"lin{l}.bias"] = _uniform(seed, key + ".b", (o,), 0.01 if l < 8 else 0.5)
        out[name] = sd
    return out


def neus_state_dict(sd):
    """The sub-dict a stage-1 NeuS checkpoint ({step:06d}.tar -> 'model') would hold."""
    pre = "implicit_network.neus_model."
    return {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}


# --------------------------------------------------------------------------------------
# camera / pixel grid (Blender convention, camera_angle_x = 0.6911, d = 0.9)
# --------------------------------------------------------------------------------------
def synth_camera(H, W, distance=0.9, camera_angle_x=0.6911):
    focal = 0.5 * W / math.tan(0.5 * camera_angle_x)
    K = np.array([[focal, 0, W / 2], [0, focal, H / 2], [0, 0, 1]], np.float32)
    pose = np.eye(4, dtype=np.float32)
    pose[2, 3] = distance
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    uv = np.stack([xs, ys], -1).reshape(-1, 2)
    return uv, pose, K


def synth_draws(seed, name, shape, kind="rand"):
    """Explicit RNG tensors that replace the torch.rand / torch.randn calls inside the forward."""
    g = _rng(seed, "draw:" + name)
    if kind == "rand":
        return g.random(shape, dtype=np.float32)
    return g.standard_normal(shape).astype(np.float32)


def pbr_draws(seed, n_hit, chunk_id=0, n_lobes=128, nsamp_diffuse=32, nsamp_spec=8):
    """The nine draws of one forward('Material') in the order the reference consumes them
    ('RNG draw order')."""
    t = "c%d:" % chunk_id
    return {
        "illum_randn": synth_draws(seed, t + "illum", (n_hit, 64), "randn"),
        "spec_randn": synth_draws(seed, t + "spec", (n_hit, 32), "randn"),
        "normal_randn": synth_draws(seed, t + "normal", (n_hit, 60), "randn"),
        "dvis_theta": synth_draws(seed, t + "dth", (n_lobes, nsamp_diffuse)),
        "dvis_phi": synth_draws(seed, t + "dph", (n_lobes, nsamp_diffuse)),
        "svis_theta_dir": synth_draws(seed, t + "sth0", (n_hit, nsamp_spec)),
        "svis_phi_dir": synth_draws(seed, t + "sph0", (n_hit, nsamp_spec)),
        "svis_theta_ind": synth_draws(seed, t + "sth1", (n_hit, nsamp_spec)),
        "svis_phi_ind": synth_draws(seed, t + "sph1", (n_hit, nsamp_spec)),
    }
