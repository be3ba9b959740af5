"""HIP MLP kernels (through the C-ABI) against the oracle on identical inputs and weights.  GPU only."""
import numpy as np
import pytest
import torch

from conftest import rel_err, load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def pts_dirs():
    g = np.random.Generator(np.random.PCG64(99))
    M = 1000   # not a multiple of the 128-row block: exercises the ragged tail
    pts = torch.from_numpy((g.standard_normal((M, 3)) * 0.25).astype(np.float32))
    dirs = torch.from_numpy(g.standard_normal((M, 3)).astype(np.float32))
    dirs = dirs / dirs.norm(dim=-1, keepdim=True)
    return pts, dirs


def test_features(dev, pts_dirs):
    from robir_amd import ops
    from robir_oracle.encoding import pe, ipe_isotropic
    pts, dirs = pts_dirs
    X = ops.feat_vis(pts.to(dev), dirs.to(dev)).cpu()
    ref = torch.cat([pe(pts, 10), pe(dirs, 10)], -1)
    assert rel_err(X[:, :126], ref) <= 1e-5
    assert float(X[:, 126:].abs().max()) == 0.0
    X = ops.feat_pe10(pts.to(dev), scale=2.0).cpu()
    assert rel_err(X[:, :63], pe(pts * 2.0, 10)) <= 1e-5
    I = ops.feat_ipe(pts.to(dev) * 300.0, 1e-5).cpu()
    assert rel_err(I[:, :60], ipe_isotropic(pts * 300.0, 1e-5)) <= 1e-4


def test_pe_tangent_rows(dev, pts_dirs):
    from robir_amd import ops
    from robir_oracle.encoding import pe
    pts, _ = pts_dirs
    p = pts[:64].double().requires_grad_(True)
    J = torch.autograd.functional.jacobian(lambda q: pe(q, 10).sum(0), p)   # [63, 64, 3]
    X = ops.feat_pe10(pts[:64].to(dev), scale=1.0, jvp=True).cpu().reshape(64, 4, 64)
    for c in range(3):
        assert rel_err(X[:, 1 + c, :63], J[:, :, c].T.float()) <= 1e-4


def test_vis_mlp(dev, pts_dirs, synth_weights, oracle_sd):
    from robir_amd import ops, packing
    from robir_oracle import nets
    pts, dirs = pts_dirs
    blob = packing.pack_vis(synth_weights, dev)
    out = ops.vis_mlp(ops.feat_vis(pts.to(dev), dirs.to(dev)), blob).cpu()
    assert rel_err(out, nets.vis_logits(oracle_sd, pts, dirs)) <= TOL
    g = load_golden("nets")
    out = ops.vis_mlp(ops.feat_vis(torch.from_numpy(g["pts"]).to(dev), torch.from_numpy(g["dirs"]).to(dev)), blob).cpu()
    assert rel_err(out, g["vis_logits"]) <= TOL          # against the reference's own output
    # split-precision (f16x3) form of the same kernel: same tolerance, and ~1e-6 from the exact-fp32 MFMA path
    blob3 = packing.pack_vis_h3(synth_weights, dev)
    X = ops.feat_vis(torch.from_numpy(g["pts"]).to(dev), torch.from_numpy(g["dirs"]).to(dev))
    out3 = ops.vis_mlp_h3(X, blob3, packing.H3_SCALE_LOG2).cpu()
    assert rel_err(out3, g["vis_logits"]) <= TOL
    assert rel_err(out3, out) <= 1e-5


def test_linear_split_layer(dev, pts_dirs, synth_weights):
    from robir_amd import ops, packing
    from robir_oracle.encoding import pe
    pts, dirs = pts_dirs
    sp = packing.pack_vis_split(synth_weights, dev)
    W0 = torch.from_numpy(synth_weights["visibility_network.vis_layer.0.weight"])
    b0 = torch.from_numpy(synth_weights["visibility_network.vis_layer.0.bias"])
    A = ops.linear_64_256(ops.feat_pe10(pts.to(dev)), sp["point"]).cpu()
    B = ops.linear_64_256(ops.feat_pe10(dirs.to(dev)), sp["dir"]).cpu()
    assert rel_err(A, pe(pts, 10) @ W0[:, :63].T + b0) <= TOL
    assert rel_err(B, pe(dirs, 10) @ W0[:, 63:].T) <= TOL


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_sdf_mlp(dev, pts_dirs, synth_weights, oracle_sd, mode):
    from robir_amd import ops, packing
    from robir_oracle import nets
    pts, _ = pts_dirs
    full = mode in (1, 3)
    blob = packing.pack_sdf(synth_weights, dev, full=full)
    X = ops.feat_pe10(pts.to(dev), scale=2.0, jvp=mode >= 2)
    out0, grad = ops.sdf_mlp(X, pts.shape[0], blob, mode, out_scale=0.5, grad_scale=1.0)
    ref = nets.implicit_forward(oracle_sd, pts)
    if full:
        assert rel_err(out0.cpu(), ref) <= TOL
    else:
        assert rel_err(out0.cpu(), ref[:, 0]) <= TOL
    if mode >= 2:
        assert rel_err(grad.cpu(), nets.implicit_gradient(oracle_sd, pts)) <= TOL


@pytest.mark.parametrize("kernel", ["ring", "v1"])
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_sdf_mlp_split_precision(dev, pts_dirs, synth_weights, oracle_sd, mode, kernel, monkeypatch):
    """f16x3 form of the SDF kernels (second generation = weight ring, first generation = phase-alternating): same tolerance
    against the oracle, and fp32-rounding-level distance from the exact path."""
    from robir_amd import ops, packing
    from robir_oracle import nets
    monkeypatch.setattr(ops, "SDF_KERNEL", kernel)
    pts, _ = pts_dirs
    full = mode in (1, 3)
    X = ops.feat_pe10(pts.to(dev), scale=2.0, jvp=mode >= 2)
    out0, grad = ops.sdf_mlp_h3(X, pts.shape[0], packing.pack_sdf_h3(synth_weights, dev, full=full), mode,
                                packing.H3_SCALE_LOG2, out_scale=0.5, grad_scale=1.0)
    ex0, exg = ops.sdf_mlp(X, pts.shape[0], packing.pack_sdf(synth_weights, dev, full=full), mode, out_scale=0.5, grad_scale=1.0)
    ref = nets.implicit_forward(oracle_sd, pts)
    assert rel_err(out0.cpu(), ref if full else ref[:, 0]) <= TOL
    assert rel_err(out0.cpu(), ex0.cpu()) <= 1e-5
    if mode >= 2:
        assert rel_err(grad.cpu(), nets.implicit_gradient(oracle_sd, pts)) <= TOL
        assert rel_err(grad.cpu(), exg.cpu()) <= 1e-5


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_sdf_ring_kernel_many_rounds_and_ragged_sizes(dev, synth_weights, mode, monkeypatch):
    """The ring kernel's workgroups are persistent: with more rounds of 128 rows than compute units every workgroup runs
    several rounds on a ring that never drains (the 4-slot ring rotates by two slots per round), and the last round is ragged.
    Against the first-generation kernel (same arithmetic) and the exact f32-MFMA kernel, for sizes around the tile / round /
    grid boundaries."""
    from robir_amd import ops, packing
    full = mode in (1, 3)
    g = torch.Generator().manual_seed(7)
    h3 = packing.pack_sdf_h3(synth_weights, dev, full=full)
    ex = packing.pack_sdf(synth_weights, dev, full=full)
    for n in (1, 5, 31, 33, 127, 129, 40000 if mode >= 2 else 150001):
        pts = ((torch.rand(n, 3, generator=g) - 0.5) * 1.6).to(dev)
        X = ops.feat_pe10(pts, scale=2.0, jvp=mode >= 2)
        monkeypatch.setattr(ops, "SDF_KERNEL", "ring")
        a0, ag = ops.sdf_mlp_h3(X, n, h3, mode, packing.H3_SCALE_LOG2, out_scale=0.5, grad_scale=1.0)
        monkeypatch.setattr(ops, "SDF_KERNEL", "v1")
        b0, bg = ops.sdf_mlp_h3(X, n, h3, mode, packing.H3_SCALE_LOG2, out_scale=0.5, grad_scale=1.0)
        e0, eg = ops.sdf_mlp(X, n, ex, mode, out_scale=0.5, grad_scale=1.0)
        assert bool(torch.isfinite(a0).all())
        assert rel_err(a0.cpu(), b0.cpu()) <= 2e-6, (n, rel_err(a0.cpu(), b0.cpu()))
        assert rel_err(a0.cpu(), e0.cpu()) <= 1e-5, (n, rel_err(a0.cpu(), e0.cpu()))
        if mode >= 2:
            assert rel_err(ag.cpu(), bg.cpu()) <= 2e-6 and rel_err(ag.cpu(), eg.cpu()) <= 1e-5, n
    ops.range_check(sync=True)


def _trained_like(synth_weights, seed):
    """The geometric initialisation zeroes the positional-encoding columns of layer 0 and of the skip connection (layer 4)
    and keeps every hidden weight matrix near-diagonal in distribution: a gradient path that only a trained net exercises --
    the Jacobian of the encoding, the skip connection's share -- needs weights away from it."""
    from robir_amd.packing import SDF
    g = np.random.Generator(np.random.PCG64(seed))
    sd = dict(synth_weights)
    for l, cols, amp in ((0, slice(3, 63), 0.08), (4, slice(193, 256), 0.05)):
        w = np.array(sd[SDF + "lin%d.weight_v" % l], dtype=np.float32, copy=True)
        damp = 1.0 / (1.0 + np.arange(w[:, cols].shape[1]) // 6)          # high frequencies weaker, like trained nets
        w[:, cols] += (amp * g.standard_normal(w[:, cols].shape) * damp[None, :]).astype(np.float32)
        sd[SDF + "lin%d.weight_v" % l] = w
    for l in (1, 2, 3, 5, 6, 7):
        w = np.array(sd[SDF + "lin%d.weight_v" % l], dtype=np.float32, copy=True)
        sd[SDF + "lin%d.weight_v" % l] = w + (0.02 * g.standard_normal(w.shape)).astype(np.float32)
    return sd


@pytest.mark.parametrize("weights", ["init", "trained_like"])
def test_sdf_value_grad_reverse_mode(dev, pts_dirs, synth_weights, weights, monkeypatch):
    """rb_sdf_value_grad (values once + one row vector back through the transposed layers) against the forward-mode rows of
    mode 3, the exact f32-MFMA kernel and the oracle's autograd -- with the synthetic initialisation and with weights that
    make the encoding's Jacobian and the skip connection carry gradient."""
    from robir_amd import ops, packing
    from robir_oracle import nets
    sd = synth_weights if weights == "init" else _trained_like(synth_weights, 5)
    osd = nets.as_torch(sd)
    pts, _ = pts_dirs
    x = pts.to(dev)
    M = x.shape[0]
    h3 = packing.pack_sdf_h3(sd, dev, full=True)
    back = packing.pack_sdf_back_h3(sd, dev)
    out, grad = ops.sdf_value_grad(x, M, h3, back, packing.H3_SCALE_LOG2, in_scale=2.0, out_scale=0.5)
    X4 = ops.feat_pe10(x, scale=2.0, jvp=True)
    f0, fg = ops.sdf_mlp_h3(X4, M, h3, 3, packing.H3_SCALE_LOG2, out_scale=0.5, grad_scale=1.0)
    e0, eg = ops.sdf_mlp(X4, M, packing.pack_sdf(sd, dev, full=True), 3, out_scale=0.5, grad_scale=1.0)
    ops.range_check(sync=True)
    assert torch.equal(out, f0)                                   # the value pass is the same kernel arithmetic
    ref_g = nets.implicit_gradient(osd, pts)
    assert float(ref_g.abs().max()) > 0.5
    if weights == "trained_like":                                 # the encoding's columns do carry gradient now
        w0 = torch.from_numpy(sd[packing.SDF + "lin0.weight_v"])
        assert float(w0[:, 3:].abs().max()) > 0.05
    for name, other, tol in (("forward-mode", fg, 2e-5), ("exact fp32", eg, 2e-5)):
        assert rel_err(grad.cpu(), other.cpu()) <= tol, (name, rel_err(grad.cpu(), other.cpu()))
    assert rel_err(grad.cpu(), ref_g) <= TOL
    assert rel_err(out.cpu(), nets.implicit_forward(osd, pts)) <= TOL


def test_sdf_value_grad_sizes_slabs_and_repeatability(dev, synth_weights, monkeypatch):
    """Persistent workgroups over many rounds, ragged last rounds, several slabs through one scratch buffer; and bit-identical
    results run after run (the sigmoid rows reach the backward kernel by LDS-DMA whose arrival is detected from the data:
    a row consumed before it landed would show up here as a difference between runs)."""
    from robir_amd import ops, packing
    sd = _trained_like(synth_weights, 6)
    h3 = packing.pack_sdf_h3(sd, dev, full=True)
    back = packing.pack_sdf_back_h3(sd, dev)
    g = torch.Generator().manual_seed(11)
    o0, g0 = ops.sdf_value_grad(torch.zeros(0, 3, device=dev), 0, h3, back, packing.H3_SCALE_LOG2, in_scale=2.0, out_scale=0.5)
    assert o0.shape == (0, 257) and g0.shape == (0, 3)
    for n in (1, 127, 129, 4097, 33000, 300001):
        x = ((torch.rand(n, 3, generator=g) - 0.5) * 1.7).to(dev)
        out, grad = ops.sdf_value_grad(x, n, h3, back, packing.H3_SCALE_LOG2, in_scale=2.0, out_scale=0.5)
        f0, fg = ops.sdf_mlp_h3(ops.feat_pe10(x, scale=2.0, jvp=True), n, h3, 3, packing.H3_SCALE_LOG2, out_scale=0.5,
                                grad_scale=1.0)
        assert torch.equal(out, f0), n
        assert bool(torch.isfinite(grad).all()) and rel_err(grad.cpu(), fg.cpu()) <= 2e-5, (n, rel_err(grad.cpu(), fg.cpu()))
    monkeypatch.setattr(ops, "SDF_GRAD_SLAB", 70000)              # 300001 points -> five slabs, the last one ragged
    o2, g2 = ops.sdf_value_grad(x, n, h3, back, packing.H3_SCALE_LOG2, in_scale=2.0, out_scale=0.5)
    assert torch.equal(o2, out) and torch.equal(g2, grad)
    monkeypatch.setattr(ops, "SDF_GRAD_SLAB", 1 << 20)
    for _ in range(4):
        o3, g3 = ops.sdf_value_grad(x, n, h3, back, packing.H3_SCALE_LOG2, in_scale=2.0, out_scale=0.5)
        assert torch.equal(o3, out) and torch.equal(g3, grad)
    ops.range_check(sync=True)


_NAN_ROWS_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from robir_amd import ops, packing, synth
dev = torch.device("cuda:0")
sd = synth.synth_state_dict(0, variance=0.3)
h3 = packing.pack_sdf_h3(sd, dev, full=True)
back = packing.pack_sdf_back_h3(sd, dev)
g = torch.Generator().manual_seed(3)
n = 40000
x = ((torch.rand(n, 3, generator=g) - 0.5) * 1.7).to(dev)
clean_o, clean_g = ops.sdf_value_grad(x, n, h3, back, packing.H3_SCALE_LOG2, in_scale=2.0, out_scale=0.5)
bad = x.clone()
bad[1000:1100] = float("nan")                 # 100 consecutive NaN points: whole 32-point wave blocks of NaN sigmoid rows
bad[5000:5040, 1] = float("inf")
bad[20000] = float("nan")                     # and a single one inside a tile
o, gr = ops.sdf_value_grad(bad, n, h3, back, packing.H3_SCALE_LOG2, in_scale=2.0, out_scale=0.5)
torch.cuda.synchronize()
isbad = torch.zeros(n, dtype=torch.bool, device=dev)
isbad[1000:1100] = True; isbad[5000:5040] = True; isbad[20000] = True
assert bool(torch.isnan(o[isbad]).all(-1).all()) and bool(torch.isnan(gr[isbad]).all(-1).all()), "NaN rows must stay NaN"
assert torch.equal(o[~isbad], clean_o[~isbad]) and torch.equal(gr[~isbad], clean_g[~isbad]), "other rows must not change"
print("nan rows ok")
"""


def test_sdf_value_grad_nan_and_inf_rows_do_not_hang():
    """The backward kernel detects the arrival of a sigmoid row from its content.  Rows of NaN sigmoids (NaN / inf input points:
    all 128 samples of a ray that grazes a cell face) must count as arrived -- the first version spun on them for ever --
    and must not disturb their neighbours.  Runs in a child process under a timeout so that a regression fails, not hangs."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _NAN_ROWS_SCRIPT, root], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0 and "nan rows ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_sdf_value_kernels_eight_and_four_waves_agree(dev, synth_weights):
    """Value rows (modes 0, 1, and the value pass + sigmoid blob of the reverse-mode gradient) as eight waves of one tile
    (csrc/sdf_ring8.hip, the split policy's default) and as four waves of two tiles (csrc/sdf_ring.hip): bit-identical outputs -- the blob through
    the backward pass that reads it."""
    from robir_amd import _lib, ops, packing
    g = torch.Generator().manual_seed(21)
    full = packing.pack_sdf_h3(synth_weights, dev, full=True)
    dist = packing.pack_sdf_h3(synth_weights, dev, full=False)
    back = packing.pack_sdf_back_h3(synth_weights, dev)
    L = _lib.legacy()                 # the split-precision SDF kernels live in the legacy library
    res = {}
    try:
        for waves in (8, 4):
            L.rb_sdf_ring_waves(waves)
            out = []
            for n in (1, 129, 40001):
                x = ((torch.rand(n, 3, generator=torch.Generator().manual_seed(n)) - 0.5) * 1.6).to(dev)
                X = ops.feat_pe10(x, scale=2.0)
                out.append(ops.sdf_mlp_h3(X, n, dist, 0, packing.H3_SCALE_LOG2, out_scale=0.5)[0])
                out.append(ops.sdf_mlp_h3(X, n, full, 1, packing.H3_SCALE_LOG2, out_scale=0.5)[0])
                out.extend(ops.sdf_value_grad(x, n, full, back, packing.H3_SCALE_LOG2, in_scale=2.0, out_scale=0.5))
            res[waves] = out
    finally:
        L.rb_sdf_ring_waves(8)
    assert len(res[8]) == len(res[4]) == 12
    for a, b in zip(res[8], res[4]):
        assert bool(torch.isfinite(a).all()) and torch.equal(a, b)
    ops.range_check(sync=True)


def test_eval_points_picks_the_reverse_mode_for_large_batches(dev, synth_weights, monkeypatch):
    from robir_amd import nets, ops, synth
    model = nets.NeuSModel(embed="PE")
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.neus_state_dict(synth_weights).items()})
    net = model.to(dev).eval().sdf_network
    g = torch.Generator().manual_seed(3)
    x = ((torch.rand(20000, 3, generator=g) - 0.5) * 1.6).to(dev)
    calls = []
    real = ops.sdf_value_grad
    monkeypatch.setattr(ops, "sdf_value_grad", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    monkeypatch.setenv("ROBIR_MLP_PRECISION", "f16x3")           # the reverse-mode pass exists in split precision
    o_r, g_r = net.eval_points(x, 2.0, 0.5, full=True, grad=True)
    assert calls == [1]
    net.eval_points(x[:1000], 2.0, 0.5, full=True, grad=True)    # small batches keep the forward-mode rows
    net.eval_points(x, 2.0, 0.5, full=False, grad=True)          # so does the distance-only form
    assert calls == [1]
    monkeypatch.setattr(ops, "SDF_GRAD", "forward")
    o_f, g_f = net.eval_points(x, 2.0, 0.5, full=True, grad=True)
    assert calls == [1] and torch.equal(o_r, o_f) and rel_err(g_r.cpu(), g_f.cpu()) <= 2e-5


def test_sdf_golden(dev, synth_weights):
    from robir_amd import ops, packing
    g = load_golden("nets")
    pts = torch.from_numpy(g["pts"]).to(dev)
    blob = packing.pack_sdf(synth_weights, dev, full=True)
    out0, grad = ops.sdf_mlp(ops.feat_pe10(pts, scale=2.0, jvp=True), pts.shape[0], blob, 3, 0.5, 1.0)
    assert rel_err(out0.cpu(), g["sdf_feat"]) <= TOL
    assert rel_err(grad.cpu(), g["grad"]) <= TOL


def test_color_mlp(dev, synth_weights):
    from robir_amd import ops, packing
    g = load_golden("nets")
    pts, dirs = torch.from_numpy(g["pts"]).to(dev), torch.from_numpy(g["dirs"]).to(dev)
    feat = torch.from_numpy(g["sdf_feat"]).to(dev)      # wrapper output: features are halved (neus_model.py:790-791)
    nrm = torch.from_numpy(g["color_normals"]).to(dev)
    X = ops.feat_color(pts, dirs, nrm, feat[:, 1:], x_scale=2.0, feat_scale=2.0)
    rgb = ops.color_mlp(X, packing.pack_color(synth_weights, dev)).cpu()
    assert rel_err(rgb, g["color"]) <= TOL
    rgb3 = ops.color_mlp_h3(X, packing.pack_color_h3(synth_weights, dev), packing.H3_SCALE_LOG2).cpu()
    assert rel_err(rgb3, g["color"]) <= TOL and rel_err(rgb3, rgb) <= 1e-5
    # the form the renderer uses: no assembled [M,304] rows -- features read in place from the SDF net's 257-float rows
    two = ops.color_mlp_h3_two(pts, dirs, nrm, feat[:, 1:], packing.pack_color_h3(synth_weights, dev), packing.H3_SCALE_LOG2,
                               x_scale=2.0, feat_scale=2.0)
    assert torch.equal(two.cpu(), rgb3)
    gen = torch.Generator().manual_seed(4)
    for n in (1, 127, 129, 70001):
        p2 = ((torch.rand(n, 3, generator=gen) - 0.5) * 1.6).to(dev)
        d2 = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1).to(dev)
        n2 = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1).to(dev)
        f2 = torch.randn(n, 257, generator=gen).to(dev) * 0.3
        a = ops.color_mlp_h3(ops.feat_color(p2, d2, n2, f2[:, 1:], x_scale=2.0, feat_scale=2.0),
                             packing.pack_color_h3(synth_weights, dev), packing.H3_SCALE_LOG2)
        b = ops.color_mlp_h3_two(p2, d2, n2, f2[:, 1:], packing.pack_color_h3(synth_weights, dev), packing.H3_SCALE_LOG2,
                                 x_scale=2.0, feat_scale=2.0)
        assert torch.equal(a, b), n


def test_illum_and_autoencoders(dev, synth_weights):
    from robir_amd import ops, packing
    g = load_golden("nets")
    pts = torch.from_numpy(g["pts"]).to(dev)
    hdr = torch.from_numpy(g["hdr"]).to(dev)
    n64, n32, n60 = (torch.from_numpy(g[k]).to(dev) for k in ("illum_noise", "spec_noise", "normal_noise"))
    X = ops.feat_pe10(pts, extra=hdr)
    sgs = ops.illum_decode(ops.illum_mlp(X, packing.pack_illum(synth_weights, dev))).cpu()
    assert rel_err(sgs, g["illum_sgs"]) <= TOL
    # integral layer: SparseAE(64->3), latent softplus, re-encode of x + 0.02*noise, |.| of the SECOND output
    enc, dec = packing.pack_sparse_ae(synth_weights, "indirect_illum_network.integral_layer", dev)
    lat, _ = ops.ae_latent(ops.ae_encode(ops.axpy(X, n64, 0.02), enc), act=1)
    integ = ops.ae_decode(lat, dec, 3, False).abs().cpu()
    assert rel_err(integ, g["illum_int"]) <= TOL
    # spec AE: latent sigmoid, second decode of latent + 0.01*noise, sigmoid out
    enc, dec = packing.pack_sparse_ae(synth_weights, "envmap_material_network.spec_brdf_encoder_layer", dev)
    lat, lat2 = ops.ae_latent(ops.ae_encode(ops.feat_pe10(pts), enc), act=0, noise=n32, noise_scale=0.01)
    brdf, brdf2 = ops.ae_decode(lat, dec, 5, True).cpu(), ops.ae_decode(lat2, dec, 5, True).cpu()
    assert rel_err(brdf[:, :3], g["mat_sg_diffuse_albedo"]) <= TOL
    assert rel_err(brdf[:, 3:4] * 0.9 + 0.09, g["mat_sg_roughness"]) <= TOL
    assert rel_err(brdf2[:, 4:5], g["mat_random_xi_metallic"]) <= TOL
    # normal AE on IPE features, re-encode of ipe + 0.02*noise
    enc, dec = packing.pack_sparse_ae(synth_weights, "envmap_material_network.normal_decoder_layer", dev)
    lat, _ = ops.ae_latent(ops.ae_encode(ops.feat_ipe(pts, 1e-5), enc), act=0)
    nm = ops.ae_decode(lat, dec, 3, False).cpu()
    nm = nm / torch.clamp(nm.norm(dim=-1, keepdim=True), 1e-4)
    assert rel_err(nm, g["mat_sg_normal_map"]) <= TOL
    lat, _ = ops.ae_latent(ops.ae_encode(ops.feat_ipe(pts, 1e-5, noise=n60, noise_scale=0.02), enc), act=0)
    nm2 = ops.ae_decode(lat, dec, 3, False).cpu()
    nm2 = nm2 / torch.clamp(nm2.norm(dim=-1, keepdim=True), 1e-4)
    assert rel_err(nm2, g["mat_random_xi_normal"]) <= TOL


def test_wide_nets_split_precision(dev, synth_weights):
    """f16x3 forms of the 512-wide nets (indirect-illumination lobe net, SparseAE encoders): golden tolerance and
    fp32-rounding-level distance from the exact kernels."""
    from robir_amd import ops, packing
    g = load_golden("nets")
    pts = torch.from_numpy(g["pts"]).to(dev)
    hdr = torch.from_numpy(g["hdr"]).to(dev)
    X = ops.feat_pe10(pts, extra=hdr)
    raw = ops.illum_mlp(X, packing.pack_illum(synth_weights, dev))
    raw3 = ops.wide_mlp_h3(X, packing.pack_illum_h3(synth_weights, dev), False, packing.H3_SCALE_LOG2)
    assert rel_err(raw3.cpu(), raw.cpu()) <= 1e-5
    assert rel_err(ops.illum_decode(raw3).cpu(), g["illum_sgs"]) <= TOL
    for prefix, feats in (("envmap_material_network.spec_brdf_encoder_layer", ops.feat_pe10(pts)),
                          ("envmap_material_network.normal_decoder_layer", ops.feat_ipe(pts, 1e-5)),
                          ("indirect_illum_network.integral_layer", X)):
        enc, _ = packing.pack_sparse_ae(synth_weights, prefix, dev)
        enc3 = packing.pack_sparse_ae_encoder_h3(synth_weights, prefix, dev)
        a = ops.ae_encode(feats, enc).cpu()
        b = ops.wide_mlp_h3(feats, enc3, True, packing.H3_SCALE_LOG2).cpu()
        assert rel_err(b, a) <= 1e-5, prefix


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
def test_illum_network_without_hdr_input(dev, precision, monkeypatch):
    """IndirctIllumNetwork(no_hdr=True) -- conf hdr_mode = -1 (implicit_differentiable_renderer.py:181-203,285): 63 embedded
    inputs, no hdr-shift column, for the lobe net and the integral SparseAE alike."""
    from robir_amd import nets as hnets
    from robir_oracle import nets as onets
    monkeypatch.setenv("ROBIR_MLP_PRECISION", precision)
    torch.manual_seed(5)
    net = hnets.IndirctIllumNetwork(multires=10, dims=[512] * 4, num_lgt_sgs=24, no_hdr=True)
    assert net.lobe_layer[0].weight.shape[1] == 63 and net.integral_layer.brdf_encoder_layer[0].weight.shape[1] == 63
    sd = {"indirect_illum_network." + k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.to(dev)
    g = torch.Generator().manual_seed(6)
    pts = torch.rand(777, 3, generator=g) * 1.6 - 0.8
    noise = torch.randn(777, 63, generator=g)
    sgs, integ = net(pts.to(dev), None, noise=noise.to(dev))
    ref_sgs, ref_int = onets.indirect_illum(sd, pts, None, noise)
    assert rel_err(sgs.cpu(), ref_sgs) <= TOL
    assert rel_err(integ.cpu(), ref_int) <= TOL


def test_side_streams_give_identical_results(dev, synth_weights, monkeypatch):
    """Small batches issue the independent 512-wide nets of one forward on side streams (nets.run_concurrently); the results are
    those of the sequential order, bit for bit, over repeated calls that recycle the side streams' allocations."""
    from robir_amd import nets as hnets, renderer
    m = renderer.build_synthetic_model(dev, build_octrees=False)
    g = torch.Generator().manual_seed(3)
    outs = {}
    for mode in (True, False, True):
        monkeypatch.setattr(hnets, "SIDE_STREAMS", mode)
        acc = []
        for rep in range(6):
            n = 700 - 37 * rep
            pts = ((torch.rand(n, 3, generator=torch.Generator().manual_seed(rep)) - 0.5) * 0.5).to(dev)
            hdr = torch.full((n, 1), 0.5, device=dev)
            nz = {"spec": torch.randn(n, 32, generator=torch.Generator().manual_seed(10 + rep)).to(dev),
                  "normal": torch.randn(n, 60, generator=torch.Generator().manual_seed(20 + rep)).to(dev)}
            ill = torch.randn(n, 64, generator=torch.Generator().manual_seed(30 + rep)).to(dev)
            sgs, integ = m.indirect_illum_network(pts, hdr, noise=ill)
            mat = m.envmap_material_network(pts, train_spec=True, noise=nz)
            acc.append(torch.cat([sgs.reshape(n, -1), integ, mat["sg_normal_map"], mat["random_xi_normal"], mat["sg_diffuse_albedo"],
                                  mat["sg_roughness"], mat["random_xi_roughness"]], -1).cpu())
        outs.setdefault(mode, []).append(acc)
    a, b, c = outs[True][0], outs[False][0], outs[True][1]
    for x, y, z in zip(a, b, c):
        assert torch.equal(x, y) and torch.equal(x, z)


def test_fused_positional_encoding_is_bit_identical_to_the_row_form(dev, synth_weights, monkeypatch):
    """k_sdf_ring8<.., FUSED>: the SDF network straight from the points (encoding evaluated inside the kernel, once per round, by
    the four lanes that share a point) against rb_feat_pe10 rows + the same kernel -- distance only, all 257 outputs and the
    value + reverse-mode-gradient op; sizes that are not multiples of the 128-row round, one point, and weights whose encoding
    columns carry real weight (the non-convex fit)."""
    from robir_amd import ops, packing, synth
    g = torch.Generator().manual_seed(31)
    for sd in (synth_weights, synth.synth_state_dict(0, scene="nonconvex")):
        full = packing.pack_sdf_h3(sd, dev, full=True)
        dist = packing.pack_sdf_h3(sd, dev, full=False)
        back = packing.pack_sdf_back_h3(sd, dev)
        for n in (1, 127, 128, 1000, 40001):
            x = ((torch.rand(n, 3, generator=g) - 0.5) * 1.9).to(dev)
            X = ops.feat_pe10(x, scale=2.0)
            r0, _ = ops.sdf_mlp_h3(X, n, dist, 0, packing.H3_SCALE_LOG2, 0.5, 1.0)
            r1, _ = ops.sdf_mlp_h3(X, n, full, 1, packing.H3_SCALE_LOG2, 0.5, 1.0)
            f0 = ops.sdf_points_h3(x, n, dist, False, packing.H3_SCALE_LOG2, 2.0, 0.5)
            f1 = ops.sdf_points_h3(x, n, full, True, packing.H3_SCALE_LOG2, 2.0, 0.5)
            assert torch.equal(f0, r0) and torch.equal(f1, r1), n
            X4 = ops.feat_pe10(x, scale=2.0, jvp=True)
            for fullm, blob in ((False, dist), (True, full)):
                ro, rg = ops.sdf_mlp_h3(X4, n, blob, 3 if fullm else 2, packing.H3_SCALE_LOG2, 0.5, 1.0)
                fo, fg = ops.sdf_points_jvp_h3(x, n, blob, fullm, packing.H3_SCALE_LOG2, 2.0, 0.5, 1.0)
                assert torch.equal(fo, ro) and torch.equal(fg, rg), (n, fullm)
        x = ((torch.rand(50000, 3, generator=g) - 0.5) * 1.9).to(dev)
        monkeypatch.setattr(ops, "SDF_FUSED_PE", True)
        of, gf = ops.sdf_value_grad(x, 50000, full, back, packing.H3_SCALE_LOG2, in_scale=2.0, out_scale=0.5)
        monkeypatch.setattr(ops, "SDF_FUSED_PE", False)
        orow, grow = ops.sdf_value_grad(x, 50000, full, back, packing.H3_SCALE_LOG2, in_scale=2.0, out_scale=0.5)
        assert torch.equal(of, orow) and torch.equal(gf, grow)
    ops.range_check(sync=True)


def test_colour_net_with_fused_encoding_is_bit_identical(dev, synth_weights):
    """k_color_mlp_h3<2>: [x | PE4(view) | normal] encoded inside the kernel against the tail-row form, ragged row counts."""
    from robir_amd import ops, packing
    blob = packing.pack_color_h3(synth_weights, dev)
    g = torch.Generator().manual_seed(77)
    for n in (1, 31, 128, 129, 5000, 33000):
        x = ((torch.rand(n, 3, generator=g) - 0.5)).to(dev)
        v = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
        nr = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
        out = torch.randn(n, 257, generator=g).to(dev)          # SDF-net rows: the 256 feature columns are read in place
        a = ops.color_mlp_h3_two(x, v, nr, out[:, 1:], blob, packing.H3_SCALE_LOG2, x_scale=2.0, feat_scale=2.0)
        b = ops.color_mlp_h3_points(x, v, nr, out[:, 1:], blob, packing.H3_SCALE_LOG2, x_scale=2.0, feat_scale=2.0, ring=False)
        c = ops.color_mlp_h3_points(x, v, nr, out[:, 1:], blob, packing.H3_SCALE_LOG2, x_scale=2.0, feat_scale=2.0, ring=True)
        assert torch.equal(a, b) and torch.equal(a, c), n
    # the chunk-stream kernel over many rounds per workgroup (persistent grid), run to run
    n = 300001
    x = ((torch.rand(n, 3, generator=g) - 0.5)).to(dev)
    v = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
    nr = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
    out = torch.randn(n, 257, generator=g).to(dev)
    a = ops.color_mlp_h3_points(x, v, nr, out[:, 1:], blob, packing.H3_SCALE_LOG2, x_scale=2.0, feat_scale=2.0, ring=False)
    for _ in range(3):
        c = ops.color_mlp_h3_points(x, v, nr, out[:, 1:], blob, packing.H3_SCALE_LOG2, x_scale=2.0, feat_scale=2.0, ring=True)
        assert torch.equal(a, c)
    ops.range_check(sync=True)


def test_f32_kernels_with_fused_encoding_are_bit_identical(dev, synth_weights):
    """rb_sdf_mlp_points (every mode: value rows, forward-mode tangent rows, library-grade activations) and rb_color_mlp_points
    against the row forms they replace."""
    from robir_amd import ops, packing, synth
    g = torch.Generator().manual_seed(41)
    for sd in (synth_weights, synth.synth_state_dict(0, scene="nonconvex")):
        full, dist = packing.pack_sdf(sd, dev, full=True), packing.pack_sdf(sd, dev, full=False)
        for n in (1, 33, 130, 3000):
            x = ((torch.rand(n, 3, generator=g) - 0.5) * 1.9).to(dev)
            for mode in (0, 1, 2, 3, 4, 6):
                blob = full if mode in (1, 3) else dist
                X = ops.feat_pe10(x, scale=2.0, jvp=(mode & 3) >= 2)
                r0, rg = ops.sdf_mlp(X, n, blob, mode, 0.5, 1.0)
                f0, fg = ops.sdf_mlp_points(x, n, blob, mode, 2.0, 0.5, 1.0)
                assert torch.equal(f0, r0), (n, mode)
                assert (rg is None and fg is None) or torch.equal(fg, rg), (n, mode)
    cb = packing.pack_color(synth_weights, dev)
    for n in (1, 31, 129, 2000):
        x = (torch.rand(n, 3, generator=g) - 0.5).to(dev)
        v = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
        nr = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
        out = torch.randn(n, 257, generator=g).to(dev)
        a = ops.color_mlp(ops.feat_color(x, v, nr, out[:, 1:], x_scale=2.0, feat_scale=2.0), cb)
        b = ops.color_mlp_points(x, v, nr, out[:, 1:], cb, x_scale=2.0, feat_scale=2.0)
        assert torch.equal(a, b), n


def test_small_nets_with_fused_encoding_are_bit_identical(dev, synth_weights):
    """Visibility MLP (both arithmetics, rep directions per point), the 64 -> 256 first-layer halves of the light-visibility net, the
    indirect-illumination lobe net ([PE10(x) | hdr_shift]) and a SparseAE encoder: straight from the points == the row forms."""
    from robir_amd import ops, packing
    g = torch.Generator().manual_seed(53)
    vis32, vis16 = packing.pack_vis(synth_weights, dev), packing.pack_vis_h3(synth_weights, dev)
    split = packing.pack_vis_split(synth_weights, dev)
    ill32, ill16 = packing.pack_illum(synth_weights, dev), packing.pack_illum_h3(synth_weights, dev)
    enc32, _ = packing.pack_sparse_ae(synth_weights, "envmap_material_network.spec_brdf_encoder_layer", dev)
    enc16 = packing.pack_sparse_ae_encoder_h3(synth_weights, "envmap_material_network.spec_brdf_encoder_layer", dev)
    for n, rep in ((1, 1), (17, 8), (130, 8), (700, 1), (4001, 3)):
        p = ((torch.rand(n, 3, generator=g) - 0.5) * 0.6).to(dev)
        d = torch.nn.functional.normalize(torch.randn(n * rep, 3, generator=g), dim=-1).to(dev)
        X = ops.feat_vis(p, d, rep=rep)
        assert torch.equal(ops.vis_mlp_points(p, d, vis32, rep), ops.vis_mlp(X, vis32)), (n, rep)
        assert torch.equal(ops.vis_mlp_points(p, d, vis16, rep, packing.H3_SCALE_LOG2), ops.vis_mlp_h3(X, vis16, packing.H3_SCALE_LOG2)), (n, rep)
        Xp = ops.feat_pe10(p)
        assert torch.equal(ops.linear_pe10_256(p, split["point"]), ops.linear_64_256(Xp, split["point"]))
        assert torch.equal(ops.linear_pe10_256(d, split["dir"]), ops.linear_64_256(ops.feat_pe10(d), split["dir"]))
        hdr = torch.rand(n, 1, generator=g).to(dev)
        Xh = ops.feat_pe10(p, extra=hdr)
        assert torch.equal(ops.wide_mlp_points(p, hdr, ill32, False), ops.illum_mlp(Xh, ill32))
        assert torch.equal(ops.wide_mlp_points(p, hdr, ill16, False, packing.H3_SCALE_LOG2), ops.wide_mlp_h3(Xh, ill16, False, packing.H3_SCALE_LOG2))
        assert torch.equal(ops.wide_mlp_points(p, None, enc32, True), ops.ae_encode(Xp, enc32))
        assert torch.equal(ops.wide_mlp_points(p, None, enc16, True, packing.H3_SCALE_LOG2), ops.wide_mlp_h3(Xp, enc16, True, packing.H3_SCALE_LOG2))
    ops.range_check(sync=True)


def test_cesr_nets_with_fused_encoding_are_bit_identical(dev):
    """shadow_net over one-hot labels and normal_net straight from the points == the same kernels on rb_feat_pe10 rows."""
    from robir_amd import ops, packing, synth
    c = synth.synth_cesr_nets(0)
    g = torch.Generator().manual_seed(61)
    sh = {"net." + k: v for k, v in c["shadow_net"].items()}
    no = {"net." + k: v for k, v in c["normal_net"].items()}
    sh32, sh16 = packing.pack_softplus512(sh, "net.", 191, dev), packing.pack_softplus512_h3(sh, "net.", 191, dev)
    no32, no16 = packing.pack_softplus512(no, "net.", 63, dev), packing.pack_softplus512_h3(no, "net.", 63, dev)
    for n in (1, 7, 100):
        p = ((torch.rand(n, 3, generator=g) - 0.5) * 0.6).to(dev)
        Xp = ops.feat_pe10(p)
        assert torch.equal(ops.cesr_net_points(p, n * 128, 2, sh32, 128), ops.cesr_net(Xp, n * 128, 2, sh32, 128))
        assert torch.equal(ops.cesr_net_points(p, n * 128, 2, sh16, 128, packing.H3_SCALE_LOG2),
                           ops.cesr_net_h3(Xp, n * 128, 2, sh16, packing.H3_SCALE_LOG2, 128))
        assert torch.equal(ops.cesr_net_points(p, n, 0, no32), ops.cesr_net(Xp, n, 0, no32))
        assert torch.equal(ops.cesr_net_points(p, n, 0, no16, 1, packing.H3_SCALE_LOG2), ops.cesr_net_h3(Xp, n, 0, no16, packing.H3_SCALE_LOG2))
    ops.range_check(sync=True)


def test_wide_ring_kernels_are_bit_identical_to_the_first_generation(dev, synth_weights):
    """The 512-wide nets on the chunk-stream machine (csrc/wide_ring.h): same products, same order, same epilogue arithmetic as
    k_softplus512_h3 / k_wide_mlp_h3 -- identical bits at ragged sizes, over many rounds per workgroup and run to run."""
    from robir_amd import ops, packing, synth
    c = synth.synth_cesr_nets(0)
    g = torch.Generator().manual_seed(67)
    s = packing.H3_SCALE_LOG2
    sh16 = packing.pack_softplus512_h3({"net." + k: v for k, v in c["shadow_net"].items()}, "net.", 191, dev)
    no16 = packing.pack_softplus512_h3({"net." + k: v for k, v in c["normal_net"].items()}, "net.", 63, dev)
    ill16 = packing.pack_illum_h3(synth_weights, dev)
    enc16 = packing.pack_sparse_ae_encoder_h3(synth_weights, "envmap_material_network.spec_brdf_encoder_layer", dev)
    for n in (1, 15, 64, 65, 1000, 40000):
        p = ((torch.rand(n, 3, generator=g) - 0.5) * 0.6).to(dev)
        hdr = torch.rand(n, 1, generator=g).to(dev)
        for nl in ((128,) if n > 1000 else (1, 3, 128)):
            if n * nl > 600000:
                continue
            a = ops.cesr_net_points(p, n * nl, 2, sh16, nl, s, ring=False)
            b = ops.cesr_net_points(p, n * nl, 2, sh16, nl, s, ring=True)
            assert torch.equal(a, b), ("shadow_net", n, nl)
        a = ops.cesr_net_points(p, n, 0, no16, 1, s, ring=False)
        for _ in range(2):
            assert torch.equal(a, ops.cesr_net_points(p, n, 0, no16, 1, s, ring=True)), ("normal_net", n)
        assert torch.equal(ops.wide_mlp_points(p, hdr, ill16, False, s, ring=False), ops.wide_mlp_points(p, hdr, ill16, False, s, ring=True)), n
        assert torch.equal(ops.wide_mlp_points(p, None, enc16, True, s, ring=False), ops.wide_mlp_points(p, None, enc16, True, s, ring=True)), n
        Xh, Xp = ops.feat_pe10(p, extra=hdr), ops.feat_pe10(p)
        assert torch.equal(ops.wide_mlp_h3(Xh, ill16, False, s, ring=False), ops.wide_mlp_h3(Xh, ill16, False, s, ring=True)), n
        assert torch.equal(ops.wide_mlp_h3(Xp, enc16, True, s, ring=False), ops.wide_mlp_h3(Xp, enc16, True, s, ring=True)), n
    p = ((torch.rand(3000, 3, generator=g) - 0.5) * 0.6).to(dev)
    a = ops.cesr_net_points(p, 3000 * 128, 2, sh16, 128, s, ring=False)
    for _ in range(3):
        assert torch.equal(a, ops.cesr_net_points(p, 3000 * 128, 2, sh16, 128, s, ring=True))
    ops.range_check(sync=True)


@pytest.mark.parametrize("weights", ["init", "trained_like"])
def test_sdf_value_grad_reverse_mode_f32(dev, synth_weights, weights):
    """The exact policy's reverse-mode gradient (k_sdf_mlp<5> + k_sdf_back_f32 + k_pe_grad_points): the 257 outputs are the value
    kernel's bit for bit, the gradient agrees with the forward-mode rows of the same engine to fp32 rounding and with the oracle's
    autograd to the value tests' tolerance -- at ragged sizes and across slabs."""
    from robir_amd import ops, packing
    from robir_oracle import nets as on
    sd = synth_weights if weights == "init" else _trained_like(synth_weights, 5)
    g = torch.Generator().manual_seed(71)
    blob, back = packing.pack_sdf(sd, dev, full=True), packing.pack_sdf_back(sd, dev)
    for n in (1, 33, 128, 129, 5000, 70000):
        x = ((torch.rand(n, 3, generator=g) - 0.5) * 1.2).to(dev)
        out, grad = ops.sdf_value_grad_f32(x, n, blob, back, in_scale=2.0, out_scale=0.5)
        ref_out, ref_grad = ops.sdf_mlp_points(x, n, blob, 3, 2.0, 0.5, 1.0)
        assert torch.equal(out, ops.sdf_mlp_points(x, n, blob, 1, 2.0, 0.5, 1.0)[0]), n
        assert torch.equal(out, ref_out), n
        assert rel_err(grad.cpu(), ref_grad.cpu()) <= 2e-5, (n, rel_err(grad.cpu(), ref_grad.cpu()))
    x = ((torch.rand(300, 3, generator=g) - 0.5) * 1.2)
    _, grad = ops.sdf_value_grad_f32(x.to(dev), 300, blob, back, in_scale=2.0, out_scale=0.5)
    ref_g = on.implicit_gradient(on.as_torch(sd), x)
    assert float(ref_g.abs().max()) > 0.5
    assert rel_err(grad.cpu(), ref_g) <= TOL
    old = ops.SDF_GRAD_SLAB
    ops.SDF_GRAD_SLAB = 4096
    try:
        x = ((torch.rand(10000, 3, generator=g) - 0.5) * 1.2).to(dev)
        a = ops.sdf_value_grad_f32(x, 10000, blob, back, 2.0, 0.5)
    finally:
        ops.SDF_GRAD_SLAB = old
    b = ops.sdf_value_grad_f32(x, 10000, blob, back, 2.0, 0.5)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


@pytest.mark.parametrize("weights", ["init", "trained_like"])
def test_sdf_exact_operand_kernel(dev, synth_weights, weights):
    """k_sdf_x6 (csrc/sdf_x6.hip: exact three-piece operands, six products per multiply-add): values against the f32-input-MFMA kernel
    and the oracle, the sdf-only form against the full one, the reverse-mode gradient through its sigmoid tiles against the fp32 one --
    ragged sizes, many rounds per workgroup, run to run."""
    from robir_amd import ops, packing
    from robir_oracle import nets as on
    sd = synth_weights if weights == "init" else _trained_like(synth_weights, 5)
    g = torch.Generator().manual_seed(91)
    b32, back = packing.pack_sdf(sd, dev, full=True), packing.pack_sdf_back(sd, dev)
    x6f, x6d = packing.pack_sdf_x6(sd, dev, full=True), packing.pack_sdf_x6(sd, dev, full=False)
    back6 = packing.pack_sdf_back_x6(sd, dev) + (packing.pack_sdf_back_x6(sd, dev, two_tile=True)[0],)     # as nets.packed_back_x6
    for n in (1, 15, 64, 65, 1000, 40000, 300001):      # 40000 rows: one tile per wave, 300001: two (ops.sdf_two_tile)
        x = ((torch.rand(n, 3, generator=g) - 0.5) * 1.2).to(dev)
        ref = ops.sdf_mlp_points(x, n, b32, 1, 2.0, 0.5, 1.0)[0]
        out = ops.sdf_points_x6(x, n, x6f, True, 2.0, 0.5)
        assert rel_err(out.cpu(), ref.cpu()) <= 2e-6, (n, rel_err(out.cpu(), ref.cpu()))
        d = ops.sdf_points_x6(x, n, x6d, False, 2.0, 0.5)
        assert torch.equal(d, out[:, 0]), n
        assert torch.equal(out, ops.sdf_points_x6(x, n, x6f, True, 2.0, 0.5)), n
        if n in (1, 65, 1000, 40000, 300001):
            o2, grad = ops.sdf_value_grad_x6(x, n, x6f, back6, 2.0, 0.5)
            assert torch.equal(o2, out), n
            _, g32 = ops.sdf_value_grad_f32(x, n, b32, back, 2.0, 0.5)
            assert rel_err(grad.cpu(), g32.cpu()) <= 2e-5, (n, rel_err(grad.cpu(), g32.cpu()))
            assert torch.equal(grad, ops.sdf_value_grad_x6(x, n, x6f, back6, 2.0, 0.5)[1]), n
    # the two forms against each other on the same rows (another fp32 summation order), on both sides of the switch and ragged
    for n in (100, 33000, 70001):
        x = ((torch.rand(n, 3, generator=g) - 0.5) * 1.2).to(dev)
        res = {}
        for rows in (0, 1 << 60):       # 0: two tiles per wave at every size | never
            old, ops.SDF_TWO_TILE_MIN_ROWS = ops.SDF_TWO_TILE_MIN_ROWS, rows
            try:
                res[rows] = (ops.sdf_points_x6(x, n, x6d, False, 2.0, 0.5),) + ops.sdf_value_grad_x6(x, n, x6f, back6, 2.0, 0.5)
            finally:
                ops.SDF_TWO_TILE_MIN_ROWS = old
        two, one = res[0], res[1 << 60]
        assert rel_err(two[0].cpu(), one[0].cpu()) <= 2e-6 and rel_err(two[1].cpu(), one[1].cpu()) <= 2e-6, n
        assert rel_err(two[2].cpu(), one[2].cpu()) <= 2e-5, (n, rel_err(two[2].cpu(), one[2].cpu()))
    x = ((torch.rand(300, 3, generator=g) - 0.5) * 1.2)
    osd = on.as_torch(sd)
    out = ops.sdf_points_x6(x.to(dev), 300, x6f, True, 2.0, 0.5)
    assert rel_err(out.cpu(), on.implicit_forward(osd, x)) <= TOL
    _, grad = ops.sdf_value_grad_x6(x.to(dev), 300, x6f, back6, 2.0, 0.5)
    assert rel_err(grad.cpu(), on.implicit_gradient(osd, x)) <= TOL
    ops.range_check(sync=True)


def test_exact_operand_kernels_report_range_overflow(dev, synth_weights):
    """The range sentinel of the exact-operand SDF / colour kernels, both forms: a hidden activation beyond the f16 range of the leading
    piece (the >= 0 activations are tracked by their raw pattern, one instruction per pair) and an input beyond it (the signed inputs of a
    round, folded in behind them) are reported by rb_range_check; the unscaled net reports nothing."""
    from robir_amd import ops, packing, _lib
    g = torch.Generator().manual_seed(7)
    n = 300
    x = ((torch.rand(n, 3, generator=g) - 0.5) * 1.2).to(dev)
    v = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
    feat = torch.randn(n, 257, generator=g).to(dev)
    big = dict(synth_weights)
    for k in list(big):                                  # layers 0 and 1 of the SDF net and of the colour net x 3000: layer 1 puts out ~1e7
        if k.endswith((".lin0.weight_g", ".lin1.weight_g", ".lin0.bias", ".lin1.bias")) and ("sdf_network" in k or "color_network" in k):
            big[k] = big[k] * 3000.0
    ops.range_check(sync=True)
    for two in (False, True):
        old, ops.SDF_TWO_TILE_MIN_ROWS = ops.SDF_TWO_TILE_MIN_ROWS, (0 if two else 1 << 60)
        try:
            ops.sdf_points_x6(x, n, packing.pack_sdf_x6(synth_weights, dev, full=True), True, 2.0, 0.5)
            ops.color_x6_points(x, v, v, feat[:, 1:], packing.pack_color_x6(synth_weights, dev))
            ops.range_check(sync=True)                                                   # in range: nothing to report
            ops.sdf_points_x6(x, n, packing.pack_sdf_x6(big, dev, full=True), True, 2.0, 0.5)
            with pytest.raises(_lib.RobirHipError, match="overflowed its activation range") as e:
                ops.range_check(sync=True)
            assert "SDF" in str(e.value), str(e.value)
            ops.color_x6_points(x, v, v, feat[:, 1:], packing.pack_color_x6(big, dev))
            with pytest.raises(_lib.RobirHipError, match="overflowed its activation range") as e:
                ops.range_check(sync=True)
            assert "colour" in str(e.value), str(e.value)
            ops.sdf_points_x6(x * 1.0e6, n, packing.pack_sdf_x6(synth_weights, dev, full=True), True, 2.0, 0.5)      # an input beyond the range
            with pytest.raises(_lib.RobirHipError, match="overflowed its activation range"):
                ops.range_check(sync=True)
            ops.color_x6_points(x, v, v, feat[:, 1:] * 1.0e6, packing.pack_color_x6(synth_weights, dev))
            with pytest.raises(_lib.RobirHipError, match="overflowed its activation range"):
                ops.range_check(sync=True)
        finally:
            ops.SDF_TWO_TILE_MIN_ROWS = old
    ops.range_check(sync=True)


def test_color_exact_operand_kernel(dev, synth_weights):
    """k_color_x6 / k_color_x6t (csrc/color_x6.hip, color_x6t.hip) against the f32-input-MFMA colour kernel on the same inputs: ragged sizes, many rounds, run to run."""
    from robir_amd import ops, packing
    g = torch.Generator().manual_seed(97)
    b32, x6 = packing.pack_color(synth_weights, dev), packing.pack_color_x6(synth_weights, dev)
    for n in (1, 15, 64, 65, 129, 5000, 300001):
        x = ((torch.rand(n, 3, generator=g) - 0.5)).to(dev)
        v = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
        nr = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
        out = torch.randn(n, 257, generator=g).to(dev)
        ref = ops.color_mlp_points(x, v, nr, out[:, 1:], b32, x_scale=2.0, feat_scale=2.0)
        for two in (False, True):       # one tile per wave (color_x6.hip) | two (color_x6t.hip): either form at every size
            a = ops.color_x6_points(x, v, nr, out[:, 1:], x6, x_scale=2.0, feat_scale=2.0, two_tile=two)
            assert float((a - ref).abs().max()) <= 2e-6, (n, two, float((a - ref).abs().max()))
            assert torch.equal(a, ops.color_x6_points(x, v, nr, out[:, 1:], x6, x_scale=2.0, feat_scale=2.0, two_tile=two)), (n, two)
    ops.range_check(sync=True)


def test_vis_exact_operand_kernel(dev, synth_weights):
    """k_vis_x6 (csrc/vis_x6.hip) against the f32-input-MFMA visibility MLP on the same points and directions: ragged sizes, several
    directions per point, many rounds, run to run."""
    from robir_amd import ops, packing
    g = torch.Generator().manual_seed(101)
    b32, x6 = packing.pack_vis(synth_weights, dev), packing.pack_vis_x6(synth_weights, dev)
    for n, rep in ((1, 1), (17, 8), (130, 8), (5000, 1), (40001, 8)):
        p = ((torch.rand(n, 3, generator=g) - 0.5) * 0.6).to(dev)
        d = torch.nn.functional.normalize(torch.randn(n * rep, 3, generator=g), dim=-1).to(dev)
        ref = ops.vis_mlp_points(p, d, b32, rep)
        a = ops.vis_x6_points(p, d, x6, rep)
        assert float((a - ref).abs().max()) <= 5e-6, (n, rep, float((a - ref).abs().max()))     # logits of O(1): fp32 rounding of either sum
        assert torch.equal(a, ops.vis_x6_points(p, d, x6, rep)), (n, rep)
    ops.range_check(sync=True)


def test_wide_exact_operand_kernels(dev, synth_weights):
    """k_wide_x6 (csrc/wide_x6.hip: half-chunk stream of the 512-wide nets) against the f32-input-MFMA kernels on the same points: the
    SparseAE encoder and the indirect-illumination decoder, ragged sizes, many rounds, run to run."""
    from robir_amd import ops, packing
    g = torch.Generator().manual_seed(103)
    ill32, ill6 = packing.pack_illum(synth_weights, dev), packing.pack_illum_x6(synth_weights, dev)
    pre = "envmap_material_network.spec_brdf_encoder_layer"
    enc32, _ = packing.pack_sparse_ae(synth_weights, pre, dev)
    enc6 = packing.pack_sparse_ae_encoder_x6(synth_weights, pre, dev)
    for n in (1, 15, 64, 65, 940, 40001):
        p = ((torch.rand(n, 3, generator=g) - 0.5) * 0.6).to(dev)
        hdr = torch.rand(n, 1, generator=g).to(dev)
        for ref, got in ((ops.wide_mlp_points(p, hdr, ill32, False), lambda: ops.wide_x6_points(p, hdr, ill6, False)),
                         (ops.wide_mlp_points(p, None, enc32, True), lambda: ops.wide_x6_points(p, None, enc6, True))):
            a = got()
            scale = max(1.0, float(ref.abs().max()))
            assert float((a - ref).abs().max()) <= 5e-6 * scale, (n, float((a - ref).abs().max()), scale)
            assert torch.equal(a, got()), n
        Xp = ops.feat_pe10(p)
        assert torch.equal(ops.wide_x6(Xp, enc6, True), ops.wide_x6_points(p, None, enc6, True)), n      # the row form: same bits
    ops.range_check(sync=True)


def test_cesr_exact_operand_kernels(dev):
    """k_cesr_x6 (csrc/cesr_x6.hip) against the f32-input-MFMA CESR kernels on the same points: normal_net, and shadow_net over one-hot
    labels -- ragged sizes, several label counts, many rounds, run to run."""
    from robir_amd import ops, packing, synth
    c = synth.synth_cesr_nets(0)
    g = torch.Generator().manual_seed(107)
    sh = {"net." + k: v for k, v in c["shadow_net"].items()}
    no = {"net." + k: v for k, v in c["normal_net"].items()}
    sh32, sh6 = packing.pack_softplus512(sh, "net.", 191, dev), packing.pack_softplus512_x6(sh, "net.", 191, dev)
    no32, no6 = packing.pack_softplus512(no, "net.", 63, dev), packing.pack_softplus512_x6(no, "net.", 63, dev)
    for n in (1, 15, 64, 65, 1000, 40001):
        p = ((torch.rand(n, 3, generator=g) - 0.5) * 0.6).to(dev)
        ref = ops.cesr_net_points(p, n, 0, no32)
        a = ops.cesr_net_x6_points(p, n, 0, no6)
        scale = max(1.0, float(ref.abs().max()))
        assert float((a - ref).abs().max()) <= 5e-6 * scale, ("normal_net", n, float((a - ref).abs().max()), scale)
        assert torch.equal(a, ops.cesr_net_x6_points(p, n, 0, no6)), n
        for nl in ((128,) if n > 1000 else (1, 3, 128)):
            if n * nl > 600000:
                continue
            ref = ops.cesr_net_points(p, n * nl, 2, sh32, nl)
            a = ops.cesr_net_x6_points(p, n * nl, 2, sh6, nl)
            scale = max(1.0, float(ref.abs().max()))
            assert float((a - ref).abs().max()) <= 5e-6 * scale, ("shadow_net", n, nl, float((a - ref).abs().max()), scale)
            assert torch.equal(a, ops.cesr_net_x6_points(p, n * nl, 2, sh6, nl)), (n, nl)
    ops.range_check(sync=True)
