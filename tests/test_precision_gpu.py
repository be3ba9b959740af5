"""Where does the split-precision (f16 hi/lo pairs on the f16 MFMA, fp32 accumulate) arithmetic sit relative to fp32?

One Material chunk: the whole per-hit chain (indirect-illumination net -> NeuS normal -> material auto-encoders -> 128-lobe
light visibility through the visibility MLP -> BRDF-lobe visibility -> SG shading) evaluated on IDENTICAL surface points,
view directions and random draws by
  (r64)  the oracle in float64            -- the anchor: what the reference's formulas give without rounding noise,
  (o32)  the oracle in float32            -- the reference's own arithmetic (PyTorch CPU fp32),
  (k32)  the HIP kernels, exact mode      -- every MLP on v_mfma_f32_16x16x4_f32 (ROBIR_MLP_PRECISION=fp32),
  (kh3)  the HIP kernels, default mode    -- split precision.
Asserted per field: err(kh3 vs r64) <= 1.25 * err(k32 vs r64) + F and <= 2 * err(o32 vs r64) + F, on the median and on the
99th percentile of the per-entry errors.  F = 16 * 2^-22 = 3.8e-6 is the representation floor of the split: an (hi, lo) half
pair carries 22 bits where fp32 carries 24, so a DIRECT network output (the NeuS normal: measured 2e-6 at p99 against 5e-7 for
the exact kernels) sits a few 2^-22 above fp32 while being 50x below the 1e-4 bar; every SHADING field (errors 1e-5..1e-3,
set by the conditioning of the SG formulas, not by the MLP arithmetic) must then be within 1.25x of the exact kernels.  The maximum is recorded but not compared: it is set by the handful of visibility samples that sit on the
n.d > 1e-6 cull (a 1e-7 change of the normal moves one sample in or out of a lobe's 32: 3e-2 of that lobe's visibility) and
those flip between ANY two evaluations that round differently -- they are counted instead.
All numbers go to gpurun_out/test_metrics.jsonl.
"""
import os

import numpy as np
import pytest
import torch

from conftest import err_entries, record_metric

pytestmark = pytest.mark.gpu

FIELDS = ("sg_rgb", "indir_rgb", "sg_diffuse_rgb", "sg_specular_rgb", "indir_diffuse_rgb", "indir_specular_rgb",
          "vis_shadow", "diffuse_albedo", "roughness", "metallic", "normals", "normal_map")


def _oracle_chain(sd, pts, view, hdr, draws, dtype):
    """IDRNetwork.forward's per-hit part (implicit_differentiable_renderer.py:340-358,386-455) + the PBR hook."""
    from robir_oracle import nets as on, renderer as orend
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        c = lambda t: t.to(dtype)
        sdd = {k: (c(v) if v.is_floating_point() else v) for k, v in sd.items()}
        dr = {k: c(v) for k, v in draws.items()}
        sgs, integ = on.indirect_illum(sdd, c(pts), c(hdr), dr["illum_randn"])
        return orend.pbr_sg_render(sdd, c(pts), c(view), sgs, integ, dr, testing=True)
    finally:
        torch.set_default_dtype(old)


@pytest.mark.parametrize("seed,variance,sharp", [(0, 0.3, False), (3, 0.6, True)])
def test_chained_error_budget(monkeypatch, seed, variance, sharp):
    from robir_amd import renderer, sg_render, synth
    from robir_oracle import nets as on
    dev = torch.device("cuda:0")
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    model = renderer.build_synthetic_model(dev, seed=seed, variance=variance, sharp_light=sharp)
    sd = on.as_torch(synth.synth_state_dict(seed, variance=variance, sharp_light=sharp))
    # 256 pixels around the image centre of the 64x64 view: all but the principal-point ray (exactly axis-parallel: NaN in the
    # reference's slab test as well) hit the object
    uv, pose, K = synth.synth_camera(64, 64)
    sel = np.concatenate([np.arange(r * 64 + 24, r * 64 + 40) for r in range(24, 40)])
    inp = {"uv": torch.from_numpy(uv[sel]).to(dev)[None], "pose": torch.from_numpy(pose).to(dev)[None],
           "intrinsics": torch.from_numpy(K).to(dev)[None], "object_mask": torch.ones(1, 256, dtype=torch.bool, device=dev),
           "hdr_shift": torch.full((256, 1), 0.5, device=dev)}
    hit = model(inp, trainstage="Illum", draws={})["network_object_mask"].cpu()
    n_hit = int(hit.sum())
    assert n_hit >= 200        # the column and the row through the principal point are axis-parallel rays (no hit, like the reference)
    draws = {k: torch.from_numpy(v) for k, v in synth.pbr_draws(seed + 40, n_hit, chunk_id=9).items()}
    dd = {k: v.to(dev) for k, v in draws.items()}
    outs = {}
    for mode, vis in (("fp32", "fp32"), ("f16x3", "f16x3-auto")):
        monkeypatch.setenv("ROBIR_MLP_PRECISION", mode)
        monkeypatch.setattr(sg_render, "VIS_PRECISION", vis)
        outs[mode] = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in
                      model(inp, trainstage="Material", train_spec=True, draws=dd).items()}
    k32, kh3 = outs["fp32"], outs["f16x3"]
    assert torch.equal(k32["network_object_mask"], hit) and torch.equal(kh3["network_object_mask"], hit)
    assert torch.equal(k32["points"][hit], kh3["points"][hit])          # the cast is geometry code: identical in both modes
    pts, view, hdr = k32["points"][hit], -k32["ray_dirs"][hit], inp["hdr_shift"].cpu()[hit]
    r64 = _oracle_chain(sd, pts, view, hdr, draws, torch.float64)
    o32 = _oracle_chain(sd, pts, view, hdr, draws, torch.float32)
    worst_ratio = 0.0
    for f in FIELDS:
        ref = r64[f].expand(-1, 3) if (f == "roughness") else r64[f]
        o = o32[f].expand(-1, 3) if (f == "roughness") else o32[f]
        e = {"o32": err_entries(o, ref), "k32": err_entries(k32[f][hit], ref), "kh3": err_entries(kh3[f][hit], ref)}
        st = {}
        for name, v in e.items():
            st[name + "_p50"], st[name + "_p99"] = float(v.quantile(0.5)), float(v.quantile(0.99))
            st[name + "_max"], st[name + "_n_gt_1e-3"] = float(v.max()), int((v > 1e-3).sum())
        record_metric(f"chained_error_budget/seed{seed}/{f}", entries=int(e["kh3"].numel()), **st)
        print(f"{f:22s} " + "  ".join(f"{n}: p50 {st[n + '_p50']:.2e} p99 {st[n + '_p99']:.2e} max {st[n + '_max']:.2e} "
                                      f"n>1e-3 {st[n + '_n_gt_1e-3']}" for n in ("o32", "k32", "kh3")))
        slack = 16 * 2.0 ** -22
        for q in ("_p50", "_p99"):
            assert st["kh3" + q] <= 1.25 * st["k32" + q] + slack, (f, q, st)
            assert st["kh3" + q] <= 2.0 * st["o32" + q] + slack, (f, q, st)
            worst_ratio = max(worst_ratio, st["kh3" + q] / (st["k32" + q] + slack))
        # cull flips: the split-precision run may not have more large outliers than fp32 arithmetic itself produces
        assert st["kh3_n_gt_1e-3"] <= max(2 * st["o32_n_gt_1e-3"], st["k32_n_gt_1e-3"] + 3, 3), (f, st)
        # and the bulk sits at north_star's bar -- or where the reference's own fp32 arithmetic sits, for the SG terms that
        # are ill-conditioned in fp32 (lambda_trick / hemisphere_int cancel large exponentials: o32 itself is at 2e-4..1e-3)
        assert st["kh3_p99"] <= max(1e-4, 2.0 * st["o32_p99"]), (f, st)
    record_metric(f"chained_error_budget/seed{seed}/worst_ratio_h3_over_fp32mfma", ratio=worst_ratio)
    # ROBIR_PRECISION=f16 (light-visibility MLP in plain f16, one product; every other net exact): NARROWER than fp32 -- measured against
    # the same float64 anchor, printed and recorded (DESIGN.md quotes the table), held only to a sanity band: no parity claim
    monkeypatch.setenv("ROBIR_MLP_PRECISION", "f16x6")
    monkeypatch.setattr(sg_render, "VIS_PRECISION", "f16x1")
    kf = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in model(inp, trainstage="Material", train_spec=True, draws=dd).items()}
    assert torch.equal(kf["network_object_mask"], hit)
    for f in FIELDS:
        ref = r64[f].expand(-1, 3) if (f == "roughness") else r64[f]
        v = err_entries(kf[f][hit], ref)
        st = dict(p50=float(v.quantile(0.5)), p99=float(v.quantile(0.99)), max=float(v.max()), n_gt_1e3=int((v > 1e-3).sum()))
        record_metric(f"f16_mode/seed{seed}/{f}", entries=int(v.numel()), **st)
        print(f"f16 mode {f:22s} p50 {st['p50']:.2e} p99 {st['p99']:.2e} max {st['max']:.2e} n>1e-3 {st['n_gt_1e3']}")
        assert st["p99"] <= 5e-2, (f, st)


def _scaled_vis_weights(sd_np, s):
    """Same visibility function, hidden activations of layer l larger by s^((l+1)/4) (the last hidden layer by s): a ReLU net
    is positively homogeneous, so scaling every hidden layer's weight by f = s^(1/4), its bias by f^(l+1), and the output
    layer's weight by 1/s leaves the logits unchanged.  (Spreading the factor keeps the WEIGHTS inside the range the
    split-precision packing accepts; the activations are what this test is about.)"""
    out = dict(sd_np)
    P = "visibility_network.vis_layer."
    f = float(s) ** 0.25
    for i, l in enumerate((0, 2, 4, 6)):
        out[P + f"{l}.weight"] = sd_np[P + f"{l}.weight"] * np.float32(f)
        out[P + f"{l}.bias"] = sd_np[P + f"{l}.bias"] * np.float32(f ** (i + 1))
    out[P + "8.weight"] = sd_np[P + "8.weight"] / np.float32(f ** 4)
    return out


def _max_hidden_activation(sd, p, d):
    from robir_oracle.encoding import pe
    h = torch.cat([pe(p, 10), pe(d, 10)], -1)
    m = 0.0
    for l in (0, 2, 4, 6):
        h = torch.relu(h @ sd[f"visibility_network.vis_layer.{l}.weight"].t() + sd[f"visibility_network.vis_layer.{l}.bias"])
        m = max(m, float(h.max()))
    return m


def test_activation_range_sentinel(monkeypatch):
    """Split-precision operands are (hi, lo) half pairs: beyond 65504 / lift they silently lose precision.  Every _h3 kernel
    tracks the largest hi half it consumed and reports through rb_range_check -> RobirHipError.  Stress weights with hidden
    activations ~1e3 must pass parity and raise nothing; activations ~2e5 must be reported (and the exact kernels, selected as
    the message says, still match the oracle)."""
    from robir_amd import nets, ops, sg_render, synth, _lib
    from robir_oracle import nets as on, sg as osg
    from conftest import rel_err
    dev = torch.device("cuda:0")
    base = synth.synth_state_dict(0, variance=0.3)
    g = torch.Generator().manual_seed(1)
    n = 96
    pts = torch.randn(n, 3, generator=g) * 0.2
    nrm = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    lgt = torch.from_numpy(base["envmap_material_network.lgtSGs"])
    u_t, u_p = torch.rand(128, 32, generator=g), torch.rand(128, 32, generator=g)
    ops.range_check(sync=True)                                       # start from a clean sentinel

    def scale_for(target):
        """s such that the largest hidden activation of the scaled net on these inputs is `target`."""
        lo, hi = 1.0, 1e12
        for _ in range(60):
            mid = (lo * hi) ** 0.5
            if _max_hidden_activation(on.as_torch(_scaled_vis_weights(base, mid)), pts, dirs) < target:
                lo = mid
            else:
                hi = mid
        return hi

    def run(s, mlp_mode, vis_mode):
        sdn = _scaled_vis_weights(base, s)
        sd = on.as_torch(sdn)
        v = nets.VisNetwork(10, 10, [256] * 4)
        v.load_state_dict({k[len("visibility_network."):]: torch.from_numpy(x) for k, x in sdn.items()
                           if k.startswith("visibility_network.")})
        v = v.to(dev).eval()
        monkeypatch.setenv("ROBIR_MLP_PRECISION", mlp_mode)
        monkeypatch.setattr(sg_render, "VIS_PRECISION", vis_mode)
        logits = v(pts.to(dev), dirs.to(dev)).cpu()
        lobes = torch.nn.functional.normalize(lgt[:, :3], dim=-1)
        vis = sg_render.get_diffuse_visibility(pts.to(dev), nrm.to(dev), v, lobes.to(dev), lgt[:, 3:4].abs().to(dev), nsamp=32,
                                               draws={"dvis_theta": u_t.to(dev), "dvis_phi": u_p.to(dev)}).cpu()
        ref_logits = on.vis_logits(sd, pts, dirs)
        ref_vis = osg.diffuse_visibility(pts, nrm, lambda p, d: on.vis_logits(sd, p, d), lobes, lgt[:, 3:4].abs(), u_t, u_p)
        return logits, vis, ref_logits, ref_vis

    s_ok = scale_for(1.0e3)                                          # hidden activations up to ~1e3
    logits, vis, ref_logits, ref_vis = run(s_ok, "f16x3", "f16x3-auto")
    ops.range_check(sync=True)                                       # nothing to report
    e1, e2 = rel_err(logits, ref_logits), rel_err(vis, ref_vis)
    record_metric("range_sentinel/act_1e3", scale=s_ok, logits=e1, light_vis=e2)
    assert e1 <= 1e-4 and e2 <= 1e-4, (e1, e2)
    s_bad = scale_for(2.0e5)                                         # beyond the f16 range of the hi halves
    run(s_bad, "f16x3", "f16x3-auto")
    with pytest.raises(_lib.RobirHipError, match="overflowed its activation range") as ei:
        ops.range_check(sync=True)
    assert "light-visibility" in str(ei.value) and "rb_vis_mlp_h3" in str(ei.value)
    ops.range_check(sync=True)                                       # reading clears the words
    run(s_bad, "f16x6", "f16x6")                                     # the exact-operand kernels carry the same f16 range on their leading piece
    with pytest.raises(_lib.RobirHipError, match="overflowed its activation range") as ei6:
        ops.range_check(sync=True)
    assert "light-visibility" in str(ei6.value) and "rb_vis_x6_points" in str(ei6.value)
    ops.range_check(sync=True)
    logits, vis, ref_logits, ref_vis = run(s_bad, "fp32", "fp32")    # the remedy the message names
    ops.range_check(sync=True)
    e1, e2 = rel_err(logits, ref_logits), rel_err(vis, ref_vis)
    record_metric("range_sentinel/act_2e5_exact_kernels", logits=e1, light_vis=e2)
    assert e1 <= 1e-4 and e2 <= 1e-4, (e1, e2)
