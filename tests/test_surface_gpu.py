"""The public helper functions the overlay gained in round 6 (every name the reference modules export) on the
GPU against the reference's OWN outputs (tests/golden/surface_helpers.npz, recorded by oracle/gen_golden_r6.py under the reference)."""
import json
import types

import numpy as np
import pytest
import torch

from conftest import rel_err, load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-4          # north_star's bound; element-wise helpers land at a few ulp


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def g():
    return load_golden("surface_helpers")


def _t(g, dev, *keys):
    return [torch.from_numpy(g[k]).to(dev) for k in keys]


def _neus(dev, synth_weights):
    from robir_amd import nets, synth
    m = nets.NeuSModel(embed="PE")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.neus_state_dict(synth_weights).items()})
    return m.to(dev).eval()


def test_tone_mapping_curve_functions(dev, g):
    """model/color_correction.py:31-73 as free functions: no clamp of t, x of any shape, t [n,1] / scalar."""
    from robir_amd import color_correction as cc
    x, t = _t(g, dev, "tm_x", "tm_t")
    assert rel_err(cc.aces_fn(x).cpu(), g["tm_aces_fn"]) <= 1e-6
    assert rel_err(cc.aces_inv(x.clamp(max=1.0)).cpu(), g["tm_aces_inv"]) <= 2e-6
    for name in ("warp_aces_inv", "warp_aces_fn", "scale_aces_inv", "scale_aces_fn", "ln_space_fn", "ln_space_inv", "identity_fn"):
        xin = x.clamp(max=0.9) if name.endswith("inv") else x
        e = rel_err(getattr(cc, name)(xin, t).cpu(), g["tm_" + name])
        assert e <= 1e-5, (name, e)
    assert rel_err(cc.scale_aces_fn(x, torch.tensor(0.37)).cpu(), g["tm_scale_aces_fn_scalar_t"]) <= 1e-5
    assert float(cc.aces_fn(0.5)) == pytest.approx(float(0.5 * (2.51 * 0.5 + 0.03) / (0.5 * (2.43 * 0.5 + 0.59) + 0.14)), rel=1e-6)   # non-tensor input (:32-33)
    # the module-level curves and the ACESToneMapping object agree where the object's clamp is inactive
    tm = cc.ACESToneMapping(hdr_mode=0).to(dev)
    tin = t.clamp(1e-4, 1.0)
    assert rel_err(tm.hdr2ldr(x, tin).cpu(), cc.scale_aces_fn(x, tin).cpu()) <= 1e-6


def test_sample_pdf_up_sample_cat_z_vals(dev, g, synth_weights):
    from robir_amd import sdf_render as rs
    neus = _neus(dev, synth_weights)
    bins, w, u = _t(g, dev, "sp_bins", "sp_w", "sp_u")
    assert rel_err(rs.sample_pdf(bins, w, 16, det=True).cpu(), g["sp_det"]) <= 1e-5
    assert rel_err(rs.sample_pdf(bins, w, 11, det=False, u=u).cpu(), g["sp_rand"]) <= 1e-5
    torch.manual_seed(3)
    a = rs.sample_pdf(bins, w, 11)                         # det=False draws on the device, inside the bins, seed-reproducible
    torch.manual_seed(3)
    assert torch.equal(a, rs.sample_pdf(bins, w, 11)) and bool((a >= bins[:, :1]).all()) and bool((a <= bins[:, -1:]).all())
    ro, rd, z, s0, zu = _t(g, dev, "ns_ro", "ns_rd", "ns_z", "ns_sdf", "ns_zu")
    assert rel_err(neus.sdf(rs._ray_points(ro, rd, z)[0]).reshape(z.shape).cpu(), g["ns_sdf"]) <= TOL
    zn = rs.up_sample(ro, rd, z, s0, 8, 128.0, neus.radius())
    assert rel_err(zn.cpu(), g["ns_up"]) <= 1e-5
    zn_ref = torch.from_numpy(g["ns_up"]).to(dev)
    z2, s2 = rs.cat_z_vals(neus, ro, rd, z, zn_ref, s0, last=False)
    assert rel_err(z2.cpu(), g["ns_cat_z"]) == 0.0 and rel_err(s2.cpu(), g["ns_cat_sdf"]) <= TOL
    z3, s3 = rs.cat_z_vals(neus, ro, rd, z, zn_ref, s0, last=True)
    assert rel_err(z3.cpu(), g["ns_cat_last_z"]) == 0.0 and torch.equal(s3, s0)          # last=True hands the old SDFs back (:122-130)
    z4, s4 = rs.cat_z_vals(neus, ro, rd, z, zu, s0, last=False)                        # unsorted new depths
    assert rel_err(z4.cpu(), g["ns_cat_u_z"]) == 0.0 and rel_err(s4.cpu(), g["ns_cat_u_sdf"]) <= TOL


def test_render_core_dict(dev, g, synth_weights):
    """render_core's ten entries (model/sdf_render.py:249-260) on the fused kernels and through the generic ISDF path."""
    from robir_amd import sdf_render as rs
    neus = _neus(dev, synth_weights)
    ro, rd, z2 = _t(g, dev, "ns_ro", "ns_rd", "ns_cat_z")
    rc = rs.render_core(ro, rd, z2, 2.0 / 24, neus, background_rgb=torch.ones(1, 3, device=dev))
    want = {k[3:]: v for k, v in g.items() if k.startswith("rc_") and k != "rc_nobg_color"}
    assert sorted(rc) == sorted(want)
    for k, tol in (("color", TOL), ("sdf", TOL), ("dists", 0.0), ("gradients", 2e-4), ("s_val", 1e-6), ("mid_z_vals", 0.0), ("weights", 5e-3),
                   ("cdf", 1e-3), ("gradient_error", TOL), ("inside_sphere", 0.0)):
        assert tuple(rc[k].shape) == tuple(want[k].shape), (k, rc[k].shape, want[k].shape)
        e = rel_err(rc[k].cpu(), want[k])
        assert e <= tol, (k, e)
    assert rel_err(rs.render_core(ro, rd, z2, 2.0 / 24, neus)["color"].cpu(), g["rc_nobg_color"]) <= TOL

    class Proto:                                            # any object with the ISDF methods (model/sdf_render.py:19-34)
        radius, dev = neus.radius, neus.dev
        sdf, sdf_and_feat, grad, color = neus.sdf, neus.sdf_and_feat, neus.grad, neus.color
    rc2 = rs.render_core(ro, rd, z2, 2.0 / 24, Proto(), background_rgb=torch.ones(1, 3, device=dev))
    for k in ("color", "weights", "gradient_error"):
        assert rel_err(rc2[k].cpu(), rc[k].cpu()) <= 2e-5, k


def test_wrap_renderer(dev, g, synth_weights):
    """model/sdf_render.py:377-426: dict layout like the reference's; values = render_neus with the colour function swapped in."""
    from robir_amd import sdf_render as rs
    neus = _neus(dev, synth_weights)
    fake = types.SimpleNamespace(implicit_network=types.SimpleNamespace(neus_model=neus))
    p, d = _t(g, dev, "wr_points", "wr_dirs")
    color_fn = lambda q: torch.sigmoid(q * 3.0)            # noqa: E731
    torch.manual_seed(11)
    out = rs.wrap_renderer(fake, color_fn, {"points": p, "dirs": d}, near=0.4, far=1.4, is_eval=True)
    assert sorted(out) == list(g["wr_keys"])
    got = [json.dumps([k, list(out[k].shape), str(out[k].dtype)]) for k in sorted(out)]
    assert got == list(g["wr_shapes"])
    ones = torch.ones(p.shape[0], 1, device=dev)
    rays = rs.Rays(p * 2.0, d, d, ones * 0.001, ones, ones * 0.4, ones * 1.4)
    torch.manual_seed(11)
    ref = rs.render_neus(rays, neus, 1.0, n_samples=32, n_importance=32, n_outside=0, up_sample_steps=2)
    assert torch.equal(out["sg_rgb"], ref["rgb"]) and torch.equal(out["normals"], ref["grad"])
    # the swapped colour really is what was composited: a constant colour function gives acc * c + (1 - acc)
    torch.manual_seed(11)
    out_c = rs.wrap_renderer(fake, lambda q: torch.full_like(q, 0.25), {"points": p, "dirs": d}, near=0.4, far=1.4)
    acc = out_c["sg_specular_rgb"][:, :1]
    assert rel_err(out_c["sg_rgb"].cpu(), (acc * 0.25 + (1 - acc)).expand(-1, 3).cpu()) <= 1e-5


def test_encodings(dev, g):
    from robir_amd import embedder as em
    x, v = _t(g, dev, "es_x", "es_var")
    y, yv = em.expected_sin(x, v)
    assert rel_err(y.cpu(), g["es_y"]) <= 2e-5 and rel_err(yv.cpu(), g["es_yvar"]) <= 2e-5          # arguments up to 1e3: sinf vs the host libm
    px, cd, cf = _t(g, dev, "ipe_x", "ipe_cov_diag", "ipe_cov_full")
    assert rel_err(em.integrated_pos_enc((px, cd), 0, 6, diag=True).cpu(), g["ipe_diag"]) <= 1e-5
    assert rel_err(em.integrated_pos_enc((px, cf), 2, 9, diag=False).cpu(), g["ipe_full"]) <= 1e-5
    ipe = em.IPE(max_deg=10)
    assert ipe.feature_dim() == 60
    assert rel_err(ipe(px, em.isotropic_cov(px, 1e-5)).cpu(), g["ipe_module"]) <= 1e-5
    fn, dim = em.ipe_embedder(10)
    assert dim == 60 and rel_err(fn(px).cpu(), g["ipe_embedder"]) <= 1e-5
    x1 = torch.from_numpy(g["pe_x1"]).to(dev)
    for tag, kw, xin in (("3_4", dict(input_dims=3, num_freq=4), px), ("1_10", dict(input_dims=1, num_freq=10), x1),
                         ("3_10", dict(input_dims=3, num_freq=10), px), ("noinp", dict(input_dims=3, num_freq=5, include_input=False), px),
                         ("lin", dict(input_dims=3, num_freq=6, log_sampling=False), px)):
        pe = em.PE(**kw)
        out = pe(xin)
        assert pe.feature_dim() == out.shape[1] == g["pe_" + tag].shape[1]
        e = rel_err(out.cpu(), g["pe_" + tag])
        assert e <= 1e-5, (tag, e)
        assert torch.equal(pe.windowed_embed(xin), out)                                     # no schedule: the plain code (:186-189)
    f4, d4 = em.get_embedder(4)
    assert d4 == 27 and rel_err(f4(px).cpu(), g["emb_get4"]) <= 1e-5
    f5, d5 = em.get_embedder_neus(5, input_dims=3)
    assert d5 == 33 and rel_err(f5(px).cpu(), g["nm_get5"]) <= 1e-5
    assert rel_err(em.PE.cosine_easing_window(0, 9, 10, torch.tensor(3.3)), g["pe_window"]) <= 1e-6
    e = em.Embedder(include_input=True, input_dims=3, max_freq_log2=3, num_freqs=4, log_sampling=True, periodic_fns=[torch.sin, torch.cos])
    assert e.out_dim == 27 and torch.equal(e.embed(px), f4(px))
    assert tuple(f4(px.reshape(1, -1, 3)).shape) == (1, px.shape[0], 27)                    # leading shape kept


def test_sparse_ae_encode(dev, g, synth_weights):
    from robir_amd import renderer
    model = renderer.build_synthetic_model(dev, seed=0, variance=0.3, build_octrees=False)
    ae = model.envmap_material_network.spec_brdf_encoder_layer
    vals, var = _t(g, dev, "ae_values", "ae_var")
    ae.var = var
    try:
        assert rel_err(ae.encode(vals).cpu(), g["ae_encode"]) <= TOL
    finally:
        ae.var = torch.zeros(32)


def _bumpy_sdf(x):
    r = torch.sqrt(x[:, 0] * x[:, 0] + x[:, 1] * x[:, 1] + x[:, 2] * x[:, 2])
    return (r - 0.5) + 0.02 * (x[:, 0] * 7.0 - x[:, 1] * 5.0 + x[:, 2] * 3.0 - 0.3).abs()


def test_raytracing_stages(dev, g):
    """RayTracing.sphere_tracing / ray_sampler / secant / minimal_sdf_points with the reference's signatures (model/ray_tracing.py:102-326)
    on an analytic SDF of element-wise torch ops."""
    from robir_amd.ray_tracing import RayTracing
    rt = RayTracing(object_bounding_sphere=1.0, sdf_threshold=5.0e-5, line_search_step=0.5, line_step_iters=3, sphere_tracing_iters=10,
                    n_steps=100, n_rootfind_steps=32).eval()
    cam, dirs, mi, si = _t(g, dev, "rt_cam", "rt_dirs", "rt_mask_intersect", "rt_sphere_intersections")
    N = dirs.shape[1]
    st = rt.sphere_tracing(1, N, _bumpy_sdf, cam, dirs, mi, si)
    assert len(st) == 6
    conv = ~torch.from_numpy(g["rt_st_unfinished"])
    assert bool((st[1].cpu() == g["rt_st_unfinished"]).all())
    for k, v in zip(("pts", "unfinished", "acc_start", "acc_end", "min_dis", "max_dis"), st):
        if k == "unfinished":
            continue
        e = rel_err(v.cpu()[conv], torch.from_numpy(g["rt_st_" + k])[conv])
        assert e <= 1e-5, (k, e)
    # the same entry / exit distances from the fused init kernel (forward's path) as from the caller's own intersections
    st0 = rt._sphere_tracing(_bumpy_sdf, cam, dirs.reshape(-1, 3).contiguous(), None, None)
    assert rel_err(st0[2].cpu(), st[2].cpu()) <= 1e-6 and torch.equal(st0[1], st[1])
    smask, mm = _t(g, dev, "rt_sampler_mask", "rt_sampler_min_max")
    obj = torch.ones(N, dtype=torch.bool, device=dev)
    sp, sh, sd = rt.ray_sampler(_bumpy_sdf, cam, obj, dirs, mm, smask)
    assert bool((sh.cpu() == g["rt_rs_hit"]).all())
    assert rel_err(sd.cpu(), g["rt_rs_dist"]) <= 1e-5 and rel_err(sp.cpu(), g["rt_rs_pts"]) <= 1e-5
    zl, zh, sl, sh2 = _t(g, dev, "rt_sec_zl", "rt_sec_zh", "rt_sec_sl", "rt_sec_sh")
    k = zl.shape[0]
    d_k = dirs[0, :k].contiguous()
    zp = rt.secant(sl, sh2, zl, zh, cam.expand(k, 3).contiguous(), d_k, _bumpy_sdf)
    assert rel_err(zp.cpu(), g["rt_sec_zpred"]) <= 1e-5
    assert rel_err(zl.cpu(), g["rt_sec_zl_after"]) <= 1e-5 and rel_err(zh.cpu(), g["rt_sec_zh_after"]) <= 1e-5      # in-place bracket, like the reference
    rt.min_sdf_steps = torch.from_numpy(g["rt_min_steps"])
    mask, mn, mx = _t(g, dev, "rt_min_mask", "rt_st_min_dis", "rt_st_max_dis")
    mp, md = rt.minimal_sdf_points(N, _bumpy_sdf, cam, dirs.reshape(-1, 3), mask, mn, mx)
    assert rel_err(md.cpu(), g["rt_min_dist"]) <= 1e-5 and rel_err(mp.cpu(), g["rt_min_pts"]) <= 1e-5


def test_idr_network_helpers(dev, g, synth_weights):
    from robir_amd import renderer
    from robir_amd.octree_tracing import OctreeVisModel
    model = renderer.build_synthetic_model(dev, seed=0, variance=0.3, build_octrees=False)
    nrm, th, ph = _t(g, dev, "sd_normals", "sd_theta", "sd_phi")
    assert rel_err(model.sample_dirs(nrm, th, ph).cpu(), g["sd_out"]) <= 1e-5
    with pytest.raises(NotImplementedError):
        model.sample_dirs(nrm[:3], th[:3], ph[:3])           # num_cam == 3: the reference's dim-less torch.cross changes meaning
    bp, bv = _t(g, dev, "bi_points", "bi_view")
    assert rel_err(model.batch_idr_forward(bp, bv, n_pixels=16).cpu(), g["bi_out"]) <= TOL
    assert model.batch_idr_forward(bp[:0], bv[:0]).shape == (0, 3)
    ovm = OctreeVisModel.__new__(OctreeVisModel)
    ip, iv = _t(g, dev, "is_points", "is_dirs")
    assert rel_err(OctreeVisModel.intersect_sphere(ovm, ip, iv).cpu(), g["is_out"]) <= 1e-5
    assert rel_err(OctreeVisModel.intersect_sphere(ovm, ip, iv, radius=0.7).cpu(), g["is_out_r07"]) <= 1e-5      # NaN rows (outside) compare equal


def test_generate_checks_sdf_fn(dev, synth_weights):
    """OctreeTracing.generate(sdf_fn): the runner's own `lambda x: implicit_network(x)[:, 0]` passes; a different field raises instead of
    being silently ignored (model/octree_tracing.py:31-41)."""
    from robir_amd import renderer
    model = renderer.build_synthetic_model(dev, seed=0, variance=0.3, build_octrees=False)
    tr = model.octree_ray_tracer
    calls = []
    import robir_amd.octree_tracing as ot
    orig = ot.OctreeSDF.build
    ot.OctreeSDF.build = classmethod(lambda cls, *a, **k: calls.append(1) or "tree")
    try:
        tr.generate(lambda x: model.implicit_network(x)[:, 0])
        assert calls == [1] and tr.sdf_octree == "tree"
        with pytest.raises(ValueError, match="differs from the bound implicit network"):
            tr.generate(lambda x: model.implicit_network(x)[:, 0] + 0.01)
        with pytest.raises(ValueError, match="probe points"):
            tr.generate(lambda x: model.implicit_network(x)[:7, 0])
        tr.generate(None)
        assert calls == [1, 1]
    finally:
        ot.OctreeSDF.build = orig
        tr.sdf_octree = None
