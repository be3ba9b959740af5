"""SG shading + visibility sampling on the GPU against the reference's own outputs (golden) and the oracle."""
import numpy as np
import pytest
import torch

from conftest import record_metric, rel_err, load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def vis_net(dev, synth_weights):
    from robir_amd import nets
    v = nets.VisNetwork(10, 10, [256] * 4)
    v.load_state_dict({k[len("visibility_network."):]: torch.from_numpy(a) for k, a in synth_weights.items()
                       if k.startswith("visibility_network.")})
    return v.to(dev).eval()


@pytest.fixture(params=["fp32", "f16x6", "f16x3", "f16x3-v2", "f16x3-v3"])
def precision(request):
    from robir_amd import sg_render
    old = sg_render.VIS_PRECISION
    sg_render.VIS_PRECISION = request.param
    yield request.param
    sg_render.VIS_PRECISION = old


@pytest.mark.parametrize("tag", ["init", "sharp"])
def test_render_with_all_sg_golden(dev, vis_net, tag, precision):
    from robir_amd import sg_render
    g = load_golden("sg_" + tag)
    t = {k: torch.from_numpy(v).to(dev) for k, v in g.items() if v.dtype.kind == "f"}
    draws = {k[5:]: t[k] for k in t if k.startswith("draw_")}
    stats = {}
    out = sg_render.render_with_all_sg(t["points"], t["normal"], t["view"], t["lgtSGs"], t["f0"], t["roughness"],
                                       t["albedo"], indir_integral=t["indir_int"], indir_lgtSGs=t["indir_sgs"],
                                       VisModel=vis_net, testing=True, draws=draws, stats=stats)
    torch.cuda.synchronize()
    assert int(stats["diffuse_vis_evals"]) > 0
    for k in ("vis_shadow", "sg_diffuse_rgb", "sg_specular_rgb", "sg_rgb", "indir_diffuse_rgb", "indir_specular_rgb",
              "indir_rgb"):
        assert rel_err(out[k].cpu(), g["out_" + k]) <= TOL, (k, rel_err(out[k].cpu(), g["out_" + k]))


def test_render_with_all_sg_multi_view_golden(dev, vis_net):
    """MULTI_VIEW shading (viewdirs [V,n,3]: model/sg_render.py:356, 375-378, 465-470; get_specular_visibility's multi_view branches)
    against the reference's outputs for two views of the sg_init points: view-independent fields [n,3], specular and totals [V,n,3]; the
    function form (fun_spec) and get_specular_visibility(multi_view=True) give the same numbers."""
    from robir_amd import sg_render
    g, mv = load_golden("sg_init"), load_golden("sg_multi_view")
    t = {k: torch.from_numpy(v).to(dev) for k, v in g.items() if v.dtype.kind == "f"}
    view = torch.from_numpy(mv["view"]).to(dev)
    draws = {k[5:]: torch.from_numpy(mv[k]).to(dev) for k in mv if k.startswith("draw_")}
    args = (t["points"], t["normal"], view, t["lgtSGs"], t["f0"], t["roughness"], t["albedo"])
    kw = dict(indir_integral=t["indir_int"], indir_lgtSGs=t["indir_sgs"], VisModel=vis_net, testing=True, draws=draws)
    out = sg_render.render_with_all_sg(*args, **kw)
    torch.cuda.synchronize()
    for k in ("vis_shadow", "sg_diffuse_rgb", "sg_specular_rgb", "sg_rgb", "indir_diffuse_rgb", "indir_specular_rgb", "indir_rgb"):
        assert tuple(out[k].shape) == mv["out_" + k].shape, k
        assert rel_err(out[k].cpu(), mv["out_" + k]) <= TOL, (k, rel_err(out[k].cpu(), mv["out_" + k]))
    fn = sg_render.render_with_all_sg(*args, fun_spec=True, **kw)
    assert torch.equal(fn["sg_rgb"], out["sg_diffuse_rgb"])
    spec = fn["sg_specular_rgb"](t["roughness"], {"svis_theta": draws["svis_theta_dir"], "svis_phi": draws["svis_phi_dir"]})
    assert torch.equal(spec, out["sg_specular_rgb"])
    bvis = sg_render.get_specular_visibility(t["points"], t["normal"], view, vis_net, None, None, nsamp=16, multi_view=True, testing=True,
                                             roughness=t["roughness"], draws=(draws["svis_theta_dir"], draws["svis_phi_dir"]))
    assert bvis.shape == (2, 40) and bool(((bvis >= 0) & (bvis <= 1)).all())
    with pytest.raises(ValueError):
        sg_render.render_with_sg(*args, VisModel=vis_net, chunk_id=torch.zeros(40, dtype=torch.int32, device=dev), n_chunks=2)


def test_diffuse_visibility_vs_oracle(dev, vis_net, oracle_sd, precision):
    """Fused kernel against the oracle on more points than one block row, two chunks with different draws."""
    from robir_amd import sg_render, synth
    from robir_oracle import nets as on, sg as osg
    g = np.random.Generator(np.random.PCG64(5))
    n = 37
    pts = torch.from_numpy((g.standard_normal((n, 3)) * 0.25).astype(np.float32))
    nrm = torch.from_numpy(g.standard_normal((n, 3)).astype(np.float32))
    nrm = nrm / nrm.norm(dim=-1, keepdim=True)
    lgt = torch.from_numpy(synth.synth_light_sgs(3, 128, sharp=True))
    u = torch.from_numpy(g.random((2, 2, 128, 32), dtype=np.float32))
    cid = torch.tensor([0] * 20 + [1] * 17, dtype=torch.int32)
    out = sg_render._diffuse_vis_core(pts.to(dev), nrm.to(dev), vis_net, lgt.to(dev), u[0].to(dev), u[1].to(dev), 1.0,
                                      False, cid.to(dev), 2, None).cpu()
    lobe = lgt[:, :3] / (lgt[:, :3].norm(dim=-1, keepdim=True) + 1e-6)
    lam = lgt[:, 3:4].abs()
    vis_fn = lambda p, d: on.vis_logits(oracle_sd, p, d)
    for c, sl in ((0, slice(0, 20)), (1, slice(20, 37))):
        ref = osg.diffuse_visibility(pts[sl], nrm[sl], vis_fn, lobe, lam, u[0, c], u[1, c]).t()
        assert rel_err(out[sl], ref) <= TOL, (c, rel_err(out[sl], ref))
        print(f"[{precision}] chunk {c}: max rel err vs oracle {rel_err(out[sl], ref):.3e}")


@pytest.mark.parametrize("vis_mode", ["f16x6", "f16x3-auto", "fp32"])
def test_diffuse_visibility_light_that_does_not_fill_whole_tiles(dev, vis_net, oracle_sd, vis_mode, monkeypatch):
    """A light with L * nsamp NOT a multiple of 16 (7 lobes x 8 samples = 56; the CESR diffuse_vis path and any direct
    get_diffuse_visibility(nsamp=8) with an odd lobe count) on a chunk-sized batch: the policies' auto rules must take the per-point
    kernels -- the tile-list forms cut a point's directions into whole 16-sample tiles and refuse this shape -- and agree with
    the oracle; the f16 throughput kernel, which exists in the tile-list form only, says so in Python instead of failing in the launcher."""
    from robir_amd import sg_render, synth
    from robir_oracle import nets as on, sg as osg
    monkeypatch.setattr(sg_render, "VIS_PRECISION", vis_mode)
    g = np.random.Generator(np.random.PCG64(35))
    n, L, nsamp = 41, 7, 8
    pts = torch.from_numpy((g.standard_normal((n, 3)) * 0.25).astype(np.float32))
    nrm = torch.nn.functional.normalize(torch.from_numpy(g.standard_normal((n, 3)).astype(np.float32)), dim=-1)
    lgt = torch.from_numpy(synth.synth_light_sgs(3, 128, sharp=True))[:L]
    lobe, lam = lgt[:, :3], lgt[:, 3:4].abs()
    u = torch.from_numpy(g.random((2, L, nsamp), dtype=np.float32))
    draws = {"dvis_theta": u[0][None].to(dev), "dvis_phi": u[1][None].to(dev)}
    out = sg_render.get_diffuse_visibility(pts.to(dev), nrm.to(dev), vis_net, lobe.to(dev), lam.to(dev), nsamp=nsamp, draws=draws).cpu()
    ref = osg.diffuse_visibility(pts, nrm, lambda p, d: on.vis_logits(oracle_sd, p, d), torch.nn.functional.normalize(lobe, dim=-1), lam, u[0], u[1])
    assert out.shape == ref.shape == (L, n)
    assert rel_err(out, ref) <= TOL, rel_err(out, ref)
    if vis_mode == "f16x6":
        monkeypatch.setattr(sg_render, "VIS_PRECISION", "f16x1")
        with pytest.raises(ValueError, match="multiple of 16"):
            sg_render.get_diffuse_visibility(pts.to(dev), nrm.to(dev), vis_net, lobe.to(dev), lam.to(dev), nsamp=nsamp, draws=draws)


@pytest.mark.parametrize("argmax_vis", [False, True])
def test_diffuse_visibility_bounding_vs_oracle(dev, vis_net, oracle_sd, argmax_vis):
    """bounding=True (sg_render.py:185-186): the per-sample visibilities [L, nsamp, n] before the lobe-weighted mean, culled pairs 0;
    and their weighted mean is what the fused kernel returns."""
    from robir_amd import sg_render, synth
    from robir_oracle import nets as on, sg as osg
    g = np.random.Generator(np.random.PCG64(25))
    n, L, nsamp = 29, 128, 8
    pts = torch.from_numpy((g.standard_normal((n, 3)) * 0.25).astype(np.float32))
    nrm = torch.nn.functional.normalize(torch.from_numpy(g.standard_normal((n, 3)).astype(np.float32)), dim=-1)
    lgt = torch.from_numpy(synth.synth_light_sgs(3, L, sharp=True))
    lobe, lam = lgt[:, :3], lgt[:, 3:4].abs()
    u = torch.from_numpy(g.random((2, L, nsamp), dtype=np.float32))
    draws = {"dvis_theta": u[0][None].to(dev), "dvis_phi": u[1][None].to(dev)}
    out = sg_render.get_diffuse_visibility(pts.to(dev), nrm.to(dev), vis_net, lobe.to(dev), lam.to(dev), nsamp=nsamp, bounding=True,
                                           argmax_vis=argmax_vis, draws=draws).cpu()
    vis_fn = lambda p, d: on.vis_logits(oracle_sd, p, d)
    ref = osg.diffuse_visibility(pts, nrm, vis_fn, lobe, lam, u[0], u[1], bounding=True, argmax_vis=argmax_vis)
    assert out.shape == ref.shape == (L, nsamp, n)
    if argmax_vis:      # 0 / 1 decisions: a logit pair closer than the arithmetic's error may flip
        assert float((out != ref).float().mean()) <= 2e-3
    else:
        assert float((out - ref).abs().max()) <= 2e-5, float((out - ref).abs().max())
        mean = sg_render.get_diffuse_visibility(pts.to(dev), nrm.to(dev), vis_net, lobe.to(dev), lam.to(dev), nsamp=nsamp,
                                                draws=draws).cpu()
        refm = osg.diffuse_visibility(pts, nrm, vis_fn, lobe, lam, u[0], u[1])
        assert rel_err(mean, refm) <= TOL


@pytest.mark.parametrize("comp_vis", [True, False])
def test_render_with_sg_fun_spec_vs_oracle(dev, vis_net, oracle_sd, comp_vis):
    """fun_spec=True (sg_render.py:413,544-551): the specular term as a function of a roughness tensor, sg_rgb = the diffuse term.
    The closure at another roughness against the oracle's on the same draws; at the call's own roughness and draws it is the
    specular term of the plain call."""
    from robir_amd import sg_render, synth
    from robir_oracle import nets as on, sg as osg
    g = np.random.Generator(np.random.PCG64(35))
    n, M = 41, 128 if comp_vis else 24
    pts = torch.from_numpy((g.standard_normal((n, 3)) * 0.25).astype(np.float32))
    nrm = torch.nn.functional.normalize(torch.from_numpy(g.standard_normal((n, 3)).astype(np.float32)), dim=-1)
    view = torch.nn.functional.normalize(nrm + 0.7 * torch.from_numpy(g.standard_normal((n, 3)).astype(np.float32)), dim=-1)
    # roughness >= 0.4: below it the reference's fp32 formulas are ill-conditioned against these sharp lights (lambda up to 500 x a
    # BRDF lobe of 2 / r^4) -- at r = 0.2 the HIP kernel and the fp32 oracle are each 5e-3 from a float64 evaluation and 1.4e-3 apart
    rough = torch.from_numpy(np.exp(g.uniform(np.log(0.4), np.log(0.9), (n, 1))).astype(np.float32))
    rough2 = torch.from_numpy(np.exp(g.uniform(np.log(0.4), np.log(0.9), (n, 1))).astype(np.float32))
    albedo = torch.from_numpy(g.uniform(0.1, 0.9, (n, 3)).astype(np.float32))
    f0 = torch.full((1, 1), 0.02)
    if comp_vis:
        lgt = torch.from_numpy(synth.synth_light_sgs(3, M, sharp=True))
        lgt_o = lgt[None].expand(n, M, 7)
        indir = None
    else:       # the indirect pass: per-point lobes, the integral replaces the diffuse term
        lgt = torch.from_numpy(np.stack([synth.synth_light_sgs(100 + i, M, sharp=False) for i in range(n)]))
        lgt_o = lgt
        indir = torch.from_numpy(g.uniform(0.0, 1.0, (n, 3)).astype(np.float32))
    u = torch.from_numpy(g.random((2, M, 32), dtype=np.float32))
    sv = torch.from_numpy(g.random((4, n, 8), dtype=np.float32))
    d0 = {"svis_theta": sv[0], "svis_phi": sv[1]}
    d1 = {"svis_theta": sv[2], "svis_phi": sv[3]}
    if comp_vis:
        d0.update(dvis_theta=u[0], dvis_phi=u[1])
    todev = lambda d: {k: v.to(dev) for k, v in d.items()}
    kw = dict(comp_vis=comp_vis, VisModel=vis_net, indir_integral=None if indir is None else indir.to(dev), testing=True)
    args = (pts.to(dev), nrm.to(dev), view.to(dev), lgt.to(dev), f0.to(dev), rough.to(dev), albedo.to(dev))
    out = sg_render.render_with_sg(*args, fun_spec=True, draws=todev(d0), **kw)
    plain = sg_render.render_with_sg(*args, draws=todev(d0), **kw)
    assert callable(out["sg_specular_rgb"])
    assert torch.equal(out["sg_rgb"], out["sg_diffuse_rgb"]) and torch.equal(out["sg_diffuse_rgb"], plain["sg_diffuse_rgb"])
    assert torch.equal(out["sg_specular_rgb"](rough.to(dev), draws=todev(d0)), plain["sg_specular_rgb"])
    vis_fn = lambda p, d: on.vis_logits(oracle_sd, p, d)
    ref = osg.render_with_sg(pts, nrm, view, lgt_o, f0, rough, albedo, d0, comp_vis=comp_vis, vis_fn=vis_fn,
                             indir_integral=indir, testing=True, fun_spec=True)
    assert rel_err(out["sg_rgb"].cpu(), ref["sg_rgb"]) <= TOL
    got = out["sg_specular_rgb"](rough2.to(dev), draws=todev(d1)).cpu()
    want = ref["sg_specular_rgb"](rough2, d1)
    # the yardstick for the closure: the oracle's formulas in float64 (the MLP stays fp32).  The specular term is a sum over lobes of
    # differences of two hemisphere integrals: where it is small (0.011 at the worst point of the indirect case) two fp32 evaluations of
    # the reference's formulas differ by a few 1e-4 of it, whatever their order of summation (kernel: wave-shuffle tree, 2.2e-4 from
    # float64 there; torch: 4e-5 on that point, 2.5e-4 on others of the direct case) -- hence 3e-4 against float64 next to TOL
    vis64 = lambda p, d: on.vis_logits(oracle_sd, p.float(), d.float()).double()
    r64 = osg.render_with_sg(pts.double(), nrm.double(), view.double(), lgt_o.double(), f0.double(), rough.double(), albedo.double(),
                             {k: v.double() for k, v in d0.items()}, comp_vis=comp_vis, vis_fn=vis64,
                             indir_integral=None if indir is None else indir.double(), testing=True, fun_spec=True)
    t64 = r64["sg_specular_rgb"](rough2.double(), {k: v.double() for k, v in d1.items()})
    e_hip, e_f32, e_pair = rel_err(got.double(), t64), rel_err(want.double(), t64), rel_err(got, want)
    print(f"fun_spec closure (comp_vis={comp_vis}): vs fp32 oracle {e_pair:.3e}; vs float64 {e_hip:.3e} (fp32 oracle: {e_f32:.3e})")
    assert e_pair <= TOL or e_hip <= max(1.25 * e_f32, 3e-4), (e_pair, e_hip, e_f32)
    if comp_vis:
        assert rel_err(out["vis_shadow"].cpu(), ref["vis_shadow"]) <= TOL


@pytest.mark.parametrize("testing,inv,argmax_vis", [(False, False, False), (True, True, False), (False, False, True)])
def test_specular_visibility_vs_oracle(dev, vis_net, oracle_sd, testing, inv, argmax_vis):
    """get_specular_visibility on its own (sg_render.py:198-301; inside render_with_sg it only shows through sg_specular_rgb):
    cone samples around the reflection direction, the front-facing cull, the lobe weights and the batch-global sharpness
    minimum, against the oracle on the same draws -- rough and mirror-like points in one batch."""
    from robir_amd import sg_render
    from robir_oracle import nets as on, sg as osg
    g = np.random.Generator(np.random.PCG64(15))
    n, nsamp = 53, 24
    pts = torch.from_numpy((g.standard_normal((n, 3)) * 0.25).astype(np.float32))
    nrm = torch.nn.functional.normalize(torch.from_numpy(g.standard_normal((n, 3)).astype(np.float32)), dim=-1)
    view = torch.nn.functional.normalize(nrm + 0.8 * torch.from_numpy(g.standard_normal((n, 3)).astype(np.float32)), dim=-1)
    rough = torch.from_numpy(np.exp(g.uniform(np.log(0.08), np.log(0.9), (n, 1))).astype(np.float32))
    u = torch.from_numpy(g.random((2, n, nsamp), dtype=np.float32))
    out = sg_render.get_specular_visibility(pts.to(dev), nrm.to(dev), view.to(dev), vis_net, None, None, nsamp=nsamp,
                                            testing=testing, inv=inv, argmax_vis=argmax_vis, roughness=rough.to(dev),
                                            draws=(u[0].to(dev), u[1].to(dev))).cpu()
    # the warped BRDF lobe the reference passes in (sg_render.py:414-437)
    vdl = (nrm * view).sum(-1, keepdim=True).clamp(min=0.0)
    w_lobe = 2 * vdl * nrm - view
    w_lobe = w_lobe / (w_lobe.norm(dim=-1, keepdim=True) + 1e-6)
    w_lam = (2.0 / rough ** 4) / (4 * vdl + 1e-6)
    ref = osg.specular_visibility(pts, nrm, view, lambda p, d: on.vis_logits(oracle_sd, p, d), w_lobe, w_lam, u[0], u[1],
                                  testing=testing, inv=inv, argmax_vis=argmax_vis)
    assert out.shape == ref.shape == (n,)
    assert float(ref.std()) > 0.02                       # a batch where the samples matter
    if argmax_vis:                                       # hard decisions: a logit pair within rounding of a tie may flip a sample
        assert float(((out - ref).abs() > 1e-4).float().mean()) <= 0.05
    else:
        assert rel_err(out, ref) <= TOL, rel_err(out, ref)


@pytest.mark.parametrize("tag,testing,inv,argmax_vis", [("plain", False, False, False), ("testing_inv", True, True, False), ("argmax", False, False, True)])
def test_specular_visibility_reference_signature_golden(dev, vis_net, oracle_sd, tag, testing, inv, argmax_vis):
    """get_specular_visibility called EXACTLY as the reference declares it (model/sg_render.py:196-197), all eleven arguments positional,
    on lobes / lambdas that are not the warped BRDF lobe of the points (un-normalised vectors, lambdas on both sides of the 0.1 .. 50
    clip): the cone opens by the PASSED lambdas (batch-global minimum, :219-223), the samples are weighted by the PASSED lobes (:281).
    Against the reference's own outputs (tests/golden/spec_vis_refsig.npz, oracle/gen_golden_r5.py) and the oracle; the two uniform draws
    are replayed by seeding torch's device generator -- the reference draws them with torch.rand at this point (:224-225)."""
    from robir_amd import sg_render
    from robir_oracle import nets as on, sg as osg
    g = load_golden("spec_vis_refsig")
    T = lambda k: torch.from_numpy(g[k]).to(dev)
    nsamp = int(g["nsamp"])
    out = sg_render.get_specular_visibility(T("points"), T("normals"), T("view"), vis_net, T("lobes"), T("lambdas"), nsamp, False, testing,
                                            inv, argmax_vis, draws=(T("u_theta"), T("u_phi"))).cpu()
    ref = torch.from_numpy(g["out_" + tag])
    assert out.shape == ref.shape
    mine = osg.specular_visibility(*(torch.from_numpy(g[k]) for k in ("points", "normals", "view")), lambda p, d: on.vis_logits(oracle_sd, p, d),
                                   torch.from_numpy(g["lobes"]), torch.from_numpy(g["lambdas"]), torch.from_numpy(g["u_theta"]),
                                   torch.from_numpy(g["u_phi"]), testing=testing, inv=inv, argmax_vis=argmax_vis)
    assert float(ref.std()) > 0.02
    if argmax_vis:
        assert float(((out - ref).abs() > 1e-4).float().mean()) <= 0.05 and float(((mine - ref).abs() > 1e-4).float().mean()) <= 0.05
    else:
        assert rel_err(mine, ref) <= 1e-5, rel_err(mine, ref)
        assert rel_err(out, ref) <= TOL, rel_err(out, ref)
    # the call with no keyword at all draws from torch's generator like the reference: same seed, same result, and a value per point
    torch.manual_seed(11)
    a = sg_render.get_specular_visibility(T("points"), T("normals"), T("view"), vis_net, T("lobes"), T("lambdas"), nsamp, False, testing, inv, argmax_vis)
    torch.manual_seed(11)
    b = sg_render.get_specular_visibility(T("points"), T("normals"), T("view"), vis_net, T("lobes"), T("lambdas"), nsamp, False, testing, inv, argmax_vis)
    assert torch.equal(a, b) and a.shape == ref.shape and bool(torch.isfinite(a).all())
    # and the warped-lobe fast path (roughness=) equals the reference-signature path fed with the warped lobe it recomputes
    rough = torch.from_numpy(np.exp(np.random.default_rng(2).uniform(np.log(0.08), np.log(0.9), (ref.shape[0], 1))).astype(np.float32))
    nrm, view = torch.from_numpy(g["normals"]), torch.from_numpy(g["view"])
    vdl = (nrm * view).sum(-1, keepdim=True).clamp(min=0.0)
    w_lobe = 2 * vdl * nrm - view
    w_lobe = w_lobe / (w_lobe.norm(dim=-1, keepdim=True) + 1e-6)
    w_lam = (2.0 / rough ** 4) / (4 * vdl + 1e-6)
    fast = sg_render.get_specular_visibility(T("points"), T("normals"), T("view"), vis_net, None, None, nsamp, testing=testing, inv=inv,
                                             argmax_vis=argmax_vis, roughness=rough.to(dev), draws=(T("u_theta"), T("u_phi")))
    slow = sg_render.get_specular_visibility(T("points"), T("normals"), T("view"), vis_net, w_lobe.to(dev), w_lam.to(dev), nsamp, False,
                                             testing, inv, argmax_vis, draws=(T("u_theta"), T("u_phi")))
    assert float((fast - slow).abs().max()) <= (0.5 if argmax_vis else 2e-5), float((fast - slow).abs().max())


def test_generic_vismodel_callable(dev, vis_net):
    """A VisModel that is not our VisNetwork goes through the generic (callable) path and must agree."""
    from robir_amd import sg_render, synth
    g = np.random.Generator(np.random.PCG64(6))
    n = 9
    pts = torch.from_numpy((g.standard_normal((n, 3)) * 0.25).astype(np.float32)).to(dev)
    nrm = torch.from_numpy(g.standard_normal((n, 3)).astype(np.float32)).to(dev)
    nrm = nrm / nrm.norm(dim=-1, keepdim=True)
    lgt = torch.from_numpy(synth.synth_light_sgs(3, 128)).to(dev)
    u = torch.from_numpy(g.random((2, 128, 32), dtype=np.float32)).to(dev)
    a = sg_render._diffuse_vis_core(pts, nrm, vis_net, lgt, u[0], u[1], 1.0, False, None, 1, None)
    b = sg_render._diffuse_vis_core(pts, nrm, lambda p, d: vis_net(p, d), lgt, u[0], u[1], 1.0, False, None, 1, None)
    assert rel_err(a.cpu(), b.cpu()) <= 1e-5


def test_fused_visibility_is_deterministic(dev, vis_net):
    """Race screen for the pipelined LDS rings (DMA staging, raw barriers, counted vmcnt): repeated launches on a busy
    chip must be bit-identical in every kernel generation, and the exact-operand kernel must sit on the fp32 MFMA kernel."""
    from robir_amd import sg_render, synth
    g = np.random.Generator(np.random.PCG64(8))
    n = 3000                                       # > 2 workgroups per CU on every CU
    pts = torch.from_numpy((g.standard_normal((n, 3)) * 0.25).astype(np.float32)).to(dev)
    nrm = torch.from_numpy(g.standard_normal((n, 3)).astype(np.float32)).to(dev)
    nrm = nrm / nrm.norm(dim=-1, keepdim=True)
    lgt = torch.from_numpy(synth.synth_light_sgs(3, 128)).to(dev)
    u = torch.from_numpy(g.random((2, 3, 128, 32), dtype=np.float32)).to(dev)
    cid = torch.arange(n, dtype=torch.int32, device=dev) % 3
    old = sg_render.VIS_PRECISION
    try:
        outs = {}
        for mode in ("f16x3", "f16x3-v2", "f16x6", "fp32"):
            sg_render.VIS_PRECISION = mode
            runs = [sg_render._diffuse_vis_core(pts, nrm, vis_net, lgt, u[0], u[1], 1.0, False, cid.contiguous(), 3, None)
                    for _ in range(4 if mode != "fp32" else 1)]
            for r in runs[1:]:
                assert torch.equal(r, runs[0]), mode
            outs[mode] = runs[0]
        # exact-operand kernel vs the f32-input MFMA kernel: identical products, different summation order only
        d6 = float((outs["f16x6"] - outs["fp32"]).abs().max())
        d3 = float((outs["f16x3-v2"] - outs["fp32"]).abs().max())
        print(f"max |vis - vis_fp32|: f16x6 {d6:.2e}, f16x3-v2 {d3:.2e}")
        assert d6 <= 2e-6, d6
    finally:
        sg_render.VIS_PRECISION = old


def test_f16_throughput_mode_error_is_measured(dev, vis_net):
    """ROBIR_PRECISION=f16 (BASELINE.json configs[4]: "fp16 MLP weights on MFMA"): the light-visibility MLP with ONE f16 product per
    multiply-add, f16 weights and f16 activations (csrc/vis_diffuse_f16t.hip).  NARROWER than fp32 by construction -- this test does
    not claim parity: it MEASURES the per-lobe visibility against the f32-input-MFMA kernel on the same points, directions and draws
    (median / 99th percentile / maximum absolute difference of values in [0, 1], recorded in gpurun_out/test_metrics.jsonl; DESIGN.md
    quotes them) and holds the mode to a sanity band and to run-to-run bit-identity."""
    from robir_amd import sg_render, synth
    g = np.random.Generator(np.random.PCG64(8))
    n = 3000
    pts = torch.from_numpy((g.standard_normal((n, 3)) * 0.25).astype(np.float32)).to(dev)
    nrm = torch.from_numpy(g.standard_normal((n, 3)).astype(np.float32)).to(dev)
    nrm = nrm / nrm.norm(dim=-1, keepdim=True)
    lgt = torch.from_numpy(synth.synth_light_sgs(3, 128)).to(dev)
    u = torch.from_numpy(g.random((2, 3, 128, 32), dtype=np.float32)).to(dev)
    cid = (torch.arange(n, dtype=torch.int32, device=dev) % 3).contiguous()
    old = sg_render.VIS_PRECISION
    try:
        outs = {}
        for mode in ("f16x1", "f16x6", "fp32"):
            sg_render.VIS_PRECISION = mode
            outs[mode] = sg_render._diffuse_vis_core(pts, nrm, vis_net, lgt, u[0], u[1], 1.0, False, cid, 3, None)
        sg_render.VIS_PRECISION = "f16x1"
        again = sg_render._diffuse_vis_core(pts, nrm, vis_net, lgt, u[0], u[1], 1.0, False, cid, 3, None)
        assert torch.equal(again, outs["f16x1"])
        # the kernel generations -- 1: round 4 (the exact-operand blob's h pieces in place), 2: two-chunk steps over the h-only blob,
        # 3: the point-block form (sixteen points x one direction per tile) -- multiply the same halves in the same order per pair: the
        # same bits.  The interleaved chunk ids above are NOT ascending: generation 3 then takes generation 2's path (ops.dvis_fused)
        from robir_amd import ops
        gen = ops.DVIS_F16_GEN
        try:
            for other_gen in (1, 2, 3):
                ops.DVIS_F16_GEN = other_gen
                other = sg_render._diffuse_vis_core(pts, nrm, vis_net, lgt, u[0], u[1], 1.0, False, cid, 3, None)
                assert torch.equal(other, outs["f16x1"]), (other_gen, float((other - outs["f16x1"]).abs().max()))
        finally:
            ops.DVIS_F16_GEN = gen
        for am in (False, True):       # the argmax form (testing=True paths): a flipped sample moves a lobe by 1 / 32
            sg_render.VIS_PRECISION = "f16x1"
            a = sg_render._diffuse_vis_core(pts, nrm, vis_net, lgt, u[0], u[1], 1.0, am, cid, 3, None)
            sg_render.VIS_PRECISION = "fp32"
            b = sg_render._diffuse_vis_core(pts, nrm, vis_net, lgt, u[0], u[1], 1.0, am, cid, 3, None)
            d = (a - b).abs().flatten().double()
            q = torch.quantile(d[torch.randperm(d.numel(), device=d.device)[:200000]], torch.tensor([0.5, 0.99], device=d.device, dtype=torch.float64))
            record_metric("f16_mode/light_visibility_" + ("argmax" if am else "softmax"), median=float(q[0]), p99=float(q[1]), max=float(d.max()),
                          exact_operand_max=float((outs["f16x6"] - outs["fp32"]).abs().max()))
            print(f"f16 mode, {'argmax' if am else 'softmax'} visibility vs f32-MFMA: median {float(q[0]):.2e}  p99 {float(q[1]):.2e}  max {float(d.max()):.2e}")
            if not am:
                assert float(q[0]) <= 2e-3 and float(q[1]) <= 2e-2 and float(d.max()) <= 0.25, (float(q[0]), float(q[1]), float(d.max()))
    finally:
        sg_render.VIS_PRECISION = old


@pytest.mark.parametrize("n,n_chunks,L,nsamp", [(3000, 3, 128, 32), (1, 1, 128, 32), (17, 2, 128, 32), (1000, 1, 128, 32), (2500, 7, 128, 32),
                                                 (700, 2, 6, 8), (300, 1, 128, 8), (40, 3, 4, 4)])
def test_f16_point_block_form_is_bit_identical(dev, vis_net, n, n_chunks, L, nsamp):
    """ROBIR_PRECISION=f16, csrc/vis_diffuse_f16p.hip: tiles of sixteen consecutive points x one direction (the rows of a round by
    whole-row LDS-DMA copies instead of a 16-line gather per load) against the per-point tile list (rb_dvis_stream_f16): ascending chunk
    ids with blocks that straddle chunk boundaries, a last block with padding lanes, a single point, empty chunks, points with a NaN
    normal (they face nothing: visibility 0), short direction lists (the CESR hook's 128 x 8, 6 x 8 = three rounds, 4 x 4 = one round) -- the same bits, and the same count of evaluated pairs."""
    from robir_amd import ops, sg_render, synth
    g = np.random.Generator(np.random.PCG64(80 + n))
    pts = torch.from_numpy((g.standard_normal((n, 3)) * 0.25).astype(np.float32)).to(dev)
    nrm = torch.from_numpy(g.standard_normal((n, 3)).astype(np.float32)).to(dev)
    nrm = nrm / nrm.norm(dim=-1, keepdim=True)
    if n > 100:
        nrm[5] = float("nan")
        nrm[40:44] = nrm[40]                                   # a run of equal normals: whole tiles kept / dropped together
    lgt = torch.from_numpy(synth.synth_light_sgs(3, L)).to(dev)
    u = torch.from_numpy(g.random((2, n_chunks, L, nsamp), dtype=np.float32)).to(dev)
    if n_chunks == 1:
        cid = None
    else:
        c = np.sort(g.integers(0, n_chunks, n)).astype(np.int32)
        if n_chunks == 7:
            c[c == 3] = 4                                        # an empty chunk in the middle
        cid = torch.from_numpy(c).to(dev)
    old, gen = sg_render.VIS_PRECISION, ops.DVIS_F16_GEN
    try:
        sg_render.VIS_PRECISION = "f16x1"
        res = {}
        for k in (2, 3):
            ops.DVIS_F16_GEN = k
            for am in (False, True):
                st = {}
                res[k, am] = (sg_render._diffuse_vis_core(pts, nrm, vis_net, lgt, u[0], u[1], 1.0, am, cid, n_chunks, st), st)
        for am in (False, True):
            assert torch.equal(res[2, am][0], res[3, am][0]), float((res[2, am][0] - res[3, am][0]).abs().max())
            assert not torch.isnan(res[3, am][0]).any()
            a, b = res[2, am][1], res[3, am][1]
            assert {k: int(v) for k, v in a.items()} == {k: int(v) for k, v in b.items()}, (a, b)
    finally:
        sg_render.VIS_PRECISION, ops.DVIS_F16_GEN = old, gen


def test_f16_point_block_entry_refuses_unsorted_chunk_ids(dev, vis_net):
    """rb_dvis_pblock_f16 called directly with chunk ids that are not ascending (the Python mirror never does: ops.dvis_fused checks and
    takes the per-point form): NaN everywhere instead of numbers for the wrong directions."""
    from robir_amd import ops, sg_render, synth
    g = np.random.Generator(np.random.PCG64(9))
    n = 200
    pts = torch.from_numpy((g.standard_normal((n, 3)) * 0.25).astype(np.float32)).to(dev)
    nrm = torch.from_numpy(g.standard_normal((n, 3)).astype(np.float32)).to(dev)
    lgt = torch.from_numpy(synth.synth_light_sgs(3, 128)).to(dev)
    u = torch.from_numpy(g.random((2, 2, 128, 32), dtype=np.float32)).to(dev)
    cid = (torch.arange(n, dtype=torch.int32, device=dev) % 2).contiguous()
    cid._robir_ascending = True                                  # a caller lying about the order
    old, gen = sg_render.VIS_PRECISION, ops.DVIS_F16_GEN
    try:
        sg_render.VIS_PRECISION, ops.DVIS_F16_GEN = "f16x1", 3
        out = sg_render._diffuse_vis_core(pts, nrm, vis_net, lgt, u[0], u[1], 1.0, False, cid, 2, None)
        assert torch.isnan(out).all()
    finally:
        sg_render.VIS_PRECISION, ops.DVIS_F16_GEN = old, gen


def test_specular_term_conditioning(dev):
    """Sharp light SGs (|lambda| up to ~500, like the shipped envmap fits) with low roughness make the reference's own
    specular formula ill-conditioned in fp32: evaluated in fp64 and fp32 it differs by percent on some rays
    (lambda_trick / hemisphere_int cancel large exponentials).  A 1e-4 bound against an fp32 oracle is meaningless
    there; what can be asked is that the kernel is as close to the fp64 evaluation as the fp32 oracle is."""
    from robir_amd import sg_render, synth
    from robir_oracle import sg as osg
    g = torch.Generator().manual_seed(0)
    n = 600
    lgt = torch.from_numpy(synth.synth_light_sgs(3, 128, sharp=True)).float()
    pts = torch.randn(n, 3, generator=g) * 0.2
    nrm = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    view = torch.nn.functional.normalize(nrm + 0.8 * torch.randn(n, 3, generator=g), dim=-1)
    rough = torch.rand(n, 1, generator=g) * 0.9 + 0.09
    alb = torch.rand(n, 3, generator=g)
    f0 = torch.full((1, 1), 0.05)
    dr = {"dvis_theta": torch.rand(128, 32, generator=g), "dvis_phi": torch.rand(128, 32, generator=g),
          "svis_theta_dir": torch.rand(n, 8, generator=g), "svis_phi_dir": torch.rand(n, 8, generator=g)}

    def oracle(dt):
        c = lambda t: t.to(dt)
        return osg.render_with_all_sg(c(pts), c(nrm), c(view), c(lgt), c(f0), c(rough), c(alb), {k: c(v) for k, v in dr.items()},
                                      vis_fn=lambda p, d: torch.zeros(p.shape[0], 2, dtype=dt), testing=True)

    o32, o64 = oracle(torch.float32), oracle(torch.float64)
    vis = lambda p, d: torch.zeros(p.shape[0], 2, device=p.device)
    d = lambda t: t.to(dev)
    out = sg_render.render_with_all_sg(d(pts), d(nrm), d(view), d(lgt), d(f0), d(rough), d(alb), VisModel=vis, testing=True,
                                       draws={k: d(v) for k, v in dr.items()})
    for k in ("sg_specular_rgb", "sg_rgb"):
        e_oracle = rel_err(o32[k].double(), o64[k])
        e_kernel = rel_err(out[k].cpu().double(), o64[k])
        assert e_oracle > 1e-3, "this test is meant to sit in the ill-conditioned regime"
        assert e_kernel <= 2.0 * e_oracle, (k, e_kernel, e_oracle)
        print(f"{k}: fp32 oracle vs fp64 {e_oracle:.2e}, kernel vs fp64 {e_kernel:.2e}")
    # the well-conditioned diffuse term keeps the strict bound
    assert rel_err(out["sg_diffuse_rgb"].cpu(), o32["sg_diffuse_rgb"]) <= TOL


def test_tone_mapping(dev):
    """ACESToneMapping (color_correction.py:31-93,116-134): hdr_mode 0 hdr2ldr = aces(x) / t^0.2, ldr2hdr = aces^-1(x t^0.2),
    t clamped to [1e-4, 1]; scalar and per-row shifts; round trip; then the hdr_mode 1 / 2 / -1 curve pairs."""
    from robir_amd import nets
    from robir_oracle import renderer as orend
    tm = nets.ACESToneMapping(0).to(dev)
    g = torch.Generator().manual_seed(9)
    x = torch.rand(513, 3, generator=g) * 4.0
    for shift in (torch.tensor([[0.37]]), torch.rand(513, 1, generator=g) * 1.4 - 0.2):      # incl. values outside [1e-4, 1]
        ldr = tm.hdr2ldr(x.to(dev), shift.to(dev)).cpu()
        assert rel_err(ldr, orend.hdr2ldr(x, shift)) <= 1e-5
        y = torch.rand(513, 3, generator=g) * 0.9
        hdr = tm.ldr2hdr(y.to(dev), shift.to(dev)).cpu()
        assert rel_err(hdr, orend.ldr2hdr(y, shift)) <= 1e-5
        back = tm.ldr2hdr(tm.hdr2ldr(x.to(dev) * 0.2, shift.to(dev)), shift.to(dev)).cpu()
        assert rel_err(back, x * 0.2) <= 1e-4
    assert float(tm.as_input()) == 0.5                                                           # adapt_illum = 0 at init
    gold = load_golden("tonemap")                                                                # the reference's own output
    gx, gy = torch.from_numpy(gold["x"]).to(dev), torch.from_numpy(gold["y"]).to(dev)
    for tag in ("rows", "scalar"):
        sh = torch.from_numpy(gold["shift_" + tag]).to(dev)
        assert rel_err(tm.hdr2ldr(gx, sh).cpu(), gold["ldr_" + tag]) <= 1e-5
        assert rel_err(tm.ldr2hdr(gy, sh).cpu(), gold["hdr_" + tag]) <= 1e-5
        for hm, key in ((1, "m1_"), (2, "m2_"), (-1, "m9_")):                      # warp_aces, ln_space, identity curves
            tmh = nets.ACESToneMapping(hm).to(dev)
            assert rel_err(tmh.hdr2ldr(gx, sh).cpu(), gold["ldr_" + key + tag]) <= 1e-5, (hm, tag)
            assert rel_err(tmh.ldr2hdr(gy * 0.7, sh).cpu(), gold["hdr_" + key + tag]) <= 1e-5, (hm, tag)
            if hm != 1:                                                            # (mode 1 divides by aces^-1(0.73 t): ill-conditioned at t = 1e-4)
                back = tmh.ldr2hdr(tmh.hdr2ldr(gx * 0.2, sh), sh).cpu()
                assert rel_err(back, gx.cpu() * 0.2) <= 1e-4, (hm, tag)


def test_envmap_sg_grid_and_lookup(dev):
    """compute_envmap / render_envmap_sg against the reference's golden grid, render_envmap (bilinear lat-long lookup)
    against torch's grid_sample with the reference's coordinate convention (sg_render.py:9-59)."""
    import torch.nn.functional as F
    from robir_amd import sg_render
    g = load_golden("envmap")
    lgt = torch.from_numpy(g["lgtSGs"]).to(dev)
    grid = sg_render.compute_envmap(lgt, 8, 16).cpu()
    assert rel_err(grid, g["grid"]) <= 1e-5
    gen = torch.Generator().manual_seed(2)
    env = torch.rand(9, 20, 3, generator=gen)
    d = torch.nn.functional.normalize(torch.randn(4000, 3, generator=gen), dim=-1)
    d[:6] = torch.tensor([[0, 0, 1.0], [0, 0, -1.0], [1.0, 0, 0], [-1.0, 0, 0], [0, 1.0, 0], [-1.0, -1e-7, 0]])   # poles, seam
    phi = torch.arccos(d[:, 2]) - 1e-6
    theta = torch.atan2(d[:, 1], d[:, 0])
    q = torch.stack((-theta / np.pi, (phi / np.pi) * 2 - 1)).permute(1, 0)[None, None]
    ref = F.grid_sample(env.permute(2, 0, 1)[None], q, align_corners=True).squeeze().permute(1, 0)
    out = sg_render.render_envmap(env.to(dev), d.to(dev)).cpu()
    assert rel_err(out, ref) <= 1e-4


def test_streaming_kernel_equals_one_point_per_workgroup_kernel(dev, vis_net):
    """The third-generation family (global tile list + persistent grid, csrc/vis_diffuse_v3.hip) puts every (point, direction)
    pair through the instruction sequence of k_dvis_v2: bit-identical visibilities -- for any number of persistent workgroups
    (1, a prime, one per CU, more than there are rounds), for points without a single front-facing direction, for a single
    point, across several chunks with their own direction tables."""
    from robir_amd import ops, sg_render, synth
    g = torch.Generator().manual_seed(12)
    lgt = torch.from_numpy(synth.synth_light_sgs(0, 128)).to(dev)
    for n, C in ((1, 1), (5, 1), (403, 3)):
        pts = ((torch.rand(n, 3, generator=g) - 0.5) * 0.5).to(dev)
        nrm = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
        if n > 3:
            nrm[3] = 0.0                                    # no front-facing direction at all
        nrm = nrm.to(dev)
        cid = (torch.arange(n) * C // n).to(torch.int32).to(dev) if C > 1 else None
        u = torch.rand(2, C, 128, 32, generator=g).to(dev)
        outs = {}
        old_p, old_w = sg_render.VIS_PRECISION, ops.DVIS_STREAM_WORKGROUPS
        try:
            for prec, wgs in (("f16x3-v2", 0), ("f16x3-v3", 0), ("f16x3-v3", 1), ("f16x3-v3", 7), ("f16x3-v3", 4096)):
                sg_render.VIS_PRECISION, ops.DVIS_STREAM_WORKGROUPS = prec, wgs
                stats = {}
                outs[(prec, wgs)] = (sg_render._diffuse_vis_core(pts, nrm, vis_net, lgt, u[0], u[1], 1.0, False, cid, C, stats),
                                     int(stats["diffuse_vis_evals"]))
        finally:
            sg_render.VIS_PRECISION, ops.DVIS_STREAM_WORKGROUPS = old_p, old_w
        ref, evals = outs[("f16x3-v2", 0)]
        assert evals > 0 and bool(torch.isfinite(ref).all())
        for k, (v, e) in outs.items():
            assert e == evals, (n, k)
            assert torch.equal(v, ref), (n, k, float((v - ref).abs().max()))
        ops.range_check(sync=True)


def test_light_visibility_arithmetic_vs_float64(dev, vis_net, oracle_sd, monkeypatch):
    """The default light-visibility kernel since round 6 forms the two outer products of the 2^-22 class (h.xl, l.xh) from bf8 copies of their
    operands (csrc/vis_diffuse_x6t.hip, XT_FP8).  Anchored on a FLOAT64 evaluation of the reference's formulas (oracle, same points,
    directions and draws): the per-lobe visibilities of the default kernel must be no farther from it than those of the f32-input-MFMA
    kernel (factor 1.0 + 2^-23, median and 99th percentile -- the bar tests/test_precision_gpu.py sets for every net), and no farther than
    1.1 x those of round 3's six-exact-products kernel (k_dvis_x6, legacy library; that leg is skipped without it)."""
    from robir_amd import _lib, sg_render, synth
    from robir_oracle import nets as on, sg as osg
    g = np.random.Generator(np.random.PCG64(61))
    n, L, nsamp = 96, 128, 32
    pts = torch.from_numpy((g.standard_normal((n, 3)) * 0.25).astype(np.float32))
    nrm = torch.nn.functional.normalize(torch.from_numpy(g.standard_normal((n, 3)).astype(np.float32)), dim=-1)
    lgt = torch.from_numpy(synth.synth_light_sgs(3, L))
    lobe, lam = lgt[:, :3], lgt[:, 3:4].abs()
    u = torch.from_numpy(g.random((2, L, nsamp), dtype=np.float32))
    draws = {"dvis_theta": u[0][None].to(dev), "dvis_phi": u[1][None].to(dev)}
    sd64 = {k: v.double() for k, v in oracle_sd.items() if k.startswith("visibility_network.")}
    # the sampled directions and the cull in fp32 like every kernel's (shared geometry), the MLP and everything behind it in float64
    r64 = osg.diffuse_visibility(pts, nrm, lambda p, d: on.vis_logits(sd64, p.double(), d.double()),
                                 torch.nn.functional.normalize(lobe, dim=-1), lam, u[0], u[1]).double()
    modes = ["f16x6", "fp32"]
    try:
        _lib.legacy()
        modes.append("f16x6-1t")
    except Exception:
        pass
    err = {}
    for mode in modes:
        monkeypatch.setattr(sg_render, "VIS_PRECISION", mode)
        out = sg_render.get_diffuse_visibility(pts.to(dev), nrm.to(dev), vis_net, lobe.to(dev), lam.to(dev), nsamp=nsamp, draws=draws).cpu().double()
        e = ((out - r64).abs() / (r64.abs() + r64.abs().mean())).flatten()
        err[mode] = (float(e.median()), float(e.kthvalue(int(0.99 * e.numel())).values), float(e.max()))
        record_metric("light_visibility_vs_float64/" + mode, median=err[mode][0], p99=err[mode][1], max=err[mode][2])
        print(f"light visibility vs float64, {mode}: median {err[mode][0]:.2e} p99 {err[mode][1]:.2e} max {err[mode][2]:.2e}")
    eps = 2.0 ** -23
    for i in (0, 1):
        assert err["f16x6"][i] <= 1.0 * err["fp32"][i] + eps, (i, err)
        if "f16x6-1t" in err:
            assert err["f16x6"][i] <= 1.1 * err["f16x6-1t"][i] + 1e-8, (i, err)
