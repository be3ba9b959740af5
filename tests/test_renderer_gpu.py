"""IDRNetwork drop-in on the GPU: forward('Material'), the batched multi-chunk renderer, forward('Illum') +
trace_radiance -- against the oracle (same octree tables, same draws) and the reference's golden outputs."""
import numpy as np
import pytest
import torch

from conftest import rel_err, bad_frac, load_golden, bounded, record_metric

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def model(dev):
    from robir_amd import renderer
    return renderer.build_synthetic_model(dev, seed=0, variance=0.3)


@pytest.fixture(scope="module")
def model_oracle_tables(dev, oracle_octree):
    """Same weights, but tracing the octree the ORACLE built: isolates everything downstream of the cast."""
    from robir_amd import renderer
    from robir_amd.octree_tracing import OctreeSDF
    m = renderer.build_synthetic_model(dev, seed=0, variance=0.3, build_octrees=False)
    m.ray_tracer.sdf_octree = OctreeSDF.from_host_tables(oracle_octree, dev, -1)
    m.octree_ray_tracer.sdf_octree = OctreeSDF.from_host_tables(oracle_octree, dev, 32)
    return m


def _inputs(dev, c, H=64, W=64):
    from robir_amd import synth
    uv, pose, K = synth.synth_camera(H, W)
    sl = slice(c * 1024, (c + 1) * 1024)
    return (torch.from_numpy(uv[sl]).to(dev), torch.from_numpy(pose).to(dev), torch.from_numpy(K).to(dev), uv, pose, K, sl)


FIELDS = ("sg_rgb", "indir_rgb", "sg_diffuse_rgb", "sg_specular_rgb", "indir_diffuse_rgb", "indir_specular_rgb",
          "vis_shadow", "diffuse_albedo", "roughness", "metallic", "normals", "normal_map", "random_xi_roughness",
          "random_xi_metallic", "random_xi_diffuse_albedo")


# fields whose only > 1e-4 differences between two fp32 evaluations are cull flips; the two specular-lobe sums are differences of
# hemisphere integrals that cancel to ~1e-3 of their terms -- there two fp32 evaluations of the reference's own formulas are a few 1e-4
# apart (tests/test_sg_gpu.py::test_render_with_sg_fun_spec_vs_oracle anchors both on float64), cull or no cull
CULL_EXACT_FIELDS = ("vis_shadow", "sg_rgb", "sg_diffuse_rgb", "indir_rgb", "indir_diffuse_rgb", "diffuse_albedo", "roughness", "metallic",
                     "normals", "normal_map", "random_xi_roughness", "random_xi_metallic", "random_xi_diffuse_albedo")


def _assert_unmarked_within_1e4(out, ref, lgt_sgs, draws, tag, hit=None, max_marked_frac=0.01, skip=(), tol_for=None):
    """-> number of UNATTRIBUTED entries beyond 1e-4 over CULL_EXACT_FIELDS (asserted 0 by the callers); prints / records the marked count."""
    from conftest import cull_marked_points, record_metric
    h = ref["network_object_mask"] if hit is None else hit
    h = torch.as_tensor(h).bool()
    marked, pairs = cull_marked_points(torch.as_tensor(lgt_sgs), draws["dvis_theta"], draws["dvis_phi"], torch.as_tensor(ref["normal_map"])[h],
                                       out["normal_map"].cpu()[h])
    n_hit = int(h.sum())
    assert int(marked.sum()) <= max(2, max_marked_frac * n_hit), (int(marked.sum()), n_hit)        # a handful of points: 0.1-0.4 % measured
    bad = 0
    worst = {}
    for k in CULL_EXACT_FIELDS:
        if k in skip:
            worst[k] = float("nan")
            continue
        a, b = out[k].cpu()[h].double(), torch.as_tensor(ref[k])[h].double()
        e = (a - b).abs() / (b.abs() + b.abs().mean())                      # the repo's floored relative error (conftest.rel_err)
        eu = e[~marked]
        worst[k] = float(eu.max()) if eu.numel() else 0.0
        tol = (tol_for or {}).get(k, 1e-4)
        bad += int((eu > tol).sum())
        if int((eu > tol).sum()) or tol != 1e-4:
            print(f"[cull attribution/{tag}] {k}: {int((eu > tol).sum())} unattributed entries > {tol:.1e}, worst {worst[k]:.2e}")
    plain = {}
    for k in ("vis_shadow", "sg_rgb"):                                       # the plain relative error of the same entries, where |ref| is not tiny
        a, b = out[k].cpu()[h].double()[~marked], torch.as_tensor(ref[k])[h].double()[~marked]
        big = b.abs() > 1e-3 * b.abs().mean()
        plain[k] = float(((a - b).abs()[big] / b.abs()[big]).max()) if bool(big.any()) else 0.0
    print(f"[cull attribution/{tag}] {int(marked.sum())} of {n_hit} hit points ({pairs} of {n_hit * 4096} pairs) on the n.d > 1e-6 cull; "
          f"{bad} unattributed entries > 1e-4; worst unmarked: vis_shadow {worst['vis_shadow']:.1e}, sg_rgb {worst['sg_rgb']:.1e}, "
          f"indir_rgb {worst['indir_rgb']:.1e} (plain relative: vis_shadow {plain['vis_shadow']:.1e}, sg_rgb {plain['sg_rgb']:.1e})")
    record_metric("cull_attribution/" + tag, marked_points=int(marked.sum()), hit_points=n_hit, marked_pairs=pairs, unattributed=bad,
                  worst_unmarked_vis_shadow=worst["vis_shadow"], worst_unmarked_sg_rgb=worst["sg_rgb"])
    return bad


def _same_tables_pair(dev, model_oracle_tables, oracle_sd, oracle_octree, c):
    """chunk c of the 64 x 64 view through the oracle and through the device path, on the oracle's octree tables and the same draws"""
    from robir_amd import synth
    from robir_oracle import renderer as orend
    from robir_oracle import octree as ooct
    uv_d, pose_d, K_d, uv, pose, K, sl = _inputs(dev, c)
    hdr = torch.full((1024, 1), 0.5)
    dirs, cam = orend.camera_rays(torch.from_numpy(uv[sl])[None], torch.from_numpy(pose)[None], torch.from_numpy(K)[None])
    _, hit, _ = ooct.trace(oracle_octree, cam, dirs, -1)
    drt = {k: torch.from_numpy(v) for k, v in synth.pbr_draws(0, int(hit.sum()), chunk_id=c).items()}
    ref = orend.forward(oracle_sd, oracle_octree, torch.from_numpy(uv[sl])[None], torch.from_numpy(pose)[None],
                        torch.from_numpy(K)[None], torch.ones(1, 1024, dtype=torch.bool), hdr, drt, "Material", testing=True)
    inp = {"uv": uv_d[None], "pose": pose_d[None], "intrinsics": K_d[None],
           "object_mask": torch.ones(1, 1024, dtype=torch.bool, device=dev), "hdr_shift": hdr.to(dev)}
    out = model_oracle_tables(inp, trainstage="Material", train_spec=True, draws={k: v.to(dev) for k, v in drt.items()})
    torch.cuda.synchronize()
    return out, ref, drt


@pytest.mark.parametrize("c", [0, 2, 3])
def test_cull_attribution_rest_of_the_view(dev, model_oracle_tables, oracle_sd, oracle_octree, c):
    """The attribution of test_forward_material_vs_oracle_same_tables (chunk 1) on the other three chunks of the 64 x 64 view: off the
    reference's n.d > 1e-6 cull (model/sg_render.py:155) EVERY entry of the non-specular fields holds north_star's 1e-4."""
    out, ref, drt = _same_tables_pair(dev, model_oracle_tables, oracle_sd, oracle_octree, c)
    assert bool((out["network_object_mask"].cpu() == ref["network_object_mask"]).all())
    if int(ref["network_object_mask"].sum()) == 0:
        pytest.skip("no hit ray in this chunk")
    assert _assert_unmarked_within_1e4(out, ref, oracle_sd["envmap_material_network.lgtSGs"], drt, "same_tables_c%d" % c) == 0


def test_forward_material_vs_oracle_same_tables(dev, model_oracle_tables, oracle_sd, oracle_octree):
    from robir_amd import synth
    from robir_oracle import renderer as orend
    c = 1
    uv_d, pose_d, K_d, uv, pose, K, sl = _inputs(dev, c)
    hdr = torch.full((1024, 1), 0.5)
    # hit count of this chunk from the oracle cast (needed to size the draws)
    dirs, cam = orend.camera_rays(torch.from_numpy(uv[sl])[None], torch.from_numpy(pose)[None], torch.from_numpy(K)[None])
    from robir_oracle import octree as ooct
    _, hit, _ = ooct.trace(oracle_octree, cam, dirs, -1)
    n_hit = int(hit.sum())
    dr = synth.pbr_draws(0, n_hit, chunk_id=c)
    drt = {k: torch.from_numpy(v) for k, v in dr.items()}
    stats_o = {}
    ref = orend.forward(oracle_sd, oracle_octree, torch.from_numpy(uv[sl])[None], torch.from_numpy(pose)[None],
                        torch.from_numpy(K)[None], torch.ones(1, 1024, dtype=torch.bool), hdr, drt, "Material",
                        testing=True, stats=stats_o)
    stats = {}
    inp = {"uv": uv_d[None], "pose": pose_d[None], "intrinsics": K_d[None],
           "object_mask": torch.ones(1, 1024, dtype=torch.bool, device=dev), "hdr_shift": hdr.to(dev)}
    out = model_oracle_tables(inp, trainstage="Material", train_spec=True,
                              draws={k: v.to(dev) for k, v in drt.items()}, stats=stats)
    torch.cuda.synchronize()
    assert bool((out["network_object_mask"].cpu() == ref["network_object_mask"]).all())
    # the n.d > 1e-6 cull may flip for a direction whose cosine is within an ulp of the threshold
    assert abs(int(stats["diffuse_vis_evals"]) - stats_o["diffuse_vis_evals"]) <= 4
    assert rel_err(out["points"].cpu(), ref["points"]) <= 1e-6
    assert rel_err(out["sdf_output"].cpu(), ref["sdf_output"]) <= 1e-4
    # chained stages (materials -> illum SGs -> visibility -> shading): each stage is within 1e-4 on identical
    # inputs (test_sg_gpu / test_mlp_gpu); the chain compounds fp32 noise, so: 99.5 % of entries within 2e-4, all < 1e-3
    for k in FIELDS:
        assert bad_frac(out[k].cpu(), ref[k], 2e-4) <= 0.005, (k, bad_frac(out[k].cpu(), ref[k], 2e-4))
        assert rel_err(out[k].cpu(), ref[k]) <= 1e-3, (k, rel_err(out[k].cpu(), ref[k]))
    # north_star's 1e-4 on EVERY entry that is not attributable to a threshold decision (VERDICT r5 task 4): the points with a sampled
    # light direction ON the reference's n.d > 1e-6 cull (model/sg_render.py:155) are identified from the oracle's own directions and the
    # two evaluations' normals (conftest.cull_marked_points); every other hit point took the same 4096 cull decisions in both, and
    # there every field but the two specular-lobe sums holds 1e-4 (floored) -- measured: 1.9e-7 for vis_shadow, whose 2.2e-4 outliers
    # are exactly the marked points.
    unattributed = _assert_unmarked_within_1e4(out, ref, oracle_sd["envmap_material_network.lgtSGs"], drt, "same_tables")
    assert unattributed == 0
    # every key / shape / dtype of the reference's return dict
    g = load_golden("forward_material_c1")
    for k in g:
        if k.startswith("out_"):
            name = k[4:]
            assert name in out, name
            assert tuple(out[name].shape) == tuple(g[k].shape), (name, out[name].shape, g[k].shape)


def test_forward_two_views_one_batch_vs_oracle(dev, model_oracle_tables, oracle_sd, oracle_octree):
    """Batch size 2 (implicit_differentiable_renderer.py:299-305,324): two views, 512 pixels each, are ONE lock-step cast over 1024 rays
    from two camera centres, every output flattened to [B N, ...] -- against the oracle on the same octree tables and draws."""
    from robir_amd import synth
    from robir_oracle import renderer as orend, octree as ooct
    uv, pose, K = synth.synth_camera(64, 64)
    a = 0.6                                  # second view: the camera turned about the y axis (still looking at the origin)
    R = np.array([[np.cos(a), 0, np.sin(a), 0], [0, 1, 0, 0], [-np.sin(a), 0, np.cos(a), 0], [0, 0, 0, 1]], np.float32)
    poses = np.stack([pose, R @ pose])
    uvs = np.stack([uv[1024 + 256:1024 + 768], uv[2048 + 128:2048 + 640]])
    Ks = np.stack([K, K])
    uv_t, pose_t, K_t = torch.from_numpy(uvs), torch.from_numpy(poses), torch.from_numpy(Ks)
    hdr = torch.full((1024, 1), 0.5)
    dirs, cam = orend.camera_rays(uv_t, pose_t, K_t)
    _, hit, _ = ooct.trace(oracle_octree, cam, dirs, -1)
    assert 0 < int(hit[:512].sum()) and 0 < int(hit[512:].sum())
    drt = {k: torch.from_numpy(v) for k, v in synth.pbr_draws(0, int(hit.sum()), chunk_id=7).items()}
    ref = orend.forward(oracle_sd, oracle_octree, uv_t, pose_t, K_t, torch.ones(2, 512, dtype=torch.bool), hdr, drt, "Material",
                        testing=True)
    inp = {"uv": uv_t.to(dev), "pose": pose_t.to(dev), "intrinsics": K_t.to(dev),
           "object_mask": torch.ones(2, 512, dtype=torch.bool, device=dev), "hdr_shift": hdr.to(dev)}
    out = model_oracle_tables(inp, trainstage="Material", train_spec=True, draws={k: v.to(dev) for k, v in drt.items()})
    torch.cuda.synchronize()
    assert tuple(out["points"].shape) == (1024, 3) and tuple(out["sg_rgb"].shape) == (1024, 3)
    assert bool((out["network_object_mask"].cpu() == ref["network_object_mask"]).all())
    assert rel_err(out["points"].cpu(), ref["points"]) <= 1e-6
    assert rel_err(out["ray_dirs"].cpu(), ref["ray_dirs"]) <= 1e-6
    # the bounds of test_forward_material_vs_oracle_same_tables; the indirect specular term (a sum of differences of hemisphere integrals,
    # small on the turned view's grazing pixels) is where two fp32 evaluations of the reference's formulas are a few 1e-4 apart
    # (test_sg_gpu.py::test_render_with_sg_fun_spec_vs_oracle measures both against float64): 0.55 % of its entries beyond 2e-4 here
    for k in FIELDS:
        lim = 0.01 if k == "indir_specular_rgb" else 0.005
        assert bad_frac(out[k].cpu(), ref[k], 2e-4) <= lim, (k, bad_frac(out[k].cpu(), ref[k], 2e-4))
        assert rel_err(out[k].cpu(), ref[k]) <= 1e-3, (k, rel_err(out[k].cpu(), ref[k]))
    # each view alone renders the same hit set (rays do not interact; only the lock-step schedule is shared)
    for b in range(2):
        one = model_oracle_tables({"uv": uv_t[b:b + 1].to(dev), "pose": pose_t[b:b + 1].to(dev), "intrinsics": K_t[b:b + 1].to(dev),
                                   "object_mask": torch.ones(1, 512, dtype=torch.bool, device=dev), "hdr_shift": hdr[:512].to(dev)},
                                  trainstage="Illum", draws={})
        assert bool((one["network_object_mask"] == out["network_object_mask"][b * 512:(b + 1) * 512]).all())


def test_forward_material_second_weight_set(dev):
    """A second synthetic checkpoint (other seed, sharper NeuS variance, light SGs shaped like the shipped fits: |lambda|
    up to ~500) through the whole Material forward against the oracle on the same octree cells."""
    from conftest import oracle_tables_from_device
    from robir_amd import renderer, synth
    from robir_oracle import nets as on, octree as ooct, renderer as orend
    m = renderer.build_synthetic_model(dev, seed=3, variance=0.6, sharp_light=True)
    sd = on.as_torch(synth.synth_state_dict(3, variance=0.6, sharp_light=True))
    T = oracle_tables_from_device(m.ray_tracer.sdf_octree.tables)
    uv_d, pose_d, K_d, uv, pose, K, sl = _inputs(dev, 2)
    uv_t, pose_t, K_t = torch.from_numpy(uv[sl])[None], torch.from_numpy(pose)[None], torch.from_numpy(K)[None]
    dirs, cam = orend.camera_rays(uv_t, pose_t, K_t)
    _, hit, _ = ooct.trace(T, cam, dirs, -1)
    n_hit = int(hit.sum())
    assert n_hit > 300
    drt = {k: torch.from_numpy(v) for k, v in synth.pbr_draws(11, n_hit, chunk_id=2).items()}
    hdr = torch.full((1024, 1), 0.35)
    ref = orend.forward(sd, T, uv_t, pose_t, K_t, torch.ones(1, 1024, dtype=torch.bool), hdr, drt, "Material", testing=True)
    inp = {"uv": uv_d[None], "pose": pose_d[None], "intrinsics": K_d[None],
           "object_mask": torch.ones(1, 1024, dtype=torch.bool, device=dev), "hdr_shift": hdr.to(dev)}
    out = m(inp, trainstage="Material", train_spec=True, draws={k: v.to(dev) for k, v in drt.items()})
    assert bool((out["network_object_mask"].cpu() == ref["network_object_mask"]).all())
    assert rel_err(out["points"].cpu(), ref["points"]) <= 1e-6
    # With these sharp lights the specular term is ill-conditioned in fp32 (tests/test_sg_gpu.py::
    # test_specular_term_conditioning: the reference's formula differs by percent between fp32 and fp64), so 1e-6 input
    # differences show up at the 1e-3 level there -- identically with the exact-fp32 kernels.  Everything else keeps the
    # chained bound.
    loose = ("sg_specular_rgb", "sg_rgb", "indir_specular_rgb")
    for k in FIELDS:
        frac, worst = bad_frac(out[k].cpu(), ref[k], 2e-4), rel_err(out[k].cpu(), ref[k])
        if k in loose:
            assert frac <= 0.1 and worst <= 5e-3, (k, frac, worst)
        else:
            assert frac <= 0.005 and worst <= 1e-3, (k, frac, worst)


SPREAD_FACTOR = 4.0


def test_forward_material_vs_reference_golden(dev, model):
    """End to end with the device-built octree against the reference's own output (looser: PE amplifies the ~1e-6
    hit-position noise of two independently built octrees 512-fold -- DESIGN.md 'Parity tolerances')."""
    g = load_golden("forward_material_c1")
    c = int(g["chunk"])
    uv_d, pose_d, K_d, *_ = _inputs(dev, c)
    draws = {k[5:]: torch.from_numpy(v).to(dev) for k, v in g.items() if k.startswith("draw_")}
    inp = {"uv": uv_d[None], "pose": pose_d[None], "intrinsics": K_d[None],
           "object_mask": torch.ones(1, 1024, dtype=torch.bool, device=dev),
           "hdr_shift": torch.from_numpy(g["hdr_shift"]).expand(1024, 1).contiguous().to(dev)}
    out = model(inp, trainstage="Material", train_spec=True, draws=draws)
    assert int((out["network_object_mask"].cpu().numpy() != g["out_network_object_mask"]).sum()) <= 2
    same = torch.from_numpy(g["out_network_object_mask"]) == out["network_object_mask"].cpu()
    for k in ("points", "sdf_output", "ray_dirs"):
        bounded("forward_material_vs_reference_golden/" + k, out[k].cpu()[same], torch.from_numpy(g["out_" + k])[same], 1e-4, 0.002)
    for k in FIELDS:
        bounded("forward_material_vs_reference_golden/" + k, out[k].cpu()[same], torch.from_numpy(g["out_" + k])[same], 2e-3, 0.003)
    # ... and against the yardstick of this comparison: what an INDEPENDENT octree build alone does to each field.  tests/golden/
    # self_spread.json (oracle/gen_golden_r4.py) holds, per field, the distance between this golden output (the reference on its own
    # octree) and the CPU oracle on the oracle's own octree build, same draws -- the oracle reproduces the reference to 0.0 on the
    # reference's octree tables, so that distance is the octree build's doing.  The HIP path (device-built octree, kernels' fp32
    # evaluation order) stays within SPREAD_FACTOR of it on every field (measured on an MI355X: 1.2-3.7, most fields 1.9-2.3 -- the hit
    # positions themselves sit 1.9 x the oracle's own-build distance from the reference's).
    import json
    import os
    spread = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "self_spread.json")))["fields"]
    for k in ("points", "sdf_output") + tuple(FIELDS):
        if spread.get(k, 0.0) > 0.0:
            m = rel_err(out[k].cpu()[same], torch.from_numpy(g["out_" + k])[same])
            record_metric("self_spread_ratio/" + k, hip_vs_reference=m, self_spread=spread[k], ratio=m / spread[k])
            assert m <= SPREAD_FACTOR * spread[k], (k, m, spread[k])


def test_batched_chunks_equal_per_chunk_forward(dev, model):
    """render_chunks over 3 chunks == three forward() calls (chunk-global quantities stay per chunk)."""
    from robir_amd import synth
    uv, pose, K = synth.synth_camera(64, 64)
    pose_d, K_d = torch.from_numpy(pose).to(dev), torch.from_numpy(K).to(dev)
    hdr = torch.full((3072, 1), 0.5, device=dev)
    uv_d = torch.from_numpy(uv[:3072]).to(dev)
    # hits per chunk (to size the draws) from a draw-free Illum pass
    pre = model.render_chunks(uv_d, pose_d, K_d, hdr, trainstage="Illum",
                              draws={"illum_randn": None, "normal_randn": None})
    hit = pre["network_object_mask"].cpu()
    counts = [int(hit[i * 1024:(i + 1) * 1024].sum()) for i in range(3)]
    per = [synth.pbr_draws(0, counts[i], chunk_id=i) for i in range(3)]
    cat = {k: torch.from_numpy(np.concatenate([p[k] for p in per])).to(dev) for k in per[0] if not k.startswith("dvis")}
    for k in ("dvis_theta", "dvis_phi"):
        cat[k] = torch.from_numpy(np.stack([p[k] for p in per])).to(dev)
    big = model.render_chunks(uv_d, pose_d, K_d, hdr, draws=cat)
    for i in range(3):
        sl = slice(i * 1024, (i + 1) * 1024)
        inp = {"uv": uv_d[None, sl], "pose": pose_d[None], "intrinsics": K_d[None],
               "object_mask": torch.ones(1, 1024, dtype=torch.bool, device=dev), "hdr_shift": hdr[sl]}
        one = model(inp, trainstage="Material", train_spec=True,
                    draws={k: torch.from_numpy(v).to(dev) for k, v in per[i].items()})
        for k in FIELDS + ("points", "sdf_output"):
            e = rel_err(big[k][sl].cpu(), one[k].cpu())
            assert e <= 1e-6, (i, k, e)
    # forward() itself with model.lockstep_chunk = 1024: one call on all 3072 pixels == the batched render
    model.lockstep_chunk = 1024
    try:
        inp = {"uv": uv_d[None], "pose": pose_d[None], "intrinsics": K_d[None],
               "object_mask": torch.ones(1, 3072, dtype=torch.bool, device=dev), "hdr_shift": hdr}
        whole = model(inp, trainstage="Material", train_spec=True, draws=cat)
    finally:
        model.lockstep_chunk = None
    assert torch.equal(whole["network_object_mask"], big["network_object_mask"])
    for k in FIELDS + ("points", "sdf_output"):
        assert rel_err(whole[k].cpu(), big[k].cpu()) == 0.0, k          # NaN-aware (axis-parallel rays)


def test_illum_and_trace_radiance_vs_golden(dev, model_oracle_tables):
    g = load_golden("trace_radiance")
    c = int(g["chunk"])
    uv_d, pose_d, K_d, *_ = _inputs(dev, c)
    inp = {"uv": uv_d[None], "pose": pose_d[None], "intrinsics": K_d[None],
           "object_mask": torch.ones(1, 1024, dtype=torch.bool, device=dev),
           "hdr_shift": torch.from_numpy(g["in_hdr_shift"]).to(dev)}
    ill = model_oracle_tables(inp, trainstage="Illum", draws={"illum_randn": torch.from_numpy(g["illum_noise"]).to(dev),
                                                              "normal_randn": torch.from_numpy(g["normal_noise"]).to(dev)})
    mask = torch.from_numpy(g["in_mask"])
    assert int((ill["network_object_mask"].cpu() != mask).sum()) <= 2
    bounded("illum_vs_reference_golden/normals", ill["normals"].cpu(), g["in_normals"], 2e-3, 0.005)
    bounded("illum_vs_reference_golden/indirect_sgs", ill["indirect_sgs"].cpu(), g["illum_sgs"], 2e-3, 0.01)
    # trace_radiance fed with the REFERENCE's forward output (stage parity)
    fwd = {"points": torch.from_numpy(g["in_points"]).to(dev), "hdr_shift": torch.from_numpy(g["in_hdr_shift"]).to(dev),
           "network_object_mask": mask.to(dev), "normals": torch.from_numpy(g["in_normals"]).to(dev)}
    out = model_oracle_tables.trace_radiance(fwd, nsamp=int(g["nsamp"]),
                                             draws=(torch.from_numpy(g["u1"]), torch.from_numpy(g["u2"])))
    assert rel_err(out["sample_dirs"].cpu(), g["out_sample_dirs"]) <= 1e-5
    assert int((out["gt_vis"].cpu().numpy() != g["out_gt_vis"]).sum()) <= 4
    assert rel_err(out["pred_vis"].cpu(), g["out_pred_vis"]) <= 1e-4
    bounded("trace_radiance_vs_reference_golden/trace_radiance", out["trace_radiance"].cpu(), g["out_trace_radiance"], 1e-3, 0.002)
    bounded("trace_radiance_vs_reference_golden/gt_integral", out["gt_integral"].cpu(), g["out_gt_integral"], 1e-3, 0.005)
    assert bool((out["indir_mask"].cpu().numpy() == g["out_indir_mask"]).mean() > 0.999)


def test_trace_radiance_second_weight_set(dev):
    """Vis-stage path (secondary rays, borrow_color through the 4-column SDF kernel, visibility MLP) with another
    checkpoint, against the oracle fed with the kernels' own Illum forward (same octree cells)."""
    from conftest import oracle_tables_from_device
    from robir_amd import renderer, synth
    from robir_oracle import nets as on, renderer as orend
    m = renderer.build_synthetic_model(dev, seed=3, variance=0.6, sharp_light=True)
    sd = on.as_torch(synth.synth_state_dict(3, variance=0.6, sharp_light=True))
    T = oracle_tables_from_device(m.ray_tracer.sdf_octree.tables)
    uv_d, pose_d, K_d, *_ = _inputs(dev, 2)
    hdr = torch.full((1024, 1), 0.35, device=dev)
    inp = {"uv": uv_d[None], "pose": pose_d[None], "intrinsics": K_d[None],
           "object_mask": torch.ones(1, 1024, dtype=torch.bool, device=dev), "hdr_shift": hdr}
    ill = m(inp, trainstage="Illum")
    n = int(ill["network_object_mask"].sum())
    g = torch.Generator().manual_seed(5)
    u1, u2 = torch.rand(n * 8, generator=g), torch.rand(n * 8, generator=g)
    fwd = {"points": ill["points"], "hdr_shift": hdr, "network_object_mask": ill["network_object_mask"], "normals": ill["normals"]}
    out = m.trace_radiance(fwd, nsamp=8, draws=(u1, u2))
    ref = orend.trace_radiance(sd, T, {k: v.cpu() for k, v in fwd.items()}, 8, u1, u2)
    assert rel_err(out["sample_dirs"].cpu(), ref["sample_dirs"]) <= 1e-5
    assert int((out["gt_vis"].cpu() != ref["gt_vis"]).sum()) <= 4
    assert rel_err(out["pred_vis"].cpu(), ref["pred_vis"]) <= 1e-4
    bounded("trace_radiance_second_ckpt/trace_radiance", out["trace_radiance"].cpu(), ref["trace_radiance"], 1e-3, 0.004)
    bounded("trace_radiance_second_ckpt/gt_integral", out["gt_integral"].cpu(), ref["gt_integral"], 1e-3, 0.01)
    assert float((out["indir_mask"].cpu() == ref["indir_mask"]).float().mean()) > 0.999
    # test_dir (implicit_differentiable_renderer.py:594-595): one given direction for every sample
    td = torch.nn.functional.normalize(torch.tensor([0.3, -0.5, 0.8]), dim=0)
    out = m.trace_radiance(fwd, nsamp=4, test_dir=td)
    ref = orend.trace_radiance(sd, T, {k: v.cpu() for k, v in fwd.items()}, 4, None, None, test_dir=td)
    assert torch.equal(out["sample_dirs"].cpu(), ref["sample_dirs"].contiguous())
    assert int((out["gt_vis"].cpu() != ref["gt_vis"]).sum()) <= 4
    assert rel_err(out["pred_vis"].cpu(), ref["pred_vis"]) <= 1e-4
    assert float((out["indir_mask"].cpu() == ref["indir_mask"]).float().mean()) > 0.999
    both = (out["gt_vis"].cpu() == ref["gt_vis"])[..., 0].all(-1)
    assert rel_err(out["trace_radiance"].cpu()[both], ref["trace_radiance"][both]) <= 2e-3
    assert rel_err(out["gt_integral"].cpu()[both], ref["gt_integral"][both]) <= 2e-3


def test_points_dirs_form_equals_uv_form(dev, model):
    """forward({'points','dirs'}) (implicit_differentiable_renderer.py:306-322) on the camera's own rays == uv form."""
    from robir_amd import synth, ops
    uv, pose, K = synth.synth_camera(64, 64)
    sl = slice(1024, 2048)
    uv_d, pose_d, K_d = torch.from_numpy(uv[sl]).to(dev), torch.from_numpy(pose).to(dev), torch.from_numpy(K).to(dev)
    hdr = torch.full((1024, 1), 0.5, device=dev)
    n0 = model.render_chunks(uv_d, pose_d, K_d, hdr, trainstage="Illum", draws={})["network_object_mask"].sum().item()
    dr = {k: torch.from_numpy(v).to(dev) for k, v in synth.pbr_draws(0, int(n0), chunk_id=1).items()}
    a = model({"uv": uv_d[None], "pose": pose_d[None], "intrinsics": K_d[None],
               "object_mask": torch.ones(1, 1024, dtype=torch.bool, device=dev), "hdr_shift": hdr},
              trainstage="Material", train_spec=True, draws=dr)
    dirs = ops.camera_rays(pose, K, uv_d)
    cam = pose_d[:3, 3].reshape(1, 3).expand(1024, 3).contiguous()
    b = model({"points": cam[None], "dirs": dirs[None], "hdr_shift": hdr}, trainstage="Material", train_spec=True, draws=dr)
    assert bool((a["network_object_mask"] == b["network_object_mask"]).all())
    for k in ("points", "sg_rgb", "indir_rgb", "vis_shadow", "normal_map"):
        assert rel_err(a[k].cpu(), b[k].cpu()) <= 1e-6, k


def test_norm_hook_and_render_view(dev, model):
    from robir_amd import renderer, render, synth
    uv, pose, K = synth.synth_camera(96, 96)
    out = render.render_view(model, uv, pose, K, chunks_per_pass=4)
    assert out["pred_rgb"].shape == (96 * 96, 3) and bool(torch.isfinite(out["pred_rgb"][out["network_object_mask"]]).all())
    # the PNG set of scripts/relight.py (original light: composite, roughness, albedo, normal)
    import os
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        names = render.save_relight_images(out, 96, 96, d, "0")
        assert names == ["albedo", "normal", "roughness", "sg_rgb_bg"]
        from PIL import Image
        im = np.asarray(Image.open(os.path.join(d, "sg_rgb_bg_0.png")))
        assert im.shape == (96, 96, 3) and im.dtype == np.uint8 and int(im.max()) > 0
    old = model.get_sg_render
    try:
        model.get_sg_render = renderer.NormHook(model)
        uv_d = torch.from_numpy(uv[:1024]).to(dev)
        o = model({"uv": uv_d[None], "pose": torch.from_numpy(pose).to(dev)[None], "intrinsics": torch.from_numpy(K).to(dev)[None],
                   "object_mask": torch.ones(1, 1024, dtype=torch.bool, device=dev),
                   "hdr_shift": torch.full((1024, 1), 0.5, device=dev)}, trainstage="Material", train_spec=True)
        hit = o["network_object_mask"]
        assert rel_err(o["diffuse_albedo"][hit].cpu(), o["normal_map"][hit].cpu()) == 0.0
        assert float(o["sg_rgb"].min()) == 1.0 and float(o["indir_rgb"][hit].abs().max()) == 0.0
    finally:
        model.__dict__.pop("get_sg_render", None)
        assert model.get_sg_render.__func__ is renderer.IDRNetwork.get_sg_render


def test_relight_with_loaded_light(dev):
    """scripts/relight.py:33-117 in small: EnvmapMaterialNetwork.load_light (sg_128.npy + .exr background read by
    robir_amd.exr), full view through render_view; the background pixels are the bilinear lat-long lookup of the decoded map
    (render_envmap, sg_render.py:45-59) and the object shading follows the new light SGs."""
    import os
    import shutil
    import tempfile
    import torch.nn.functional as F
    from conftest import GOLD
    from robir_amd import renderer, render, synth, exr, ops
    m = renderer.build_synthetic_model(dev, seed=0, variance=0.3)
    uv, pose, K = synth.synth_camera(96, 96)
    base = render.render_view(m, uv, pose, K, chunks_per_pass=4)
    with tempfile.TemporaryDirectory() as d:
        light = os.path.join(d, "envmapT")
        os.makedirs(light)
        sgs = m.envmap_material_network.lgtSGs.detach().cpu().numpy().copy()
        sgs[:, -3:] *= 0.5                                                  # half the energy of the fitted light
        np.save(os.path.join(light, "sg_128.npy"), sgs)
        shutil.copy(os.path.join(GOLD, "envmap6_rows0_31.exr"), light + ".exr")
        m.envmap_material_network.load_light(light)
        env = torch.from_numpy(exr.read_exr(light + ".exr")[:, :, :3].copy())
    out = render.render_view(m, uv, pose, K, chunks_per_pass=4)
    hit = out["network_object_mask"].cpu()
    assert torch.equal(hit, base["network_object_mask"].cpu()) and 0.2 < float(hit.float().mean()) < 0.9
    # background: grid_sample of the decoded map with the reference's coordinate convention
    dirs = ops.camera_rays(pose, K, torch.from_numpy(uv).to(dev)).cpu()
    phi = torch.arccos(dirs[:, 2]) - 1e-6
    theta = torch.atan2(dirs[:, 1], dirs[:, 0])
    q = torch.stack((-theta / np.pi, (phi / np.pi) * 2 - 1)).permute(1, 0)[None, None]
    ref = F.grid_sample(env.permute(2, 0, 1)[None], q, align_corners=True).squeeze().permute(1, 0)
    assert rel_err(out["bg_rgb"].cpu()[~hit], ref[~hit]) <= 1e-4
    assert rel_err(out["pred_rgb"].cpu()[~hit], ref[~hit]) <= 1e-4
    # direct shading is linear in the light amplitudes (same directions drawn: the lobes did not move)
    torch.manual_seed(3)
    a = render.render_view(m, uv, pose, K, chunks_per_pass=4)["sg_rgb"].cpu()[hit]
    m.envmap_material_network.lgtSGs.data[:, -3:] *= 2.0
    torch.manual_seed(3)
    b = render.render_view(m, uv, pose, K, chunks_per_pass=4)["sg_rgb"].cpu()[hit]
    assert rel_err(2.0 * a, b) <= 1e-4


@pytest.mark.parametrize("env_id", [6, 12])
def test_relight_forward_vs_reference_golden(dev, env_id):
    """SURVEY 8(f)1 pinned on the reference: the relight serve loop's forward (scripts/relight.py:33-60) under a LOADED light.  The golden
    (oracle/gen_golden_r5.py) is the reference's own IDRNetwork.forward('Material') after `load_light` semantics -- lgtSGs.data := the
    SHIPPED fit envmaps/envmap{6,12}/sg_128.npy (un-normalised lobes, |lambda| up to 505), .envmap := a decoded background map -- on chunk 1
    with recorded draws.  Here: EnvmapMaterialNetwork.load_light(path) itself (sg_128.npy + <path>.exr through robir_amd.exr), then
    forward() -- against that golden (device-built octree: the end-to-end bound) and against the oracle on the SAME octree cells (chained
    bound).  bg_rgb = the reference's render_envmap of the map along every ray."""
    import os
    import shutil
    import tempfile
    from conftest import GOLD, oracle_tables_from_device
    from robir_amd import renderer, synth
    from robir_oracle import nets as on, renderer as orend, octree as ooct
    g = load_golden("forward_relit_%d" % env_id)
    m = renderer.build_synthetic_model(dev, seed=0, variance=0.3)
    with tempfile.TemporaryDirectory() as d:
        light = os.path.join(d, "envmap%d" % env_id)
        os.makedirs(light)
        np.save(os.path.join(light, "sg_128.npy"), g["lgtSGs"])
        shutil.copy(os.path.join(GOLD, str(g["env_fixture"])), light + ".exr")
        m.envmap_material_network.load_light(light)
    assert float(m.envmap_material_network.lgtSGs.detach()[:, 3].abs().max()) > 400.0          # the shipped fit, not the synthetic light
    c = int(g["chunk"])
    uv_d, pose_d, K_d, uv, pose, K, sl = _inputs(dev, c)
    draws = {k[5:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("draw_")}
    hdr = torch.from_numpy(g["hdr_shift"]).expand(1024, 1).contiguous()
    inp = {"uv": uv_d[None], "pose": pose_d[None], "intrinsics": K_d[None],
           "object_mask": torch.ones(1, 1024, dtype=torch.bool, device=dev), "hdr_shift": hdr.to(dev)}
    out = m(inp, trainstage="Material", train_spec=True, draws={k: v.to(dev) for k, v in draws.items()})
    out = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in out.items()}
    hit_ref = torch.from_numpy(g["out_network_object_mask"])
    assert int((out["network_object_mask"] != hit_ref).sum()) <= 2
    same = hit_ref == out["network_object_mask"]
    tag = "forward_relit_%d_vs_reference_golden/" % env_id
    # the background is a function of the ray directions alone: every ray, hit or not
    assert rel_err(out["ray_dirs"], g["out_ray_dirs"]) <= 1e-6
    assert rel_err(out["bg_rgb"], g["out_bg_rgb"]) <= 1e-4, rel_err(out["bg_rgb"], g["out_bg_rgb"])
    assert float(np.abs(g["out_bg_rgb"] - 1.0).max()) > 0.1                    # ... and it IS the map, not the ones pre-fill
    for k in ("points", "sdf_output"):
        bounded(tag + k, out[k][same], torch.from_numpy(g["out_" + k])[same], 1e-4, 0.002)
    # sharp shipped lights: the specular SG terms are ill-conditioned in fp32 (test_sg_gpu.py::test_specular_term_conditioning), so 1e-6
    # differences of the hit positions show at the 1e-3 level there, with any fp32 evaluation -- the same split of the fields as
    # test_forward_material_second_weight_set; each comparison additionally holds a recorded cap (conftest.bounded)
    loose = ("sg_specular_rgb", "sg_rgb", "indir_specular_rgb")
    for k in FIELDS:
        bounded(tag + k, out[k][same], torch.from_numpy(g["out_" + k])[same], 2e-3, 0.1 if k in loose else 0.003)
    # same octree cells, same draws: the oracle with the loaded light
    sd = on.as_torch(synth.synth_state_dict(0, variance=0.3))
    sd["envmap_material_network.lgtSGs"] = torch.from_numpy(g["lgtSGs"])
    T = oracle_tables_from_device(m.ray_tracer.sdf_octree.tables)
    uv_t, pose_t, K_t = torch.from_numpy(uv[sl])[None], torch.from_numpy(pose)[None], torch.from_numpy(K)[None]
    dirs, cam = orend.camera_rays(uv_t, pose_t, K_t)
    _, hit, _ = ooct.trace(T, cam, dirs, -1)
    # The recorded draws are sized by the REFERENCE's hit count.  Should this octree (device-built) disagree by a ray, the per-hit draws are
    # re-sized (recorded rows kept, rows of extra hits drawn from robir_amd.synth) and the kernel forward is repeated with them, so that this
    # leg ALWAYS runs (it used to be skipped silently on a one-ray difference).
    n_here = int(hit.sum())
    if n_here != int(g["n_hit"]):
        extra = synth.pbr_draws(1234, n_here, chunk_id=c)
        draws = {k: (torch.cat([v[:n_here], torch.from_numpy(extra[k])[v.shape[0]:n_here]]) if v.shape[0] == int(g["n_hit"]) and k.split("_")[0] in ("illum", "spec", "normal", "svis") else v)
                 for k, v in draws.items()}
        out = m(inp, trainstage="Material", train_spec=True, draws={k: v.to(dev) for k, v in draws.items()})
        out = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in out.items()}
    env = m.envmap_material_network.envmap.cpu()
    ref = orend.forward(sd, T, uv_t, pose_t, K_t, torch.ones(1, 1024, dtype=torch.bool), hdr, draws, "Material", testing=True, envmap=env)
    assert bool((out["network_object_mask"] == ref["network_object_mask"]).all())
    assert rel_err(out["bg_rgb"], ref["bg_rgb"]) <= 1e-4
    for k in FIELDS:
        frac, worst = bad_frac(out[k], ref[k], 2e-4), rel_err(out[k], ref[k])
        record_metric("forward_relit_%d_vs_oracle_same_cells/%s" % (env_id, k), frac_gt_2e4=frac, max=worst)
        if k in loose:
            assert frac <= 0.1 and worst <= 5e-3, (k, frac, worst)
        else:
            assert frac <= 0.005 and worst <= 1e-3, (k, frac, worst)
    # the cull attribution under the shipped (sharp, un-normalised) light: sg_rgb joins the ill-conditioned specular fields here (see
    # `loose`), every other cull-exact field holds 1e-4 on the points off the cull
    # ... and the DIFFUSE light term is a sum of exp(lambda (n.l - 1)) lobes: the chain's difference of the shading normal (the normal
    # auto-encoder's output, ~1e-5 between two fp32 evaluations) is amplified by lambda -- up to 505 in the shipped fits -- before any
    # arithmetic of the SG stage happens (on IDENTICAL stage inputs the term holds 1e-4: tests/test_sg_gpu.py).  Its bound here is therefore
    # max(1e-4, lambda_max x the measured normal difference), printed with the worst entry.
    lam_max = float(sd["envmap_material_network.lgtSGs"][:, 3].abs().max())
    dn = rel_err(out["normal_map"][ref["network_object_mask"]], ref["normal_map"][ref["network_object_mask"]])
    bad = _assert_unmarked_within_1e4(out, ref, sd["envmap_material_network.lgtSGs"], draws, "relit_%d" % env_id, max_marked_frac=0.02,
                                      skip=("sg_rgb",), tol_for={"sg_diffuse_rgb": max(1e-4, lam_max * dn)})
    assert bad == 0


def test_exact_and_split_precision_forward_agree(dev, model, monkeypatch):
    """The three arithmetics of the MLP layers on a whole Material forward with the same random draws: the DEFAULT policy (exact three-piece
    f16 operands, "f16x6": not narrower than fp32) and the split-precision policy ((hi, lo) f16 pairs, 22-bit operands; legacy library)
    against the f32-input-MFMA kernels.  Every stage is the same fp32 computation to summation order (exact) / to ~2^-22 (split), so the
    images differ only by the chained-stage noise any two fp32 evaluations show (DESIGN 'Parity tolerances'); the default sits closer."""
    from robir_amd import sg_render, synth
    uv_d, pose_d, K_d, *_ = _inputs(dev, 1)
    inp = {"uv": uv_d[None], "pose": pose_d[None], "intrinsics": K_d[None],
           "object_mask": torch.ones(1, 1024, dtype=torch.bool, device=dev), "hdr_shift": torch.full((1024, 1), 0.5, device=dev)}
    probe = model(inp, trainstage="Material", train_spec=True)
    n_hit = int(probe["network_object_mask"].sum())
    draws = {k: torch.from_numpy(v).to(dev) for k, v in synth.pbr_draws(0, n_hit, chunk_id=1).items()}
    outs = {}
    for mode, vis in (("f16x6", "f16x6"), ("f16x3", "f16x3-auto"), ("fp32", "fp32")):
        monkeypatch.setenv("ROBIR_MLP_PRECISION", mode)
        monkeypatch.setattr(sg_render, "VIS_PRECISION", vis)
        outs[mode] = model(inp, trainstage="Material", train_spec=True, draws=draws)
    b = outs["fp32"]
    assert n_hit > 300
    for mode in ("f16x6", "f16x3"):
        a = outs[mode]
        assert bool((a["network_object_mask"] == b["network_object_mask"]).all()), mode
        assert rel_err(a["points"].cpu(), b["points"].cpu()) <= 1e-6
        for k in FIELDS:
            frac, worst = bad_frac(a[k].cpu(), b[k].cpu(), 1e-4), rel_err(a[k].cpu(), b[k].cpu())
            record_metric(f"forward_{mode}_vs_fp32_mfma/{k}", frac_gt_1e4=frac, max=worst)
            assert frac <= 0.005, (mode, k, frac)
            assert worst <= 1e-3, (mode, k, worst)


@pytest.mark.gpu
def test_forward_fun_spec_and_tex_uv(dev, model):
    """forward(fun_spec=True) returns the two specular terms as functions of a per-pixel roughness (implicit_differentiable_renderer.py:
    417-427,455-463); input['tex_uv'] reaches the hook as its hit rows (:390-392,408)."""
    from robir_amd import synth
    uv, pose, K = synth.synth_camera(64, 64)
    sl = slice(1024, 2048)
    inp = {"uv": torch.from_numpy(uv[sl])[None].to(dev), "pose": torch.from_numpy(pose)[None].to(dev),
           "intrinsics": torch.from_numpy(K)[None].to(dev), "object_mask": torch.ones(1, 1024, dtype=torch.bool, device=dev),
           "hdr_shift": torch.full((1024, 1), 0.5, device=dev)}
    pre = model(inp, trainstage="Illum", draws={"illum_randn": None, "normal_randn": None})
    hit = pre["network_object_mask"]
    nhit = int(hit.sum())
    assert nhit > 100
    dr = {k: torch.from_numpy(v).to(dev) for k, v in synth.pbr_draws(0, nhit, chunk_id=1).items()}
    plain = model(inp, trainstage="Material", train_spec=True, draws=dr)
    fun = model(inp, trainstage="Material", train_spec=True, fun_spec=True, draws=dr)
    assert callable(fun["sg_specular_rgb"]) and callable(fun["indir_specular_rgb"])
    rough = plain["roughness"][:, :1].contiguous()
    for k, tag in (("sg_specular_rgb", "dir"), ("indir_specular_rgb", "ind")):
        # the closure re-draws the specular cone's samples: hand it the draws the plain call used for this term
        got = fun[k](rough, draws={"svis_theta": dr["svis_theta_" + tag], "svis_phi": dr["svis_phi_" + tag]})
        assert got.shape == plain[k].shape
        assert rel_err(got.cpu(), plain[k].cpu()) <= 1e-6, k
        assert bool((got[~hit] == 1.0).all())                      # non-hit rows keep the buffer's ones
    assert rel_err(fun["sg_diffuse_rgb"].cpu(), plain["sg_diffuse_rgb"].cpu()) <= 1e-6
    # tex_uv: a runner-style hook sees the hit rows of the per-pixel coordinates
    seen = {}
    orig = model.get_sg_render

    def hook(*a, **k):
        seen["tex_uv"] = k.get("tex_uv")
        return orig(*a, **k)
    hook.robir_native = True
    tex = torch.rand(1, 1024, 2, device=dev)
    model.get_sg_render = hook
    try:
        model(dict(inp, tex_uv=tex), trainstage="Material", train_spec=True, draws=dr)
        assert seen["tex_uv"] is not None and torch.equal(seen["tex_uv"], tex[:, hit])
        model(inp, trainstage="Material", train_spec=True, draws=dr)
        assert seen["tex_uv"] is None                                # nothing lingers from the previous call
        with pytest.raises(ValueError):
            model(dict(inp, tex_uv=tex[:, :100]), trainstage="Material", train_spec=True, draws=dr)
    finally:
        del model.get_sg_render
