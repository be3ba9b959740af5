"""OctreeVisModel (model/octree_tracing.py:63-85) as a real mode: traced visibility instead of the visibility MLP, the switch
the reference's runners call `trace_vis` (training/train_pbr.py:409-410).  Against the reference's own outputs
(tests/golden/octree_vis.npz, recorded by oracle/gen_golden_r2.py), the oracle, and the generic pair-materialising path."""
import numpy as np
import pytest
import torch

from conftest import rel_err, bad_frac, load_golden, bounded, record_metric

pytestmark = pytest.mark.gpu

KEYS = ("sg_rgb", "sg_specular_rgb", "sg_diffuse_rgb", "vis_shadow", "indir_rgb", "indir_diffuse_rgb", "indir_specular_rgb")


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def tracer(dev, oracle_octree):
    """Secondary tracer on the octree the oracle built (the reference's build differs from it in one hit cell of 233 k)."""
    from robir_amd.octree_tracing import OctreeTracing, OctreeSDF, OctreeVisModel
    tr = OctreeTracing(max_iter=32)
    tr.sdf_octree = OctreeSDF.from_host_tables(oracle_octree, dev, 32)
    return OctreeVisModel(tr)


def test_direct_logits_vs_reference(dev, tracer):
    g = load_golden("octree_vis")
    p, d = torch.from_numpy(g["direct_points"]).to(dev), torch.from_numpy(g["direct_dirs"]).to(dev)
    lg = tracer(p, d).cpu()
    ref = torch.from_numpy(g["direct_logits"])
    assert lg.shape == (512, 2) and torch.equal(lg.sum(-1), torch.ones(512))
    assert int((lg != ref).any(-1).sum()) <= 2                      # the one differing hit cell / threshold rays
    assert 100 < int(ref[:, 0].sum()) < 400


def test_render_with_traced_visibility_vs_reference(dev, tracer):
    """render_with_all_sg(VisModel=OctreeVisModel) on 64 surface points: 131 k culled pairs = ONE lock-step batch beyond 100 000
    rays (step 0.01), then two 8-sample BRDF-lobe batches, against the reference's output."""
    from robir_amd import sg_render
    g = load_golden("octree_vis")
    t = {k: torch.from_numpy(v).to(dev) for k, v in g.items() if v.dtype.kind == "f"}
    draws = {k[5:]: t[k] for k in t if k.startswith("draw_")}
    stats = {}
    out = sg_render.render_with_all_sg(t["points"], t["normal"], t["view"], t["lgtSGs"], t["f0"], t["roughness"], t["albedo"],
                                       indir_integral=t["indir_int"], indir_lgtSGs=t["indir_sgs"], VisModel=tracer,
                                       testing=True, draws=draws, stats=stats)
    assert int(stats["diffuse_vis_evals"]) == int(g["cast_sizes"][0])          # same cull: same number of traced pairs
    for k in KEYS:
        # a ray that flips (threshold cell) moves one of 32 samples of one lobe of one point
        bounded("octree_vis_vs_reference/" + k, out[k].cpu(), g["out_" + k], 1e-5, 0.05)


def test_fused_equals_pair_materialising_path_and_oracle(dev, tracer, oracle_octree, monkeypatch):
    """Fused cull + grouped cast == the generic path that builds the (point, direction) pair tensors and calls the VisModel
    in batches like the reference -- the same hits, visibilities equal to summation order --, also when the batch size makes several lock-step groups per chunk (sizes on
    both sides of the 100 000-ray step switch) and with three chunks in one call; and both follow the oracle."""
    from robir_amd import sg_render, synth
    from robir_oracle import octree as ooct, sg as osg
    g = load_golden("octree_vis")
    gen = torch.Generator().manual_seed(3)
    pts_all, nrm_all = torch.from_numpy(g["points"]), torch.from_numpy(g["normal"])
    lgt = torch.from_numpy(synth.synth_light_sgs(0, 128))
    for batch, C in ((2000000, 1), (30000, 1), (120000, 3)):
        monkeypatch.setattr(sg_render, "OCTREE_VIS_BATCH", batch)
        n = 60
        pts, nrm = pts_all[:n].to(dev), nrm_all[:n].to(dev)
        cid = (torch.arange(n) * C // n).to(torch.int32).to(dev) if C > 1 else None
        u = torch.rand(2, C, 128, 32, generator=gen)
        stats = {}
        fused = sg_render._diffuse_vis_core(pts, nrm, tracer, lgt.to(dev), u[0].to(dev), u[1].to(dev), 1.0, False, cid, C, stats)
        dirs, wdir, wsum = __import__("robir_amd").ops.dvis_dirs(lgt.to(dev), u[0].to(dev), u[1].to(dev), 1.0)
        generic = []
        for c in range(C):                  # the generic path is one get_diffuse_visibility call per chunk, like the reference
            sel = (cid == c).nonzero()[:, 0] if C > 1 else torch.arange(n, device=dev)
            generic.append(sg_render._diffuse_vis_generic(pts[sel], nrm[sel], tracer, dirs[c * 4096:(c + 1) * 4096],
                                                          wdir[c * 4096:(c + 1) * 4096], wsum[c * 128:(c + 1) * 128], None, 1,
                                                          128, 32, False))
        generic = torch.cat(generic)
        # same hits (a flipped ray would move an entry by ~1e-2); the per-lobe weighted sums differ in summation order only
        assert rel_err(fused.cpu(), generic.cpu()) <= 1e-6, (batch, C, float((fused - generic).abs().max()))
        assert int(stats["diffuse_vis_evals"]) > 50 * 1000
        lobe = lgt[:, :3] / (lgt[:, :3].norm(dim=-1, keepdim=True) + 1e-6)
        ref = []
        for c in range(C):
            sel = (cid.cpu() == c).nonzero()[:, 0] if C > 1 else torch.arange(n)
            ref.append(osg.diffuse_visibility(pts_all[:n][sel], nrm_all[:n][sel],
                                              lambda p, d: ooct.octree_vis_logits(oracle_octree, p, d), lobe, lgt[:, 3:4].abs(),
                                              u[0, c], u[1, c], batch=batch).t())
        ref = torch.cat(ref)
        bounded("octree_vis_vs_oracle/batch%d_C%d" % (batch, C), fused.cpu(), ref, 1e-6, 0.002)


def test_grouped_cast_equals_separate_casts(dev, tracer):
    g = load_golden("octree_vis")
    p, d = torch.from_numpy(g["direct_points"]).to(dev), torch.from_numpy(g["direct_dirs"]).to(dev)
    p, d = p.repeat(5, 1), d.repeat(5, 1).flip(0).contiguous()
    cuts = [0, 300, 1500, 1500, 2560]                                            # an empty group among them
    gs = torch.tensor(cuts, dtype=torch.int64, device=dev)
    lg = tracer.forward_groups(p, d, gs)
    for a, b in zip(cuts[:-1], cuts[1:]):
        if b > a:
            assert torch.equal(lg[a:b], tracer(p[a:b].contiguous(), d[a:b].contiguous())), (a, b)


def test_traced_visibility_through_the_renderer(dev):
    """trace_vis on the model: visibility_network = OctreeVisModel(octree_ray_tracer) (train_pbr.py:409-410).  The batched
    multi-chunk render equals chunk-by-chunk forward() bit for bit (every chunk keeps its own lock-step groups)."""
    from robir_amd import renderer, synth
    from robir_amd.octree_tracing import OctreeVisModel
    m = renderer.build_synthetic_model(dev, seed=0, variance=0.3)
    m.visibility_network = OctreeVisModel(m.octree_ray_tracer)
    uv, pose, K = synth.synth_camera(64, 64)
    pose_d, K_d = torch.from_numpy(pose).to(dev), torch.from_numpy(K).to(dev)
    uv_d = torch.from_numpy(uv[:3072]).to(dev)
    hdr = torch.full((3072, 1), 0.5, device=dev)
    hit = m.render_chunks(uv_d, pose_d, K_d, hdr, trainstage="Illum", draws={})["network_object_mask"].cpu()
    counts = [int(hit[i * 1024:(i + 1) * 1024].sum()) for i in range(3)]
    per = [synth.pbr_draws(0, counts[i], chunk_id=i) for i in range(3)]
    cat = {k: torch.from_numpy(np.concatenate([q[k] for q in per])).to(dev) for k in per[0] if not k.startswith("dvis")}
    for k in ("dvis_theta", "dvis_phi"):
        cat[k] = torch.from_numpy(np.stack([q[k] for q in per])).to(dev)
    stats = {}
    big = m.render_chunks(uv_d, pose_d, K_d, hdr, draws=cat, stats=stats)
    assert int(stats["diffuse_vis_evals"]) > 1000000
    for i in range(3):
        sl = slice(i * 1024, (i + 1) * 1024)
        one = m({"uv": uv_d[None, sl], "pose": pose_d[None], "intrinsics": K_d[None],
                 "object_mask": torch.ones(1, 1024, dtype=torch.bool, device=dev), "hdr_shift": hdr[sl]},
                trainstage="Material", train_spec=True, draws={k: torch.from_numpy(v).to(dev) for k, v in per[i].items()})
        for k in KEYS:
            assert rel_err(big[k][sl].cpu(), one[k].cpu()) == 0.0, (i, k)
    # traced visibility really differs from the MLP's (an untrained network here): not the same image
    m2 = renderer.build_synthetic_model(dev, seed=0, variance=0.3, build_octrees=False)
    m2.ray_tracer.sdf_octree, m2.octree_ray_tracer.sdf_octree = m.ray_tracer.sdf_octree, m.octree_ray_tracer.sdf_octree
    mlp = m2.render_chunks(uv_d, pose_d, K_d, hdr, draws=cat)
    hitm = big["network_object_mask"]
    assert torch.equal(hitm, mlp["network_object_mask"])
    assert float((big["vis_shadow"][hitm] - mlp["vis_shadow"][hitm]).abs().mean()) > 1e-2
    assert rel_err(big["diffuse_albedo"].cpu(), mlp["diffuse_albedo"].cpu()) == 0.0


def test_compacted_iterations_and_chunk_groups_change_nothing(dev, tracer, monkeypatch):
    """Round 3: the lock-step iterations walk a stably compacted list of the rays still active (csrc/octree_vis.hip,
    rb_dvis_octree_compact) and a view's chunks are traced in groups that bound the scratch.  Neither may move a bit: the plain
    walk over all pairs, the compacted walk, and the compacted walk with the seven chunks cut into groups of 1 / 2 / 3 chunks."""
    from robir_amd import ops, sg_render, synth
    g = load_golden("octree_vis")
    gen = torch.Generator().manual_seed(17)
    C, n = 7, 64
    pts, nrm = torch.from_numpy(g["points"])[:n].to(dev), torch.from_numpy(g["normal"])[:n].to(dev)
    cid = (torch.arange(n) * C // n).to(torch.int32).to(dev)
    lgt = torch.from_numpy(synth.synth_light_sgs(0, 128)).to(dev)
    u = torch.rand(2, C, 128, 32, generator=gen).to(dev)
    outs = {}
    for name, compact, per_call in (("plain", False, 48), ("compact", True, 48), ("groups1", True, 1), ("groups2", True, 2), ("groups3", False, 3)):
        monkeypatch.setattr(ops, "OVIS_COMPACT", compact)
        monkeypatch.setattr(ops, "OVIS_CHUNKS_PER_CALL", per_call)
        stats = {}
        outs[name] = (sg_render._diffuse_vis_core(pts, nrm, tracer, lgt, u[0], u[1], 1.0, False, cid, C, stats), int(stats["diffuse_vis_evals"]),
                      [int(v) for v in ops.LAST_OCTREE_VIS_LAYOUT.cpu()])
    ref, evals, lay = outs["plain"]
    assert evals > 100000 and bool(torch.isfinite(ref).all()) and 0.0 < float(ref.mean()) < 1.0
    for name, (v, e, l) in outs.items():
        assert e == evals and torch.equal(v, ref), name
        assert l[0] == lay[0] and l[2] == lay[2] and l[3] == lay[3], (name, l, lay)       # pairs, node records read, ray-iterations
