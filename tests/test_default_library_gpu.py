"""The default precision policy runs on the DEFAULT library alone (round 5: the split-precision family and the other retired kernel
generations moved into librobir_hip_legacy.so, include/robir_hip_legacy.h).  Every call shape of the hot path -- forward('Material') with
the PBR hook and the CESR hook, forward('Illum') + trace_radiance, render_chunks, render_neus (both sampling modes), borrow_color,
get_neus_surface, the traced-visibility mode -- with robir_amd._lib.legacy() made to fail."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_default_policy_never_touches_the_legacy_library(monkeypatch):
    from robir_amd import _lib, nets, ops, precision, renderer, sdf_render, sg_render, synth
    from robir_amd.octree_tracing import OctreeVisModel
    if precision.policy() != "exact" or precision.mlp_precision() != "f16x6":
        pytest.skip("this run selects another precision policy")

    def refuse():
        raise AssertionError("the default policy asked for the legacy library")

    monkeypatch.setattr(_lib, "legacy", refuse)
    monkeypatch.setattr(_lib, "legacy_loaded", lambda: False)      # (an earlier test of this process may have loaded it: range_check would then read its sentinels too)
    monkeypatch.setattr(sg_render, "VIS_PRECISION", "f16x6")
    dev = torch.device("cuda:0")
    m = renderer.build_synthetic_model(dev, seed=0, variance=0.3)            # octree build included
    uv, pose, K = synth.synth_camera(64, 64)
    uv_d, pose_d, K_d = torch.from_numpy(uv).to(dev), torch.from_numpy(pose).to(dev), torch.from_numpy(K).to(dev)
    hdr = torch.full((4096, 1), 0.5, device=dev)
    inp = {"uv": uv_d[None, 1024:2048], "pose": pose_d[None], "intrinsics": K_d[None],
           "object_mask": torch.ones(1, 1024, dtype=torch.bool, device=dev), "hdr_shift": hdr[:1024]}
    m.deferred_chunks = 0
    out = m(inp, trainstage="Material", train_spec=True)
    assert bool(torch.isfinite(out["sg_rgb"][out["network_object_mask"]]).all())
    ill = m(inp, trainstage="Illum")
    tr = m.trace_radiance(ill, nsamp=8)
    assert tr["trace_radiance"].shape == (1024, 8, 3)
    whole = m.render_chunks(uv_d, pose_d, K_d, hdr, chunk=1024)
    assert whole["sg_rgb"].shape == (4096, 3)
    m.__dict__.pop("deferred_chunks", None)
    lazy = m(inp, trainstage="Material", train_spec=True)                    # recorded chunk + recorded trace
    trl = m.trace_radiance(lazy, nsamp=8)
    assert bool(torch.isfinite(trl["gt_integral"]).all().cpu())
    # CESR hook
    c = synth.synth_cesr_nets(0)
    shadow, normal = nets.SDFNetwork(63 + 128, 2, 512, 8, [4], 0), nets.SDFNetwork(63, 3, 512, 8, [4], 0)
    shadow.load_state_dict({k: torch.from_numpy(v) for k, v in c["shadow_net"].items()})
    normal.load_state_dict({k: torch.from_numpy(v) for k, v in c["normal_net"].items()})
    m.get_sg_render = renderer.CESRHook(m, shadow.to(dev).eval(), normal.to(dev).eval(), is_training=False, cur_iter=100000, prefit="explore")
    try:
        oc = m(inp, trainstage="Material", lin_diff=True, train_spec=True)
        assert bool(torch.isfinite(oc["sg_rgb"][oc["network_object_mask"]]).all().cpu())
    finally:
        m.__dict__.pop("get_sg_render", None)
    # traced visibility as the VisModel
    mlp_vis = m.visibility_network
    m.visibility_network = OctreeVisModel(m.octree_ray_tracer)
    try:
        ov = m.render_chunks(uv_d[1024:2048], pose_d, K_d, hdr[:1024], chunk=1024)
        assert bool(torch.isfinite(ov["vis_shadow"][ov["network_object_mask"]]).all())
    finally:
        m.visibility_network = mlp_vis
    # NeuS ray-march, both sampling modes; borrow_color; get_neus_surface
    neus = m.implicit_network.neus_model
    dirs = ops.camera_rays(pose_d, K_d, uv_d[1024 + 16 * 64:1024 + 16 * 64 + 48].contiguous())
    o = (pose_d[:3, 3] * 2.0).expand(48, 3).contiguous()
    rays = sdf_render.Rays(o, dirs, dirs, None, None, torch.full((48, 1), 0.8, device=dev), torch.full((48, 1), 2.8, device=dev))
    for kw in (dict(is_eval=True), dict()):
        r = sdf_render.render_neus(rays, neus, 1.0, n_samples=32, n_importance=32, n_outside=0, up_sample_steps=2, **kw)
        assert bool(torch.isfinite(r["rgb"]).all())
    hit = out["network_object_mask"]
    col = m.implicit_network.batch_borrow_color(out["points"][hit][:64].contiguous(), (-out["ray_dirs"][hit][:64]).contiguous())
    assert col.shape == (64, 3)
    x, n_, ge = sdf_render.get_neus_surface(m.implicit_network, out["points"][hit][:64].contiguous(), out["ray_dirs"][hit][:64].contiguous(),
                                           out["normals"][hit][:64].contiguous())
    assert x.shape == (64, 3) and bool(torch.isfinite(ge))
    ops.range_check(sync=True)


def test_f16_vis_policy_runs_on_the_default_library_alone(monkeypatch):
    """ROBIR_PRECISION=f16-vis (round 4's meaning of `f16`, kept selectable: plain f16 in the light-visibility MLP, every other net on the
    exact-operand kernels) needs no legacy library; with ROBIR_CESR_PRECISION=f16x1 on top the CESR hook runs its one-product kernel --
    still on the default library.  A labelled NARROWER mode: only sanity bands here (the light visibility within 2e-2 of the exact render)."""
    from robir_amd import _lib, nets, precision, renderer, sg_render, synth

    def refuse():
        raise AssertionError("the f16-vis policy asked for the legacy library")

    monkeypatch.setattr(_lib, "legacy", refuse)
    monkeypatch.setattr(_lib, "legacy_loaded", lambda: False)      # (an earlier test of this process may have loaded it: range_check would then read its sentinels too)
    dev = torch.device("cuda:0")
    m = renderer.build_synthetic_model(dev, seed=0, variance=0.3)
    m.deferred_chunks = 0
    uv, pose, K = synth.synth_camera(64, 64)
    inp = {"uv": torch.from_numpy(uv).to(dev)[None, 1024:2048], "pose": torch.from_numpy(pose).to(dev)[None],
           "intrinsics": torch.from_numpy(K).to(dev)[None], "object_mask": torch.ones(1, 1024, dtype=torch.bool, device=dev),
           "hdr_shift": torch.full((1024, 1), 0.5, device=dev)}
    monkeypatch.delenv("ROBIR_MLP_PRECISION", raising=False)
    monkeypatch.delenv("ROBIR_VIS_PRECISION", raising=False)
    monkeypatch.delenv("ROBIR_CESR_PRECISION", raising=False)
    monkeypatch.setenv("ROBIR_PRECISION", "exact")
    monkeypatch.setattr(sg_render, "VIS_PRECISION", "f16x6")
    hit = m(inp, trainstage="Material", train_spec=True)["network_object_mask"]          # sizes the recorded draws
    draws = {k: torch.from_numpy(v).to(dev) for k, v in synth.pbr_draws(0, int(hit.sum()), chunk_id=1).items()}
    ref = m(inp, trainstage="Material", train_spec=True, draws=draws)
    monkeypatch.setenv("ROBIR_PRECISION", "f16-vis")
    assert (precision.vis_precision(), precision.mlp_precision(), precision.cesr_precision()) == ("f16x1", "f16x6", "f16x6")
    monkeypatch.setattr(sg_render, "VIS_PRECISION", precision.vis_precision())
    out = m(inp, trainstage="Material", train_spec=True, draws=draws)
    torch.cuda.synchronize()
    assert bool((out["network_object_mask"] == hit).all())
    assert torch.equal(out["normal_map"], ref["normal_map"]) and torch.equal(out["diffuse_albedo"], ref["diffuse_albedo"])      # the exact-operand nets ran
    dv = (out["vis_shadow"][hit] - ref["vis_shadow"][hit]).abs()
    assert 0.0 < float(dv.max()) <= 2e-2, float(dv.max())                                                             # the plain-f16 visibility kernel ran
    # the CESR nets' one-product kernel under the same policy
    monkeypatch.setenv("ROBIR_CESR_PRECISION", "f16x1")
    c = synth.synth_cesr_nets(0)
    shadow, normal = nets.SDFNetwork(63 + 128, 2, 512, 8, [4], 0), nets.SDFNetwork(63, 3, 512, 8, [4], 0)
    shadow.load_state_dict({k: torch.from_numpy(v) for k, v in c["shadow_net"].items()})
    normal.load_state_dict({k: torch.from_numpy(v) for k, v in c["normal_net"].items()})
    m.get_sg_render = renderer.CESRHook(m, shadow.to(dev).eval(), normal.to(dev).eval(), is_training=False, cur_iter=100000, prefit="explore")
    try:
        oc = m(inp, trainstage="Material", lin_diff=True, train_spec=True)
        assert bool(torch.isfinite(oc["sg_rgb"][oc["network_object_mask"]]).all().cpu())
    finally:
        m.__dict__.pop("get_sg_render", None)
