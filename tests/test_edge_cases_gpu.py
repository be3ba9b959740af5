"""Edge cases on the GPU: empty inputs, all-miss chunks, ragged batches, degenerate rays (NaN propagation like the
reference's IEEE 0*inf in intersect_box)."""
import numpy as np
import pytest
import torch

from conftest import rel_err, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def model(dev):
    from robir_amd import renderer
    return renderer.build_synthetic_model(dev)


def test_empty_inputs(dev, model):
    from robir_amd import ops, sg_render
    z3 = torch.zeros(0, 3, device=dev)
    assert model.implicit_network(z3).numel() == 0
    assert model.implicit_network.gradient(z3).numel() == 0
    assert model.visibility_network(z3, z3).shape == (0, 2)
    assert model.implicit_network.batch_borrow_color(z3, z3).shape == (0, 3)
    x, hit, t = model.ray_tracer.sdf_octree.cast_full(z3, z3)
    assert x.shape == (0, 3) and hit.numel() == 0


def test_all_miss_chunk_and_ragged_batch(dev, model):
    """Rays looking away from the object: no hits -> every per-ray output keeps its prefill of 1.0; a batch whose size
    is not a multiple of the chunk (2500 rays, chunk 1024) renders like its three chunks rendered separately."""
    from robir_amd import synth
    uv, pose, K = synth.synth_camera(64, 64)
    pose_away = pose.copy()
    pose_away[:3, :3] = np.diag([1, 1, -1]).astype(np.float32) @ pose[:3, :3]      # look along +z, away from the origin
    hdr = torch.full((1024, 1), 0.5, device=dev)
    out = model.render_chunks(torch.from_numpy(uv[:1024]).to(dev), torch.from_numpy(pose_away).to(dev),
                              torch.from_numpy(K).to(dev), hdr)
    assert int(out["network_object_mask"].sum()) == 0
    for k in ("sg_rgb", "indir_rgb", "diffuse_albedo", "roughness", "metallic", "vis_shadow"):
        assert float(out[k].min()) == 1.0 and float(out[k].max()) == 1.0, k
    # ragged: 2500 rays
    uv_d = torch.from_numpy(uv[1024:3524]).to(dev)
    pose_d, K_d = torch.from_numpy(pose).to(dev), torch.from_numpy(K).to(dev)
    hdr = torch.full((2500, 1), 0.5, device=dev)
    big = model.render_chunks(uv_d, pose_d, K_d, hdr, trainstage="Illum", draws={})
    for i, (a, b) in enumerate(((0, 1024), (1024, 2048), (2048, 2500))):
        one = model.render_chunks(uv_d[a:b], pose_d, K_d, hdr[a:b], trainstage="Illum", draws={})
        assert bool((big["network_object_mask"][a:b] == one["network_object_mask"]).all()), i
        assert rel_err(big["points"][a:b].cpu(), one["points"].cpu()) <= 1e-6


def test_axis_parallel_rays_propagate_nan_like_reference(dev, oracle_octree):
    """d.x = 0 and o.x on a cell boundary gives 0*inf = NaN in the slab test; torch.minimum/maximum propagate it and
    the reference reports such rays as misses with NaN distance.  The HIP tracer must do the same (golden chunk 1 of the
    64x64 view contains the x = 32 pixel column)."""
    from robir_amd.octree_tracing import OctreeSDF
    from robir_oracle import octree as ooct
    od = OctreeSDF.from_host_tables(oracle_octree, dev, -1)
    o = torch.tensor([[0.0, 0.0, 0.9]]).expand(64, 3).contiguous()
    d = torch.zeros(64, 3)
    d[:, 1] = torch.linspace(-0.3, 0.3, 64)
    d[:, 2] = -1.0
    d = d / d.norm(dim=-1, keepdim=True)                # d.x == 0 exactly
    t_ref, hit_ref = ooct.cast(oracle_octree, o, d, -1)
    x, hit, t = od.cast_full(o.to(dev), d.to(dev))
    assert bool((hit.cpu() == hit_ref).all())
    assert bool((torch.isnan(t.cpu()) == torch.isnan(t_ref)).all())
    assert rel_err(t.cpu(), t_ref) <= 1e-6


def test_secondary_all_miss_and_octree_vis_model(dev, model):
    from robir_amd.octree_tracing import OctreeVisModel, OctreeTracing
    tr = OctreeTracing(max_iter=32)
    tr.sdf_octree = type(model.ray_tracer.sdf_octree)(model.ray_tracer.sdf_octree.tables, 32)
    vm = OctreeVisModel(tr)
    p = torch.tensor([[0.0, 0.0, 0.6]], device=dev).expand(100, 3).contiguous()
    d = torch.tensor([[0.0, 0.1, 1.0]], device=dev).expand(100, 3).contiguous()
    d = d / d.norm(dim=-1, keepdim=True)
    v = vm(p, d)                                        # pointing away: nothing hit -> [0, 1]
    assert v.shape == (100, 2) and float(v[:, 0].max()) == 0.0 and float(v[:, 1].min()) == 1.0
    # towards the object, off the cell-boundary planes (an exactly axis-parallel ray through x = y = 0 is the NaN case above)
    p2 = torch.tensor([[0.013, 0.021, 0.6]], device=dev).expand(100, 3).contiguous()
    d2 = -p2 / p2.norm(dim=-1, keepdim=True)
    v2 = vm(p2, d2.contiguous())
    assert float(v2[:, 0].min()) == 1.0 and float(v2[:, 1].max()) == 0.0


def test_oversized_launch_fails_loudly(dev):
    """A dispatch holds its grid size in work-items as a 32-bit number; a request beyond that must come back as an error,
    not as a grid whose tail never runs (that is how a 20 M-point colour batch went wrong before the guard)."""
    import ctypes
    from robir_amd import _lib
    x = torch.zeros(16, 3, device=dev)
    shift = torch.full((1, 1), 0.5, device=dev)
    y = torch.zeros(16, 3, device=dev)
    with pytest.raises(_lib.RobirHipError):
        _lib.call("rb_tonemap", _lib.ptr(x), ctypes.c_long(1 << 40), _lib.ptr(shift), ctypes.c_int(0), ctypes.c_int(0),
                  _lib.ptr(y), _lib.stream_ptr())
    torch.cuda.synchronize()
    # the library is usable afterwards
    from robir_amd import ops
    assert ops.tonemap(x + 0.25, shift, 0).shape == (16, 3)


def test_split_precision_packing_rejects_out_of_range_weights(dev):
    """|w| * 2^8 must fit the f16 hi half; such a checkpoint has to use the exact-fp32 kernels."""
    from robir_amd import packing
    W = torch.zeros(16, 32)
    W[3, 5] = 300.0
    with pytest.raises(ValueError):
        packing.pack_layers_h3([dict(W=W, b=None, n_pad=16, k_pad=32)], dev)
    W[3, 5] = 200.0
    assert packing.pack_layers_h3([dict(W=W, b=None, n_pad=16, k_pad=32)], dev).numel() > 0


def test_visibility_kernel_size_limits(dev):
    """The fused light-visibility kernels keep a point's directions in LDS: L * nsamp <= 4096 (128 lobes x 32 samples is
    the reference's maximum, sg_render.py:389); beyond that the call must fail, not truncate."""
    from robir_amd import _lib, ops, renderer, sg_render
    m = renderer.build_synthetic_model(dev, build_octrees=False)
    lgt = m.envmap_material_network.lgtSGs.detach()
    n = 5
    g = torch.Generator(device=dev).manual_seed(0)
    pts = (torch.rand(n, 3, device=dev, generator=g) - 0.5) * 0.4
    nrm = torch.nn.functional.normalize(torch.rand(n, 3, device=dev, generator=g) - 0.5, dim=-1)
    for ns in (8, 32):                                           # CESR (8) and PBR (32) sample counts
        u = torch.rand(2, 128, ns, device=dev, generator=g)
        v = sg_render._diffuse_vis_core(pts, nrm, m.visibility_network, lgt, u[0], u[1], 1.0, False, None, 1, None)
        assert v.shape == (n, 128) and float(v.min()) >= 0.0 and float(v.max()) <= 1.0 + 1e-6
    u = torch.rand(2, 128, 33, device=dev, generator=g)          # 128 * 33 > 4096
    with pytest.raises(_lib.RobirHipError):
        sg_render._diffuse_vis_core(pts, nrm, m.visibility_network, lgt, u[0], u[1], 1.0, False, None, 1, None)


def test_split_precision_kernels_ragged_row_counts(dev, synth_weights):
    """Row counts that are not multiples of the 128- / 64-row workgroup tiles (and empty inputs): the split-precision
    kernels must agree with the exact ones row for row."""
    from robir_amd import ops, packing, synth
    g = torch.Generator(device=dev).manual_seed(2)
    sdf3, sdf = packing.pack_sdf_h3(synth_weights, dev, full=False), packing.pack_sdf(synth_weights, dev, full=False)
    vis3, vis = packing.pack_vis_h3(synth_weights, dev), packing.pack_vis(synth_weights, dev)
    ill3, ill = packing.pack_illum_h3(synth_weights, dev), packing.pack_illum(synth_weights, dev)
    for M in (0, 1, 17, 129, 1000):
        x = (torch.rand(M, 3, device=dev, generator=g) - 0.5) * 0.8
        d = torch.nn.functional.normalize(torch.rand(M, 3, device=dev, generator=g) - 0.5, dim=-1) if M else x
        a, ga = ops.sdf_mlp_h3(ops.feat_pe10(x, scale=2.0, jvp=True), M, sdf3, 2, packing.H3_SCALE_LOG2, 0.5, 1.0)
        b, gb = ops.sdf_mlp(ops.feat_pe10(x, scale=2.0, jvp=True), M, sdf, 2, 0.5, 1.0)
        assert a.shape == (M,) and ga.shape == (M, 3)
        X = ops.feat_vis(x, d)
        v3, v = ops.vis_mlp_h3(X, vis3, packing.H3_SCALE_LOG2), ops.vis_mlp(X, vis)
        Xi = ops.feat_pe10(x, extra=torch.full((M, 1), 0.5, device=dev))
        i3, i = ops.wide_mlp_h3(Xi, ill3, False, packing.H3_SCALE_LOG2), ops.illum_mlp(Xi, ill)
        if M:
            assert rel_err(a.cpu(), b.cpu()) <= 1e-5 and rel_err(ga.cpu(), gb.cpu()) <= 1e-5
            assert rel_err(v3.cpu(), v.cpu()) <= 1e-5 and rel_err(i3.cpu(), i.cpu()) <= 1e-5
        else:
            assert v3.shape == (0, 2) and i3.shape == (0, 144)


def test_visibility_point_without_front_facing_directions(dev, oracle_sd):
    """A point whose normal has no direction with n.d > 1e-6 (here: a zero normal) evaluates nothing and gets
    visibility 0 on every lobe, like the reference's scatter into zeros (sg_render.py:155-183); its neighbours in the
    batch are unaffected."""
    from robir_amd import renderer, sg_render, synth
    from robir_oracle import nets as on, sg as osg
    m = renderer.build_synthetic_model(dev, build_octrees=False)
    lgt = torch.from_numpy(synth.synth_light_sgs(0, 128))
    g = torch.Generator().manual_seed(4)
    pts = (torch.rand(6, 3, generator=g) - 0.5) * 0.4
    nrm = torch.nn.functional.normalize(torch.rand(6, 3, generator=g) - 0.5, dim=-1)
    nrm[2] = 0.0
    u = torch.rand(2, 128, 32, generator=g)
    old_default = sg_render.VIS_PRECISION
    for prec in ("f16x6", "f16x3-v3", "f16x3-v2", "f16x3", "fp32"):
        sg_render.VIS_PRECISION = prec
        try:
            v = sg_render._diffuse_vis_core(pts.to(dev), nrm.to(dev), m.visibility_network, lgt.to(dev), u[0].to(dev),
                                            u[1].to(dev), 1.0, False, None, 1, None).cpu()
        finally:
            sg_render.VIS_PRECISION = old_default
        assert float(v[2].abs().max()) == 0.0, prec
        lobe = lgt[:, :3] / (lgt[:, :3].norm(dim=-1, keepdim=True) + 1e-6)
        ref = osg.diffuse_visibility(pts, nrm, lambda p, d: on.vis_logits(oracle_sd, p, d), lobe, lgt[:, 3:4].abs(), u[0], u[1]).t()
        assert rel_err(v, ref) <= 1e-4, prec


def test_round3_entry_points_on_empty_and_tiny_inputs(dev, synth_weights):
    """Zero rows / one row through the fused-encoding entry points, the exact-operand visibility kernel with argmax and a
    small lobe table, and the traced visibility without chunk ids: no launch on empty input, shapes kept, finite results."""
    from robir_amd import ops, packing, renderer, sg_render, synth
    from robir_amd.octree_tracing import OctreeVisModel
    full = packing.pack_sdf_h3(synth_weights, dev, full=True)
    full32 = packing.pack_sdf(synth_weights, dev, full=True)
    e3 = torch.zeros(0, 3, device=dev)
    assert ops.sdf_points_h3(e3, 0, full, True, packing.H3_SCALE_LOG2).shape == (0, 257)
    o, g = ops.sdf_points_jvp_h3(e3, 0, full, True, packing.H3_SCALE_LOG2)
    assert o.shape == (0, 257) and g.shape == (0, 3)
    o, g = ops.sdf_mlp_points(e3, 0, full32, 3)
    assert o.shape == (0, 257) and g.shape == (0, 3)
    assert ops.vis_mlp_points(e3, e3, packing.pack_vis(synth_weights, dev)).shape == (0, 2)
    assert ops.linear_pe10_256(e3, packing.pack_vis_split(synth_weights, dev)["point"]).shape == (0, 256)
    one = torch.tensor([[0.05, -0.1, 0.2]], device=dev)
    assert bool(torch.isfinite(ops.sdf_points_h3(one, 1, full, True, packing.H3_SCALE_LOG2)).all())
    # exact-operand visibility: 8 lobes x 4 samples, argmax and softmax, a single point
    m = renderer.build_synthetic_model(dev)
    lgt = torch.from_numpy(synth.synth_light_sgs(0, 128))[:8].to(dev)
    gen = torch.Generator().manual_seed(2)
    u = torch.rand(2, 8, 4, generator=gen).to(dev)
    nrm = torch.tensor([[0.0, 0.0, 1.0]], device=dev)
    old = sg_render.VIS_PRECISION
    try:
        outs = {}
        for prec in ("f16x6", "fp32"):
            sg_render.VIS_PRECISION = prec
            outs[prec] = [sg_render._diffuse_vis_core(one, nrm, m.visibility_network, lgt, u[0], u[1], 1.0, am, None, 1, None) for am in (False, True)]
        for a, b in zip(outs["f16x6"], outs["fp32"]):
            assert a.shape == (1, 8) and float((a - b).abs().max()) <= 2e-6
        assert sg_render._diffuse_vis_core(e3, e3, m.visibility_network, lgt, u[0], u[1], 1.0, False, None, 1, None).shape == (0, 8)
    finally:
        sg_render.VIS_PRECISION = old
    # traced visibility: one chunk without chunk ids, compacted and plain
    tr = OctreeVisModel(m.octree_ray_tracer)
    pts = torch.tensor([[0.0, 0.0, 0.26], [0.1, 0.0, 0.24]], device=dev)
    nr = torch.nn.functional.normalize(pts, dim=-1)
    lg = torch.from_numpy(synth.synth_light_sgs(0, 128)).to(dev)
    uu = torch.rand(2, 128, 32, generator=gen).to(dev)
    a = sg_render._diffuse_vis_core(pts, nr, tr, lg, uu[0], uu[1], 1.0, False, None, 1, None)
    old_c = ops.OVIS_COMPACT
    try:
        ops.OVIS_COMPACT = False
        b = sg_render._diffuse_vis_core(pts, nr, tr, lg, uu[0], uu[1], 1.0, False, None, 1, None)
    finally:
        ops.OVIS_COMPACT = old_c
    assert torch.equal(a, b) and bool(torch.isfinite(a).all())
    ops.range_check(sync=True)


def test_round3_second_half_entry_points_on_empty_tiny_and_bad_inputs(dev, synth_weights):
    """The chunk-stream / exact-operand / one-launch entry points of the round's second half: zero rows launch nothing and keep their
    shapes, one row gives finite results equal to the reference kernels' within their bounds, null pointers and bad arguments come
    back as RobirHipError (never a device fault), rb_scatter_rows covers no-hit, broadcast and default-only blocks."""
    import ctypes
    from robir_amd import _lib, ops, packing, synth
    s = packing.H3_SCALE_LOG2
    e3 = torch.zeros(0, 3, device=dev)
    one = torch.tensor([[0.05, -0.1, 0.2]], device=dev)
    c = synth.synth_cesr_nets(0)
    sh16 = packing.pack_softplus512_h3({"net." + k: v for k, v in c["shadow_net"].items()}, "net.", 191, dev)
    ill16 = packing.pack_illum_h3(synth_weights, dev)
    col16, col6, col32 = packing.pack_color_h3(synth_weights, dev), packing.pack_color_x6(synth_weights, dev), packing.pack_color(synth_weights, dev)
    x6f, back6 = packing.pack_sdf_x6(synth_weights, dev, full=True), packing.pack_sdf_back_x6(synth_weights, dev)
    b32, back = packing.pack_sdf(synth_weights, dev, full=True), packing.pack_sdf_back(synth_weights, dev)
    feat0, feat1 = torch.zeros(0, 257, device=dev), torch.randn(1, 257, generator=torch.Generator().manual_seed(3)).to(dev)
    # zero rows
    assert ops.cesr_net_points(e3, 0, 2, sh16, 128, s, ring=True).shape == (0, 2)
    assert ops.wide_mlp_points(e3, None, ill16, False, s, ring=True).shape == (0, 144)
    assert ops.wide_mlp_h3(torch.zeros(0, 64, device=dev), ill16, False, s, ring=True).shape == (0, 144)
    assert ops.color_mlp_h3_points(e3, e3, e3, feat0[:, 1:], col16, s, ring=True).shape == (0, 3)
    assert ops.color_x6_points(e3, e3, e3, feat0[:, 1:], col6).shape == (0, 3)
    assert ops.sdf_points_x6(e3, 0, x6f, True).shape == (0, 257)
    for fn, args in ((ops.sdf_value_grad_x6, (x6f, back6)), (ops.sdf_value_grad_f32, (b32, back))):
        o, g = fn(e3, 0, *args)
        assert o.shape == (0, 257) and g.shape == (0, 3)
    # one row: the exact-operand kernels against the f32-input MFMA
    v = torch.nn.functional.normalize(torch.tensor([[0.3, -0.2, 0.9]], device=dev), dim=-1)
    a, b = ops.color_x6_points(one, v, v, feat1[:, 1:], col6), ops.color_mlp_points(one, v, v, feat1[:, 1:], col32)
    assert float((a - b).abs().max()) <= 2e-6
    (o6, g6), (o32, g32) = ops.sdf_value_grad_x6(one, 1, x6f, back6, 2.0, 0.5), ops.sdf_value_grad_f32(one, 1, b32, back, 2.0, 0.5)
    assert rel_err(o6.cpu(), o32.cpu()) <= 1e-5 and rel_err(g6.cpu(), g32.cpu()) <= 1e-4, (rel_err(o6.cpu(), o32.cpu()), rel_err(g6.cpu(), g32.cpu()))
    # bad arguments through the C ABI
    L = _lib.lib()
    nul = ctypes.c_void_p(0)
    y = torch.empty(4, 3, device=dev)
    for name, args in (
            ("rb_cesr_net_ring_points", (nul, ctypes.c_long(4), ctypes.c_int(2), ctypes.c_int(128), ops.ptr(sh16), ctypes.c_int(s), ops.ptr(y), ctypes.c_int(0), nul)),
            ("rb_cesr_net_ring_points", (ops.ptr(y), ctypes.c_long(4), ctypes.c_int(1), ctypes.c_int(128), ops.ptr(sh16), ctypes.c_int(s), ops.ptr(y), ctypes.c_int(0), nul)),
            ("rb_cesr_net_ring_points", (ops.ptr(y), ctypes.c_long(4), ctypes.c_int(2), ctypes.c_int(200), ops.ptr(sh16), ctypes.c_int(s), ops.ptr(y), ctypes.c_int(0), nul)),
            ("rb_sdf_x6_points", (ops.ptr(y), ctypes.c_long(4), ctypes.c_float(1.0), ops.ptr(x6f), ctypes.c_int(3), ctypes.c_float(1.0), ops.ptr(y), ctypes.c_int(0), ctypes.c_int(0), nul)),
            ("rb_color_x6_points", (nul, ctypes.c_long(257), ctypes.c_float(1.0), ops.ptr(y), ctypes.c_float(1.0), ops.ptr(y), ops.ptr(y), ctypes.c_long(4), ops.ptr(col6), ops.ptr(y), ctypes.c_int(0), ctypes.c_int(0), nul)),
            ("rb_octree_cast_coop", (nul,) * 6 + (ops.ptr(y), ops.ptr(y), ctypes.c_long(4), ctypes.c_int(32), ctypes.c_double(0.005), ctypes.c_int(64), ctypes.c_float(0.1)) + (nul,) * 9)):
        L, fn = _lib.resolve(name)                # the default library, or the legacy one for a retired entry point
        assert fn(*args) != 0, name
        assert b"null pointer" in L.rb_last_error() or b"kind" in L.rb_last_error() or b"n_label" in L.rb_last_error() or b"mode" in L.rb_last_error(), (name, L.rb_last_error())
    # rb_scatter_rows
    idx = torch.tensor([5, 0, 3], device=dev)
    src3, src1 = torch.arange(9, dtype=torch.float32, device=dev).reshape(3, 3), torch.tensor([[7.0], [8.0], [9.0]], device=dev)
    o3, ob, od = ops.scatter_rows([src3, src1, None], [3, 3, 1], idx, 6)
    exp3 = torch.ones(6, 3, device=dev)
    exp3[idx] = src3
    expb = torch.ones(6, 3, device=dev)
    expb[idx] = src1.expand(-1, 3)
    assert torch.equal(o3, exp3) and torch.equal(ob, expb) and torch.equal(od, torch.ones(6, 1, device=dev))
    none = ops.scatter_rows([src3[:0], None], [3, 1], idx[:0], 4, fill=0.5)
    assert all(bool((t == 0.5).all()) for t in none) and none[0].shape == (4, 3)
    ops.range_check(sync=True)


def test_round4_entry_points_on_empty_tiny_and_ragged_inputs(dev, synth_weights):
    """The two-tile exact-operand kernels through the C ABI at the edges: zero rows (no launch), one row, rows that do not fill a
    128-row round or a 16-row tile, a point whose normal culls every direction, bad arguments (an error code and a message, no launch);
    the per-point and the tile-list form of the light-visibility kernel on the same tiny input; the f16 throughput mode on it."""
    import ctypes
    from robir_amd import _lib, ops, packing, renderer, sg_render, synth
    x6f, x6d = packing.pack_sdf_x6(synth_weights, dev, full=True), packing.pack_sdf_x6(synth_weights, dev, full=False)
    back6 = packing.pack_sdf_back_x6(synth_weights, dev) + (packing.pack_sdf_back_x6(synth_weights, dev, two_tile=True)[0],)
    g = torch.Generator().manual_seed(11)
    old_min = ops.SDF_TWO_TILE_MIN_ROWS
    try:
        for n in (0, 1, 17, 127, 129, 300):
            x = ((torch.rand(n, 3, generator=g) - 0.5) * 1.0).to(dev)
            ops.SDF_TWO_TILE_MIN_ROWS = 1 << 60
            ref_v, ref_g = ops.sdf_value_grad_x6(x, n, x6f, back6, 2.0, 0.5)
            ref_d = ops.sdf_points_x6(x, n, x6d, False, 2.0, 0.5)
            ops.SDF_TWO_TILE_MIN_ROWS = 0            # force the two-tile kernels at every size
            v, gr = ops.sdf_value_grad_x6(x, n, x6f, back6, 2.0, 0.5)
            d = ops.sdf_points_x6(x, n, x6d, False, 2.0, 0.5)
            assert v.shape == (n, 257) and gr.shape == (n, 3) and d.shape == (n,)
            if n:
                assert rel_err(v.cpu(), ref_v.cpu()) <= 2e-6 and rel_err(d.cpu(), ref_d.cpu()) <= 2e-6, n
                assert rel_err(gr.cpu(), ref_g.cpu()) <= 2e-5, n
                assert torch.equal(d, v[:, 0]) and torch.equal(v, ops.sdf_points_x6(x, n, x6f, True, 2.0, 0.5)), n
    finally:
        ops.SDF_TWO_TILE_MIN_ROWS = old_min
    L = _lib.lib()
    one = torch.zeros(1, 3, device=dev)
    out = torch.zeros(257, device=dev)
    rc = L.rb_sdf_x6_points(_lib.ptr(one), ctypes.c_long(1), ctypes.c_float(1.0), _lib.ptr(x6f), ctypes.c_int(7), ctypes.c_float(1.0),
                            _lib.ptr(out), ctypes.c_int(1), ctypes.c_int(0), _lib.stream_ptr())
    assert rc != 0 and b"mode" in L.rb_last_error()
    rc = L.rb_sdf_x6_points(None, ctypes.c_long(1), ctypes.c_float(1.0), _lib.ptr(x6f), ctypes.c_int(1), ctypes.c_float(1.0),
                            _lib.ptr(out), ctypes.c_int(1), ctypes.c_int(0), _lib.stream_ptr())
    assert rc != 0 and b"null" in L.rb_last_error()
    assert L.rb_sdf_x6_points(None, ctypes.c_long(0), ctypes.c_float(1.0), None, ctypes.c_int(1), ctypes.c_float(1.0), None,
                              ctypes.c_int(1), ctypes.c_int(0), _lib.stream_ptr()) == 0          # zero rows: nothing to do, not an error
    # the two-tile colour kernel: zero rows, one row, rows that do not fill a round; null pointers refused
    col6 = packing.pack_color_x6(synth_weights, dev)
    for n in (0, 1, 17, 129):
        x = (torch.rand(n, 3, generator=g) - 0.5).to(dev)
        v = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
        feat = torch.randn(n, 257, generator=g).to(dev)
        a, b = (ops.color_x6_points(x, v, v, feat[:, 1:], col6, two_tile=t) for t in (False, True))
        assert a.shape == b.shape == (n, 3) and (n == 0 or float((a - b).abs().max()) <= 2e-6), n
    y3 = torch.zeros(3, device=dev)
    rc = L.rb_color_x6_points(None, ctypes.c_long(256), ctypes.c_float(1.0), _lib.ptr(y3), ctypes.c_float(1.0), _lib.ptr(y3), _lib.ptr(y3),
                              ctypes.c_long(1), _lib.ptr(col6), _lib.ptr(y3), ctypes.c_int(1), ctypes.c_int(0), _lib.stream_ptr())
    assert rc != 0 and b"null" in L.rb_last_error()
    assert L.rb_color_x6_points(None, ctypes.c_long(256), ctypes.c_float(1.0), None, ctypes.c_float(1.0), None, None, ctypes.c_long(0), None,
                                None, ctypes.c_int(1), ctypes.c_int(0), _lib.stream_ptr()) == 0
    # light visibility: 8 lobes x 4 samples, three points (one whose normal faces away from every direction: all culled)
    m = renderer.build_synthetic_model(dev, build_octrees=False)
    lgt = torch.from_numpy(synth.synth_light_sgs(0, 128))[:8].to(dev)
    u = torch.rand(2, 8, 4, generator=g).to(dev)
    pts = torch.tensor([[0.05, -0.1, 0.2], [0.0, 0.1, 0.25], [0.1, 0.1, 0.1]], device=dev)
    nrm = torch.tensor([[0.0, 0.0, 1.0], [0.0, 0.0, 0.0], [0.6, 0.0, 0.8]], device=dev)
    old, old_form = sg_render.VIS_PRECISION, ops.DVIS_X6_FORM
    try:
        res = {}
        for form in ("f16x6-pt", "f16x6-stream", "f16x6-1t", "fp32", "f16x1"):
            sg_render.VIS_PRECISION = form
            res[form] = [sg_render._diffuse_vis_core(pts, nrm, m.visibility_network, lgt, u[0], u[1], 1.0, am, None, 1, None) for am in (False, True)]
            assert res[form][0].shape == (3, 8) and float(res[form][0][1].abs().max()) == 0.0, form     # the culled point: zeros, like the reference
            e3 = torch.zeros(0, 3, device=dev)
            assert sg_render._diffuse_vis_core(e3, e3, m.visibility_network, lgt, u[0], u[1], 1.0, False, None, 1, None).shape == (0, 8)
        for am in (0, 1):
            assert torch.equal(res["f16x6-pt"][am], res["f16x6-stream"][am])                         # two forms of one kernel body
            assert float((res["f16x6-pt"][am] - res["fp32"][am]).abs().max()) <= 2e-6
            assert float((res["f16x6-1t"][am] - res["fp32"][am]).abs().max()) <= 2e-6
        assert float((res["f16x1"][0] - res["fp32"][0]).abs().max()) <= 5e-3                         # the narrower mode: sane, not equal
    finally:
        sg_render.VIS_PRECISION, ops.DVIS_X6_FORM = old, old_form


def test_round6_entry_points_on_empty_tiny_and_bad_inputs(dev):
    """The entry points added in round 6 (csrc/surface.hip, csrc/cesr_f16.hip) through the C ABI at the edges: zero rows (status 0, no
    launch, null pointers allowed), one row against the closed forms, bad arguments (an error code and a message, no launch)."""
    import ctypes
    import math
    from robir_amd import _lib, ops, packing, synth
    L = _lib.lib()
    c_long, c_int, c_float, ptr, sp = ctypes.c_long, ctypes.c_int, ctypes.c_float, _lib.ptr, _lib.stream_ptr
    z1 = torch.zeros(8, device=dev)
    # zero rows: nothing to do, not an error -- whatever the pointers
    assert L.rb_pe_encode(None, c_long(0), c_int(3), None, c_int(4), c_int(1), None, sp()) == 0
    assert L.rb_expected_sin(None, None, c_long(0), None, None, sp()) == 0
    assert L.rb_tonemap_curve(None, c_long(0), c_int(1), None, c_int(0), c_int(2), None, sp()) == 0
    assert L.rb_sample_pdf(None, None, c_long(0), c_int(8), None, c_long(0), c_int(4), None, None, sp()) == 0
    assert L.rb_neus_core_aux(None, c_long(1), None, None, c_long(0), c_int(8), c_float(1.0), c_float(1.0), c_float(0.1), None, None, None, sp()) == 0
    assert L.rb_sample_dirs(None, None, None, c_long(0), None, sp()) == 0
    assert L.rb_intersect_sphere(None, None, c_long(0), c_float(1.0), None, sp()) == 0
    assert L.rb_cesr_net_f16_points(None, c_long(0), c_int(0), c_int(1), None, None, c_int(3), c_int(0), sp()) == 0
    # ... and the wrappers return empty tensors of the right shape
    e3 = torch.zeros(0, 3, device=dev)
    fr = torch.tensor([1.0, 2.0, 4.0], device=dev)
    assert ops.pe_encode(e3, fr).shape == (0, 21) and ops.pe_encode(e3, fr, include_input=False).shape == (0, 18)
    assert ops.tonemap_curve(torch.zeros(0, 3, device=dev), torch.zeros(0, 1, device=dev), 2).shape == (0, 3)
    assert ops.sample_dirs(e3, torch.zeros(0, device=dev), torch.zeros(0, device=dev)).shape == (0, 3)
    assert ops.intersect_sphere(e3, e3, 1.0).shape == (0, 3)
    # null pointers / bad arguments: refused with a message
    assert L.rb_pe_encode(None, c_long(1), c_int(3), ptr(fr), c_int(3), c_int(1), ptr(z1), sp()) != 0 and b"null" in L.rb_last_error()
    assert L.rb_pe_encode(ptr(z1), c_long(1), c_int(0), ptr(fr), c_int(3), c_int(1), ptr(z1), sp()) != 0 and b"d >= 1" in L.rb_last_error()
    assert L.rb_tonemap_curve(ptr(z1), c_long(3), c_int(1), None, c_int(0), c_int(9), ptr(z1), sp()) != 0 and b"curve" in L.rb_last_error()
    assert L.rb_tonemap_curve(ptr(z1), c_long(3), c_int(1), None, c_int(0), c_int(2), ptr(z1), sp()) != 0 and b"shift" in L.rb_last_error()
    assert L.rb_sample_pdf(ptr(z1), ptr(z1), c_long(1), c_int(1), ptr(z1), c_long(0), c_int(4), ptr(z1), ptr(z1), sp()) != 0 and b"bins" in L.rb_last_error()
    assert L.rb_sample_pdf(ptr(z1), ptr(z1), c_long(1), c_int(4), ptr(z1), c_long(3), c_int(4), ptr(z1), ptr(z1), sp()) != 0 and b"stride" in L.rb_last_error()
    assert L.rb_sample_dirs(ptr(z1), None, ptr(z1), c_long(1), ptr(z1), sp()) != 0 and b"null" in L.rb_last_error()
    assert L.rb_intersect_sphere(ptr(z1), ptr(z1), c_long(1), c_float(1.0), None, sp()) != 0 and b"null" in L.rb_last_error()
    # one row against the closed forms
    x = torch.tensor([[0.3, -0.2, 0.5]], device=dev)
    pe = ops.pe_encode(x, fr).cpu()[0]
    want = [0.3, -0.2, 0.5]
    for f in (1.0, 2.0, 4.0):
        want += [math.sin(f * v) for v in (0.3, -0.2, 0.5)] + [math.cos(f * v) for v in (0.3, -0.2, 0.5)]
    assert float((pe - torch.tensor(want)).abs().max()) <= 1e-6
    o, d = torch.tensor([[0.0, 0.0, 0.0]], device=dev), torch.tensor([[0.0, 0.6, 0.8]], device=dev)
    assert float((ops.intersect_sphere(o, d, 2.0).cpu()[0] - torch.tensor([0.0, 1.2, 1.6])).abs().max()) <= 1e-6
    bins = torch.tensor([[0.0, 1.0, 2.0, 3.0]], device=dev)
    w = torch.tensor([[1.0, 0.0, 1.0]], device=dev)
    s, cdf = ops.sample_pdf(bins, w, torch.tensor([0.25, 0.75], device=dev))
    assert s.shape == (1, 2) and cdf.shape == (1, 4) and abs(float(cdf[0, -1]) - 1.0) <= 1e-6
    assert 0.0 <= float(s[0, 0]) <= 1.0 and 2.0 <= float(s[0, 1]) <= 3.0            # the empty middle bin is never sampled
    # the plain-f16 CESR kernel: a tile count the build does not carry, an unknown kind, too many labels -- refused; one row runs
    cz = synth.synth_cesr_nets(0)
    blob = packing.pack_softplus512_f16({"net." + k: v for k, v in cz["normal_net"].items()}, "net.", 63, dev)
    y = torch.zeros(3, device=dev)
    for tiles in (0, 1, 7):
        assert L.rb_cesr_net_f16_points(ptr(x), c_long(1), c_int(0), c_int(1), ptr(blob), ptr(y), c_int(tiles), c_int(0), sp()) != 0
        assert b"tile" in L.rb_last_error()
    assert L.rb_cesr_net_f16_points(ptr(x), c_long(1), c_int(1), c_int(1), ptr(blob), ptr(y), c_int(ops.CESR_F16_TILES), c_int(0), sp()) != 0
    assert b"kind" in L.rb_last_error()
    assert L.rb_cesr_net_f16_points(ptr(x), c_long(1), c_int(2), c_int(129), ptr(blob), ptr(y), c_int(ops.CESR_F16_TILES), c_int(0), sp()) != 0
    assert b"n_label" in L.rb_last_error()
    assert L.rb_cesr_net_f16_points(None, c_long(1), c_int(0), c_int(1), ptr(blob), ptr(y), c_int(ops.CESR_F16_TILES), c_int(0), sp()) != 0
    one = ops.cesr_net_f16_points(x, 1, 0, blob, 1)
    torch.cuda.synchronize()
    assert one.shape == (1, 3) and bool(torch.isfinite(one).all())
    # the two-tile light-visibility kernel names its weight layout through scale_log2 (8 = the bf8 layout of packing.repack_x6_chunks_fp8 since
    # round 6, 0 = rb_pack_layer_x6's own for a library built with -DXT_FP8=0): the other name is refused before any launch
    if ops.DVIS_X6_FP8:
        z = torch.zeros(4096, device=dev)
        rc = L.rb_dvis_fused_x6t(ptr(z), None, c_long(1), ptr(z), ptr(z), ptr(z), ptr(z), ptr(z), ptr(z), c_int(8), c_int(4), c_int(0), c_int(0),
                                 ptr(z), None, sp())
        assert rc != 0 and b"bf8 weight layout" in L.rb_last_error()
