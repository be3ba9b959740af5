"""Octree build + lock-step cast on the GPU against the oracle / the reference's golden hits."""
import numpy as np
import pytest
import torch

from conftest import rel_err, bad_frac, load_golden, bounded

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def neus(dev, synth_weights):
    from robir_amd import nets, synth
    m = nets.NeuSModel(embed="PE")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.neus_state_dict(synth_weights).items()})
    return m.to(dev).eval()


@pytest.fixture(scope="module")
def dev_octree(neus):
    from robir_amd.octree_tracing import OctreeSDF
    return OctreeSDF.build(neus.sdf_network, [[-1.0] * 3, [1.0] * 3])


def _oracle_dev(oracle_octree, dev, max_iter=-1):
    from robir_amd.octree_tracing import OctreeSDF
    return OctreeSDF.from_host_tables(oracle_octree, dev, max_iter)


def test_build_matches_oracle(dev_octree, oracle_octree):
    T, O = dev_octree.tables, oracle_octree
    assert T.B == O.box_min.shape[0]
    node = T.node.cpu()
    assert rel_err(node[:, 0:3], O.box_min) == 0.0 and rel_err(node[:, 4:7], O.box_size) == 0.0
    fc = node[:, 3].view(torch.int32).long()
    exp = torch.where(O.is_split, O.child[:, 0], torch.full_like(fc, -1))
    assert bool((fc == exp).all())
    assert rel_err(node[:, 7], O.sdf_val) <= TOL
    assert float((T.nrm.cpu() - O.sdf_nrm).abs().max()) <= 1e-4
    hit_dev, hit_or = node[:, 7] <= 1e-4, O.hit
    assert int((hit_dev != hit_or).sum()) <= 8


def test_primary_cast_bit_parity_with_oracle_tables(dev, oracle_octree):
    """Same tables, same rays: hits, t and the lock-step schedule must match the oracle exactly (t to 1 ulp-ish)."""
    from robir_oracle import octree as ooct
    g = load_golden("cast_primary")
    od = _oracle_dev(oracle_octree, dev)
    cam, dirs = torch.from_numpy(g["cam"]), torch.from_numpy(g["dirs"])
    x, hit, t = od.cast_chunks(cam.to(dev), dirs.to(dev), chunk=1024, sched_cap=128)
    sched = od.last_sched.cpu()
    for i in range(2):
        log = []
        xo, ho, to = ooct.trace(oracle_octree, cam, dirs[None, i * 1024:(i + 1) * 1024], -1, log)
        sl = slice(i * 1024, (i + 1) * 1024)
        assert bool((hit[sl].cpu() == ho).all())
        assert [int(v) for v in sched[i, :len(log), 1]] == [m for _, m in log]
        assert [int(v) for v in sched[i, :len(log), 0]] == [n for n, _ in log]
        assert int(sched[i, len(log), 0]) == 0
        assert rel_err(t[sl].cpu(), to) <= 1e-6
        assert rel_err(x[sl].cpu(), xo) <= 1e-6


def test_primary_cast_vs_reference_golden(dev, dev_octree):
    """Device-built octree against the reference's own hits for two chunks."""
    g = load_golden("cast_primary")
    cam, dirs = torch.from_numpy(g["cam"]).to(dev), torch.from_numpy(g["dirs"]).to(dev)
    x, hit, t = dev_octree.cast_chunks(cam, dirs, chunk=1024, sched_cap=128)
    sched = dev_octree.last_sched.cpu()
    for i, c in enumerate((1, 2)):
        sl = slice(i * 1024, (i + 1) * 1024)
        rh, rt = torch.from_numpy(g["hit"][i]), torch.from_numpy(g["t"][i])
        assert int((hit[sl].cpu() != rh).sum()) <= 2
        both = hit[sl].cpu() & rh
        bounded("cast_primary_vs_reference_golden/t_c%d" % c, t[sl].cpu()[both], rt[both], TOL, 0.005)
        ref_m = list(g["sched_m_c%d" % c])
        assert [int(v) for v in sched[i, :len(ref_m), 1]] == ref_m


def test_secondary_and_general_path(dev, oracle_octree):
    """max_iter=32 mode, > 1024 rays (multi-launch path) and <= 1024 rays (single-workgroup path) vs the oracle."""
    from robir_oracle import octree as ooct
    g = np.random.Generator(np.random.PCG64(11))
    od = _oracle_dev(oracle_octree, dev, max_iter=32)
    n = 3000
    o = g.standard_normal((n, 3)).astype(np.float32)
    o = 0.26 * o / np.linalg.norm(o, axis=1, keepdims=True)
    d = g.standard_normal((n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o_t, d_t = torch.from_numpy(o), torch.from_numpy(d)
    for cnt in (n, 700):
        to, ho = ooct.cast(oracle_octree, o_t[:cnt], d_t[:cnt], 32)
        x, hit, t = od.cast_full(o_t[:cnt].to(dev), d_t[:cnt].to(dev))
        assert bool((hit.cpu() == ho).all()), (cnt, int((hit.cpu() != ho).sum()))
        assert rel_err(t.cpu(), to) <= 1e-6
    # primary mode through the multi-launch path
    od1 = _oracle_dev(oracle_octree, dev, max_iter=-1)
    o2 = torch.tensor([[0.0, 0.0, 0.9]]).expand(n, 3).contiguous()
    d2 = d_t.clone()
    d2[:, 2] = -d2[:, 2].abs() - 1.5
    d2 = d2 / d2.norm(dim=-1, keepdim=True)
    to, ho = ooct.cast(oracle_octree, o2, d2, -1)
    x, hit, t = od1.cast_full(o2.to(dev), d2.to(dev))
    assert bool((hit.cpu() == ho).all())
    assert rel_err(t.cpu(), to) <= 1e-6


def test_one_launch_cast_equals_per_iteration_launches(dev, oracle_octree):
    """k_cast_coop (persistent grid, grid-wide arrival counters) == k_cast_init / k_cast_iter / k_cast_finish bit for bit, in both
    modes, below and above one ray per thread of the capped grid, and the per-iteration active counts with it."""
    from robir_amd import ops
    g = np.random.Generator(np.random.PCG64(12))
    for max_iter, n in ((32, 1500), (32, 40000), (32, 300001), (-1, 5000), (0, 5000)):
        od = _oracle_dev(oracle_octree, dev, max_iter=max_iter)
        o = g.standard_normal((n, 3)).astype(np.float32)
        o = (0.26 if max_iter > 0 else 0.9) * o / np.linalg.norm(o, axis=1, keepdims=True)
        d = g.standard_normal((n, 3)).astype(np.float32)
        if max_iter <= 0:
            d = -o + 0.3 * d
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        o_t, d_t = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
        step = od.step_size(n)
        a = ops.octree_cast_general(od.tables, o_t, d_t, od.max_iter, step, one_launch=False)
        for _ in range(2):
            b = ops.octree_cast_general(od.tables, o_t, d_t, od.max_iter, step, one_launch=True)
            for u, v in zip(a[:3], b[:3]):
                assert torch.equal(u, v), (max_iter, n)
            m = min(a[3].numel(), b[3].numel())
            ca, cb = a[3][:m].cpu(), b[3][:m].cpu()
            last = int((cb > 0).nonzero().max()) + 2 if bool((cb > 0).any()) else 1
            assert torch.equal(ca[:last], cb[:last]), (max_iter, n)


def test_camera_rays(dev):
    from robir_amd import ops, synth
    from robir_oracle import renderer
    uv, pose, K = synth.synth_camera(64, 64)
    K[0, 1] = 0.3     # exercise the skew term
    d = ops.camera_rays(pose, K, torch.from_numpy(uv).to(dev)).cpu()
    ref, cam = renderer.camera_rays(torch.from_numpy(uv)[None], torch.from_numpy(pose)[None], torch.from_numpy(K)[None])
    assert rel_err(d, ref[0]) <= 1e-6


def test_camera_rays_quaternion_pose_vs_reference_golden(dev):
    """get_camera_params' 7-vector pose form (utils/rend_util.py:52-57 + quat_to_rot): reference-recorded directions / centres
    (tests/golden/camera_quat.npz, oracle/gen_golden_r4.py), host-array and device-tensor forms, and forward() given the 7-vector."""
    from robir_amd import ops
    g = load_golden("camera_quat")
    for b in range(2):
        uv = torch.from_numpy(g["uv"][b]).to(dev)
        for pose, K in ((g["pose7"][b], g["K"][b]), (torch.from_numpy(g["pose7"][b]).to(dev), torch.from_numpy(g["K"][b]).to(dev))):
            d = ops.camera_rays(pose, K, uv).cpu()
            assert rel_err(d, torch.from_numpy(g["ray_dirs"][b])) <= 1e-6
            m = ops.pose_matrix(pose)
            loc = m[:3, 3].cpu() if isinstance(m, torch.Tensor) else torch.from_numpy(m[:3, 3])
            assert rel_err(loc, torch.from_numpy(g["cam_loc"][b])) <= 1e-7
    assert ops.pose_matrix(np.eye(4, dtype=np.float32)).shape == (4, 4)      # a matrix passes through


def test_build_second_sdf_and_mesh_box(dev):
    """Another SDF (other seed) and the mesh-bounding-box form of generate() (octree_tracing.py:33-39: root = a box
    tighter than [-1,1]^3): the device-built node structure must again equal the oracle's exactly."""
    from robir_amd import nets, synth
    from robir_amd.octree_tracing import OctreeSDF
    from robir_oracle import nets as on, octree as ooct
    import os
    w = synth.synth_state_dict(3, variance=0.6)
    m = nets.NeuSModel(embed="PE")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.neus_state_dict(w).items()})
    m = m.to(dev).eval()
    sd = on.as_torch(w)
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    box = ([-0.42, -0.40, -0.45], [0.41, 0.43, 0.40])
    O = ooct.build(lambda x: on.implicit_forward(sd, x)[:, 0], lambda x: on.implicit_gradient(sd, x), box[0], box[1])
    T = OctreeSDF.build(m.sdf_network, [box[0], box[1]]).tables
    node = T.node.cpu()
    # A cell splits when |sdf(centre)| is below a size-dependent threshold; the device and the CPU evaluate the SDF with
    # different summation orders (~1e-7), so a cell sitting on the threshold may split on one side only.  Such a flip
    # renumbers every later node, hence the comparison by cell set: the two trees must agree except for at most a few
    # threshold cells (and their 8 children).
    def cells(bmin, bsize):
        a = torch.cat([bmin, bsize], 1).contiguous().numpy()
        return set(map(bytes, a.view(np.uint8).reshape(a.shape[0], -1)))
    dev_cells, ora_cells = cells(node[:, 0:3], node[:, 4:7]), cells(O.box_min, O.box_size)
    only = len(dev_cells ^ ora_cells)
    print("nodes", T.B, O.box_min.shape[0], "cells in one tree only:", only)
    assert only <= 4 * 9 and abs(T.B - O.box_min.shape[0]) <= 4 * 8
