"""Ray sharding + tile gather with 2 processes on the gloo backend (CPU)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_chunks, chunk, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from robir_amd import parallel
    full = torch.arange(n_chunks * chunk * 3, dtype=torch.float32).reshape(n_chunks * chunk, 3)
    ids = parallel.shard_chunks(n_chunks, rank, world)
    local = torch.cat([full[i * chunk:(i + 1) * chunk] for i in ids])
    img = parallel.gather_image(local, n_chunks, chunk)
    ok1 = bool(torch.equal(img, full))
    tiles = torch.full((5, 2), float(rank))
    ag = parallel.all_gather_tiles(tiles)
    ok2 = bool(torch.equal(ag[:5], torch.zeros(5, 2)) and torch.equal(ag[5:], torch.ones(5, 2)))
    t = torch.tensor([1.0 + rank])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    q.put((rank, ok1, ok2, float(t)))
    dist.destroy_process_group()


def test_shard_chunks_partition():
    from robir_amd import parallel
    for world in (1, 2, 3, 8):
        for inter in (True, False):
            got = sorted(sum((parallel.shard_chunks(625, r, world, inter) for r in range(world)), []))
            assert got == list(range(625))


def test_gather_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 7, 4, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(r[1] and r[2] and r[3] == 2.0 for r in res), res


class _StubModel:
    """render_chunks of a 'renderer' whose 17 tile floats are plain functions of the pixel coordinates: what the sharding / gather code
    moves must come back as the single-process image, whatever the rank layout."""

    def render_chunks(self, uv, pose, K, hdr_shift, chunk=1024, stats=None, draws=None):
        assert uv.shape[0] <= chunk or uv.shape[0] % chunk == 0          # whole chunks, or ONE ragged chunk as its own pass
        x, y = uv[:, :1], uv[:, 1:2]
        f = lambda a, b, c: torch.cat([x * a + y, x - y * b, x * y * c], -1)
        return {"sg_rgb": f(1.0, 2.0, 1e-3), "indir_rgb": f(3.0, 4.0, 2e-3), "diffuse_albedo": f(5.0, 6.0, 3e-3),
                "roughness": f(7.0, 8.0, 4e-3), "vis_shadow": f(9.0, 10.0, 5e-3) + hdr_shift, "normal_map": f(11.0, 12.0, 6e-3),
                "network_object_mask": (x + y).remainder(3.0).squeeze(-1) < 1.0}


def _worker8(rank, world, port, cases, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from robir_amd import parallel
    res = []
    for H, W, chunk, per_pass in cases:
        N = H * W
        yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
        uv = torch.stack([xx, yy], -1).reshape(N, 2)
        hdr = torch.full((N, 1), 0.5)
        plan = parallel.plan_view(N, chunk, uv.device, True, per_pass)
        mine = sorted(c for grp, _, _ in plan["passes"] for c in grp)
        img = parallel.render_view_sharded(_StubModel(), uv, None, None, hdr, chunk=chunk, chunks_per_pass=per_pass, plan=plan)
        ref = parallel.pack_tiles(_StubModel().render_chunks(uv, None, None, hdr, chunk=N))         # the whole view in one process
        n_chunks = (N + chunk - 1) // chunk
        res.append((bool(torch.equal(img, ref)), mine == parallel.shard_chunks(n_chunks, rank, world), len(mine), n_chunks,
                    sum(1 for _, _, ragged in plan["passes"] if ragged)))
    q.put((rank, res))
    dist.destroy_process_group()


def test_sharded_view_world8_gloo():
    """the sharded render + tile gather at WORLD SIZE 8 (gloo, CPU): the 625 chunks of an 800 x 800 view with a ragged
    last chunk (N = 639 800), passes of 7 chunks, and a view with FEWER chunks than ranks (ranks 6 and 7 own nothing and only join the
    gather) -- every rank ends with the single-process image bit for bit."""
    world = 8
    cases = [(700, 914, 1024, None),          # 639 800 px = 624 full chunks + 824: 625 chunks, ragged last one (rank 0 owns it: 624 % 8 == 0)
             (700, 914, 1024, 7),                # the same view in passes of 7 chunks
             (61, 100, 1024, None),              # 6100 px = 5 full chunks + 980: six chunks on eight ranks
             (3, 50, 1024, None)]                # one ragged chunk only: seven ranks idle
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, cases, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(60)
    assert sorted(res) == list(range(world))
    for r in range(world):
        for ci, (ok, ids_ok, n_mine, n_chunks, n_ragged) in enumerate(res[r]):
            assert ok and ids_ok, (r, ci)
    assert [res[r][0][2] for r in range(world)] == [79] + [78] * 7 and res[0][0][3] == 625          # 625 = 79 + 7 * 78
    assert sum(res[r][0][4] for r in range(world)) == 1 and res[0][0][4] == 1                        # the ragged chunk 624 is rank 0's
    assert [res[r][2][2] for r in range(world)] == [1, 1, 1, 1, 1, 1, 0, 0]
    assert [res[r][3][2] for r in range(world)] == [1, 0, 0, 0, 0, 0, 0, 0]
