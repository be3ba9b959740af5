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
