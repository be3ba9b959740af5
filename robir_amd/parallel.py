"""Multi-GPU: one process per GPU, rays sharded at CHUNK granularity, one all-gather of the rendered tiles.

The per-ray path has no exchange step: the only chunk-global quantities (octree tracer schedule,
specular-cone minimum) live inside a 1024-pixel chunk, so rank r renders whole chunks and nothing crosses GPUs
during rendering.  The single collective is the gather of the output tiles (RCCL all-gather over xGMI on MI355X;
`backend="nccl"` is RCCL on ROCm; gloo on CPU for the tests).
"""
import torch
import torch.distributed as dist


def shard_chunks(n_chunks, rank, world, interleave=True):
    """Chunk ids rendered by `rank`.  Interleaved assignment balances the per-chunk hit fraction (rows of an image
    near the object are expensive, background rows are cheap)."""
    if interleave:
        return list(range(rank, n_chunks, world))
    per = (n_chunks + world - 1) // world
    return list(range(rank * per, min(n_chunks, (rank + 1) * per)))


def all_gather_tiles(tiles):
    """tiles [n, F] (same n on every rank) -> [world*n, F], rank-major."""
    world = dist.get_world_size()
    out = torch.empty((world * tiles.shape[0],) + tuple(tiles.shape[1:]), dtype=tiles.dtype, device=tiles.device)
    if dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(out, tiles.contiguous())
    else:
        parts = list(out.chunk(world, 0))
        dist.all_gather(parts, tiles.contiguous())
    return out


def gather_image(local_tiles, n_chunks, chunk, interleave=True):
    """Strong-scaling form: every rank rendered `shard_chunks(...)` of ONE image; returns the full image
    [n_chunks*chunk, F] on every rank (ranks holding fewer chunks pad with one dummy chunk)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    per = (n_chunks + world - 1) // world
    F = local_tiles.shape[1]
    pad = torch.zeros(per * chunk, F, dtype=local_tiles.dtype, device=local_tiles.device)
    pad[: local_tiles.shape[0]] = local_tiles
    allt = all_gather_tiles(pad).reshape(world, per, chunk, F)
    img = torch.empty(n_chunks, chunk, F, dtype=local_tiles.dtype, device=local_tiles.device)
    for r in range(world):
        ids = shard_chunks(n_chunks, r, world, interleave)
        img[ids] = allt[r, : len(ids)]
    return img.reshape(n_chunks * chunk, F)
