"""Multi-GPU: one process per GPU, rays sharded at CHUNK granularity, one all-gather of the rendered tiles.

The per-ray path has no exchange step: the only chunk-global quantities (octree tracer schedule,
specular-cone minimum) live inside a 1024-pixel chunk, so rank r renders whole chunks and nothing crosses GPUs
during rendering.  The single collective is the gather of the output tiles (RCCL all-gather over xGMI on MI355X;
`backend="nccl"` is RCCL on ROCm; gloo on CPU for the tests).
"""
import os

import torch
import torch.distributed as dist


def bind_device(share_gpu=False):
    """This rank's GPU: cuda:LOCAL_RANK, as torch.distributed.run exports it (one process per GPU); share_gpu: every rank on cuda:0 (the
    one-GPU functional runs over gloo).  Sets the current device and returns it."""
    local = 0 if share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if local >= torch.cuda.device_count():
        raise RuntimeError(f"LOCAL_RANK={local} but {torch.cuda.device_count()} GPU(s) are visible: start one rank per GPU")
    torch.cuda.set_device(local)
    return torch.device("cuda", local)


def init_distributed(backend="nccl", device=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run's variables; the address
    defaults to 127.0.0.1).  backend "nccl" is RCCL on ROCm and is bound to `device` (default: bind_device()); "gloo" for CPU tests and
    ranks that share a GPU.  Returns the device."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        device = device or bind_device()
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group(backend)
    return device


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
        # gloo (CPU tests; two ranks sharing one GPU): stage through host memory, which every gloo build supports
        host = tiles.detach().cpu().contiguous()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host)
        out.copy_(torch.cat(parts, 0))
    return out


def gather_image(local_tiles, n_chunks, chunk, interleave=True):
    """Strong-scaling form: every rank rendered `shard_chunks(...)` of ONE image; returns the full image
    [n_chunks*chunk, F] on every rank (ranks holding fewer chunks pad with one dummy chunk)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    per = (n_chunks + world - 1) // world
    F = local_tiles.shape[1]
    pad = torch.zeros(per * chunk, F, dtype=local_tiles.dtype, device=local_tiles.device)
    pad[: local_tiles.shape[0]] = local_tiles
    allt = all_gather_tiles(pad).reshape(world * per, chunk, F)
    # chunk c sits at slot (rank, position in the rank's list): one gather through an index built once per (layout, device)
    key = (n_chunks, world, bool(interleave), str(local_tiles.device))
    src = _GATHER_INDEX.get(key)
    if src is None:
        slot = [0] * n_chunks
        for r in range(world):
            for k, c in enumerate(shard_chunks(n_chunks, r, world, interleave)):
                slot[c] = r * per + k
        src = _GATHER_INDEX[key] = torch.tensor(slot, dtype=torch.int64, device=local_tiles.device)
    return allt.index_select(0, src).reshape(n_chunks * chunk, F)


_GATHER_INDEX = {}

TILE_FIELDS = ("sg_rgb", "indir_rgb", "diffuse_albedo", "roughness", "vis_shadow", "normal_map", "network_object_mask")


def pack_tiles(o):
    """The 17 floats per ray a consumer of a rendered view needs (plot_to_disk / relight keep exactly these)."""
    return torch.cat([o["sg_rgb"], o["indir_rgb"], o["diffuse_albedo"], o["roughness"][:, :1], o["vis_shadow"],
                      o["normal_map"], o["network_object_mask"][:, None].float()], -1)


def plan_view(N, chunk, device, interleave=True, chunks_per_pass=None):
    """Row indices of the passes this rank renders of an N-pixel view: [(chunk_ids, rows[int64] | slice, ragged)]."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    n_chunks = (N + chunk - 1) // chunk
    ids = shard_chunks(n_chunks, rank, world, interleave)
    full = [c for c in ids if (c + 1) * chunk <= N]
    ragged = [c for c in ids if (c + 1) * chunk > N]            # at most the last chunk of the view
    per = chunks_per_pass or max(1, len(full))
    passes = []
    for s in range(0, len(full), per):
        grp = full[s:s + per]
        if grp == list(range(grp[0], grp[0] + len(grp))):
            rows = slice(grp[0] * chunk, (grp[-1] + 1) * chunk)
        else:
            rows = (torch.tensor(grp, device=device)[:, None] * chunk + torch.arange(chunk, device=device)[None]).reshape(-1)
        passes.append((grp, rows, False))
    for c in ragged:
        passes.append(([c], slice(c * chunk, N), True))
    return {"passes": passes, "n_chunks": n_chunks, "world": world, "N": N, "chunk": chunk, "interleave": interleave}


def render_view_sharded(model, uv, pose, K, hdr_shift, chunk=1024, draws_for=None, interleave=True, stats=None,
                        chunks_per_pass=None, plan=None):
    """Strong-scaling render of ONE view: this rank renders chunks {c : c mod world = rank} through
    IDRNetwork.render_chunks -- every chunk keeps its own lock-step tracer schedule and specular-cone minimum, i.e. exactly
    what the reference computes rendering the chunks one after another -- then one all-gather of the 17-float tiles puts the
    whole image [N, 17] on every rank.  Works without an initialised process group (world = 1).
    uv [N,2], hdr_shift [N,1] on the device; draws_for(chunk_ids) -> explicit draws dict for those chunks (tests) or None."""
    N = uv.shape[0]
    plan = plan or plan_view(N, chunk, uv.device, interleave, chunks_per_pass)
    parts = []
    for grp, rows, is_ragged in plan["passes"]:
        o = model.render_chunks(uv[rows], pose, K, hdr_shift[rows], chunk=chunk, stats=stats,
                                draws=draws_for(grp) if draws_for else None)
        t = pack_tiles(o)
        if is_ragged:                                           # its own lock-step batch, like the reference's last chunk
            t = torch.cat([t, t.new_zeros(chunk - t.shape[0], t.shape[1])])
        parts.append(t)
    if not parts:                                               # fewer chunks than ranks: this rank only joins the gather
        local = uv.new_zeros(0, 17)
    else:
        local = torch.cat(parts) if len(parts) != 1 else parts[0]
    if plan["world"] == 1:
        return local[:N]
    return gather_image(local, plan["n_chunks"], chunk, plan["interleave"])[:N]
