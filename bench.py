#!/usr/bin/env python
"""bench.py -- PBR-stage rays/s of the MI355X-native renderer (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (config 4 of BASELINE.json, the one the metric is quoted on; it fits one GPU): full PBR forward of a
synthetic 800x800 view -- camera rays, octree sphere trace of 625 lock-step chunks of 1024 pixels, SDF at every
ray, per-hit indirect-illumination / material networks, 128-lobe light-SG visibility (32 samples per lobe through the
visibility MLP), 8+8 BRDF-lobe visibility samples, SG shading (128 direct + 24 indirect lobes), scatter to per-ray
outputs.  Synthetic weights (robir_amd.synth, seed 0), random draws generated on the device inside the step.
One step = one such image per GPU ("weak": rank r renders view r of an N-view job, tiles all-gathered over RCCL at
the end of the step).  The octree build is one-off set-up and is reported separately.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel: fused light-SG visibility, fp32 MFMA bound) and, at
N=1, `cpu_baseline` (the oracle restatement timed on this box's host cores on a bounded sample of the same workload).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VIS_MACS_PER_EVAL = 229376          # 126*256 + 3*256*256 + 256*2
PEAK_FP32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2, dense
PEAK_F16_MFMA_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense f16/bf16 MFMA (not the 2:1-sparse marketing figure)
H = W = 800
CHUNK = 1024


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--chunks-per-batch", type=int, default=625,
                    help="1024-pixel chunks rendered per kernel pass (625 = the whole 800x800 view; 2.7 GB of tables)")
    ap.add_argument("--cpu-baseline-chunks", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exact", action="store_true",
                    help="skip the exact-fp32 cross-check step that follows the timed region (rank 0, N = 1)")
    ap.add_argument("--vis-precision", default="f16x3-v2", choices=["fp32", "f16x3-v2", "f16x3", "f16x3-regstage", "f16x3-nt2"],
                    help="hidden layers of the fused light-visibility kernel: exact f32-input MFMA, or the error-compensated "
                         "hi/lo half split on the f16 MFMA (fp32 accumulate, same measured parity)")
    return ap.parse_args()


def view_pose(rank, distance=0.9):
    """Camera `rank` of the multi-view job: rotate the synthetic camera about the y axis."""
    a = 0.35 * rank
    c, s = math.cos(a), math.sin(a)
    R = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float32)
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = R
    pose[:3, 3] = R @ np.array([0, 0, distance], np.float32)
    return pose


class KernelTimer:
    """HIP-event timing of one kernel family on the stream it is launched on (torch's current stream)."""

    def __init__(self):
        self.pairs = []
        self.on = False

    def wrap(self, fn):
        def inner(*a, **k):
            if not self.on:
                return fn(*a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = fn(*a, **k)
            e.record()
            self.pairs.append((s, e))
            return out
        return inner

    def stats(self):
        ms = [s.elapsed_time(e) for s, e in self.pairs]
        return (sum(ms) / len(ms), len(ms)) if ms else (0.0, 0)


def render_image(model, uv_d, pose_d, K_d, hdr, chunks_per_batch, stats):
    """One full image in passes of `chunks_per_batch` chunks.  Returns the per-ray tiles a consumer needs."""
    N = uv_d.shape[0]
    per = chunks_per_batch * CHUNK
    outs = []
    for s in range(0, N, per):
        o = model.render_chunks(uv_d[s:s + per], pose_d, K_d, hdr[s:s + per], chunk=CHUNK, stats=stats)
        outs.append(torch.cat([o["sg_rgb"], o["indir_rgb"], o["diffuse_albedo"], o["roughness"][:, :1],
                               o["vis_shadow"], o["normal_map"], o["network_object_mask"][:, None].float()], -1))
    return torch.cat(outs, 0)                                        # [N, 17]


def cpu_baseline(model, n_chunks, uv, pose, K):
    """Oracle (CPU restatement of the reference) on `n_chunks` central chunks of the same image, all host cores."""
    from robir_oracle import nets as on, octree as ooct, renderer as orend
    from robir_amd import synth
    cores = min(16, os.cpu_count() or 1)      # more threads make the many small tensor ops of the reference slower
    torch.set_num_threads(cores)
    sd = on.as_torch(synth.synth_state_dict(0, variance=0.3))
    # geometry: the octree the device built, converted to the oracle's table format (same cells, no 10 s CPU rebuild)
    Td = model.ray_tracer.sdf_octree.tables
    node = Td.node.cpu()
    T = ooct.OctreeTables()
    T.root_min, T.root_size = torch.from_numpy(Td.root_min.copy()), torch.from_numpy(Td.root_size.copy())
    T.box_min, T.box_size, T.sdf_val = node[:, 0:3].contiguous(), node[:, 4:7].contiguous(), node[:, 7].contiguous()
    fc = node[:, 3].contiguous().view(torch.int32).long()
    T.is_split = fc >= 0
    T.child = torch.where(T.is_split[:, None], fc[:, None] + torch.arange(8)[None, :], torch.full((1, 8), -1))
    res = [int(v) for v in Td.res]
    T.base_index = torch.arange(res[0] * res[1] * res[2]).reshape(*res)
    T.sdf_nrm, T.centre = Td.nrm.cpu(), T.box_min + T.box_size * 0.5
    T.hit, T.min_step = T.sdf_val <= 1e-4, Td.min_step
    first = (H // 2) * W // CHUNK - n_chunks // 2            # chunks around the image centre (hit fraction ~ image mean x2)
    t0 = time.time()
    rays = hits = 0
    for c in range(first, first + n_chunks):
        sl = slice(c * CHUNK, (c + 1) * CHUNK)
        uv_t = torch.from_numpy(uv[sl])[None]
        dirs, cam = orend.camera_rays(uv_t, torch.from_numpy(pose)[None], torch.from_numpy(K)[None])
        _, hit, _ = ooct.trace(T, cam, dirs, -1)
        n_hit = int(hit.sum())
        dr = {k: torch.from_numpy(v) for k, v in synth.pbr_draws(7, n_hit, chunk_id=c).items()}
        orend.forward(sd, T, uv_t, torch.from_numpy(pose)[None], torch.from_numpy(K)[None],
                      torch.ones(1, CHUNK, dtype=torch.bool), torch.full((CHUNK, 1), 0.5), dr, "Material", testing=True)
        rays += CHUNK
        hits += n_hit
    dt = time.time() - t0
    return {"value": rays / dt, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": f"{n_chunks} central 1024-px chunks of the 800x800 view ({hits} hit rays, {dt:.1f} s, "
                      f"oracle = PyTorch-CPU restatement of the reference, torch threads = {cores})",
            "hit_rays_per_s": hits / dt}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # ROBIR_SHARE_GPU=1 + ROBIR_DIST_BACKEND=gloo: all ranks on cuda:0 (lets the multi-rank code path be exercised on a
    # 1-GPU box; RCCL refuses two ranks on one device).  Normal runs: one rank per GPU over RCCL ("nccl" on ROCm).
    if os.environ.get("ROBIR_SHARE_GPU") == "1":
        local = 0
    backend = os.environ.get("ROBIR_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from robir_amd import ops, renderer, synth, sg_render
    from robir_amd.parallel import all_gather_tiles
    if "ROBIR_VIS_PRECISION" not in os.environ:
        sg_render.VIS_PRECISION = args.vis_precision
    precision = sg_render.VIS_PRECISION
    if precision == "fp32":        # exact f32-input MFMA everywhere, not only in the visibility kernel
        os.environ.setdefault("ROBIR_MLP_PRECISION", "fp32")
    t0 = time.time()
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):     # the octree build prints like the reference; stdout carries only the JSON line
        model = renderer.build_synthetic_model(dev, seed=0, variance=0.3)
    torch.cuda.synchronize()
    build_s = time.time() - t0

    timer = KernelTimer()
    ops.dvis_fused = timer.wrap(ops.dvis_fused)
    uv, _, K = synth.synth_camera(H, W)
    pose = view_pose(rank)
    uv_d, pose_d, K_d = torch.from_numpy(uv).to(dev), torch.from_numpy(pose).to(dev), torch.from_numpy(K).to(dev)
    hdr = torch.full((H * W, 1), 0.5, device=dev)
    stats = {}

    def step():
        tiles = render_image(model, uv_d, pose_d, K_d, hdr, args.chunks_per_batch, stats)
        return all_gather_tiles(tiles) if world > 1 else tiles

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    stats.clear()
    timer.on = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    timer.on = False
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    rays_total = world * H * W * args.steps
    hit_frac = float(out[: H * W, 16].mean())
    evals = int(stats["diffuse_vis_evals"]) if "diffuse_vis_evals" in stats else 0
    k_ms, k_n = timer.stats()
    flops_per_launch = 2.0 * VIS_MACS_PER_EVAL * evals / max(k_n, 1)
    achieved = flops_per_launch / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0

    if rank == 0:
        h3 = precision.startswith("f16x3")
        # peak of the pipe the dominant kernel runs on, per ALGORITHMIC flop: exact mode = dense f32-input MFMA;
        # f16x3 = dense f16 MFMA (2.5 PFLOP/s) / 3 products per algorithmic multiply-add
        peak = PEAK_F16_MFMA_TFLOPS / 3.0 if h3 else PEAK_FP32_MFMA_TFLOPS
        roofline = {"bound": "mfma", "kernel": ("k_dvis_v2" if precision == "f16x3-v2" else "k_dvis_fused") + " (light-SG visibility MLP)", "achieved": achieved, "peak": peak,
                    "unit": "TFLOP/s", "frac": achieved / peak,
                    # HBM bytes per launch: 30 B per (point, direction) pair measured with rocprofv3 --pmc FETCH_SIZE (x2 gfx950
                    # correction) + WRITE_SIZE on this kernel at 16 and at 128 chunks per launch (28-31 B/pair,
                    # profiles/r01_dvis_f16x3_pmc.md), scaled to this launch's pair count: ~0.4 % of the HBM roofline --
                    # the bound is the matrix pipe
                    "traffic": 30.0 * evals / max(k_n, 1), "traffic_unit": "B/launch (scaled from the PMC profile)",
                    "precision": precision,
                    "peak_note": ("dense f16 MFMA 2500 TFLOP/s / 3 (hi*hi, hi*lo, lo*hi products per multiply-add)" if h3
                                  else "dense f32-input MFMA"),
                    "frac_of_fp32_mfma_peak": achieved / PEAK_FP32_MFMA_TFLOPS,
                    "launches": k_n, "avg_launch_ms": k_ms, "evals_per_launch": evals / max(k_n, 1),
                    "flops_per_eval": 2 * VIS_MACS_PER_EVAL}
        line = {
            "metric": "PBR-stage rays/sec (128 SG lobes, 32 visibility samples/lobe), full forward render",
            "value": rays_total / dt, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if precision == "fp32" else "f32 (MLP layers as 3x f16 hi/lo-split MFMA products, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": "hotdog-like synthetic 800x800 full PBR forward (BASELINE.json configs[3]): "
                                   "625 lock-step chunks of 1024 px per view, one view per GPU",
                       "image": [H, W], "chunk": CHUNK, "chunks_per_pass": args.chunks_per_batch,
                       "hit_fraction": round(hit_frac, 4), "octree_build_s": round(build_s, 2),
                       "octree_nodes": model.ray_tracer.sdf_octree.tables.B, "parallelism": f"ray-shard x{world} (views)"},
            "roofline": roofline,
        }
        if world == 1 and precision != "fp32" and not args.no_exact:
            # Outside the timed region: the same view with every MLP on the exact f32-input MFMA and the same random
            # draws -- the conservative rate, and how far the split-precision image is from it.
            torch.manual_seed(20260928)
            split_img = step()
            sg_render.VIS_PRECISION = "fp32"
            os.environ["ROBIR_MLP_PRECISION"] = "fp32"
            torch.manual_seed(20260928)
            step()                                      # packs the fp32 weight blobs
            torch.cuda.synchronize()
            torch.manual_seed(20260928)
            t1 = time.perf_counter()
            exact_img = step()
            torch.cuda.synchronize()
            t_exact = time.perf_counter() - t1
            ok = torch.isfinite(exact_img).all(-1) & torch.isfinite(split_img).all(-1)
            a, b = split_img[ok], exact_img[ok]
            rel = (a - b).abs() / (b.abs() + b.abs().mean(0, keepdim=True))
            line["exact_fp32"] = {"value": H * W / t_exact, "unit": "rays/s", "ms_per_step": t_exact * 1e3,
                                  "max_rel_diff_of_split_precision_image": float(rel.max()),
                                  "frac_entries_beyond_1e-4": float((rel > 1e-4).float().mean()),
                                  "rays_beyond_1e-3": int((rel > 1e-3).any(-1).sum()),
                                  "note": "same view and draws, every MLP on v_mfma_f32_16x16x4_f32; diff = |a-b|/(|b|+mean|b|) "
                                          "over the 17 output channels of all rays; the outliers are rays where a sampled "
                                          "direction sits on the n.d > 1e-6 cull (a 1e-6 change of the normal flips one "
                                          "visibility sample), the same effect as between two exact runs on different hardware"}
            sg_render.VIS_PRECISION = precision
            os.environ["ROBIR_MLP_PRECISION"] = "f16x3"
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(model, args.cpu_baseline_chunks, uv, pose, K)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    main()
