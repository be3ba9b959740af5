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
One step = ONE such image rendered by all N GPUs together ("strong": rank r renders chunks {c : c mod N = r} of the view,
robir_amd.parallel.render_view_sharded, and the 17-float tiles are all-gathered over RCCL at the end of the step -- the
partition SURVEY.md 8e / north_star name).  `value` = rays of that image / step time.  For N > 1 a secondary figure times the
weak form (rank r renders its own view r).  The octree build is one-off set-up and is reported separately.

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
TRAFFIC_B_PER_PAIR = 24.0          # profiles/r02_dvis_pmc.md: (2 x FETCH_SIZE + WRITE_SIZE) / pairs of k_dvis_v2 at 32 chunks per launch (r01: 28-31)
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
                    help="skip the exact-fp32 leg (N = 1): the same view with every MLP on the f32-input MFMA, timed over "
                         "--exact-steps steps of its own, and the image distance of the default run from it")
    ap.add_argument("--exact-steps", type=int, default=3)
    ap.add_argument("--cpu-baseline-1thread-pixels", type=int, default=256,
                    help="pixels of the single-thread CPU baseline sample (the runners force torch.set_num_threads(1))")
    ap.add_argument("--vis-precision", default="f16x3-auto", choices=["fp32", "f16x3-auto", "f16x3-v4", "f16x3-v3", "f16x3-v2", "f16x3", "f16x3-regstage", "f16x3-nt2"],
                    help="hidden layers of the fused light-visibility kernel: exact f32-input MFMA, or the error-compensated "
                         "hi/lo half split on the f16 MFMA (fp32 accumulate, same measured parity)")
    ap.add_argument("--vis", default="mlp", choices=["mlp", "octree"],
                    help="light / BRDF-lobe visibility model: the visibility MLP (default, the reference's default) or traced "
                         "visibility = OctreeVisModel(octree_ray_tracer), the reference's `trace_vis` switch "
                         "(training/train_pbr.py:409-410): the MLP's 98 %% of the PBR FLOPs become octree gathers")
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


def cpu_baseline(model, n_chunks, uv, pose, K, cores=None, pixels=CHUNK):
    """Oracle (CPU restatement of the reference) on `n_chunks` central chunks of `pixels` pixels of the same image."""
    from robir_oracle import nets as on, octree as ooct, renderer as orend
    from robir_amd import synth
    if cores is None:
        cores = min(16, os.cpu_count() or 1)  # more threads make the many small tensor ops of the reference slower
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
        sl = slice(c * CHUNK, c * CHUNK + pixels)
        uv_t = torch.from_numpy(uv[sl])[None]
        dirs, cam = orend.camera_rays(uv_t, torch.from_numpy(pose)[None], torch.from_numpy(K)[None])
        _, hit, _ = ooct.trace(T, cam, dirs, -1)
        n_hit = int(hit.sum())
        dr = {k: torch.from_numpy(v) for k, v in synth.pbr_draws(7, n_hit, chunk_id=c).items()}
        orend.forward(sd, T, uv_t, torch.from_numpy(pose)[None], torch.from_numpy(K)[None],
                      torch.ones(1, pixels, dtype=torch.bool), torch.full((pixels, 1), 0.5), dr, "Material", testing=True)
        rays += pixels
        hits += n_hit
    dt = time.time() - t0
    return {"value": rays / dt, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": f"{n_chunks} central {pixels}-px chunks of the 800x800 view ({hits} hit rays, {dt:.1f} s, "
                      f"oracle = PyTorch-CPU restatement of the reference, torch threads = {cores})",
            "hit_rays_per_s": hits / dt}


class PowerProbe:
    """Samples `rocm-smi --showpower` in a thread while the timed region runs: the dominant kernel sits on the package power
    cap, and the bench line should say so with a number.  Best effort: no rocm-smi, no field."""

    def __init__(self, period=0.4):
        import re
        import shutil
        import subprocess
        import threading
        self.samples, self.cap, self._stop = [], None, threading.Event()
        exe = shutil.which("rocm-smi") or ("/opt/rocm/bin/rocm-smi" if os.path.exists("/opt/rocm/bin/rocm-smi") else None)
        dev = int(os.environ.get("LOCAL_RANK", "0"))

        def read(extra):
            out = subprocess.run([exe, "-d", str(dev), "--showpower"] + extra, capture_output=True, text=True, timeout=3).stdout
            m = re.search(r"Package Power \(W\): ([0-9.]+)", out.split("Max")[0] if "Max" in out else out)
            c = re.search(r"Max Graphics Package Power \(W\): ([0-9.]+)", out)
            return (float(m.group(1)) if m else None), (float(c.group(1)) if c else None)

        def loop():
            try:
                _, self.cap = read(["--showmaxpower"])
                while not self._stop.wait(period):
                    w, _ = read([])
                    if w is not None:
                        self.samples.append(w)
            except Exception:
                pass

        self._thread = threading.Thread(target=loop, daemon=True) if exe else None
        if self._thread:
            self._thread.start()

    def stop(self):
        if not self._thread:
            return None
        self._stop.set()
        self._thread.join(timeout=5)
        if not self.samples:
            return None
        s = sorted(self.samples)
        return {"cap_w": self.cap, "max_w": s[-1], "median_w": s[len(s) // 2], "samples": len(s),
                "note": "rocm-smi package power sampled during the timed steps (samples between two launches read lower)"}


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

    from robir_amd import ops, renderer, synth, sg_render, parallel
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
    if args.vis == "octree":
        from robir_amd.octree_tracing import OctreeVisModel
        model.visibility_network = OctreeVisModel(model.octree_ray_tracer)
        ops.dvis_octree = timer.wrap(ops.dvis_octree)
    else:
        ops.dvis_fused = timer.wrap(ops.dvis_fused)
    uv, _, K = synth.synth_camera(H, W)
    pose = view_pose(0)                                  # every rank works on the SAME view: the N = 1 workload
    uv_d, pose_d, K_d = torch.from_numpy(uv).to(dev), torch.from_numpy(pose).to(dev), torch.from_numpy(K).to(dev)
    hdr = torch.full((H * W, 1), 0.5, device=dev)
    stats = {}
    plan = parallel.plan_view(H * W, CHUNK, dev, interleave=True, chunks_per_pass=args.chunks_per_batch)

    def step():
        """One image: this rank's chunks (all of them at N = 1), then the all-gather of the tiles -> [H*W, 17] on every rank."""
        return parallel.render_view_sharded(model, uv_d, pose_d, K_d, hdr, CHUNK, stats=stats, plan=plan)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps):
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = fn()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax)
        return dt, out

    for _ in range(args.warmup):
        step()
    barrier()
    stats.clear()
    timer.on = True
    power = PowerProbe() if rank == 0 else None      # package power while the timed steps run (rocm-smi, best effort)
    dt, out = timed(step, args.steps)
    power_line = power.stop() if power is not None else None
    timer.on = False
    ops.range_check(sync=True)                           # split-precision activation-range sentinel: raises on overflow
    rays_total = H * W * args.steps
    hit_frac = float(out[:, 16].mean())
    evals = int(stats["diffuse_vis_evals"]) if "diffuse_vis_evals" in stats else 0
    k_ms, k_n = timer.stats()
    flops_per_launch = 2.0 * VIS_MACS_PER_EVAL * evals / max(k_n, 1)
    achieved = flops_per_launch / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0

    weak = None
    if world > 1:
        # secondary figure: the weak form (rank r renders its own whole view r, tiles all-gathered)
        pose_r = torch.from_numpy(view_pose(rank)).to(dev)
        plan1 = dict(parallel.plan_view(H * W, CHUNK, dev), world=1)
        plan1["passes"] = [(list(range(H * W // CHUNK)), slice(0, H * W), False)]

        def weak_step():
            t = parallel.render_view_sharded(model, uv_d, pose_r, K_d, hdr, CHUNK, plan=plan1)
            return parallel.all_gather_tiles(t)
        weak_step()
        wdt, _ = timed(weak_step, 2)
        weak = {"value": world * H * W * 2 / wdt, "unit": "rays/s", "steps": 2, "ms_per_step": wdt / 2 * 1e3,
                "note": "one whole view per GPU (views differ in cost: hit fraction 0.65 .. 0.53)"}

    octree_line = None
    if args.vis == "octree":
        lay = [int(v) for v in ops.LAST_OCTREE_VIS_LAYOUT.cpu()]
        pairs, fetches, ray_steps = lay[0], lay[2], lay[3]
        # algorithmic bytes of one traced-visibility launch: the 32-byte octree records its lock-step iterations read, plus the
        # per-pair state written once and read once (pair 6 B, t 4, leaf 4, active 1, group 4) and the active flag re-read by
        # every one of the 33 iterations
        bytes_launch = 32.0 * fetches + pairs * (2 * 19.0 + 33.0)
        octree_line = {"bound": "hbm", "kernel": "k_ovis_iter x33 + cull / fill / reduce (traced light visibility)",
                       "achieved": bytes_launch / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0, "peak": 8000.0, "unit": "GB/s",
                       "traffic": None, "launches": k_n, "avg_launch_ms": k_ms, "pairs_per_launch": pairs,
                       "octree_records_read_per_ray": fetches / max(pairs, 1), "iterations_per_ray": ray_steps / max(pairs, 1),
                       "gathered_bytes_per_ray": 32.0 * fetches / max(pairs, 1),
                       "note": "gathers are dependent 32-byte reads of a 29 MB table (L2 / MALL resident): the bound is "
                               "latency x occupancy, not HBM bandwidth; frac is against the HBM peak as the metric asks"}
        octree_line["frac"] = octree_line["achieved"] / 8000.0
    if rank == 0:
        h3 = precision.startswith("f16x3")
        # peak of the pipe the dominant kernel runs on, per ALGORITHMIC flop: exact mode = dense f32-input MFMA;
        # f16x3 = dense f16 MFMA (2.5 PFLOP/s) / 3 products per algorithmic multiply-add
        peak = PEAK_F16_MFMA_TFLOPS / 3.0 if h3 else PEAK_FP32_MFMA_TFLOPS
        roofline = {"bound": "mfma", "kernel": ops.DVIS_KERNEL_NAMES.get(precision, "k_dvis_fused") + " (light-SG visibility MLP)",
                    "achieved": achieved, "peak": peak,
                    "unit": "TFLOP/s", "frac": achieved / peak,
                    # HBM bytes per launch: B per (point, direction) pair measured with rocprofv3 --pmc FETCH_SIZE (x2 gfx950
                    # correction) + WRITE_SIZE on this kernel (profiles/), scaled to this launch's pair count: well under 1 % of
                    # the HBM roofline -- the bound is the matrix pipe
                    "traffic": TRAFFIC_B_PER_PAIR * evals / max(k_n, 1), "traffic_unit": "B/launch (scaled from the PMC profile)",
                    "precision": precision,
                    "peak_note": ("dense f16 MFMA 2500 TFLOP/s / 3 (hi*hi, hi*lo, lo*hi products per multiply-add)" if h3
                                  else "dense f32-input MFMA"),
                    "frac_of_fp32_mfma_peak": achieved / PEAK_FP32_MFMA_TFLOPS,
                    "launches": k_n, "avg_launch_ms": k_ms, "evals_per_launch": evals / max(k_n, 1),
                    "flops_per_eval": 2 * VIS_MACS_PER_EVAL}
        if power_line is not None:
            roofline["package_power"] = power_line
        line = {
            "metric": "PBR-stage rays/sec (128 SG lobes, 32 visibility samples/lobe), full forward render",
            "value": rays_total / dt, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if precision == "fp32" else "f32 (MLP layers as 3x f16 hi/lo-split MFMA products, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": "hotdog-like synthetic 800x800 full PBR forward (BASELINE.json configs[3]): ONE view = "
                                   "625 lock-step chunks of 1024 px, chunks sharded over the GPUs (c mod N), tiles all-gathered",
                       "image": [H, W], "chunk": CHUNK, "chunks_per_pass": args.chunks_per_batch,
                       "hit_fraction": round(hit_frac, 4), "octree_build_s": round(build_s, 2),
                       "octree_nodes": model.ray_tracer.sdf_octree.tables.B,
                       "parallelism": f"chunk-shard x{world} of one view + all-gather"},
            "roofline": roofline if octree_line is None else octree_line,
        }
        if args.vis == "octree":
            line["metric"] = "PBR-stage rays/sec (128 SG lobes, 32 visibility samples/lobe), traced visibility (trace_vis)"
            line["config"]["visibility"] = "OctreeVisModel (secondary lock-step octree cast, max_iter 32, 2 M-pair batches)"
        if weak is not None:
            line["weak_views"] = weak
    if world == 1 and precision != "fp32" and not args.no_exact and args.vis == "mlp":
        # Second first-class figure: the same view with EVERY MLP on the exact f32-input MFMA (v_mfma_f32_16x16x4_f32),
        # timed over its own loop -- the rate at the reference's own precision -- and how far the default image is from it
        # under identical random draws.
        torch.manual_seed(20260928)
        split_img = step()
        sg_render.VIS_PRECISION = "fp32"
        os.environ["ROBIR_MLP_PRECISION"] = "fp32"
        torch.manual_seed(20260928)
        exact_img = step()                              # warm-up: packs the fp32 weight blobs
        timer.pairs.clear()
        timer.on = True
        stats.clear()
        t_exact, _ = timed(step, args.exact_steps)
        timer.on = False
        e_ms, e_n = timer.stats()
        e_evals = int(stats["diffuse_vis_evals"]) if "diffuse_vis_evals" in stats else 0
        e_tflops = 2.0 * VIS_MACS_PER_EVAL * e_evals / max(e_n, 1) / (e_ms * 1e-3) / 1e12 if e_ms > 0 else 0.0
        ok = torch.isfinite(exact_img).all(-1) & torch.isfinite(split_img).all(-1)
        a, b = split_img[ok], exact_img[ok]
        rel = (a - b).abs() / (b.abs() + b.abs().mean(0, keepdim=True))
        line["exact_fp32"] = {"value": H * W * args.exact_steps / t_exact, "unit": "rays/s", "steps": args.exact_steps,
                              "warmup": 1, "ms_per_step": t_exact / args.exact_steps * 1e3, "dtype": "f32",
                              "roofline": {"bound": "mfma", "kernel": "k_dvis_fused<fp32>", "achieved": e_tflops,
                                           "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                           "frac": e_tflops / PEAK_FP32_MFMA_TFLOPS, "avg_launch_ms": e_ms, "launches": e_n},
                              "max_rel_diff_of_split_precision_image": float(rel.max()),
                              "frac_entries_beyond_1e-4": float((rel > 1e-4).float().mean()),
                              "rays_beyond_1e-3": int((rel > 1e-3).any(-1).sum()),
                              "note": "same view and draws, every MLP on v_mfma_f32_16x16x4_f32; diff = |a-b|/(|b|+mean|b|) "
                                      "over the 17 output channels of all rays; the outliers are rays where a sampled "
                                      "direction sits on the n.d > 1e-6 cull (tests/test_precision_gpu.py anchors both modes "
                                      "on a float64 evaluation of the reference's formulas)"}
        sg_render.VIS_PRECISION = precision
        os.environ["ROBIR_MLP_PRECISION"] = "f16x3"
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(model, args.cpu_baseline_chunks, uv, pose, K)
            # the reference's runners pin torch to one thread (training/train_pbr.py:24): that figure too, on a smaller sample
            one = cpu_baseline(model, 1, uv, pose, K, cores=1, pixels=args.cpu_baseline_1thread_pixels)
            line["cpu_baseline"]["single_thread"] = {k: one[k] for k in ("value", "unit", "cores", "sample", "hit_rays_per_s")}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    main()
