"""Times the ACTUAL reference (/root/reference under oracle/ref_shim.py) on CPU in the build container -- BASELINE.md section 3(1):
the full PBR forward (`IDRNetwork.forward('Material')` with the PBR runner's hook, 1024-pixel chunks in raster order) on the synthetic
64 x 64 view and on central chunks of the 400 x 400 view, with 1 thread (the runners pin torch to one, training/train_pbr.py:24) and with
all host cores; one warm-up, then the MEDIAN of >= 3 runs; the octree build is excluded and listed separately.

TEST INFRASTRUCTURE (build container only).  Writes profiles/reference_cpu_container.json.
    python oracle/time_reference_cpu.py [--runs 3] [--chunks400 2]
"""
import argparse
import json
import os
import platform
import statistics
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_shim  # noqa: E402

ref_shim.install()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--chunks400", type=int, default=2)
    args = ap.parse_args()
    import gen_golden as g1
    from robir_amd import synth
    cores = os.cpu_count() or 1
    sd_np = synth.synth_state_dict(0, variance=0.3)
    try:
        cpu = open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t")
    except (OSError, IndexError):
        cpu = platform.processor()
    out = {"host": {"cpu": cpu,
                    "cores": cores, "torch": torch.__version__},
           "what": "the reference's own IDRNetwork.forward('Material') + PBRTrainRunner.get_sg_render through oracle/ref_shim.py, synthetic "
                   "weights (robir_amd.synth seed 0), 1024-px chunks, torch's own random draws; median of the timed runs after one warm-up",
           "runs": args.runs, "cases": []}
    with ref_shim.CpuMode():
        torch.set_num_threads(cores)
        net = g1.build_reference(sd_np, "v03")
        g1.install_pbr_hook(net)
        impl = net.implicit_network
        sdf_fn = lambda x: impl(x)[:, 0]      # noqa: E731
        t0 = time.time()
        net.ray_tracer.generate(sdf_fn)
        out["octree_build_s"] = {"threads": cores, "seconds": time.time() - t0, "nodes": int(net.ray_tracer.sdf_octree.boxes.shape[0])
                                 if hasattr(net.ray_tracer.sdf_octree, "boxes") else None}
        net.octree_ray_tracer.sdf_octree = net.ray_tracer.sdf_octree

        def forward_chunks(H, W, chunk_ids):
            uv, pose, K = synth.synth_camera(H, W)
            rays = hits = 0
            for c in chunk_ids:
                sl = slice(c * 1024, (c + 1) * 1024)
                n = uv[sl].shape[0]
                inp = {"uv": torch.from_numpy(uv[sl])[None], "pose": torch.from_numpy(pose)[None], "intrinsics": torch.from_numpy(K)[None],
                       "object_mask": torch.ones(1, n, dtype=torch.bool), "hdr_shift": torch.full((n, 1), 0.5)}
                with torch.no_grad():
                    o = net(inp, trainstage="Material", train_spec=True)
                rays += n
                hits += int(o["network_object_mask"].sum())
            return rays, hits

        cases = [("64x64 view, all 4 chunks", 64, 64, [0, 1, 2, 3])]
        n400 = 400 * 400 // 1024
        first = (200 * 400) // 1024 - args.chunks400 // 2
        cases.append((f"400x400 view, {args.chunks400} central chunks of {n400 + 1}", 400, 400, list(range(first, first + args.chunks400))))
        for name, H, W, ids in cases:
            for threads in (cores, 1):
                torch.set_num_threads(threads)
                torch.manual_seed(0)
                forward_chunks(H, W, ids[:1])                  # warm-up
                ts = []
                for _ in range(args.runs):
                    t0 = time.time()
                    rays, hits = forward_chunks(H, W, ids)
                    ts.append(time.time() - t0)
                med = statistics.median(ts)
                out["cases"].append({"case": name, "threads": threads, "rays": rays, "hit_rays": hits, "seconds_runs": ts, "seconds_median": med,
                                     "rays_per_s": rays / med, "hit_rays_per_s": hits / med})
                print(out["cases"][-1], flush=True)
                json.dump(out, open(os.path.join(ROOT, "profiles", "reference_cpu_container.json"), "w"), indent=1)
    print("wrote profiles/reference_cpu_container.json")


if __name__ == "__main__":
    main()
