#!/usr/bin/env python
"""Throughput of the BASELINE.json configurations that are not the bench.py line (they are parity-test cases; this
script times them for DESIGN.md and is the workload of their rocprofv3 summaries under profiles/).
`python tools/bench_configs.py [config ...]` on one GPU, synthetic weights (default: 1 2 3 5):
  config 1  SDF network forward on a 64x64 crop x 64 samples
  config 2  NeuS ray-march 400x400, 64+64 samples per ray (render_neus, hierarchical sampling, colour net)
  config 3  800x800 'Illum' forward + trace_radiance(nsamp=8)  (secondary rays, borrow_color, visibility MLP)
  config 5  CESR hook (shadow_net x 128 labels, normal_net, 8-sample light visibility) on a band of a 1600x1200 view
RB_CONFIG_REPS=n limits the timed repetitions (profiling runs use 1)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robir_amd import nets, ops, renderer, sdf_render, synth  # noqa: E402

dev = torch.device("cuda:0")
REPS = int(os.environ.get("RB_CONFIG_REPS", "3"))


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(min(reps, REPS)):
        t0 = time.time()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.time() - t0)
    return best


def config1(model):
    neus = model.implicit_network.neus_model
    uv, pose, K = synth.synth_camera(64, 64)
    dirs = ops.camera_rays(pose, K, torch.from_numpy(uv).to(dev))
    z = torch.linspace(0.8, 2.8, 64, device=dev)
    pts = ((torch.from_numpy(pose[:3, 3]).to(dev) * 2.0)[None, None, :] + z[None, :, None] * dirs[:, None, :]).reshape(-1, 3).contiguous()
    t = timed(lambda: neus.sdf_network(pts))
    print(f"config 1: SDF forward, {pts.shape[0]} points (64x64 rays x 64 samples): {t * 1e3:.2f} ms = {4096 / t:.3g} rays/s, "
          f"{pts.shape[0] / t:.3g} points/s")


def config2(model):
    neus = model.implicit_network.neus_model
    uv, pose, K = synth.synth_camera(400, 400)
    dirs = ops.camera_rays(pose, K, torch.from_numpy(uv).to(dev))
    R = dirs.shape[0]
    ro = (torch.from_numpy(pose[:3, 3]).to(dev) * 2.0).expand(R, 3).contiguous()
    near, far = torch.full((R, 1), 0.8, device=dev), torch.full((R, 1), 2.8, device=dev)
    rays = sdf_render.Rays(ro, dirs, dirs, None, None, near, far)
    t = timed(lambda: sdf_render.render_neus(rays, neus, 1.0, n_samples=64, n_importance=64, up_sample_steps=4, is_eval=True), reps=2)
    # 112 SDF evaluations for the sampling + 128 x (SDF+features, gradient, colour) = 0.59 GFLOP per ray
    print(f"config 2: render_neus 400x400, 128 samples/ray: {t:.3f} s = {R / t:.3g} rays/s = {0.59e9 * R / t / 1e12:.0f} "
          "algorithmic TFLOP/s")
    if os.environ.get("RB_CONFIG_REPS") == "1":
        return                       # profiling run: the default path only
    # the same without the (unused in stage 2) eikonal term: gradient + colour only where the weight is non-zero -- identical
    # rgb / dist / acc / grad / weights.  Depends on the sharpness of the SDF: the untrained init (variance 0.3, inv_s 20)
    # keeps every sample, a trained-like sharpness (variance 0.6, inv_s 403) drops the samples behind the surface
    var = neus.deviation_network.variance
    old = float(var)
    for v in (old, 0.6):
        var.fill_(v)
        o = sdf_render.render_neus(rays, neus, 1.0, is_eval=True, need_grad_error=False)
        kept = float((o["weights"] != 0).float().mean())
        t0 = timed(lambda: sdf_render.render_neus(rays, neus, 1.0, is_eval=True), reps=2)
        t1 = timed(lambda: sdf_render.render_neus(rays, neus, 1.0, is_eval=True, need_grad_error=False), reps=2)
        print(f"config 2 (variance {v:g}, inv_s {neus.inv_s():.0f}): full {t0:.3f} s = {R / t0:.3g} rays/s; without grad_error "
              f"{t1:.3f} s = {R / t1:.3g} rays/s ({100 * kept:.0f} % of the samples carry weight)")
    var.fill_(old)


def config3(model):
    uv, pose, K = synth.synth_camera(800, 800)
    uv_d, pose_d, K_d = torch.from_numpy(uv).to(dev), torch.from_numpy(pose).to(dev), torch.from_numpy(K).to(dev)
    N = uv_d.shape[0]
    hdr = torch.full((N, 1), 0.5, device=dev)

    def illum():
        o = model.render_chunks(uv_d, pose_d, K_d, hdr, chunk=1024, trainstage="Illum")
        o["hdr_shift"] = hdr
        return model.trace_radiance(o, nsamp=8, chunk=1024)      # every chunk its own lock-step batch, like the reference

    t = timed(illum, reps=2)
    print(f"config 3: 800x800 Illum forward + trace_radiance(nsamp=8): {t:.3f} s = {N / t:.3g} primary rays/s")


def config5(model):
    # band of 125 chunks through the image centre, chunk by chunk like the reference's forward()
    c = synth.synth_cesr_nets(0)
    shadow = nets.SDFNetwork(63 + 128, 2, 512, 8, [4], 0)
    normal = nets.SDFNetwork(63, 3, 512, 8, [4], 0)
    shadow.load_state_dict({k: torch.from_numpy(v) for k, v in c["shadow_net"].items()})
    normal.load_state_dict({k: torch.from_numpy(v) for k, v in c["normal_net"].items()})
    model.get_sg_render = renderer.CESRHook(model, shadow.to(dev).eval(), normal.to(dev).eval(), is_training=False,
                                            cur_iter=100000, prefit="explore")
    uv, pose, K = synth.synth_camera(1200, 1600)
    pose_d, K_d = torch.from_numpy(pose).to(dev)[None], torch.from_numpy(K).to(dev)[None]
    first, nch = 875, 125
    uv_b = torch.from_numpy(uv[first * 1024:(first + nch) * 1024]).to(dev)

    def cesr():
        hits = []
        for i in range(nch):
            inp = {"uv": uv_b[None, i * 1024:(i + 1) * 1024], "pose": pose_d, "intrinsics": K_d,
                   "object_mask": torch.ones(1, 1024, dtype=torch.bool, device=dev), "hdr_shift": torch.full((1024, 1), 0.5, device=dev)}
            o = model(inp, trainstage="Material", lin_diff=True, train_spec=True)
            hits.append(o["network_object_mask"].sum())
        return int(torch.stack(hits).sum())

    h = cesr()
    t = timed(cesr, reps=2)
    print(f"config 5: CESR forward, {nch} central chunks of 1600x1200 ({h} hit rays): {t:.3f} s = {nch * 1024 / t:.3g} rays/s, "
          f"{h / t:.3g} hit rays/s")
    model.__dict__.pop("get_sg_render", None)


if __name__ == "__main__":
    only = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 5]
    with torch.no_grad():
        m = renderer.build_synthetic_model(dev)
        for k in only:
            {1: config1, 2: config2, 3: config3, 5: config5}[k](m)
