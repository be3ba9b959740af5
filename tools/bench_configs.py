#!/usr/bin/env python
"""The BASELINE.json configurations other than the headline one (configs[1..3, 5] in the order of SURVEY.md 8d; the
headline, config 4, is bench.py's own step).  Each function times its workload on the current device with synthetic
weights and returns a dict with the rate and a `roofline` for the op that dominates it, measured with HIP events around
every call of that op on torch's current stream (the stream the kernels are launched on):

  config 1  SDF network forward on a 64x64 crop x 64 samples                      (A1, A2)
  config 2  NeuS ray-march 400x400, 64+64 samples per ray, hierarchical sampling   (A1-A6)   op: SDF net (PE + value [+ gradient])
  config 3  800x800 'Illum' forward + trace_radiance(nsamp=8) per 1024-px chunk    (+A7, A13-A15, A20)   op: SDF net
  config 5  CESR hook on a 1600x1200 view, chunk by chunk like the runner, trace_radiance(nsamp=8) per chunk
            (training/train_cesr.py:321-326)                                       op: shadow_net over 128 labels

`python tools/bench_configs.py [config ...]` prints one line per config (used for the rocprofv3 summaries under
profiles/); bench.py imports the functions for its `--config N` lines and the `configs` extras of the default line.
RB_CONFIG_REPS=n limits the timed repetitions (profiling runs use 1)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from robir_amd import nets, ops, renderer, sdf_render, synth  # noqa: E402

REPS = int(os.environ.get("RB_CONFIG_REPS", "3"))
PEAK_FP32_MFMA = 157.3          # TFLOP/s, dense f32-input MFMA (MI355X_MICROARCH.md)
PEAK_F16_MFMA = 2500.0          # TFLOP/s, dense f16 MFMA
# algorithmic MACs per evaluation, SURVEY.md 8d
MAC_SDF_ONLY, MAC_SDF_FULL, MAC_SDF_GRAD, MAC_SHADOW = 459008, 524544, 524544, 1836032


class OpTimer:
    """HIP events around every call of one op + its algorithmic flops."""

    def __init__(self):
        self.pairs, self.flops, self.on = [], 0.0, False

    def clear(self):
        self.pairs, self.flops = [], 0.0

    def record(self, fn, flops):
        if not self.on:
            return fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = fn()
        e.record()
        self.pairs.append((s, e))
        self.flops += flops
        return out

    def roofline(self, kernel, precision, x6=False):
        ms = sum(s.elapsed_time(e) for s, e in self.pairs)
        split = precision == "f16x3"
        if precision == "f16x1":      # plain f16: ONE MFMA product per multiply-add (csrc/cesr_f16.hip) -- NARROWER than fp32
            ach = self.flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            return {"bound": "mfma", "kernel": kernel, "achieved": ach, "peak": PEAK_F16_MFMA, "unit": "TFLOP/s", "frac": ach / PEAK_F16_MFMA,
                    "traffic": None, "calls": len(self.pairs), "op_ms_total": ms,
                    "peak_note": "dense f16 MFMA 2500 TFLOP/s, one product per multiply-add (plain f16 operands: NARROWER than fp32, a labelled throughput mode)",
                    "flops_note": "algorithmic MACs of SURVEY.md 8d x evaluations executed"}
        if x6 and not split:      # the op runs on exact three-piece operands: six f16 MFMA products per multiply-add
            ach = self.flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            peak = PEAK_F16_MFMA / 6.0
            return {"bound": "mfma", "kernel": kernel, "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                    "traffic": None, "calls": len(self.pairs), "op_ms_total": ms,
                    "peak_note": "dense f16 MFMA 2500 / 6 products per multiply-add (exact three-piece operands: not narrower than fp32)",
                    "flops_note": "algorithmic MACs of SURVEY.md 8d x evaluations executed"}
        peak = PEAK_F16_MFMA / 3.0 if split else PEAK_FP32_MFMA
        ach = self.flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        return {"bound": "mfma", "kernel": kernel, "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                "traffic": None, "calls": len(self.pairs), "op_ms_total": ms,
                "peak_note": ("dense f16 MFMA 2500 / 3 products per multiply-add (22-bit operand pairs)" if split
                              else "dense f32-input MFMA"),
                "flops_note": "algorithmic MACs of SURVEY.md 8d x evaluations executed"}


SDF_TIMER, SHADOW_TIMER = OpTimer(), OpTimer()


def _cesr_precision():
    from robir_amd.precision import cesr_precision
    return cesr_precision()
_installed = False


def install_timers():
    """Wrap SDFNetwork.eval_points (NeuS shape: encoding + net + gradient) and .eval_point_labels (CESR shadow_net)."""
    global _installed
    if _installed:
        return
    _installed = True
    ev, el = nets.SDFNetwork.eval_points, nets.SDFNetwork.eval_point_labels

    def eval_points(self, x, in_scale=1.0, out_scale=1.0, full=True, grad=False, precise=False):
        mac = (MAC_SDF_FULL if full else MAC_SDF_ONLY) + (MAC_SDF_GRAD if grad else 0)
        return SDF_TIMER.record(lambda: ev(self, x, in_scale, out_scale, full, grad, precise), 2.0 * mac * x.shape[0])

    def eval_point_labels(self, Xp, n_label=128):
        return SHADOW_TIMER.record(lambda: el(self, Xp, n_label), 2.0 * MAC_SHADOW * Xp.shape[0] * n_label)

    nets.SDFNetwork.eval_points = eval_points
    nets.SDFNetwork.eval_point_labels = eval_point_labels


def timed(fn, reps=3, timers=()):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(max(1, min(reps, REPS))):
        for t in timers:
            t.clear()
            t.on = True
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        for t in timers:
            t.on = False
        best = min(best, dt)
    return best


def config1(model, reps=3):
    dev = next(model.parameters()).device
    neus = model.implicit_network.neus_model
    uv, pose, K = synth.synth_camera(64, 64)
    dirs = ops.camera_rays(pose, K, torch.from_numpy(uv).to(dev))
    z = torch.linspace(0.8, 2.8, 64, device=dev)
    pts = ((torch.from_numpy(pose[:3, 3]).to(dev) * 2.0)[None, None, :] + z[None, :, None] * dirs[:, None, :]).reshape(-1, 3).contiguous()
    t = timed(lambda: neus.sdf_network(pts), reps, (SDF_TIMER,))
    return {"config": 1, "workload": "SDF network forward (PE + 9 layers, 257 outputs), 64x64 rays x 64 samples",
            "value": 4096 / t, "unit": "rays/s", "ms": t * 1e3, "points_per_s": pts.shape[0] / t,
            "roofline": SDF_TIMER.roofline("SDF net (encoding + value rows)", nets.mlp_precision(), nets.mlp_precision() == "f16x6")}


def config2(model, reps=2):
    dev = next(model.parameters()).device
    neus = model.implicit_network.neus_model
    uv, pose, K = synth.synth_camera(400, 400)
    dirs = ops.camera_rays(pose, K, torch.from_numpy(uv).to(dev))
    R = dirs.shape[0]
    ro = (torch.from_numpy(pose[:3, 3]).to(dev) * 2.0).expand(R, 3).contiguous()
    near, far = torch.full((R, 1), 0.8, device=dev), torch.full((R, 1), 2.8, device=dev)
    rays = sdf_render.Rays(ro, dirs, dirs, None, None, near, far)
    t = timed(lambda: sdf_render.render_neus(rays, neus, 1.0, n_samples=64, n_importance=64, n_outside=0, up_sample_steps=4, is_eval=True),
              reps, (SDF_TIMER,))
    # 112 SDF evaluations for the sampling + 128 x (SDF+features, gradient, colour) = 0.59 GFLOP per ray
    return {"config": 2, "workload": "render_neus 400x400, 64+64 samples/ray, 4 up-sampling steps, colour net (Norm stage)",
            "value": R / t, "unit": "rays/s", "ms": t * 1e3, "algorithmic_tflops": 0.59e9 * R / t / 1e12,
            "roofline": SDF_TIMER.roofline("SDF net (encoding + value rows + reverse-mode gradient)", nets.mlp_precision(), nets.mlp_precision() == "f16x6")}


def config3(model, reps=2):
    dev = next(model.parameters()).device
    uv, pose, K = synth.synth_camera(800, 800)
    uv_d, pose_d, K_d = torch.from_numpy(uv).to(dev), torch.from_numpy(pose).to(dev), torch.from_numpy(K).to(dev)
    N = uv_d.shape[0]
    hdr = torch.full((N, 1), 0.5, device=dev)

    def illum():
        o = model.render_chunks(uv_d, pose_d, K_d, hdr, chunk=1024, trainstage="Illum")
        o["hdr_shift"] = hdr
        return model.trace_radiance(o, nsamp=8, chunk=1024)      # every chunk its own lock-step batch, like the reference

    t = timed(illum, reps, (SDF_TIMER,))
    return {"config": 3, "workload": "800x800 forward('Illum') + trace_radiance(nsamp=8), every 1024-px chunk its own lock-step batch",
            "value": N / t, "unit": "primary rays/s", "ms": t * 1e3,
            "roofline": SDF_TIMER.roofline("SDF net (borrow_color: value rows + reverse-mode gradient)", nets.mlp_precision(), nets.mlp_precision() == "f16x6")}


def config5(model, reps=1, first=875, nch=125, trace=True):
    """CESR stage on `nch` chunks (first chunk `first`) of the 1600x1200 view, chunk by chunk like train_cesr.py:319-326:
    forward('Material') with the CESR hook, then trace_radiance(out, nsamp=8).  first=0, nch=1875 is the whole view."""
    dev = next(model.parameters()).device
    c = synth.synth_cesr_nets(0)
    shadow = nets.SDFNetwork(63 + 128, 2, 512, 8, [4], 0)
    normal = nets.SDFNetwork(63, 3, 512, 8, [4], 0)
    shadow.load_state_dict({k: torch.from_numpy(v) for k, v in c["shadow_net"].items()})
    normal.load_state_dict({k: torch.from_numpy(v) for k, v in c["normal_net"].items()})
    model.get_sg_render = renderer.CESRHook(model, shadow.to(dev).eval(), normal.to(dev).eval(), is_training=False,
                                            cur_iter=100000, prefit="explore")
    uv, pose, K = synth.synth_camera(1200, 1600)
    pose_d, K_d = torch.from_numpy(pose).to(dev)[None], torch.from_numpy(K).to(dev)[None]
    uv_b = torch.from_numpy(uv[first * 1024:(first + nch) * 1024]).to(dev)
    ones = torch.ones(1, 1024, dtype=torch.bool, device=dev)
    hdr = torch.full((1024, 1), 0.5, device=dev)
    state = {}

    def cesr():
        hits, vis = [], []
        for i in range(nch):
            inp = {"uv": uv_b[None, i * 1024:(i + 1) * 1024], "pose": pose_d, "intrinsics": K_d, "object_mask": ones, "hdr_shift": hdr}
            o = model(inp, trainstage="Material", lin_diff=True, train_spec=True)
            if trace:
                tr = model.trace_radiance(o, nsamp=8)
                vis.append(tr["pred_vis"].argmax(-1).float().mean())
            hits.append(o["network_object_mask"].sum())
        state["hits"] = int(torch.stack(hits).sum())
        return state["hits"]

    try:
        t = timed(cesr, reps, (SHADOW_TIMER,))
    finally:
        model.__dict__.pop("get_sg_render", None)
    from robir_amd import deferred
    dc = int(model.__dict__.get("deferred_chunks", deferred.DEFAULT_CHUNKS))
    return {"config": 5, "workload": f"CESR forward + trace_radiance(nsamp=8) per chunk, {nch} chunks of 1600x1200 chunk by chunk"
                                     + ("" if trace else " (no trace_radiance)")
                                     + (f"; default settings: the chunk forwards and their trace_radiance calls are recorded and run as passes of {dc} chunks "
                                        "(robir_amd/deferred.py), every chunk its own lock-step batch" if dc else "; every call runs at once (ROBIR_DEFER_CHUNKS=0)"),
            "deferred_chunks": dc,
            "value": nch * 1024 / t, "unit": "rays/s", "ms": t * 1e3, "chunks": nch, "hit_rays": state["hits"],
            "hit_rays_per_s": state["hits"] / t,
            "roofline": SHADOW_TIMER.roofline("shadow_net (512 x 8 softplus net over 128 one-hot labels)", _cesr_precision(), _cesr_precision() == "f16x6")}


CONFIGS = {1: config1, 2: config2, 3: config3, 5: config5}


if __name__ == "__main__":
    only = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 5]
    dev = torch.device("cuda:0")
    install_timers()
    with torch.no_grad():
        m = renderer.build_synthetic_model(dev)
        for k in only:
            r = CONFIGS[k](m)
            rf = r["roofline"]
            print(f"config {k}: {r['workload']}: {r['ms']:.2f} ms = {r['value']:.4g} {r['unit']}; {rf['kernel']}: "
                  f"{rf['achieved']:.0f} TFLOP/s = {rf['frac']:.2f} of {rf['peak']:.0f}")
