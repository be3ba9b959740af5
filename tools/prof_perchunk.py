import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from robir_amd import renderer, synth
dev = torch.device("cuda:0")
model = renderer.build_synthetic_model(dev)
model.deferred_chunks = 0        # the IMMEDIATE call shape: every forward() runs at once (the default since round 4 records eval-mode chunks:
                                 # tools/prof_deferred.py and bench.py's `extras.runner_plot_loop` time that form)
uv, pose, K = synth.synth_camera(800, 800)
uv_d = torch.from_numpy(uv).to(dev); pose_d = torch.from_numpy(pose).to(dev)[None]; K_d = torch.from_numpy(K).to(dev)[None]
om = torch.ones(1, 1024, dtype=torch.bool, device=dev); hdr = torch.full((1024, 1), 0.5, device=dev)
def run(n0, n):
    for c in range(n0, n0 + n):
        o = model({"uv": uv_d[None, c * 1024:(c + 1) * 1024], "pose": pose_d, "intrinsics": K_d, "object_mask": om, "hdr_shift": hdr},
                  trainstage="Material", train_spec=True)
    torch.cuda.synchronize()
run(280, 8)
t0 = time.time(); run(280, 64); t = time.time() - t0
print("per-chunk forward(): %.2f ms per 1024-px chunk (%.3g rays/s)" % (t / 64 * 1e3, 64 * 1024 / t))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); run(280, 16); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
