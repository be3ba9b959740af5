import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from robir_amd import ops, renderer, synth
dev = torch.device("cuda:0")
model = renderer.build_synthetic_model(dev)
uv, pose, K = synth.synth_camera(800, 800)
c = 300
uv_d = torch.from_numpy(uv[c*1024:(c+1)*1024]).to(dev)
dirs = ops.camera_rays(pose, K, uv_d)
cam = torch.from_numpy(pose[:3, 3]).to(dev).reshape(1, 3)
oct_ = model.ray_tracer.sdf_octree
for rep in range(3):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    _, hit, dist = oct_.cast_chunks(cam, dirs, chunk=1024, sched_cap=256)
    e.record(); torch.cuda.synchronize()
    print("cast ms", s.elapsed_time(e))
sc = oct_.last_sched[0].cpu()
n = int((sc[:, 0] > 0).sum())
print("iterations", n, "hit", int(hit.sum()))
print([tuple(int(v) for v in r) for r in sc[:n]])
print("max_iter", oct_.max_iter, "step", oct_.step_size(1024), "res", oct_.tables.res)
