import sys, torch
sys.path.insert(0, "/root/repo")
from robir_amd import ops, packing, synth
dev = torch.device("cuda:0")
w = synth.synth_state_dict(0, variance=0.3)
b0, b1 = packing.pack_sdf_x6(w, dev, full=False), packing.pack_sdf_x6(w, dev, full=True)
back = packing.pack_sdf_back_x6(w, dev) + (packing.pack_sdf_back_x6(w, dev, two_tile=True)[0],)
col = packing.pack_color_x6(w, dev)
g = torch.Generator().manual_seed(1)
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best
for n in (4096, 8192, 12288, 16384, 24576, 32768, 49152, 65536):
    p = ((torch.rand(n, 3, generator=g) - 0.5) * 0.6).to(dev)
    v = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
    feat = torch.randn(n, 257, generator=g).to(dev)
    row = [f"{n:6d}"]
    for rows in (1 << 60, 0):
        ops.SDF_TWO_TILE_MIN_ROWS = rows
        row.append("%s dist %.3f vg %.3f col %.3f" % ("two" if rows == 0 else "one", timed(lambda: ops.sdf_points_x6(p, n, b0, False)),
                   timed(lambda: ops.sdf_value_grad_x6(p, n, b1, back)), timed(lambda: ops.color_x6_points(p, v, v, feat[:, 1:], col))))
    print(" | ".join(row), flush=True)
