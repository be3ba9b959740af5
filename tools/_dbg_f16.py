import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robir_amd import ops, packing, synth
dev = torch.device("cuda:0")
c = synth.synth_cesr_nets(0)
g = np.random.Generator(np.random.PCG64(11))
pts = torch.from_numpy((g.standard_normal((203, 3)) * 0.25).astype(np.float32)).to(dev)
shf = packing.pack_softplus512_f16({"net." + k: v for k, v in c["shadow_net"].items()}, "net.", 191, dev)
full = ops.cesr_net_f16_points(pts, 203 * 128, 2, shf, 128)
full2 = ops.cesr_net_f16_points(pts, 203 * 128, 2, shf, 128)
print("deterministic:", torch.equal(full, full2))
for m in (1, 16, 48, 49, 128, 129, 191, 192, 193, 384, 717, 203 * 128 - 1):
    a = ops.cesr_net_f16_points(pts[: (m + 127) // 128].contiguous(), m, 2, shf, 128)
    d = (a != full[:m]).any(-1).nonzero().flatten()
    print(m, "rows differing:", d.numel(), d[:8].tolist(), d[-3:].tolist() if d.numel() else "")
print("part2", flush=True)
sh6 = packing.pack_softplus512_x6({"net." + k: v for k, v in c["shadow_net"].items()}, "net.", 191, dev)
ex = ops.cesr_net_x6_points(pts, 203 * 128, 2, sh6, 128); torch.cuda.synchronize(); print("x6 ok", flush=True)
a = ops.cesr_net_f16_points(pts[:1].contiguous(), 128, 2, shf, 128); torch.cuda.synchronize(); print("f16 ragged ok", flush=True)
for r in (110, 111, 112, 113, 127):
    print(r, "exact", ex[r].tolist(), "full", full[r].tolist(), "ragged128", a[r].tolist())
# is the ragged result of rows 112.. what a launch with point 1 := point 0 gives?
p2 = pts.clone(); p2[1] = p2[0]
b = ops.cesr_net_f16_points(p2, 203 * 128, 2, shf, 128)
print("rows 112..127 with point1:=point0 equal full:", torch.equal(b[112:128], full[112:128]), "equal ragged:", torch.equal(b[112:128], a[112:128]))
