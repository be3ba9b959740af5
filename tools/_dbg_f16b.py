import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robir_amd import ops, packing, synth
dev = torch.device("cuda:0")
c = synth.synth_cesr_nets(0)
g = np.random.Generator(np.random.PCG64(11))
pts = torch.from_numpy((g.standard_normal((203, 3)) * 0.25).astype(np.float32)).to(dev)
which = sys.argv[1]
if which == "shadow":
    blob = packing.pack_softplus512_f16({"net." + k: v for k, v in c["shadow_net"].items()}, "net.", 191, dev); kind, nl = 2, 128
else:
    blob = packing.pack_softplus512_f16({"net." + k: v for k, v in c["normal_net"].items()}, "net.", 63, dev); kind, nl = 0, 1
M = int(sys.argv[2])
npts = int(sys.argv[4]) if len(sys.argv) > 4 else (M + nl - 1) // nl
ref = None
bad = 0
for it in range(int(sys.argv[3])):
    a = ops.cesr_net_f16_points(pts[:npts].contiguous(), M, kind, blob, nl)
    torch.cuda.synchronize()
    if ref is None: ref = a
    elif not torch.equal(a, ref): bad += 1
print(which, M, "launches ok; differing from the first:", bad, flush=True)
