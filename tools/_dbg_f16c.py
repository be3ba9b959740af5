import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robir_amd import ops, packing, synth
dev = torch.device("cuda:0")
c = synth.synth_cesr_nets(0)
g = np.random.Generator(np.random.PCG64(11))
pts = torch.from_numpy((g.standard_normal((203, 3)) * 0.25).astype(np.float32)).to(dev)
blob = packing.pack_softplus512_f16({"net." + k: v for k, v in c["shadow_net"].items()}, "net.", 191, dev)
full = ops.cesr_net_f16_points(pts, 203 * 128, 2, blob, 128)
for M in (191, 190, 176, 175, 160, 129, 127):
    outs = []
    for it in range(6):
        outs.append(ops.cesr_net_f16_points(pts[:2].contiguous(), M, 2, blob, 128).clone())
        torch.cuda.synchronize()
    for it in range(1, 6):
        d = (outs[it] != outs[it - 1]).any(-1).nonzero().flatten().tolist()
        df = (outs[it] != full[:M]).any(-1).nonzero().flatten().tolist()
        print(M, "launch", it, "rows differing from previous:", d[:6], "from full:", df[:6], flush=True)
