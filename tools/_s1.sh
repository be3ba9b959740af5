python -m pytest tests/test_mlp_gpu.py -m gpu -x -q 2>&1 | tail -5
python - <<'PY'
import torch, time, numpy as np
from robir_amd import ops, packing
from robir_amd import synth
dev='cuda:0'
w = synth.synth_state_dict(0, variance=0.3)
blob = packing.pack_color_h3(w, dev)
g=torch.Generator().manual_seed(1)
for n in (1<<17, 1<<20, 3*(1<<20)):
    x=((torch.rand(n,3,generator=g)-0.5)).to(dev)
    v=torch.nn.functional.normalize(torch.randn(n,3,generator=g),dim=-1).to(dev)
    nr=torch.nn.functional.normalize(torch.randn(n,3,generator=g),dim=-1).to(dev)
    out=torch.randn(n,257,generator=g).to(dev)
    for ring in (False, True):
        for _ in range(3): ops.color_mlp_h3_points(x,v,nr,out[:,1:],blob,packing.H3_SCALE_LOG2,ring=ring)
        torch.cuda.synchronize(); t=time.time()
        for _ in range(10): ops.color_mlp_h3_points(x,v,nr,out[:,1:],blob,packing.H3_SCALE_LOG2,ring=ring)
        torch.cuda.synchronize(); dt=(time.time()-t)/10
        fl = 2*3*(320*256+3*256*256+256*16)*n
        print(n, 'ring' if ring else 'gen1', f'{dt*1e3:.3f} ms  {fl/dt/1e12:.1f} TF/s (of 833)')
PY
