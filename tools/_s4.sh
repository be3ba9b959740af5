cat > tools/_t.py <<'PY'
import torch, time, sys, os
sys.path.insert(0, os.getcwd())
from robir_amd import _lib
if len(sys.argv) > 1: _lib.LIB_PATH = os.path.join(os.getcwd(), sys.argv[1])
from robir_amd import ops, packing, synth
dev='cuda:0'
w = synth.synth_state_dict(0, variance=0.3)
x6f, back6 = packing.pack_sdf_x6(w, dev, full=True), packing.pack_sdf_back_x6(w, dev)
g=torch.Generator().manual_seed(1)
n=1<<20
x=((torch.rand(n,3,generator=g)-0.5)*1.2).to(dev)
def tm(f, k=7):
    for _ in range(2): f()
    torch.cuda.synchronize(); t=time.time()
    for _ in range(k): f()
    torch.cuda.synchronize(); return (time.time()-t)/k*1e3
a=tm(lambda: ops.sdf_value_grad_x6(x, n, x6f, back6, 2.0, 0.5)); b=tm(lambda: ops.sdf_points_x6(x, n, x6f, True, 2.0, 0.5))
print(sys.argv[1:], 'value+grad %.3f ms  values %.3f ms -> gradient pass alone ~%.3f ms' % (a, b, a-b))
PY
for l in "" robir_amd/librobir_hip_bnobar.so robir_amd/librobir_hip_bnodma.so robir_amd/librobir_hip_bnone.so ""; do python tools/_t.py $l; done
