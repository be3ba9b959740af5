timeout 600 python -m pytest tests/test_mlp_gpu.py -m gpu -x -q -k "wide_ring or fused" 2>&1 | tail -8
timeout 600 python - <<'PY'
import torch, time
from robir_amd import ops, packing, synth
dev='cuda:0'
c = synth.synth_cesr_nets(0)
w = synth.synth_state_dict(0, variance=0.3)
s = packing.H3_SCALE_LOG2
sh16 = packing.pack_softplus512_h3({"net." + k: v for k, v in c["shadow_net"].items()}, "net.", 191, dev)
no16 = packing.pack_softplus512_h3({"net." + k: v for k, v in c["normal_net"].items()}, "net.", 63, dev)
ill16 = packing.pack_illum_h3(w, dev)
enc16 = packing.pack_sparse_ae_encoder_h3(w, "envmap_material_network.spec_brdf_encoder_layer", dev)
g=torch.Generator().manual_seed(1)
def tm(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t=time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time()-t)/n
for npts in (940, 4000):
    p=((torch.rand(npts,3,generator=g)-0.5)*0.6).to(dev)
    M=npts*128
    fl = 6*(192*512+2*512*512+512*336+544*512+3*512*512+512*16)*M
    for ring in (False, True):
        dt=tm(lambda: ops.cesr_net_points(p, M, 2, sh16, 128, s, ring=ring))
        print('shadow', M, 'ring' if ring else 'gen1', f'{dt*1e3:.3f} ms {fl/dt/1e12/2500:.3f} of bound')
for npts in (16384, 120000, 1000000):
    p=((torch.rand(npts,3,generator=g)-0.5)*0.6).to(dev)
    hdr=torch.rand(npts,1,generator=g).to(dev)
    fl0 = 6*(64*512+2*512*512+512*464+544*512+3*512*512+512*16)*npts
    fl1 = 6*(64*512+3*512*512+512*144)*npts
    fl2 = 6*(64*512+3*512*512+512*32)*npts
    for ring in (False, True):
        dt=tm(lambda: ops.cesr_net_points(p, npts, 0, no16, 1, s, ring=ring))
        print('normal', npts, 'ring' if ring else 'gen1', f'{dt*1e3:.3f} ms {fl0/dt/1e12/2500:.3f}')
        dt=tm(lambda: ops.wide_mlp_points(p, hdr, ill16, False, s, ring=ring))
        print('decoder', npts, 'ring' if ring else 'gen1', f'{dt*1e3:.3f} ms {fl1/dt/1e12/2500:.3f}')
        dt=tm(lambda: ops.wide_mlp_points(p, None, enc16, True, s, ring=ring))
        print('encoder', npts, 'ring' if ring else 'gen1', f'{dt*1e3:.3f} ms {fl2/dt/1e12/2500:.3f}')
PY
