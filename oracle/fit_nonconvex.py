#!/usr/bin/env python
"""Recipe of the NON-CONVEX synthetic scene ("missing" 6): the NeuS SDF network fitted on the CPU to the union of
two overlapping spheres and a torus around them, so that secondary rays re-hit the surface, the lock-step schedules see
concavities / a hole, and the encoding columns and the skip connection of the network carry real weight.

    python oracle/fit_nonconvex.py [steps]      ->  robir_amd/data/nonconvex_sdf.npz  (27 tensors, reference key names)

Test infrastructure: the fit runs the oracle's restatement of SDFNetwork.forward (robir_oracle.nets.sdf_raw, model/neus_model.py:
385-417) under torch autograd, starting from synth.synth_state_dict(0) (geometric initialisation).  Seeded and deterministic on one
machine; the committed .npz is the fixture (robir_amd.synth.synth_state_dict(scene="nonconvex") loads it).  NeuS units (stage-2
units are half of these): spheres c = (-0.20, 0, 0), r = 0.32 and c = (0.22, 0.08, 0), r = 0.28; torus about the y axis through the
origin, R = 0.42, r = 0.09."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from robir_amd import synth            # noqa: E402
from robir_oracle import nets          # noqa: E402

SDF = nets.SDF


def target_sdf(x):
    """Union of the three shapes: exact distance outside, a lower bound inside."""
    a = (x - torch.tensor([-0.20, 0.0, 0.0])).norm(dim=-1) - 0.32
    b = (x - torch.tensor([0.22, 0.08, 0.0])).norm(dim=-1) - 0.28
    q = torch.stack([torch.sqrt(x[:, 0] ** 2 + x[:, 2] ** 2) - 0.42, x[:, 1]], -1)
    t = q.norm(dim=-1) - 0.09
    return torch.minimum(torch.minimum(a, b), t)


def sample(g, n):
    """30 % over the whole box, 30 % around the object, 40 % within +-0.05 of the surface (projected by two Newton steps)."""
    n0, n1 = int(0.3 * n), int(0.3 * n)
    far = (torch.rand(n0, 3, generator=g) - 0.5) * 4.0
    mid = (torch.rand(n1, 3, generator=g) - 0.5) * 1.6
    s = (torch.rand(n - n0 - n1, 3, generator=g) - 0.5) * 1.3
    for _ in range(2):
        with torch.enable_grad():
            s = s.requires_grad_(True)
            d = target_sdf(s)
            (gr,) = torch.autograd.grad(d.sum(), s)
        s = (s - d[:, None] * gr / (gr.norm(dim=-1, keepdim=True) ** 2 + 1e-9)).detach()
    s = s + torch.randn(s.shape, generator=g) * 0.02
    return torch.cat([far, mid, s])


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
    torch.manual_seed(0)
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    sd = nets.as_torch(synth.synth_state_dict(0, variance=0.3))
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith(SDF)}
    work = dict(sd)
    work.update(params)
    opt = torch.optim.Adam(list(params.values()), lr=1e-3)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, steps, eta_min=2e-5)
    g = torch.Generator().manual_seed(1)
    t0 = time.time()
    for it in range(steps):
        x = sample(g, 6144)
        y = target_sdf(x)
        pred = nets.sdf_raw(work, x)[:, 0]
        w = 1.0 + 4.0 * (y.abs() < 0.1).float()              # the surface neighbourhood matters most
        loss = (w * (pred - y).abs()).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        sched.step()
        if it % 100 == 0 or it == steps - 1:
            print(f"step {it:5d} loss {float(loss):.5f}  ({time.time() - t0:.0f} s)", flush=True)
    x = sample(torch.Generator().manual_seed(2), 200000)
    with torch.no_grad():
        y = target_sdf(x)
        pred = nets.sdf_raw(work, x)[:, 0]
        near = y.abs() < 0.05
        print(f"fit: mean |err| {float((pred - y).abs().mean()):.5f}, near the surface {float((pred - y).abs()[near].mean()):.5f}, "
              f"max near {float((pred - y).abs()[near].max()):.4f}; sign agreement {float(((pred > 0) == (y > 0)).float().mean()):.4f}")
    out = {k: v.detach().numpy().astype(np.float32) for k, v in params.items()}
    os.makedirs(os.path.join(ROOT, "robir_amd", "data"), exist_ok=True)
    path = os.path.join(ROOT, "robir_amd", "data", "nonconvex_sdf.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
