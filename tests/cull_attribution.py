"""Diagnostic behind tests/test_renderer_gpu.py's cull attribution (VERDICT r5 task 4): one 1024-pixel chunk of the synthetic 64 x 64
view, HIP forward vs the oracle on the same octree tables and draws; per output field the error of the points whose sampled light
directions sit on the n.d > 1e-6 cull (conftest.cull_marked_points) and of all the other points.
    python tests/cull_attribution.py [chunk]        (GPU box; ~1 minute incl. the oracle's octree build)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    import conftest as ct
    from robir_amd import renderer, synth
    from robir_amd.octree_tracing import OctreeSDF
    from robir_oracle import nets as on, octree as ooct, renderer as orend
    c = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    dev = torch.device("cuda:0")
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    sd = on.as_torch(synth.synth_state_dict(0, variance=0.3))
    T = ooct.build(lambda x: on.implicit_forward(sd, x)[:, 0], lambda x: on.implicit_gradient(sd, x), [-1.0] * 3, [1.0] * 3)
    m = renderer.build_synthetic_model(dev, seed=0, variance=0.3, build_octrees=False)
    m.ray_tracer.sdf_octree = OctreeSDF.from_host_tables(T, dev, -1)
    m.octree_ray_tracer.sdf_octree = OctreeSDF.from_host_tables(T, dev, 32)
    uv, pose, K = synth.synth_camera(64, 64)
    sl = slice(c * 1024, (c + 1) * 1024)
    uv_t, pose_t, K_t = torch.from_numpy(uv[sl])[None], torch.from_numpy(pose)[None], torch.from_numpy(K)[None]
    dirs, cam = orend.camera_rays(uv_t, pose_t, K_t)
    _, hit, _ = ooct.trace(T, cam, dirs, -1)
    dr = {k: torch.from_numpy(v) for k, v in synth.pbr_draws(0, int(hit.sum()), chunk_id=c).items()}
    hdr = torch.full((1024, 1), 0.5)
    with torch.no_grad():
        ref = orend.forward(sd, T, uv_t, pose_t, K_t, torch.ones(1, 1024, dtype=torch.bool), hdr, dr, "Material", testing=True)
        out = m({"uv": uv_t.to(dev), "pose": pose_t.to(dev), "intrinsics": K_t.to(dev), "object_mask": torch.ones(1, 1024, dtype=torch.bool, device=dev),
                 "hdr_shift": hdr.to(dev)}, trainstage="Material", train_spec=True, draws={k: v.to(dev) for k, v in dr.items()})
    h = ref["network_object_mask"]
    assert bool((out["network_object_mask"].cpu() == h).all())
    for ulps in (1, 4, 16):
        marked, pairs = ct.cull_marked_points(sd["envmap_material_network.lgtSGs"], dr["dvis_theta"], dr["dvis_phi"], ref["normal_map"][h],
                                              out["normal_map"].cpu()[h], ulps=ulps)
        print(f"ulps {ulps}: {int(marked.sum())} of {int(h.sum())} hit points marked ({pairs} pairs of {int(h.sum()) * 4096})")
    marked, _ = ct.cull_marked_points(sd["envmap_material_network.lgtSGs"], dr["dvis_theta"], dr["dvis_phi"], ref["normal_map"][h],
                                      out["normal_map"].cpu()[h])
    for k in ("vis_shadow", "sg_rgb", "sg_diffuse_rgb", "sg_specular_rgb", "indir_rgb", "indir_diffuse_rgb", "indir_specular_rgb", "diffuse_albedo",
              "roughness", "normal_map", "normals"):
        a, b = out[k].cpu()[h].double(), ref[k][h].double()
        e = (a - b).abs() / (b.abs() + b.abs().mean())
        ep = (a - b).abs() / b.abs().clamp(min=1e-30)
        em, eu = e[marked], e[~marked]
        print(f"{k:22s} marked: max {float(em.max()) if em.numel() else 0:.2e}   unmarked: max {float(eu.max()):.2e}  >1e-4: {int((eu > 1e-4).sum())} of {eu.numel()}"
              f"   plain-rel unmarked max {float(ep[~marked].max()):.2e}")


if __name__ == "__main__":
    main()
