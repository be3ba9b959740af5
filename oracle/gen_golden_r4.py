"""Round-4 golden vectors: get_camera_params with the QUATERNION pose form, recorded by running the REFERENCE (/root/reference) on CPU
under oracle/ref_shim.py.

TEST INFRASTRUCTURE ONLY (build container; /root/reference does not exist on the GPU box).  Writes
  tests/golden/camera_quat.npz   two 7-vector poses [qr qi qj qk | cam_loc] (one unnormalised: utils/rend_util.py:109 normalises), a skewed
                                 intrinsics matrix, 512 pixel coordinates -> ray_dirs, cam_loc of utils/rend_util.py:51-97 (the branch of
                                 lines 52-57 + quat_to_rot, :107-124)
  tests/golden/self_spread.json  per output field of forward('Material'), chunk 1: the distance between the REFERENCE's recorded output
                                 (tests/golden/forward_material_c1.npz: its own octree) and the oracle evaluated on an INDEPENDENTLY built
                                 octree of the same SDF (the oracle's own CPU build) with the same draws.  On the reference's own octree
                                 tables the oracle reproduces that output to 0.0 on every shading field (PINNING.json), so this is what
                                 an independent octree build alone does to the outputs: the yardstick the GPU test holds the HIP path
                                 (device-built octree) to.  Needs no reference run: the golden file and the oracle.
  tests/golden/raytracing_train_{r1,r045}.npz   RayTracing.forward with the module in TRAINING mode (model/ray_tracing.py:68-100, 256,
                                 299-326) on the rays of raytracing_{r1,r045}.npz, an object mask with holes, the uniform draws of
                                 minimal_sdf_points replayed from the seed -> points, hit, dist
  oracle/PINNING_r4.json         oracle-vs-reference distances of this run

  tests/golden/sg_multi_view.npz render_with_all_sg with viewdirs [V=2, n, 3] (MULTI_VIEW, model/sg_render.py:356,375-378,465-470 and
                                 get_specular_visibility's multi_view branches :227-231,247-258) on the inputs of sg_init.npz: two views,
                                 one set of specular draws [n,16] per pass, the reference's own visibility network -> the seven outputs

    python oracle/gen_golden_r4.py          # a few minutes (reference model build, one CPU octree build, one oracle forward)
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_shim  # noqa: E402

ref_shim.install()
from robir_oracle import renderer as orend  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    from utils import rend_util
    g = np.random.default_rng(4)
    q = g.normal(size=(2, 4)).astype(np.float32)
    q[0] /= np.linalg.norm(q[0])                       # a unit quaternion and one that is not (the reference normalises)
    q[1] *= 1.7
    loc = g.uniform(-2.0, 2.0, (2, 3)).astype(np.float32)
    pose7 = np.concatenate([q, loc], 1)
    K = np.array([[[220.0, 1.5, 31.5], [0.0, 215.0, 30.5], [0.0, 0.0, 1.0]]] * 2, dtype=np.float32)
    uv = g.uniform(0.0, 64.0, (2, 512, 2)).astype(np.float32)
    with ref_shim.CpuMode():
        rd, cl = rend_util.get_camera_params(torch.from_numpy(uv), torch.from_numpy(pose7), torch.from_numpy(K))
    od, oc = orend.camera_rays(torch.from_numpy(uv), torch.from_numpy(pose7), torch.from_numpy(K))
    rep = {"camera_quat": {"dirs": float((od - rd).abs().max()), "cam": float((oc - cl).abs().max())}}
    print(rep)
    np.savez(os.path.join(GOLD, "camera_quat.npz"), pose7=pose7, K=K, uv=uv, ray_dirs=rd.numpy(), cam_loc=cl.numpy())

    # ------------------------------------------------------------------ MULTI_VIEW shading (sg_render.py:356)
    from robir_amd import synth
    from robir_oracle import nets as on, sg as osg
    import gen_golden as g1
    sd_np = synth.synth_state_dict(0, variance=0.3)
    sdt = on.as_torch(sd_np)
    si = dict(np.load(os.path.join(GOLD, "sg_init.npz"), allow_pickle=False))
    assert str(si["weights"]) == g1.weights_checksum(sd_np)
    T = lambda k: torch.from_numpy(si[k])
    n, V = si["points"].shape[0], 2      # not 3: the reference's torch.cross(z_axis, ref_dir) has no dim= and takes the FIRST axis of size 3
    view = g.standard_normal((V, n, 3)).astype(np.float32) * 0.7 + si["normal"][None]
    view[0] = si["view"]
    view /= np.linalg.norm(view, axis=-1, keepdims=True)
    dr = {k[5:]: si[k] for k in si if k.startswith("draw_dvis")}
    for k in ("svis_theta_dir", "svis_phi_dir", "svis_theta_ind", "svis_phi_ind"):
        dr[k] = g.uniform(0.0, 1.0, (n, 16)).astype(np.float32)
    keys = ["sg_rgb", "sg_specular_rgb", "sg_diffuse_rgb", "vis_shadow", "indir_rgb", "indir_diffuse_rgb", "indir_specular_rgb"]
    with ref_shim.CpuMode():
        net = g1.build_reference(sd_np, "v03")
        from model import sg_render as rsg
        q = [("rand", dr[k]) for k in ("dvis_theta", "dvis_phi", "svis_theta_dir", "svis_phi_dir", "svis_theta_ind", "svis_phi_ind")]
        with g1.DrawQueue(q):
            ref = rsg.render_with_all_sg(T("points"), T("normal"), torch.from_numpy(view), T("lgtSGs"), T("f0"), T("roughness"), T("albedo"),
                                         indir_integral=T("indir_int"), indir_lgtSGs=T("indir_sgs"), VisModel=net.visibility_network,
                                         testing=True)
    mine = osg.render_with_all_sg(T("points"), T("normal"), torch.from_numpy(view), T("lgtSGs"), T("f0"), T("roughness"), T("albedo"),
                                  {k: torch.from_numpy(v) for k, v in dr.items()}, indir_integral=T("indir_int"),
                                  indir_lgt_sgs=T("indir_sgs"), vis_fn=lambda p, d: on.vis_logits(sdt, p, d), testing=True)
    rep["sg_multi_view"] = {k: g1.relerr(mine[k], ref[k]) for k in keys}
    rep["sg_multi_view"]["shapes"] = {k: list(ref[k].shape) for k in keys}
    print(rep["sg_multi_view"])
    np.savez_compressed(os.path.join(GOLD, "sg_multi_view.npz"), weights=si["weights"], view=view,
                        **{"draw_" + k: v for k, v in dr.items()}, **{"out_" + k: ref[k].detach().numpy() for k in keys})

    # ------------------------------------------------------------------ RayTracing in TRAINING mode (ray_tracing.py:68-100, 256, 299-326)
    from robir_oracle import raytracing as ort
    from model.ray_tracing import RayTracing
    with ref_shim.CpuMode():
        rt_conf = ref_shim.hotdog_model_conf().get_config("ray_tracer")
        rt = RayTracing(**{k: rt_conf[k] for k in rt_conf.keys()})
        rt.train()
        impl = net.implicit_network
        sdf_ref = lambda x: impl(x)[:, 0]
        for tag in ("r1", "r045"):
            rg = dict(np.load(os.path.join(GOLD, "raytracing_%s.npz" % tag), allow_pickle=False))
            radius = float(rg["radius"])
            rt.object_bounding_sphere = radius
            cam, dirs = torch.from_numpy(rg["cam"]), torch.from_numpy(rg["dirs"])
            obj = torch.from_numpy(g.uniform(0.0, 1.0, dirs.shape[0]) > 0.3)          # an object mask with holes: in / out rays exist
            seed = 41 if tag == "r1" else 42
            torch.manual_seed(seed)
            with torch.no_grad():
                rx, rh, rdist = rt(sdf=sdf_ref, cam_loc=cam[None], object_mask=obj, ray_directions=dirs[None])
            torch.manual_seed(seed)
            steps_u = torch.empty(rt.n_steps).uniform_(0.0, 1.0)                       # the only draw of the call (:305), replayed
            ox, oh, od = ort.trace(lambda x: on.implicit_forward(sdt, x)[:, 0], cam, dirs, obj, r=radius, training=True, steps_u=steps_u)
            rep["raytracing_train_" + tag] = dict(hits=int(rh.sum()), hit_mismatch=int((rh != oh).sum()), dist=g1.relerr(od, rdist),
                                                  pts=g1.relerr(ox, rx), no_surface_rays=int((~rh).sum()))
            print(tag, rep["raytracing_train_" + tag])
            np.savez_compressed(os.path.join(GOLD, "raytracing_train_%s.npz" % tag), weights=si["weights"], radius=radius, object_mask=obj.numpy(),
                                steps_u=steps_u.numpy(), points=rx.numpy(), hit=rh.numpy(), dist=rdist.numpy())

    # ------------------------------------------------------------------ self-spread of forward('Material') under an independent octree build
    from robir_oracle import octree as ooct
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import rel_err
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    gold = dict(np.load(os.path.join(GOLD, "forward_material_c1.npz"), allow_pickle=False))
    sd = on.as_torch(synth.synth_state_dict(0, variance=0.3))
    T = ooct.build(lambda x: on.implicit_forward(sd, x)[:, 0], lambda x: on.implicit_gradient(sd, x), [-1.0] * 3, [1.0] * 3)
    Hh, Ww, c = int(gold["H"]), int(gold["W"]), int(gold["chunk"])
    uvv, pose, Kk = synth.synth_camera(Hh, Ww)
    sl = slice(c * 1024, (c + 1) * 1024)
    draws = {k[5:]: torch.from_numpy(v) for k, v in gold.items() if k.startswith("draw_")}
    own = orend.forward(sd, T, torch.from_numpy(uvv[sl])[None], torch.from_numpy(pose)[None], torch.from_numpy(Kk)[None],
                        torch.ones(1, 1024, dtype=torch.bool), torch.from_numpy(gold["hdr_shift"]).expand(1024, 1), draws, "Material",
                        testing=True)
    same = torch.from_numpy(gold["out_network_object_mask"]) == own["network_object_mask"]
    spread = {}
    for k, v in gold.items():
        if not k.startswith("out_") or k[4:] not in own or not isinstance(own[k[4:]], torch.Tensor):
            continue
        a, b = own[k[4:]], torch.from_numpy(v)
        if a.dtype != torch.float32 or a.dim() == 0 or a.shape != b.shape:
            continue
        spread[k[4:]] = rel_err(a[same], b[same])
    rep["forward_material_c1_self_spread"] = dict(hit_mismatch=int((~same).sum()), **spread)
    print(rep["forward_material_c1_self_spread"])
    json.dump({"_doc": "max rel_err (tests/conftest.py) per field between the reference's forward('Material') of chunk 1 and the oracle on an "
                       "independently built octree, same draws: oracle/gen_golden_r4.py", "hit_mismatch": int((~same).sum()), "fields": spread},
              open(os.path.join(GOLD, "self_spread.json"), "w"), indent=1, sort_keys=True)
    json.dump(rep, open(os.path.join(HERE, "PINNING_r4.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
