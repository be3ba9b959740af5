"""Round-5 golden vectors, recorded by running the REFERENCE (/root/reference) on CPU under oracle/ref_shim.py.

TEST INFRASTRUCTURE ONLY (build container; /root/reference does not exist on the GPU box).  Writes
  tests/golden/forward_relit_{6,12}.npz   the relight serve loop's forward (scripts/relight.py:33-60) in small: the reference's IDRNetwork with
                                 the PBR runner's hook, `envmap_material_network.lgtSGs.data` := the SHIPPED fit envmaps/envmap{6,12}/sg_128.npy
                                 exactly as EnvmapMaterialNetwork.load_light assigns it (model/sg_envmap_material.py:257-265: un-normalised lobe
                                 vectors, |lambda| up to 505, no clamping), `.envmap` := a background map (the decoded rows of
                                 tests/golden/envmap6_rows0_31.exr -- the reference's imageio reader is absent from the image, so the ARRAY is
                                 handed over and the reference's own render_envmap, model/sg_render.py:45-59, samples it),
                                 forward(trainstage='Material') on chunk 1 of the 64x64 view with recorded draws -> every output incl. bg_rgb.
                                 The 3.5 KB light fit travels inside the fixture (`lgtSGs`): data, like the EXR rows.
  tests/golden/spec_vis_refsig.npz        get_specular_visibility called with the reference's OWN positional signature
                                 (points, normals, viewdirs, VisModel, lgtSGLobes, lgtSGLambdas, nsamp, multi_view, testing, inv, argmax_vis),
                                 model/sg_render.py:196-301, on lobes / lambdas that are NOT the warped BRDF lobe of the points (un-normalised
                                 vectors, lambdas on both sides of the 0.1 .. 50 clip): three flag combinations
  tests/golden/render_neus_perturb.npz    render_neus with perturb = 1.0, is_eval = False (model/sdf_render.py:293-295: what the only stage-2
                                 caller wrap_renderer gets by default, :397-399), the one torch.rand([R,1]) draw recorded
  oracle/PINNING_r5.json                  oracle-vs-reference distances of this run

    python oracle/gen_golden_r5.py [spec_vis] [perturb] [relit]     # default: all three; a few minutes (reference model + octree build, two forwards)
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
from robir_oracle import renderer as orend, nets as on, sg as osg, neus as oneus, octree as ooct  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    import gen_golden as g1
    from robir_amd import synth, exr
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    only = set(sys.argv[1:]) or {"spec_vis", "perturb", "relit"}
    rep_path = os.path.join(HERE, "PINNING_r5.json")
    rep = json.load(open(rep_path)) if os.path.exists(rep_path) else {}
    g = np.random.default_rng(5)
    sd_np = synth.synth_state_dict(0, variance=0.3)
    sd = on.as_torch(sd_np)
    wsum = g1.weights_checksum(sd_np)
    vis_fn = lambda p, d: on.vis_logits(sd, p, d)

    with ref_shim.CpuMode():
        net = g1.build_reference(sd_np, "v03")
        g1.install_pbr_hook(net)
        impl = net.implicit_network

        # ------------------------------------------------------------------ get_specular_visibility, the reference's own signature
        from model import sg_render as rsg
        n, nsamp = 61, 24
        pts = torch.from_numpy((g.standard_normal((n, 3)) * 0.25).astype(np.float32))
        nrm = torch.nn.functional.normalize(torch.from_numpy(g.standard_normal((n, 3)).astype(np.float32)), dim=-1)
        view = torch.nn.functional.normalize(nrm + 0.8 * torch.from_numpy(g.standard_normal((n, 3)).astype(np.float32)), dim=-1)
        lobes = torch.from_numpy((g.standard_normal((n, 3)) * 0.9).astype(np.float32)) + nrm        # NOT unit, NOT the reflection
        # ... but |lobe| <= 2.4: the weight exp(sharp (d . lobe - 1)) with sharp <= 50 then stays finite in fp32 (inf / inf = NaN drops the
        # reference into ipdb.set_trace(), model/sg_render.py:296-299: not a call a caller can make)
        lobes = lobes * torch.clamp(2.4 / lobes.norm(dim=-1, keepdim=True), max=1.0)
        lams = torch.from_numpy(np.exp(g.uniform(np.log(0.03), np.log(400.0), (n, 1))).astype(np.float32))   # both sides of the clip
        u = g.random((2, n, nsamp), dtype=np.float32)
        sv = dict(points=pts.numpy(), normals=nrm.numpy(), view=view.numpy(), lobes=lobes.numpy(), lambdas=lams.numpy(), u_theta=u[0],
                  u_phi=u[1], nsamp=nsamp, weights=wsum)
        rep["spec_vis_refsig"] = {}
        for tag, (testing, inv, amax) in {"plain": (False, False, False), "testing_inv": (True, True, False),
                                          "argmax": (False, False, True)}.items():
            with g1.DrawQueue([("rand", u[0]), ("rand", u[1])]):
                ref = rsg.get_specular_visibility(pts, nrm, view, net.visibility_network, lobes, lams, nsamp, False, testing, inv, amax)
            mine = osg.specular_visibility(pts, nrm, view, vis_fn, lobes, lams, torch.from_numpy(u[0]), torch.from_numpy(u[1]),
                                           testing=testing, inv=inv, argmax_vis=amax)
            assert bool(torch.isfinite(ref).all()), tag
            rep["spec_vis_refsig"][tag] = g1.relerr(mine, ref.detach())
            sv["out_" + tag] = ref.detach().numpy()
        print("spec_vis_refsig", rep["spec_vis_refsig"])
        if "spec_vis" in only:
            np.savez_compressed(os.path.join(GOLD, "spec_vis_refsig.npz"), **sv)

        # ------------------------------------------------------------------ render_neus with perturb > 0
        from model.sdf_render import render_neus, Rays
        from utils import rend_util
        H = W = 64
        uv, pose, K = synth.synth_camera(H, W)
        uv_t, pose_t, K_t = torch.from_numpy(uv)[None], torch.from_numpy(pose)[None], torch.from_numpy(K)[None]
        rd, cl = rend_util.get_camera_params(uv_t, pose_t, K_t)
        R = 48
        ro = (cl.expand(R, 3) * 2.0).contiguous()
        rdd = rd[0, 1024 + 16 * 64: 1024 + 16 * 64 + R].contiguous()
        near, far = torch.full((R, 1), 0.8), torch.full((R, 1), 2.8)
        rays = Rays(ro, rdd, rdd, None, None, near, far)
        t_rand = g.random((R, 1), dtype=np.float32)
        with g1.DrawQueue([("rand", t_rand)]):
            ref_rn = render_neus(rays, impl.neus_model, 1.0, n_samples=32, n_importance=32, n_outside=0, up_sample_steps=2)   # wrap_renderer's call
        mine_rn = oneus.render_neus(sd, ro, rdd, near, far, n_samples=32, n_importance=32, up_sample_steps=2, t_rand=torch.from_numpy(t_rand))
        rk = ["rgb", "dist", "acc", "grad", "weights", "grad_error"]
        rep["render_neus_perturb"] = {k: g1.relerr(mine_rn[k], ref_rn[k].detach()) for k in rk}
        print("render_neus_perturb", rep["render_neus_perturb"])
        if "perturb" in only:
          np.savez_compressed(os.path.join(GOLD, "render_neus_perturb.npz"), weights=wsum, rays_o=ro.numpy(), rays_d=rdd.numpy(), near=near.numpy(),
                            far=far.numpy(), t_rand=t_rand, n_samples=32, n_importance=32, up_sample_steps=2,
                            **{"out_" + k: ref_rn[k].detach().numpy() for k in rk})

        # ------------------------------------------------------------------ relight: forward('Material') under a LOADED light
        if "relit" not in only:
            json.dump(rep, open(rep_path, "w"), indent=1)
            return
        sdf_fn = lambda x: impl(x)[:, 0]
        net.ray_tracer.generate(sdf_fn)
        net.octree_ray_tracer.generate(sdf_fn)
        roct = net.ray_tracer.sdf_octree
        Tref = ooct.OctreeTables()      # the reference's own tables in the oracle's format
        Tref.root_min, Tref.root_size = roct.octree.whole_box[:3].clone(), roct.octree.whole_box[3:].clone()
        Tref.box_min, Tref.box_size = roct.octree.boxes[:, :3].clone(), roct.octree.boxes[:, 3:].clone()
        Tref.child, Tref.is_split = roct.octree.links.clone(), roct.octree.non_leaf[:, 0].bool()
        Tref.base_index = roct.octree.cache_index.clone()
        Tref.sdf_val, Tref.sdf_nrm, Tref.centre = roct.sdf_val.clone(), roct.sdf_grad.clone(), roct.centers.clone()
        Tref.hit, Tref.min_step = roct.hit_ptr.clone(), roct.min_step
        env = np.ascontiguousarray(exr.read_exr(os.path.join(GOLD, "envmap6_rows0_31.exr"))[:, :, :3])      # [32, 1024, 3] background map
        hdr_in = net.gamma.hdr_shift.as_input()
        obj_mask = torch.ones(1, H * W, dtype=torch.bool)
        c = 1
        sl = slice(c * 1024, (c + 1) * 1024)
        _, h_r, _ = net.ray_tracer(sdf=None, cam_loc=cl, object_mask=None, ray_directions=rd[:, sl])
        n_hit = int(h_r.sum())
        for e in (6, 12):
            lgt = np.load(os.path.join(ref_shim.REF_ROOT, "envmaps", "envmap%d" % e, "sg_128.npy"))
            emn = net.envmap_material_network
            emn.lgtSGs.data = torch.from_numpy(lgt).to(emn.lgtSGs.data.device)          # load_light, sg_envmap_material.py:258-261
            emn.envmap = torch.from_numpy(env)                                          # :266-268 with the array in place of imageio.imread
            dr = synth.pbr_draws(50 + e, n_hit, chunk_id=c)
            q = [("randn", dr["illum_randn"]), ("randn", dr["spec_randn"]), ("randn", dr["normal_randn"]),
                 ("rand", dr["dvis_theta"]), ("rand", dr["dvis_phi"]), ("rand", dr["svis_theta_dir"]),
                 ("rand", dr["svis_phi_dir"]), ("rand", dr["svis_theta_ind"]), ("rand", dr["svis_phi_ind"])]
            inp = {"uv": uv_t[:, sl], "pose": pose_t, "intrinsics": K_t, "object_mask": obj_mask[:, sl], "hdr_shift": hdr_in.expand(1024, 1)}
            with g1.DrawQueue(q):
                ref = net(inp, trainstage="Material", fun_spec=False, lin_diff=False, train_spec=True)
            sd_e = dict(sd)
            sd_e["envmap_material_network.lgtSGs"] = torch.from_numpy(lgt)
            mine = orend.forward(sd_e, Tref, uv_t[:, sl], pose_t, K_t, obj_mask[:, sl], hdr_in.expand(1024, 1),
                                 {k: torch.from_numpy(v) for k, v in dr.items()}, "Material", testing=True, envmap=torch.from_numpy(env))
            keys = [k for k in ref if isinstance(ref[k], torch.Tensor) and ref[k].dtype == torch.float32 and ref[k].dim() > 0 and k in mine]
            errs = {k: g1.relerr(mine[k], ref[k].detach()) for k in keys}
            rep["forward_relit_%d" % e] = dict(n_hit=n_hit, worst=max(errs.values()), worst_key=max(errs, key=errs.get), **errs)
            print("forward_relit_%d" % e, {k: rep["forward_relit_%d" % e][k] for k in ("n_hit", "worst", "worst_key", "bg_rgb", "sg_rgb")})
            np.savez_compressed(os.path.join(GOLD, "forward_relit_%d.npz" % e), weights=wsum, H=H, W=W, chunk=c, n_hit=n_hit, lgtSGs=lgt,
                                env_fixture="envmap6_rows0_31.exr", hdr_shift=hdr_in.detach().numpy(),
                                **{"draw_" + k: v for k, v in dr.items()},
                                **{"out_" + k: ref[k].detach().numpy() for k in ref if isinstance(ref[k], torch.Tensor)})
    json.dump(rep, open(rep_path, "w"), indent=1)


if __name__ == "__main__":
    main()
