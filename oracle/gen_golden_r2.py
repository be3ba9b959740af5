"""Round-2 golden vectors, recorded by running the REFERENCE (/root/reference) on CPU under oracle/ref_shim.py.
Separate from gen_golden.py so the round-1 fixtures stay byte-identical.

TEST INFRASTRUCTURE ONLY (build container; /root/reference does not exist on the GPU box).  Writes
  tests/golden/sg_helpers.npz        hemisphere_int / lambda_trick (model/sg_render.py:62-104) on random inputs
  tests/golden/octree_vis.npz        OctreeVisModel (model/octree_tracing.py:63-85) as the VisModel: direct logits on a
                                     <=1024-ray batch, and render_with_all_sg(VisModel=OctreeVisModel) on 64 surface
                                     points (131 k culled pairs: the R > 100000 step branch of utils/octree.py:542-546)
                                     with the lock-step schedule of the diffuse batch
  tests/golden/readers_layout.json   key layout of a NeuS `{step:06d}.tar` written by neus/optimization/log.py:75-88 and
                                     of a stage `latest.pth` written by training/train_pbr.py:215-233 (names, shapes,
                                     dtypes -- no weights), both produced by the reference's own savers
  tests/golden/transforms_test.json  a 2-frame Blender camera file (made up here) ...
  tests/golden/syn_dataset.npz       ... and what datasets/syn_dataset.py:25-130 makes of it (uv, intrinsics, pose)
  oracle/PINNING_r2.json             oracle-vs-reference distances of this run

    python oracle/gen_golden_r2.py          # ~1.5 minutes on 8 cores
"""
import json
import os
import sys
import tempfile
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_shim  # noqa: E402

ref_shim.install()
import gen_golden as G  # noqa: E402   (helpers only: DrawQueue, relerr, save, build_reference, ...)
from gen_golden import synth, on, osg, ooct, orend  # noqa: E402

GOLD = G.GOLD
REPORT = {}


def report(name, **errs):
    REPORT[name] = errs
    print(f"[pin] {name}: " + ", ".join(f"{k}={v:.3g}" if isinstance(v, float) else f"{k}={v}" for k, v in errs.items()),
          flush=True)


def layout(sd):
    return {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()}


def main():
    t_start = time.time()
    torch.set_num_threads(8)
    seed = 0
    sd_np = synth.synth_state_dict(seed, variance=0.3)
    sd = on.as_torch(sd_np)
    wsum = G.weights_checksum(sd_np)
    g = np.random.Generator(np.random.PCG64(4321))
    with ref_shim.CpuMode():
        net = G.build_reference(sd_np, "v03")
        G.install_pbr_hook(net)
        impl = net.implicit_network
        from model import sg_render as rsg

        # ------------------------------------------------------------------ SG algebra helpers
        n = 300
        lam = torch.from_numpy(np.exp(g.uniform(np.log(1e-3), np.log(2e3), (n, 1))).astype(np.float32))
        cb = torch.from_numpy(g.uniform(-1, 1, (n, 1)).astype(np.float32))
        cb[:8] = torch.tensor([[-1.0], [1.0], [0.0], [-0.0], [1e-7], [-1e-7], [0.5], [-0.5]])
        h_ref = rsg.hemisphere_int(lam, cb)
        l1 = torch.from_numpy(g.standard_normal((n, 3)).astype(np.float32))
        l2 = torch.from_numpy(g.standard_normal((n, 3)).astype(np.float32))
        lam1 = torch.from_numpy(np.exp(g.uniform(np.log(1e-2), np.log(50), (n, 1))).astype(np.float32))
        lam2 = torch.from_numpy(np.exp(g.uniform(np.log(1.0), np.log(2e3), (n, 1))).astype(np.float32))
        mu1 = torch.from_numpy(g.uniform(0, 3, (n, 3)).astype(np.float32))
        mu2 = torch.from_numpy(g.uniform(0, 3, (n, 3)).astype(np.float32))
        t_ref = rsg.lambda_trick(l1, lam1, mu1, l2, lam2, mu2)
        t_or = osg.sg_product(l1, lam1, mu1, l2, lam2, mu2)
        report("sg_helpers", hemisphere_int=G.relerr(osg.hemisphere_int(lam, cb), h_ref),
               **{"lambda_trick_%d" % i: G.relerr(t_or[i], t_ref[i]) for i in range(3)})
        G.save("sg_helpers", lam=lam, cos_beta=cb, hemi=h_ref, lobe1=l1, lobe2=l2, lam1=lam1, lam2=lam2, mu1=mu1, mu2=mu2,
               out_lobe=t_ref[0], out_lam=t_ref[1], out_mu=t_ref[2])

        # ------------------------------------------------------------------ octrees (reference build) + OctreeVisModel
        sdf_fn = lambda x: impl(x)[:, 0]
        net.ray_tracer.generate(sdf_fn)
        net.octree_ray_tracer.generate(sdf_fn)
        roct = net.ray_tracer.sdf_octree
        Tref = ooct.OctreeTables()
        Tref.root_min, Tref.root_size = roct.octree.whole_box[:3].clone(), roct.octree.whole_box[3:].clone()
        Tref.box_min, Tref.box_size = roct.octree.boxes[:, :3].clone(), roct.octree.boxes[:, 3:].clone()
        Tref.child, Tref.is_split = roct.octree.links.clone(), roct.octree.non_leaf[:, 0].bool()
        Tref.base_index = roct.octree.cache_index.clone()
        Tref.sdf_val, Tref.sdf_nrm, Tref.centre = roct.sdf_val.clone(), roct.sdf_grad.clone(), roct.centers.clone()
        Tref.hit, Tref.min_step = roct.hit_ptr.clone(), roct.min_step

        from model.octree_tracing import OctreeVisModel
        vis_ref = OctreeVisModel(net.octree_ray_tracer)            # sets sdf_octree.max_iter = 32 (octree_tracing.py:68)
        soct = net.octree_ray_tracer.sdf_octree

        H = W = 64
        uv, pose, K = synth.synth_camera(H, W)
        uv_t, pose_t, K_t = torch.from_numpy(uv)[None], torch.from_numpy(pose)[None], torch.from_numpy(K)[None]
        from utils import rend_util
        rd, cl = rend_util.get_camera_params(uv_t, pose_t, K_t)
        sl = slice(1024, 2048)
        x_r, h_r, t_r = net.ray_tracer(sdf=None, cam_loc=cl, object_mask=None, ray_directions=rd[:, sl])
        pts_all = (cl + t_r[:, None] * rd[0, sl])[h_r]
        sel = torch.from_numpy(g.permutation(int(h_r.sum()))[:64].copy())
        sp = pts_all[sel].contiguous()
        nrm = impl.gradient(sp.clone())[:, 0, :].detach()
        geo_n = (nrm / nrm.norm(dim=-1, keepdim=True)).contiguous()
        # shading normal: strongly perturbed (the PBR hook shades with the material net's normal map, train_pbr.py:373), so
        # that a good share of the sampled directions points INTO the convex test surface and the traced visibility sees hits
        sn = geo_n + 0.9 * torch.from_numpy(g.standard_normal((64, 3)).astype(np.float32))
        sn = (sn / sn.norm(dim=-1, keepdim=True)).contiguous()
        sv = (-rd[0, sl][h_r][sel]).contiguous()

        # (a) direct call on 512 rays (one lock-step batch <= 1024; R <= 100000 -> step 0.005)
        dd = torch.from_numpy(g.standard_normal((512, 3)).astype(np.float32))
        dd = dd / dd.norm(dim=-1, keepdim=True)
        pp = sp.repeat(8, 1).contiguous()                  # about half the rays start into the surface
        sched = []
        orig = soct.fast_volume_render
        soct.fast_volume_render = lambda o, d, m, s, _o=orig: (sched.append((int(o.shape[0]), int(m))), _o(o, d, m, s))[1]
        lg_ref = vis_ref(pp, dd)
        soct.fast_volume_render = orig
        log = []
        t_o, h_o = ooct.cast(Tref, pp, dd, 32, log)
        lg_or = torch.stack([h_o, ~h_o], -1).float()
        report("octree_vis_direct", rays=512, hits=int(lg_ref[:, 0].sum()), mismatch=int((lg_ref != lg_or).any(-1).sum()),
               iters_ref=len(sched), iters_oracle=len(log), sched_equal=bool([m for _, m in sched] == [m for _, m in log]))

        # (b) render_with_all_sg with the traced visibility (train_pbr.py:409-410 installs it as visibility_network)
        nsp = sp.shape[0]
        rough = torch.from_numpy(g.uniform(0.09, 0.99, (nsp, 1)).astype(np.float32))
        alb = torch.from_numpy(g.uniform(0, 1, (nsp, 3)).astype(np.float32))
        hdr = torch.full((nsp, 1), 0.5)
        n64 = synth.synth_draws(seed, "ovis:illum", (nsp, 64), "randn")
        with G.DrawQueue([("randn", n64)]):
            ind_sgs, ind_int = net.indirect_illum_network(sp, hdr)
        ind_sgs, ind_int = ind_sgs.detach(), ind_int.detach() * 2 * np.pi
        f0 = torch.full((1, 1), 0.05)
        lsg = sd["envmap_material_network.lgtSGs"]
        dr = synth.pbr_draws(seed + 2, nsp, chunk_id=78)
        q = [("rand", dr["dvis_theta"]), ("rand", dr["dvis_phi"]), ("rand", dr["svis_theta_dir"]),
             ("rand", dr["svis_phi_dir"]), ("rand", dr["svis_theta_ind"]), ("rand", dr["svis_phi_ind"])]
        sched2, sizes = [], []
        orig_cast = soct.cast
        soct.cast = lambda o, d, **k: (sizes.append(int(o.shape[0])), orig_cast(o, d, **k))[1]
        soct.fast_volume_render = lambda o, d, m, s, _o=orig: (sched2.append((len(sizes), int(o.shape[0]), int(m))), _o(o, d, m, s))[1]
        t0 = time.time()
        with G.DrawQueue(q):
            ref = rsg.render_with_all_sg(sp, sn, sv, lsg, f0, rough, alb, indir_integral=ind_int, indir_lgtSGs=ind_sgs,
                                         VisModel=vis_ref, testing=True)
        t_ref_s = time.time() - t0
        soct.fast_volume_render, soct.cast = orig, orig_cast
        drt = {k: torch.from_numpy(v) for k, v in dr.items()}

        def vis_or(p, d):
            _, h = ooct.cast(Tref, p, d, 32)
            return torch.stack([h, ~h], -1).float()

        mine = osg.render_with_all_sg(sp, sn, sv, lsg, f0, rough, alb, drt, indir_integral=ind_int, indir_lgt_sgs=ind_sgs,
                                      vis_fn=vis_or, testing=True)
        keys = ["sg_rgb", "sg_specular_rgb", "sg_diffuse_rgb", "vis_shadow", "indir_rgb", "indir_diffuse_rgb",
                "indir_specular_rgb"]
        report("octree_vis_sg", ref_s=t_ref_s, cast_sizes=str(sizes), **{k: G.relerr(mine[k], ref[k]) for k in keys})
        G.save("octree_vis", weights=wsum, direct_points=pp, direct_dirs=dd, direct_logits=lg_ref,
               direct_sched_m=np.array([m for _, m in sched]), points=sp, normal=sn, view=sv, lgtSGs=lsg, f0=f0,
               roughness=rough, albedo=alb, indir_sgs=ind_sgs, indir_int=ind_int, cast_sizes=np.array(sizes),
               diffuse_sched_m=np.array([m for c, _, m in sched2 if c == 1]),
               **{"draw_" + k: dr[k] for k in ("dvis_theta", "dvis_phi", "svis_theta_dir", "svis_phi_dir",
                                                "svis_theta_ind", "svis_phi_ind")},
               **{"out_" + k: ref[k].detach() for k in keys})

        # ------------------------------------------------------------------ checkpoint layouts from the reference's own savers
        tmp = tempfile.mkdtemp()
        for name in ("torch.utils.tensorboard",):
            ref_shim._mod(name)
        for name in ("absl", "absl.flags", "absl.app", "absl.logging"):
            ref_shim._mod(name)
        sys.path.insert(0, os.path.join(ref_shim.REF_ROOT, "neus"))
        from optimization.log import Logger
        os.makedirs(os.path.join(tmp, "exp"))
        fake_log = types.SimpleNamespace(log_dir=tmp, exp_name="exp", _global_step=200000, _modules={},
                                         time_cost=lambda: 12.5)
        Logger.save_state(fake_log, model=impl.neus_model)
        tar_path = os.path.join(tmp, "exp", "200000.tar")
        tar = torch.load(tar_path)
        from training.train_pbr import PBRTrainRunner
        os.makedirs(os.path.join(tmp, "ckpt", "ModelParameters"))
        os.makedirs(os.path.join(tmp, "ckpt", "OptimizerParameters"))
        os.makedirs(os.path.join(tmp, "ckpt", "SchedulerParameters"))
        opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1e-3)
        fake_run = types.SimpleNamespace(model=net, checkpoints_path=os.path.join(tmp, "ckpt"),
                                         model_params_subdir="ModelParameters", sg_optimizer_params_subdir="OptimizerParameters",
                                         sg_scheduler_params_subdir="SchedulerParameters", c_optimizer=opt,
                                         c_scheduler=torch.optim.lr_scheduler.MultiStepLR(opt, [10], 0.5))
        PBRTrainRunner.save_checkpoints(fake_run, 7)
        pth_path = os.path.join(tmp, "ckpt", "ModelParameters", "latest.pth")
        pth = torch.load(pth_path)
        lay = {"neus_tar": {"file": "{:06d}.tar", "top": {k: (type(v).__name__) for k, v in tar.items()},
                            "global_step": int(tar["global_step"]), "model": layout(tar["model"])},
               "stage_pth": {"file": "ModelParameters/latest.pth", "top": {k: type(v).__name__ for k, v in pth.items()},
                             "epoch": int(pth["epoch"]), "model_state_dict": layout(pth["model_state_dict"])}}
        json.dump(lay, open(os.path.join(GOLD, "readers_layout.json"), "w"), indent=0, sort_keys=True)
    # our loaders read the files the reference wrote (outside CpuMode: plain torch)
    sys.path.insert(0, ROOT)
    import warnings
    from robir_amd import nets as hnets, render as hrender, renderer as hrenderer
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = hrenderer.IDRNetwork(hrenderer.hotdog_conf())
    for p in m.parameters():
        torch.nn.init.constant_(p, 0.123)
    step = hnets.load_neus_checkpoint(m.implicit_network.neus_model, tar_path)
    res = hrender.load_stage_checkpoint(m, pth_path)
    mine_sd = m.state_dict()
    worst = max(float((mine_sd[k] - torch.from_numpy(v)).abs().max()) for k, v in sd_np.items())
    report("readers", neus_step=step, stage_missing=len(res.missing_keys), stage_unexpected=len(res.unexpected_keys),
           keys=len(mine_sd), max_abs_diff_after_loading_reference_files=worst)
    assert step == 200000 and not res.missing_keys and not res.unexpected_keys and worst == 0.0

    # ------------------------------------------------------------------ SynDataset on a made-up 2-frame camera file
    from PIL import Image
    data = os.path.join(tmp, "scene")
    os.makedirs(os.path.join(data, "test"))
    os.makedirs(os.path.join(data, "test_rli"))
    Hh, Ww = 6, 8

    def rot(ax, ay, t):
        cx, sx, cy, sy = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay)
        R = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
        M = np.eye(4)
        M[:3, :3] = R
        M[:3, 3] = R @ np.array(t)
        return M.tolist()

    meta = {"camera_angle_x": 0.6911112070083618,
            "frames": [{"file_path": "test/r_0", "rotation": 0.031, "transform_matrix": rot(-0.4, 0.3, [0.1, -0.2, 4.031])},
                       {"file_path": "test/r_1", "rotation": 0.031, "transform_matrix": rot(0.25, 2.1, [0.0, 0.0, 4.031])}]}
    json.dump(meta, open(os.path.join(data, "transforms_test.json"), "w"), indent=1)
    json.dump(meta, open(os.path.join(GOLD, "transforms_test.json"), "w"), indent=1)
    for i in range(2):
        rgba = (g.uniform(0, 255, (Hh, Ww, 4))).astype(np.uint8)
        Image.fromarray(rgba, "RGBA").save(os.path.join(data, "test", "r_%d_rgba.png" % i))
        for e in ("envmap6", "envmap12"):
            Image.fromarray(rgba[..., :3], "RGB").save(os.path.join(data, "test_rli", "%s_r_%d.png" % (e, i)))
    import imageio
    imageio.imread = lambda p: np.asarray(Image.open(p))
    sys.modules.pop("datasets.syn_dataset", None)
    sys.modules.pop("datasets", None)
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_syn_dataset", os.path.join(ref_shim.REF_ROOT, "datasets", "syn_dataset.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ds = mod.SynDataset(data, 1, split="test")
    items = [ds[i] for i in range(2)]
    cams = [hrender.blender_camera(os.path.join(GOLD, "transforms_test.json"), i, Hh, Ww) for i in range(2)]
    report("syn_dataset", uv=max(float((torch.from_numpy(c[0]) - it[1]["uv"]).abs().max()) for c, it in zip(cams, items)),
           pose=max(float((torch.from_numpy(c[1]) - it[1]["pose"]).abs().max()) for c, it in zip(cams, items)),
           K=max(float((torch.from_numpy(c[2]) - it[1]["intrinsics"]).abs().max()) for c, it in zip(cams, items)))
    G.save("syn_dataset", H=Hh, W=Ww, img_res=np.array(ds.img_res), total_pixels=ds.total_pixels,
           uv=items[0][1]["uv"], pose=torch.stack([it[1]["pose"] for it in items]),
           intrinsics=torch.stack([it[1]["intrinsics"] for it in items]), idx=np.array([it[0] for it in items]))

    REPORT["_meta"] = {"torch": torch.__version__, "weights": wsum, "seconds": time.time() - t_start}
    json.dump(REPORT, open(os.path.join(HERE, "PINNING_r2.json"), "w"), indent=1)
    print("done in %.1fs" % (time.time() - t_start))


if __name__ == "__main__":
    main()
