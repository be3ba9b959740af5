"""Record golden vectors by running the REFERENCE (ingra14m/RobIR at /root/reference) on CPU, and pin the
oracle restatement (oracle/robir_oracle) against it in the same run.

Runs only in the build container (needs /root/reference).  Writes small .npz fixtures (inputs, explicit RNG
draws, reference outputs -- no reference source) into tests/golden/ and a pinning report to
oracle/PINNING.json.  Weights are NOT stored: they are regenerated from robir_amd.synth (seeded); each
fixture carries a checksum of the weights it was produced with.

    python oracle/gen_golden.py            # ~3-4 minutes on 8 cores
"""
import hashlib
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
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("robir_synth", os.path.join(ROOT, "robir_amd", "synth.py"))
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)
import robir_oracle as O  # noqa: E402
from robir_oracle import nets as on, sg as osg, neus as oneus, octree as ooct, renderer as orend, raytracing as ort  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)
REPORT = {}


def weights_checksum(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(np.ascontiguousarray(sd[k]).tobytes())
    return h.hexdigest()[:16]


class DrawQueue:
    """Replaces torch.rand / torch.randn inside the reference with pre-drawn tensors, in call order."""

    def __init__(self, items):
        self.items = list(items)
        self.log = []

    def _pop(self, kind, shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        shape = tuple(int(s) for s in shape)
        want_kind, arr = self.items.pop(0)
        assert want_kind == kind and tuple(arr.shape) == shape, (kind, shape, want_kind, arr.shape)
        self.log.append((kind, shape))
        return torch.from_numpy(np.ascontiguousarray(arr)).clone()

    def __enter__(self):
        self._r, self._n = torch.rand, torch.randn
        torch.rand = lambda *s, **k: self._pop("rand", s)
        torch.randn = lambda *s, **k: self._pop("randn", s)
        return self

    def __exit__(self, *a):
        torch.rand, torch.randn = self._r, self._n
        assert not self.items, f"{len(self.items)} draws not consumed"


def _err(a, b):
    """|a-b| / (|b| + mean|b|): relative error with the tensor's own scale as floor; NaN==NaN, inf==inf."""
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    same = (a == b) | (torch.isnan(a) & torch.isnan(b))
    fin = torch.isfinite(b)
    scale = b[fin].abs().mean() if fin.any() else torch.tensor(1.0, dtype=torch.float64)
    e = (a - b).abs() / (b.abs() + scale + 1e-30)
    e = torch.where(same, torch.zeros_like(e), e)
    return torch.nan_to_num(e, nan=float("inf"))


def relerr(a, b):
    if torch.as_tensor(a).numel() == 0:
        return 0.0
    return float(_err(a, b).max())


def frac_bad(a, b, tol=1e-4):
    if torch.as_tensor(a).numel() == 0:
        return 0.0
    return float((_err(a, b) > tol).double().mean())


def report(name, **errs):
    REPORT[name] = errs
    print(f"[pin] {name}: " + ", ".join(f"{k}={v:.3g}" if isinstance(v, float) else f"{k}={v}" for k, v in errs.items()),
          flush=True)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)


def build_reference(sd_np, variance_tag):
    from confs_sg import env_path
    tmp = tempfile.mkdtemp()
    neus_sd = {k: torch.from_numpy(v) for k, v in synth.neus_state_dict(sd_np).items()}
    torch.save({"global_step": 1, "model": neus_sd}, os.path.join(tmp, "000001.tar"))
    env_path.set_path(tmp, 1)
    from model.implicit_differentiable_renderer import IDRNetwork
    net = IDRNetwork(ref_shim.hotdog_model_conf())
    missing = net.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)
    net.eval()
    return net


def install_pbr_hook(net):
    from training.train_pbr import PBRTrainRunner
    runner = types.SimpleNamespace(model=net, train_spec=True, no_normal=False, is_training=False)
    net.get_sg_render = types.MethodType(PBRTrainRunner.get_sg_render, runner)


def main():
    t_start = time.time()
    torch.set_num_threads(8)
    seed = 0
    sd_np = synth.synth_state_dict(seed, variance=0.3)
    sd = on.as_torch(sd_np)
    wsum = weights_checksum(sd_np)
    print("weights checksum", wsum)
    with ref_shim.CpuMode():
        net = build_reference(sd_np, "v03")
        install_pbr_hook(net)
        impl = net.implicit_network
        g = np.random.Generator(np.random.PCG64(1234))

        # ------------------------------------------------------------------ encodings + single networks
        M = 96
        pts = torch.from_numpy((g.standard_normal((M, 3)) * 0.2).astype(np.float32))
        dirs = torch.from_numpy(g.standard_normal((M, 3)).astype(np.float32))
        dirs = dirs / dirs.norm(dim=-1, keepdim=True)
        from model.embedder import get_embedder, ipe_embedder
        pe10 = get_embedder(10)[0](pts)
        pe4 = get_embedder(4)[0](dirs)
        ipe = ipe_embedder(10, 1e-5)[0](pts)
        big = pts * 300.0                                                   # exercises the mod-100pi wrap
        ipe_big = ipe_embedder(10, 1e-5)[0](big)
        from robir_oracle.encoding import pe, ipe_isotropic
        report("encoding", pe10=relerr(pe(pts, 10), pe10), pe4=relerr(pe(dirs, 4), pe4),
               ipe=relerr(ipe_isotropic(pts, 1e-5), ipe), ipe_big=relerr(ipe_isotropic(big, 1e-5), ipe_big))
        save("encoding", pts=pts, dirs=dirs, pe10=pe10, pe4=pe4, ipe=ipe, ipe_big_in=big, ipe_big=ipe_big)

        ref_sdf = impl(pts).detach()
        ref_grad = impl.gradient(pts.clone())[:, 0, :].detach()
        feat = ref_sdf[:, 1:] * 2.0
        nrm = ref_grad / ref_grad.norm(dim=-1, keepdim=True)
        ref_col = impl.color(pts, nrm, dirs, feat).detach()
        ref_vis = net.visibility_network(pts, dirs).detach()
        report("sdf", fwd=relerr(on.implicit_forward(sd, pts), ref_sdf), grad=relerr(on.implicit_gradient(sd, pts), ref_grad))
        report("color", col=relerr(on.color_raw(sd, pts * 2.0, nrm, dirs, feat), ref_col))
        report("vis", logits=relerr(on.vis_logits(sd, pts, dirs), ref_vis))
        hdr = torch.full((M, 1), 0.5)
        n64 = synth.synth_draws(seed, "nets:illum", (M, 64), "randn")
        n32 = synth.synth_draws(seed, "nets:spec", (M, 32), "randn")
        n60 = synth.synth_draws(seed, "nets:normal", (M, 60), "randn")
        with DrawQueue([("randn", n64)]):
            ref_sgs, ref_int = net.indirect_illum_network(pts, hdr)
        with DrawQueue([("randn", n32), ("randn", n60)]):
            ref_mat = net.envmap_material_network(pts, train_spec=True)
        o_sgs, o_int = on.indirect_illum(sd, pts, hdr, torch.from_numpy(n64))
        o_mat = on.materials(sd, pts, torch.from_numpy(n32), torch.from_numpy(n60))
        report("illum", sgs=relerr(o_sgs, ref_sgs), integral=relerr(o_int, ref_int))
        mat_keys = ["sg_roughness", "sg_metallic", "sg_normal_map", "sg_diffuse_albedo", "random_xi_roughness",
                    "random_xi_metallic", "random_xi_diffuse_albedo", "random_xi_normal"]
        report("materials", **{k: relerr(o_mat[k], ref_mat[k]) for k in mat_keys})
        save("nets", weights=wsum, pts=pts, dirs=dirs, sdf_feat=ref_sdf, grad=ref_grad, color_normals=nrm,
             color=ref_col, vis_logits=ref_vis, hdr=hdr, illum_noise=n64, spec_noise=n32, normal_noise=n60,
             illum_sgs=ref_sgs.detach(), illum_int=ref_int.detach(),
             **{"mat_" + k: ref_mat[k].detach() for k in mat_keys})

        # ------------------------------------------------------------------ SG shading (stand-alone)
        from model import sg_render as rsg
        n = 40
        sp = pts[:n]
        sn = torch.from_numpy(g.standard_normal((n, 3)).astype(np.float32))
        sn = sn / sn.norm(dim=-1, keepdim=True)
        sv = sn + 0.7 * torch.from_numpy(g.standard_normal((n, 3)).astype(np.float32))
        sv = sv / sv.norm(dim=-1, keepdim=True)
        rough = torch.from_numpy(g.uniform(0.09, 0.99, (n, 1)).astype(np.float32))
        alb = torch.from_numpy(g.uniform(0, 1, (n, 3)).astype(np.float32))
        ind_sgs = ref_sgs[:n].detach()
        ind_int = ref_int[:n].detach() * 2 * np.pi
        f0 = torch.full((1, 1), 0.05)
        for tag, lsg in (("init", sd["envmap_material_network.lgtSGs"]),
                         ("sharp", torch.from_numpy(synth.synth_light_sgs(seed, 128, sharp=True)))):
            dr = synth.pbr_draws(seed + 1, n, chunk_id=77)
            q = [("rand", dr["dvis_theta"]), ("rand", dr["dvis_phi"]), ("rand", dr["svis_theta_dir"]),
                 ("rand", dr["svis_phi_dir"]), ("rand", dr["svis_theta_ind"]), ("rand", dr["svis_phi_ind"])]
            with DrawQueue(q):
                ref = rsg.render_with_all_sg(sp, sn, sv, lsg, f0, rough, alb, indir_integral=ind_int,
                                             indir_lgtSGs=ind_sgs, VisModel=net.visibility_network, testing=True)
            drt = {k: torch.from_numpy(v) for k, v in dr.items()}
            mine = osg.render_with_all_sg(sp, sn, sv, lsg, f0, rough, alb, drt, indir_integral=ind_int,
                                          indir_lgt_sgs=ind_sgs, vis_fn=lambda p, d: on.vis_logits(sd, p, d),
                                          testing=True)
            keys = ["sg_rgb", "sg_specular_rgb", "sg_diffuse_rgb", "vis_shadow", "indir_rgb", "indir_diffuse_rgb",
                    "indir_specular_rgb"]
            report("sg_" + tag, **{k: relerr(mine[k], ref[k]) for k in keys})
            save("sg_" + tag, weights=wsum, points=sp, normal=sn, view=sv, lgtSGs=lsg, f0=f0, roughness=rough,
                 albedo=alb, indir_sgs=ind_sgs, indir_int=ind_int,
                 **{"draw_" + k: dr[k] for k in ("dvis_theta", "dvis_phi", "svis_theta_dir", "svis_phi_dir",
                                                  "svis_theta_ind", "svis_phi_ind")},
                 **{"out_" + k: ref[k].detach() for k in keys})
        # envmap helper
        env_ref = rsg.compute_envmap(sd["envmap_material_network.lgtSGs"], 8, 16)
        report("envmap", grid=relerr(osg.envmap_grid(sd["envmap_material_network.lgtSGs"], 8, 16), env_ref))
        save("envmap", lgtSGs=sd["envmap_material_network.lgtSGs"], grid=env_ref)

        # ------------------------------------------------------------------ octree build + casts
        t0 = time.time()
        sdf_fn = lambda x: impl(x)[:, 0]
        net.ray_tracer.generate(sdf_fn)
        net.octree_ray_tracer.generate(sdf_fn)
        t_ref_build = time.time() - t0
        roct = net.ray_tracer.sdf_octree
        t0 = time.time()
        T = ooct.build(lambda x: on.implicit_forward(sd, x)[:, 0], lambda x: on.implicit_gradient(sd, x),
                       [-1.0] * 3, [1.0] * 3)
        t_my_build = time.time() - t0
        Tref = ooct.OctreeTables()      # the reference's own tables in the oracle's format (bit-identical cast input)
        Tref.root_min, Tref.root_size = roct.octree.whole_box[:3].clone(), roct.octree.whole_box[3:].clone()
        Tref.box_min, Tref.box_size = roct.octree.boxes[:, :3].clone(), roct.octree.boxes[:, 3:].clone()
        Tref.child, Tref.is_split = roct.octree.links.clone(), roct.octree.non_leaf[:, 0].bool()
        Tref.base_index = roct.octree.cache_index.clone()
        Tref.sdf_val, Tref.sdf_nrm, Tref.centre = roct.sdf_val.clone(), roct.sdf_grad.clone(), roct.centers.clone()
        Tref.hit, Tref.min_step = roct.hit_ptr.clone(), roct.min_step
        same_struct = (T.box_min.shape[0] == roct.octree.boxes.shape[0])
        report("octree_build", nodes=int(T.box_min.shape[0]), ref_nodes=int(roct.octree.boxes.shape[0]),
               same_count=bool(same_struct),
               box=relerr(torch.cat([T.box_min, T.box_size], -1), roct.octree.boxes) if same_struct else -1.0,
               links_equal=bool((T.child[T.is_split] == roct.octree.links[T.is_split]).all() and (T.is_split == roct.octree.non_leaf[:, 0].bool()).all()) if same_struct else False,
               sdf_val=relerr(T.sdf_val, roct.sdf_val) if same_struct else -1.0,
               sdf_nrm=float((T.sdf_nrm - roct.sdf_grad).abs().max()) if same_struct else -1.0,
               hit_equal=int((T.hit != roct.hit_ptr).sum()) if same_struct else -1,
               ref_build_s=t_ref_build, oracle_build_s=t_my_build)
        oct_stats = dict(nodes=int(roct.octree.boxes.shape[0]), split=int(roct.octree.non_leaf.sum()),
                         hit=int(roct.hit_ptr.sum()), sdf_sum=float(roct.sdf_val.double().sum()),
                         sdf_abs_sum=float(roct.sdf_val.double().abs().sum()), min_step=float(roct.min_step))

        H = W = 64
        uv, pose, K = synth.synth_camera(H, W)
        uv_t = torch.from_numpy(uv)[None]
        pose_t, K_t = torch.from_numpy(pose)[None], torch.from_numpy(K)[None]
        from utils import rend_util
        rd, cl = rend_util.get_camera_params(uv_t, pose_t, K_t)
        od, oc = orend.camera_rays(uv_t, pose_t, K_t)
        report("camera", dirs=float((od - rd).abs().max()), cam=float((oc - cl).abs().max()))
        # primary cast, chunk 1 of 4 (rows 16..31) -- the lock-step schedule is per chunk
        prim = {}
        for c in (1, 2):
            sl = slice(c * 1024, (c + 1) * 1024)
            # record the reference's per-iteration schedule by wrapping fast_volume_render
            sched = []
            orig = roct.fast_volume_render
            roct.fast_volume_render = lambda o, d, m, s, _orig=orig: (sched.append((int(o.shape[0]), int(m))), _orig(o, d, m, s))[1]
            x_r, h_r, t_r = net.ray_tracer(sdf=None, cam_loc=cl, object_mask=None, ray_directions=rd[:, sl])
            roct.fast_volume_render = orig
            log = []
            x_o, h_o, t_o = ooct.trace(Tref, oc, od[:, sl], -1, log)
            x_b, h_b, t_b = ooct.trace(T, oc, od[:, sl], -1)
            report(f"cast_primary_c{c}", hit_ref=int(h_r.sum()), hit_mismatch=int((h_r != h_o).sum()),
                   t_bad_frac=frac_bad(t_o[h_r & h_o], t_r[h_r & h_o]), t_max=relerr(t_o[h_r & h_o], t_r[h_r & h_o]),
                   iters_ref=len(sched), iters_oracle=len(log),
                   sched_equal=bool([m for _, m in sched] == [m for _, m in log]),
                   ownbuild_hit_mismatch=int((h_b != h_r).sum()), ownbuild_t_bad_frac=frac_bad(t_b[h_r & h_b], t_r[h_r & h_b]),
                   ownbuild_t_max=relerr(t_b[h_r & h_b], t_r[h_r & h_b]))
            prim[c] = (x_r, h_r, t_r, sched)
        save("cast_primary", weights=wsum, cam=cl, dirs=rd[0, 1024:3072], t=torch.stack([prim[1][2], prim[2][2]]),
             hit=torch.stack([prim[1][1], prim[2][1]]), x=torch.stack([prim[1][0], prim[2][0]]),
             sched_m_c1=np.array([m for _, m in prim[1][3]]), sched_m_c2=np.array([m for _, m in prim[2][3]]),
             **{"oct_" + k: v for k, v in oct_stats.items()})

        # ------------------------------------------------------------------ forward('Material'), chunks 1 and 2
        hdr_in = net.gamma.hdr_shift.as_input()
        report("tonemap", as_input=float((orend.hdr_shift_as_input(sd) - hdr_in).abs().max()))
        obj_mask = torch.ones(1, H * W, dtype=torch.bool)
        fw_keys = None
        fw_store = {}
        for c in (1, 2):
            sl = slice(c * 1024, (c + 1) * 1024)
            n_hit = int(prim[c][1].sum())
            dr = synth.pbr_draws(seed, n_hit, chunk_id=c)
            q = [("randn", dr["illum_randn"]), ("randn", dr["spec_randn"]), ("randn", dr["normal_randn"]),
                 ("rand", dr["dvis_theta"]), ("rand", dr["dvis_phi"]), ("rand", dr["svis_theta_dir"]),
                 ("rand", dr["svis_phi_dir"]), ("rand", dr["svis_theta_ind"]), ("rand", dr["svis_phi_ind"])]
            inp = {"uv": uv_t[:, sl], "pose": pose_t, "intrinsics": K_t, "object_mask": obj_mask[:, sl],
                   "hdr_shift": hdr_in.expand(1024, 1)}
            t0 = time.time()
            with DrawQueue(q):
                ref = net(inp, trainstage="Material", fun_spec=False, lin_diff=False, train_spec=True)
            t_ref = time.time() - t0
            drt = {k: torch.from_numpy(v) for k, v in dr.items()}
            stats = {}
            t0 = time.time()
            mine = orend.forward(sd, Tref, uv_t[:, sl], pose_t, K_t, obj_mask[:, sl], hdr_in.expand(1024, 1), drt,
                                 "Material", testing=True, stats=stats)
            own = orend.forward(sd, T, uv_t[:, sl], pose_t, K_t, obj_mask[:, sl], hdr_in.expand(1024, 1), drt,
                                "Material", testing=True)
            t_or = time.time() - t0
            fw_keys = [k for k in ref if isinstance(ref[k], torch.Tensor) and ref[k].dtype == torch.float32 and ref[k].dim() > 0]
            hitm = ref["network_object_mask"]
            errs = {k: relerr(mine[k], ref[k].detach()) for k in fw_keys if k in mine}
            missing = [k for k in ref if k not in mine]
            report(f"forward_material_c{c}", n_hit=n_hit, ref_s=t_ref, oracle_s=t_or,
                   vis_evals=stats.get("diffuse_vis_evals", 0), missing=str(missing), worst=max(errs.values()),
                   worst_key=max(errs, key=errs.get), sg_rgb=errs["sg_rgb"], vis_shadow=errs["vis_shadow"],
                   points=errs["points"], ownbuild_sg_rgb=relerr(own["sg_rgb"], ref["sg_rgb"].detach()),
                   ownbuild_sg_rgb_bad_frac_1e3=frac_bad(own["sg_rgb"], ref["sg_rgb"].detach(), 1e-3),
                   ownbuild_vis_shadow=relerr(own["vis_shadow"], ref["vis_shadow"].detach()))
            fw_store[c] = (ref, dr, n_hit, stats.get("diffuse_vis_evals", 0))
        c = 1
        ref, dr, n_hit, nev = fw_store[c]
        save("forward_material_c1", weights=wsum, H=H, W=W, chunk=c, n_hit=n_hit, diffuse_vis_evals=nev,
             hdr_shift=hdr_in, **{"draw_" + k: v for k, v in dr.items()},
             **{"out_" + k: ref[k].detach() for k in ref if isinstance(ref[k], torch.Tensor)})

        # ------------------------------------------------------------------ IDR sphere tracer (use_octree=False), chunk 1
        from model.ray_tracing import RayTracing
        rt_conf = ref_shim.hotdog_model_conf().get_config("ray_tracer")
        rt = RayTracing(**{k: rt_conf[k] for k in rt_conf.keys()})
        rt.eval()
        sl = slice(1024, 2048)
        # a bounding sphere that some rays miss, so every branch is exercised (r = 0.45 instead of the conf's 1.0)
        for tag, radius in (("r1", 1.0), ("r045", 0.45)):
            rt.object_bounding_sphere = radius
            t0 = time.time()
            rx, rh, rd_ = rt(sdf=sdf_fn, cam_loc=cl, object_mask=obj_mask[0, sl], ray_directions=rd[:, sl])
            t_ref = time.time() - t0
            ox, oh, od_ = ort.trace(lambda x: on.implicit_forward(sd, x)[:, 0], cl[0], rd[0, sl], obj_mask[0, sl], r=radius)
            both = rh & oh
            report("raytracing_" + tag, ref_s=t_ref, hits=int(rh.sum()), hit_mismatch=int((rh != oh).sum()),
                   dist=relerr(od_[both], rd_.detach()[both]), pts=relerr(ox[both], rx.detach()[both]),
                   miss_dist=relerr(od_[~rh & ~oh], rd_.detach()[~rh & ~oh]))
            save("raytracing_" + tag, weights=wsum, radius=radius, cam=cl[0], dirs=rd[0, sl], points=rx.detach(), hit=rh,
                 dist=rd_.detach())

        # ------------------------------------------------------------------ CESR hook (shadow_net / normal_net), chunk 1
        from training.train_cesr import ClusteredAlbedoTrainRunner
        from model.neus_model import SDFNetwork as RefSDFNetwork
        cesr_np = synth.synth_cesr_nets(seed)
        shadow_net = RefSDFNetwork(63 + 128, 2, 512, 8, [4], 0)
        normal_net = RefSDFNetwork(63, 3, 512, 8, [4], 0)
        shadow_net.load_state_dict({k: torch.from_numpy(v) for k, v in cesr_np["shadow_net"].items()})
        normal_net.load_state_dict({k: torch.from_numpy(v) for k, v in cesr_np["normal_net"].items()})
        cesr_conf = types.SimpleNamespace(get_bool=lambda k: False)
        runner = types.SimpleNamespace(model=net, train_spec=True, is_training=False, cur_iter=100000, conf=cesr_conf,
                                       shadow_embed=get_embedder(10)[0], shadow_net=shadow_net, normal_net=normal_net,
                                       prefit_option=lambda: "explore", white_light=False)
        net.get_sg_render = types.MethodType(ClusteredAlbedoTrainRunner.get_sg_render, runner)
        c = 1
        sl = slice(c * 1024, (c + 1) * 1024)
        n_hit = int(prim[c][1].sum())
        dr = synth.pbr_draws(seed + 5, n_hit, chunk_id=c, nsamp_diffuse=8)
        q = [("randn", dr["illum_randn"]), ("randn", dr["spec_randn"]), ("randn", dr["normal_randn"]),
             ("rand", dr["dvis_theta"]), ("rand", dr["dvis_phi"]), ("rand", dr["svis_theta_dir"]),
             ("rand", dr["svis_phi_dir"]), ("rand", dr["svis_theta_ind"]), ("rand", dr["svis_phi_ind"])]
        inp = {"uv": uv_t[:, sl], "pose": pose_t, "intrinsics": K_t, "object_mask": obj_mask[:, sl],
               "hdr_shift": hdr_in.expand(1024, 1)}
        t0 = time.time()
        with DrawQueue(q):
            ref = net(inp, trainstage="Material", fun_spec=False, lin_diff=True, train_spec=True)
        t_ref = time.time() - t0
        drt = {k: torch.from_numpy(v) for k, v in dr.items()}
        cesr_t = ({k: torch.from_numpy(v) for k, v in cesr_np["shadow_net"].items()},
                  {k: torch.from_numpy(v) for k, v in cesr_np["normal_net"].items()})
        mine = orend.forward(sd, Tref, uv_t[:, sl], pose_t, K_t, obj_mask[:, sl], hdr_in.expand(1024, 1), drt,
                             "Material", testing=True, cesr=cesr_t)
        ck = ["sg_rgb", "indir_rgb", "sg_diffuse_rgb", "sg_specular_rgb", "vis_shadow", "normal_map", "diffuse_albedo",
              "roughness", "gradient_error"]
        report("forward_cesr_c1", n_hit=n_hit, ref_s=t_ref, **{k: relerr(mine[k], ref[k].detach()) for k in ck})
        save("forward_cesr_c1", weights=wsum, H=H, W=W, chunk=c, n_hit=n_hit, hdr_shift=hdr_in,
             **{"draw_" + k: v for k, v in dr.items()},
             **{"out_" + k: ref[k].detach() for k in ref if isinstance(ref[k], torch.Tensor)})
        # the two nets alone
        xs_n = torch.from_numpy(g.standard_normal((64, 63)).astype(np.float32))
        xs_s = torch.cat([xs_n, torch.eye(128)[torch.arange(64) % 128]], -1)
        ref_n, ref_s = normal_net(xs_n).detach(), shadow_net(xs_s).detach()
        report("cesr_nets", normal=relerr(on.softplus_net512(cesr_t[1], xs_n), ref_n),
               shadow=relerr(on.softplus_net512(cesr_t[0], xs_s), ref_s))
        save("cesr_nets", x_normal=xs_n, x_shadow=xs_s, y_normal=ref_n, y_shadow=ref_s)
        install_pbr_hook(net)

        # ------------------------------------------------------------------ forward('Illum') + trace_radiance(nsamp=8)
        c = 1
        sl = slice(c * 1024, (c + 1) * 1024)
        n_hit = int(prim[c][1].sum())
        d_ill = synth.synth_draws(seed, "illumstage:illum", (n_hit, 64), "randn")
        d_nrm = synth.synth_draws(seed, "illumstage:normal", (n_hit, 60), "randn")
        inp = {"uv": uv_t[:, sl], "pose": pose_t, "intrinsics": K_t, "object_mask": obj_mask[:, sl],
               "hdr_shift": hdr_in.expand(1024, 1)}
        with DrawQueue([("randn", d_ill), ("randn", d_nrm)]):
            ref_ill = net(inp, trainstage="Illum")
        mine_ill = orend.forward(sd, Tref, uv_t[:, sl], pose_t, K_t, obj_mask[:, sl], hdr_in.expand(1024, 1),
                                 {"illum_randn": torch.from_numpy(d_ill), "normal_randn": torch.from_numpy(d_nrm)},
                                 "Illum")
        report("forward_illum", normals=relerr(mine_ill["normals"], ref_ill["normals"].detach()),
               sgs=relerr(mine_ill["indirect_sgs"], ref_ill["indirect_sgs"].detach()),
               integral=relerr(mine_ill["indir_integral"], ref_ill["indir_integral"].detach()))
        nsamp = 8
        u1 = synth.synth_draws(seed, "trace:u1", (n_hit * nsamp,))
        u2 = synth.synth_draws(seed, "trace:u2", (n_hit * nsamp,))
        T2 = Tref  # same geometry; the secondary tracer differs only by max_iter (octree_tracing.py:40-41)
        ref_in = {k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in ref_ill.items()}
        t0 = time.time()
        with DrawQueue([("rand", u1), ("rand", u2)]):
            ref_tr = net.trace_radiance(ref_in, nsamp=nsamp)
        t_ref = time.time() - t0
        mine_tr = orend.trace_radiance(sd, T2, {k: ref_in[k] for k in ("points", "hdr_shift", "network_object_mask", "normals")},
                                       nsamp, torch.from_numpy(u1), torch.from_numpy(u2))
        report("trace_radiance", ref_s=t_ref, sec_hits=int(ref_tr["gt_vis"].sum()),
               gt_vis_mismatch=int((ref_tr["gt_vis"] != mine_tr["gt_vis"]).sum()),
               radiance_bad_frac=frac_bad(mine_tr["trace_radiance"], ref_tr["trace_radiance"]),
               radiance=relerr(mine_tr["trace_radiance"], ref_tr["trace_radiance"]),
               pred_vis=relerr(mine_tr["pred_vis"], ref_tr["pred_vis"].detach()),
               gt_integral=relerr(mine_tr["gt_integral"], ref_tr["gt_integral"]),
               dirs=relerr(mine_tr["sample_dirs"], ref_tr["sample_dirs"]))
        save("trace_radiance", weights=wsum, chunk=c, nsamp=nsamp, u1=u1, u2=u2, illum_noise=d_ill, normal_noise=d_nrm,
             in_points=ref_in["points"], in_hdr_shift=ref_in["hdr_shift"], in_mask=ref_in["network_object_mask"],
             in_normals=ref_in["normals"], illum_sgs=ref_in["indirect_sgs"], illum_int=ref_in["indir_integral"],
             **{"out_" + k: v.detach() for k, v in ref_tr.items()})

        # ------------------------------------------------------------------ borrow_color / neus_surface / render_neus
        hp = ref_in["points"][ref_in["network_object_mask"]][:64]
        hv = dirs[:64]
        ref_bc = impl.batch_borrow_color(hp, hv)
        report("borrow_color", rgb=relerr(oneus.borrow_color(sd, hp, hv), ref_bc))
        from training.train_normal import NormalTrainRunner
        runner = types.SimpleNamespace(model=net)
        hn = ref_in["normals"][ref_in["network_object_mask"]][:64]
        rdirs = rd[0, sl][ref_in["network_object_mask"]][:64]
        rx, rn, rge = NormalTrainRunner.get_neus_surface(runner, hp.clone(), rdirs, hn)
        ox, on_, oge = oneus.neus_surface(sd, hp, rdirs, hn)
        report("neus_surface", x=relerr(ox, rx.detach()), n=relerr(on_, rn.detach()), gerr=relerr(oge, rge.detach()))
        save("neus_misc", weights=wsum, bc_points=hp, bc_view=hv, bc_rgb=ref_bc, ns_points=hp, ns_dirs=rdirs,
             ns_normals=hn, ns_x=rx.detach(), ns_n=rn.detach(), ns_gerr=rge.detach())

        from model.sdf_render import render_neus, Rays
        for tag, var in (("v03", 0.3), ("v06", 0.6)):
            if var != 0.3:
                sd2_np = dict(sd_np)
                sd2_np["implicit_network.neus_model.deviation_network.variance"] = np.array(var, np.float32)
                sd2 = on.as_torch(sd2_np)
                impl.neus_model.deviation_network.variance.data.fill_(var)
            else:
                sd2 = sd
            R = 48
            ro = (cl.expand(R, 3) * 2.0).contiguous()
            rdd = rd[0, 1024 + 16 * 64: 1024 + 16 * 64 + R].contiguous()      # a row through the sphere
            near, far = torch.full((R, 1), 0.8), torch.full((R, 1), 2.8)
            rays = Rays(ro, rdd, rdd, None, None, near, far)
            t0 = time.time()
            ref_rn = render_neus(rays, impl.neus_model, 1.0, n_samples=64, n_importance=64, n_outside=0,
                                 up_sample_steps=4, is_eval=True)
            t_ref = time.time() - t0
            mine_rn = oneus.render_neus(sd2, ro, rdd, near, far)
            rk = ["rgb", "dist", "acc", "grad", "weights", "grad_error"]
            report("render_neus_" + tag, ref_s=t_ref, **{k: relerr(mine_rn[k], ref_rn[k].detach()) for k in rk})
            save("render_neus_" + tag, weights=wsum, variance=var, rays_o=ro, rays_d=rdd, near=near, far=far,
                 **{"out_" + k: ref_rn[k].detach() for k in rk})
        impl.neus_model.deviation_network.variance.data.fill_(0.3)

        # tone mapping (model/color_correction.py, hdr_mode 0), scalar and per-row shifts incl. values outside [1e-4, 1]
        tmr = net.gamma.hdr_shift
        gt = torch.Generator().manual_seed(9)
        tx = torch.rand(257, 3, generator=gt) * 4.0
        ty = torch.rand(257, 3, generator=gt) * 0.9
        tsh = torch.rand(257, 1, generator=gt) * 1.4 - 0.2
        tone = {"x": tx, "y": ty, "shift_rows": tsh, "shift_scalar": torch.tensor([[0.37]])}
        for tag, sh in (("rows", tsh), ("scalar", tone["shift_scalar"])):
            tone["ldr_" + tag] = tmr.hdr2ldr(tx, sh).detach()
            tone["hdr_" + tag] = tmr.ldr2hdr(ty, sh).detach()
        report("tonemap", **{k: relerr(f(a, tone["shift_" + tag]), tone[o + tag]) for tag in ("rows", "scalar")
                             for k, f, a, o in (("hdr2ldr_" + tag, orend.hdr2ldr, tx, "ldr_"), ("ldr2hdr_" + tag, orend.ldr2hdr, ty, "hdr_"))})
        # the other curve pairs of the class (hdr_mode 1 warp_aces, 2 ln_space, -1 identity)
        from model.color_correction import ACESToneMapping as RefTone
        for hm in (1, 2, -1):
            rt = RefTone(hdr_mode=hm)
            for tag, sh in (("rows", tsh), ("scalar", tone["shift_scalar"])):
                key = "m%d_%s" % (hm if hm >= 0 else 9, tag)
                tone["ldr_" + key] = rt.hdr2ldr(tx, sh).detach()
                tone["hdr_" + key] = rt.ldr2hdr(ty * 0.7, sh).detach()
                report("tonemap_" + key, hdr2ldr=relerr(orend.hdr2ldr(tx, sh, hm), tone["ldr_" + key]),
                       ldr2hdr=relerr(orend.ldr2hdr(ty * 0.7, sh, hm), tone["hdr_" + key]))
        save("tonemap", **tone)

        # stage-1 renderer (neus/volume_render/sdf_render.py: cos-annealed alpha), same model, same rays
        for name in ("absl", "absl.flags", "absl.app", "absl.logging"):
            ref_shim._mod(name)
        sys.path.insert(0, os.path.join(ref_shim.REF_ROOT, "neus"))
        from volume_render.sdf_render import render_neus as render_neus_stage1
        R = 48
        ro = (cl.expand(R, 3) * 2.0).contiguous()
        rdd = rd[0, 1024 + 16 * 64: 1024 + 16 * 64 + R].contiguous()
        near, far = torch.full((R, 1), 0.8), torch.full((R, 1), 2.8)
        rays = Rays(ro, rdd, rdd, None, None, near, far)
        for tag, ratio in (("c03", 0.3), ("c10", 1.0)):
            ref_rn = render_neus_stage1(rays, impl.neus_model, ratio, n_samples=64, n_importance=64, n_outside=0,
                                        up_sample_steps=4, is_eval=True)
            mine_rn = oneus.render_neus(sd, ro, rdd, near, far, cos_anneal_ratio=ratio)
            rep = {k: relerr(mine_rn[k], ref_rn[k].detach()) for k in ("rgb", "dist", "acc", "weights")}
            rep["grad_error"] = relerr(mine_rn["grad_error"], ref_rn["sim_or_grad"].detach())
            report("render_neus_stage1_" + tag, **rep)
            save("render_neus_stage1_" + tag, weights=wsum, ratio=ratio, rays_o=ro, rays_d=rdd, near=near, far=far,
                 out_rgb=ref_rn["rgb"].detach(), out_dist=ref_rn["dist"].detach(), out_acc=ref_rn["acc"].detach(),
                 out_weights=ref_rn["weights"].detach(), out_grad_error=ref_rn["sim_or_grad"].detach(),
                 out_means=ref_rn["means"].detach())

    REPORT["_meta"] = {"torch": torch.__version__, "threads": torch.get_num_threads(), "weights": wsum,
                       "seconds": time.time() - t_start}
    with open(os.path.join(HERE, "PINNING.json"), "w") as f:
        json.dump(REPORT, f, indent=1)
    print("done in %.1fs" % (time.time() - t_start))


if __name__ == "__main__":
    main()
