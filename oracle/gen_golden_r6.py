"""Round-6 golden vectors, recorded by running the REFERENCE (/root/reference) on CPU under oracle/ref_shim.py.

TEST INFRASTRUCTURE ONLY (build container; /root/reference does not exist on the GPU box).  Covers the public helper names the overlay
gained in round 6 (VERDICT r05 "What's missing" 1): every entry is (inputs, the reference function's own outputs).  Writes
  tests/golden/surface_helpers.npz
      color_correction (:31-73)      aces_fn / aces_inv / warp_aces_* / scale_aces_* / ln_space_* on x in [0, 1.5], t in (0, 1.3] (no clamp)
      sdf_render (:37-260)           sample_pdf det / random (the torch.rand draw recorded), up_sample, cat_z_vals (last False / True, unsorted
                                     new depths), render_core's whole dict, on the synthetic NeuS weights
      neus_model (:14-94,136-309)    expected_sin, integrated_pos_enc (diag True / False), IPE(max_deg=10), PE for (3, 4) / (1, 10) / (3, 10) /
                                     no-input / linear bands, get_embedder; model/embedder.py get_embedder(4), ipe_embedder(10)
      sg_envmap_material (:96-99)    SparseAE.encode (spec_brdf_encoder_layer with a non-zero var)
      ray_tracing (:102-326)         sphere_tracing / ray_sampler / secant / minimal_sdf_points on an ANALYTIC bumpy-sphere SDF written with
                                     element-wise torch ops only (bit-reproducible on any device), uniform draws recorded
      implicit_differentiable_renderer (:531-564)  IDRNetwork.sample_dirs, batch_idr_forward
      octree_tracing (:70-76)        OctreeVisModel.intersect_sphere
      sdf_render.wrap_renderer (:377-426)  key list / shapes / dtypes of the returned dict (its torch.rand jitter is drawn on the device)
  oracle/PINNING_r6.json             which functions were recorded, with output checksums

    python oracle/gen_golden_r6.py          # ~1 minute
"""
import hashlib
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_shim  # noqa: E402

ref_shim.install()

GOLD = os.path.join(ROOT, "tests", "golden")


def bumpy_sdf(x):
    """Analytic SDF used for the RayTracing stage goldens: element-wise ops only (same bits on CPU and GPU)."""
    r = torch.sqrt(x[:, 0] * x[:, 0] + x[:, 1] * x[:, 1] + x[:, 2] * x[:, 2])
    return (r - 0.5) + 0.02 * (x[:, 0] * 7.0 - x[:, 1] * 5.0 + x[:, 2] * 3.0 - 0.3).abs()


def main():
    import gen_golden as g1
    from robir_amd import synth
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    g = np.random.default_rng(6)
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)      # noqa: E731
    T = torch.from_numpy
    out = {}
    sd_np = synth.synth_state_dict(0, variance=0.3)
    out["weights"] = g1.weights_checksum(sd_np)

    with ref_shim.CpuMode():
        # ---------------------------------------------------------------- tone-mapping curves
        from model import color_correction as cc
        x = f32(g.uniform(0.0, 1.5, (97, 3)))
        t = f32(g.uniform(0.02, 1.3, (97, 1)))
        out["tm_x"], out["tm_t"] = x, t
        out["tm_aces_fn"] = cc.aces_fn(T(x)).numpy()
        out["tm_aces_inv"] = cc.aces_inv(T(np.minimum(x, 1.0))).numpy()
        for name in ("warp_aces_inv", "warp_aces_fn", "scale_aces_inv", "scale_aces_fn", "ln_space_fn", "ln_space_inv", "identity_fn"):
            xin = np.minimum(x, 0.9) if name.endswith("inv") else x
            out["tm_" + name] = getattr(cc, name)(T(xin), T(t)).numpy()
        out["tm_scale_aces_fn_scalar_t"] = cc.scale_aces_fn(T(x), torch.tensor(0.37)).numpy()

        # ---------------------------------------------------------------- NeuS sampling helpers
        net = g1.build_reference(sd_np, "v03")
        neus = net.implicit_network.neus_model
        from model import sdf_render as rs
        R, n = 37, 24
        bins = np.sort(f32(g.uniform(0.5, 3.0, (R, n))), -1)
        w = f32(g.uniform(0.0, 1.0, (R, n - 1)) ** 3)
        w[3] = 0.0                                            # an all-zero row: the +1e-5 alone
        u = f32(g.uniform(0.0, 1.0, (R, 11)))
        out["sp_bins"], out["sp_w"], out["sp_u"] = bins, w, u
        out["sp_det"] = rs.sample_pdf(T(bins), T(w), 16, det=True).numpy()
        with g1.DrawQueue([("rand", u)]):
            out["sp_rand"] = rs.sample_pdf(T(bins), T(w), 11, det=False).numpy()

        ro = f32(g.standard_normal((R, 3)) * 0.05 + np.array([0.0, 0.0, 1.8]))
        rd = f32(-ro / np.linalg.norm(ro, axis=-1, keepdims=True) + g.standard_normal((R, 3)) * 0.15)
        rd = f32(rd / np.linalg.norm(rd, axis=-1, keepdims=True))
        z = np.sort(f32(g.uniform(0.8, 2.8, (R, n))), -1)
        with torch.no_grad():
            pts = T(ro)[:, None, :] + T(rd)[:, None, :] * T(z)[..., :, None]
            s0 = neus.sdf(pts.reshape(-1, 3)).reshape(R, n)
            zn = rs.up_sample(T(ro), T(rd), T(z), s0, 8, 128.0, neus.radius())
            z2, s2 = rs.cat_z_vals(neus, T(ro), T(rd), T(z), zn, s0, last=False)
            z3, s3 = rs.cat_z_vals(neus, T(ro), T(rd), T(z), zn, s0, last=True)
            zu = f32(g.uniform(0.8, 2.8, (R, 5)))              # unsorted new depths (what sample_pdf(det=False) would give)
            z4, s4 = rs.cat_z_vals(neus, T(ro), T(rd), T(z), T(zu), s0, last=False)
        out.update(ns_ro=ro, ns_rd=rd, ns_z=z, ns_sdf=s0.numpy(), ns_up=zn.numpy(), ns_cat_z=z2.numpy(), ns_cat_sdf=s2.numpy(),
                   ns_cat_last_z=z3.numpy(), ns_cat_last_sdf=s3.numpy(), ns_zu=zu, ns_cat_u_z=z4.numpy(), ns_cat_u_sdf=s4.numpy())
        rc = rs.render_core(T(ro), T(rd), z2, 2.0 / 24, neus, background_rgb=torch.ones(1, 3))
        for k, v in rc.items():
            out["rc_" + k] = v.detach().numpy()
        rc0 = rs.render_core(T(ro), T(rd), z2, 2.0 / 24, neus)
        out["rc_nobg_color"] = rc0["color"].detach().numpy()

        # wrap_renderer: structure only (the jitter is a device draw)
        fake = types.SimpleNamespace(implicit_network=net.implicit_network)
        orig_color = neus.__dict__.get("color")
        wr = rs.wrap_renderer(fake, lambda p: torch.sigmoid(p * 3.0), {"points": T(ro) * 0.5, "dirs": T(rd)}, near=0.4, far=1.4, is_eval=True)
        if orig_color is None:
            neus.__dict__.pop("color", None)
        out["wr_keys"] = np.array(sorted(wr.keys()))
        out["wr_shapes"] = np.array([json.dumps([k, list(wr[k].shape), str(wr[k].dtype)]) for k in sorted(wr.keys())])
        out["wr_points"], out["wr_dirs"] = f32(ro * 0.5), rd

        # ---------------------------------------------------------------- encodings
        from model import neus_model as nm, embedder as emb
        ex = f32(g.standard_normal((53, 3)) * 40.0)
        ev = f32(np.abs(g.standard_normal((53, 3))) * 2.0)
        y, yv = nm.expected_sin(T(ex * 10), T(ev))            # includes |x| > 100 pi rows (safe_trig_helper wrap)
        out.update(es_x=ex * 10, es_var=ev, es_y=y.numpy(), es_yvar=yv.numpy())
        px = f32(g.standard_normal((41, 3)) * 0.6)
        cov = f32(np.abs(g.standard_normal((41, 3))) * 1e-3)
        out["ipe_x"], out["ipe_cov_diag"] = px, cov
        out["ipe_diag"] = nm.integrated_pos_enc((T(px), T(cov)), 0, 6, diag=True).numpy()
        full = np.zeros((41, 3, 3), np.float32)
        full[:, [0, 1, 2], [0, 1, 2]] = cov
        full[:, 0, 1] = full[:, 1, 0] = 3e-4                  # off-diagonal terms drop out of diag(B^T C B)
        out["ipe_cov_full"] = full
        out["ipe_full"] = nm.integrated_pos_enc((T(px), T(full)), 2, 9, diag=False).numpy()
        ipe = nm.IPE(max_deg=10)
        out["ipe_module"] = ipe(T(px), nm.isotropic_cov(T(px), 1e-5)).numpy()
        fn, dim = emb.ipe_embedder(10)
        out["ipe_embedder"] = fn(T(px)).numpy()
        assert dim == 60
        out["pe_x1"] = f32(g.standard_normal((29, 1)))
        for tag, kw, xin in (("3_4", dict(input_dims=3, num_freq=4), px), ("1_10", dict(input_dims=1, num_freq=10), out["pe_x1"]),
                             ("3_10", dict(input_dims=3, num_freq=10), px), ("noinp", dict(input_dims=3, num_freq=5, include_input=False), px),
                             ("lin", dict(input_dims=3, num_freq=6, log_sampling=False), px)):
            pe = nm.PE(**kw)
            out["pe_" + tag] = pe(T(xin)).numpy()
            assert pe.feature_dim() == out["pe_" + tag].shape[1]
        f4, d4 = emb.get_embedder(4)
        out["emb_get4"] = f4(T(px)).numpy()
        f5, d5 = nm.get_embedder(5, input_dims=3)
        out["nm_get5"] = f5(T(px)).numpy()
        out["pe_window"] = nm.PE.cosine_easing_window(0, 9, 10, torch.tensor(3.3)).numpy()

        # ---------------------------------------------------------------- SparseAE.encode
        ae = net.envmap_material_network.spec_brdf_encoder_layer
        ae.var = T(f32(g.uniform(0.0, 0.3, 32)))
        vals = f32(g.standard_normal((33, 63)) * 0.5)
        with torch.no_grad():
            out["ae_values"], out["ae_var"], out["ae_encode"] = vals, ae.var.numpy(), ae.encode(T(vals)).numpy()
        ae.var = torch.zeros(32)

        # ---------------------------------------------------------------- RayTracing stages on the analytic SDF
        from model.ray_tracing import RayTracing
        from utils import rend_util
        rt = RayTracing(object_bounding_sphere=1.0, sdf_threshold=5.0e-5, line_search_step=0.5, line_step_iters=3, sphere_tracing_iters=10,
                        n_steps=100, n_rootfind_steps=32)
        rt.eval()
        N = 211
        cam = f32([[0.1, -0.2, 1.7]])
        dirs = f32(-cam + g.standard_normal((N, 3)) * 0.45)
        dirs = f32(dirs / np.linalg.norm(dirs, axis=-1, keepdims=True))[None]
        si, mi = rend_util.get_sphere_intersection(T(cam), T(dirs), r=1.0)
        st = rt.sphere_tracing(1, N, bumpy_sdf, T(cam), T(dirs), mi, si)
        names = ("pts", "unfinished", "acc_start", "acc_end", "min_dis", "max_dis")
        out.update(rt_cam=cam, rt_dirs=dirs, rt_mask_intersect=mi.numpy(), rt_sphere_intersections=si.numpy())
        for k, v in zip(names, st):
            out["rt_st_" + k] = v.numpy()
        # the sampler on a hand-made mask (the sphere tracer itself converges on most of these rays): every 3rd intersecting ray
        smask = mi.reshape(-1).clone()
        smask[torch.arange(N) % 3 != 0] = False
        mm = torch.zeros(1, N, 2)
        mm.reshape(-1, 2)[smask, 0] = si.reshape(-1, 2)[smask, 0]
        mm.reshape(-1, 2)[smask, 1] = si.reshape(-1, 2)[smask, 1]
        obj = torch.ones(N, dtype=torch.bool)
        sp, sh, sdist = rt.ray_sampler(bumpy_sdf, T(cam), obj, T(dirs), mm, smask)
        out.update(rt_sampler_mask=smask.numpy(), rt_sampler_min_max=mm.numpy(), rt_rs_pts=sp.numpy(), rt_rs_hit=sh.numpy(), rt_rs_dist=sdist.numpy())
        k = 64
        zl, zh = f32(g.uniform(0.6, 1.0, k)), f32(g.uniform(1.6, 2.4, k))
        d_k = T(dirs[0, :k].copy())
        c_k = T(cam).expand(k, 3).contiguous()
        sl, sh2 = bumpy_sdf(c_k + T(zl)[:, None] * d_k), bumpy_sdf(c_k + T(zh)[:, None] * d_k)
        out.update(rt_sec_zl=zl, rt_sec_zh=zh, rt_sec_sl=sl.numpy(), rt_sec_sh=sh2.numpy())
        zl_t, zh_t, sl_t, sh_t = T(zl.copy()), T(zh.copy()), sl.clone(), sh2.clone()
        zp = rt.secant(sl_t, sh_t, zl_t, zh_t, c_k, d_k, bumpy_sdf)
        out.update(rt_sec_zpred=zp.numpy(), rt_sec_zl_after=zl_t.numpy(), rt_sec_zh_after=zh_t.numpy())
        steps = f32(g.uniform(0.0, 1.0, 100))
        mask = mi.reshape(-1) & (torch.arange(N) % 4 == 1)
        orig_empty = torch.empty

        class _Steps:
            def uniform_(self, a, b):
                return T(steps.copy())
        torch.empty = lambda *a, **kw: _Steps() if a == (100,) and not kw else orig_empty(*a, **kw)
        try:
            mp, md = rt.minimal_sdf_points(N, bumpy_sdf, T(cam), T(dirs).reshape(-1, 3), mask, st[4].clone(), st[5].clone())
        finally:
            torch.empty = orig_empty
        out.update(rt_min_steps=steps, rt_min_mask=mask.numpy(), rt_min_pts=mp.numpy(), rt_min_dist=md.numpy())

        # ---------------------------------------------------------------- IDRNetwork.sample_dirs / batch_idr_forward, intersect_sphere
        nrm = f32(g.standard_normal((19, 5, 3)))
        th, ph = f32(g.uniform(0, 2 * np.pi, (19, 5))), f32(g.uniform(0, np.pi / 2, (19, 5)))
        out.update(sd_normals=nrm, sd_theta=th, sd_phi=ph, sd_out=net.sample_dirs(T(nrm), T(th), T(ph)).numpy())
        bp = f32(g.standard_normal((45, 3)) * 0.2)
        bv = f32(g.standard_normal((45, 3)))
        bv = f32(bv / np.linalg.norm(bv, axis=-1, keepdims=True))
        out.update(bi_points=bp, bi_view=bv, bi_out=net.batch_idr_forward(T(bp), T(bv), n_pixels=16).detach().numpy())
        from model.octree_tracing import OctreeVisModel
        ovm = OctreeVisModel.__new__(OctreeVisModel)
        ip = f32(g.standard_normal((31, 3)) * 0.3)
        iv = f32(g.standard_normal((31, 3)) * 2.0)
        out.update(is_points=ip, is_dirs=iv, is_out=OctreeVisModel.intersect_sphere(ovm, T(ip), T(iv), radius=1.0).numpy(),
                   is_out_r07=OctreeVisModel.intersect_sphere(ovm, T(ip), T(iv), radius=0.7).numpy())

    np.savez_compressed(os.path.join(GOLD, "surface_helpers.npz"), **out)
    rep = {k: hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest()[:12] for k, v in out.items() if isinstance(v, np.ndarray)}
    json.dump({"recorded_from": "the reference's own functions under oracle/ref_shim.py (CPU)", "entries": rep},
              open(os.path.join(HERE, "PINNING_r6.json"), "w"), indent=1)
    print("wrote surface_helpers.npz:", len(out), "entries,", os.path.getsize(os.path.join(GOLD, "surface_helpers.npz")) // 1024, "KB")


if __name__ == "__main__":
    main()
