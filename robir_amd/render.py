"""Full-image forward rendering entry point (the 'serve' path; cf. the reference's scripts/relight.py:33-117 and
PBRTrainRunner.plot_to_disk, training/train_pbr.py:235-311): all chunks of one view through IDRNetwork.render_chunks,
optionally under a new light (EnvmapMaterialNetwork.load_light), tone-mapped to LDR.

    python -m robir_amd.render --synthetic --size 800 --out /tmp/view.npz
    python -m robir_amd.render --neus-ckpt logs/.../200000.tar --stage-ckpt exps/.../ModelParameters/latest.pth \
        --cameras data/hotdog/transforms_test.json --index 0 --size 800 --light envmaps/envmap3 --out view.npz
"""
import argparse
import json
import math

import numpy as np
import torch


def blender_camera(transforms_json, index, H, W):
    """SynDataset conventions (datasets/syn_dataset.py:25-84): focal from camera_angle_x, translation / 2."""
    meta = json.load(open(transforms_json))
    focal = 0.5 * W / math.tan(0.5 * float(meta["camera_angle_x"]))
    pose = np.array(meta["frames"][index]["transform_matrix"])
    pose[..., 3] /= 2.0            # the whole 4th column, homogeneous 1 included, exactly like syn_dataset.py:56-58
    pose = pose.astype(np.float32)
    K = np.array([[focal, 0, W / 2], [0, focal, H / 2], [0, 0, 1]], np.float32)
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    return np.stack([xs, ys], -1).reshape(-1, 2), pose, K       # uv = (column, row) as floats (syn_dataset.py:124-126)


def load_stage_checkpoint(model, path, only=None):
    """{'epoch', 'model_state_dict'} written by the stage runners (training/train_pbr.py:215-233), loaded with
    strict=False like the runners do (train_pbr.py:155-203).  `only`: substrings selecting the keys to take (the PBR runner
    takes only `normal_decoder_layer` from a Norm checkpoint and `indirect_illum_network` / `visibility_network` from a
    Vis checkpoint).  Missing / unexpected keys are returned AND reported: strict=False hides a key mismatch otherwise."""
    import warnings
    blob = torch.load(path, map_location="cpu", weights_only=False)
    if "model_state_dict" not in blob:
        raise KeyError(f"{path}: no 'model_state_dict' entry (found {sorted(blob)}) -- not a stage checkpoint")
    sd = blob["model_state_dict"]
    if only is not None:
        sd = {k: v for k, v in sd.items() if any(s in k for s in only)}
        if not sd:
            raise KeyError(f"{path}: no key matches {only}")
    res = model.load_state_dict(sd, strict=False)
    if res.unexpected_keys:
        warnings.warn(f"{path}: {len(res.unexpected_keys)} checkpoint tensors have no counterpart in the model "
                      f"(e.g. {res.unexpected_keys[:4]})", RuntimeWarning, stacklevel=2)
    if only is None and res.missing_keys:
        warnings.warn(f"{path}: {len(res.missing_keys)} model tensors are not in the checkpoint and keep their current "
                      f"values (e.g. {res.missing_keys[:4]})", RuntimeWarning, stacklevel=2)
    return res


def render_view(model, uv, pose, K, chunks_per_pass=125, chunk=1024, hdr_shift=None):
    dev = next(model.parameters()).device
    uv_d = torch.from_numpy(uv).to(dev)
    pose_d, K_d = torch.from_numpy(pose).to(dev), torch.from_numpy(K).to(dev)
    N = uv_d.shape[0]
    shift = model.gamma.hdr_shift.as_input().detach() if hdr_shift is None else torch.tensor([[hdr_shift]], device=dev)
    hdr = shift.expand(N, 1).contiguous()
    per = chunks_per_pass * chunk
    keys = ("sg_rgb", "indir_rgb", "diffuse_albedo", "roughness", "vis_shadow", "normal_map", "normals",
            "network_object_mask", "bg_rgb")
    acc = {k: [] for k in keys}
    for s in range(0, N, per):
        o = model.render_chunks(uv_d[s:s + per], pose_d, K_d, hdr[s:s + per], chunk=chunk)
        for k in keys:
            acc[k].append(o[k])
    out = {k: torch.cat(v) for k, v in acc.items()}
    from . import ops
    ops.range_check(sync=True)          # a finished view is a natural sync point: raise if split precision overflowed
    tm = model.gamma.hdr_shift
    hit = out["network_object_mask"][:, None]
    pred = tm.hdr2ldr(out["sg_rgb"] + out["indir_rgb"], shift)
    out["pred_rgb"] = torch.where(hit, pred, out["bg_rgb"])
    return out


def save_relight_images(out, H, W, images_dir, name, light_type="origin"):
    """The image set scripts/relight.py:60-113 writes per view: `sg_rgb_bg_<name>.png` (sg_rgb [+ indir_rgb for the
    original light], x^(1/2.2), envmap background behind the object), and for the original light `roughness_`, `albedo_`
    (x^(1/2.2)) and `normal_` ((n+1)/2).  Needs PIL."""
    import os
    from PIL import Image

    def img(x, tonemap):
        x = x.detach().float().cpu().numpy().reshape(H, W, -1)
        if x.shape[-1] == 1:
            x = np.repeat(x, 3, -1)
        if tonemap:
            x = np.power(np.clip(x, 0.0, None), 1.0 / 2.2)
        return np.clip(x, 0.0, 1.0)

    os.makedirs(images_dir, exist_ok=True)
    hit = out["network_object_mask"].detach().cpu().numpy().reshape(H, W)
    rgb = out["sg_rgb"] + out["indir_rgb"] if light_type == "origin" else out["sg_rgb"]
    rgb, bg = img(rgb, True), img(out["bg_rgb"], True)
    rgb[~hit] = bg[~hit]
    files = {"sg_rgb_bg": rgb}
    if light_type == "origin":
        files.update(roughness=img(out["roughness"], False), albedo=img(out["diffuse_albedo"], True),
                     normal=img((out["normals"] + 1.0) / 2.0, False))
    for k, v in files.items():
        Image.fromarray((v * 255).astype(np.uint8)).save(os.path.join(images_dir, "%s_%s.png" % (k, name)))
    return sorted(files)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--size", type=int, default=800)
    ap.add_argument("--neus-ckpt")
    ap.add_argument("--stage-ckpt")
    ap.add_argument("--cameras")
    ap.add_argument("--index", type=int, default=0)
    ap.add_argument("--light", help="directory holding sg_128.npy (and optionally <dir>.exr)")
    ap.add_argument("--out", required=True)
    ap.add_argument("--png-dir", help="also write the PNG set of scripts/relight.py into this directory")
    a = ap.parse_args()
    from . import renderer, synth
    dev = torch.device("cuda:0")
    H = W = a.size
    if a.synthetic:
        model = renderer.build_synthetic_model(dev)
        uv, pose, K = synth.synth_camera(H, W)
    else:
        import warnings
        with warnings.catch_warnings():  # the NeuS checkpoint is loaded explicitly on the next lines
            warnings.simplefilter("ignore", RuntimeWarning)
            model = renderer.IDRNetwork(renderer.hotdog_conf())
        from .nets import load_neus_checkpoint
        load_neus_checkpoint(model.implicit_network.neus_model, a.neus_ckpt)
        if a.stage_ckpt:
            load_stage_checkpoint(model, a.stage_ckpt)
        model = model.to(dev).eval()
        model.ray_tracer.generate(None)
        uv, pose, K = blender_camera(a.cameras, a.index, H, W)
    if a.light:
        model.envmap_material_network.load_light(a.light)
    out = render_view(model, uv, pose, K)
    np.savez_compressed(a.out, **{k: v.detach().cpu().numpy().reshape(H, W, -1) for k, v in out.items()})
    print("wrote", a.out, "hit fraction %.3f" % float(out["network_object_mask"].float().mean()))
    if a.png_dir:
        names = save_relight_images(out, H, W, a.png_dir, str(a.index), "relit" if a.light else "origin")
        print("wrote", ", ".join("%s_%d.png" % (n, a.index) for n in names), "to", a.png_dir)


if __name__ == "__main__":
    main()
