"""Cut a small OpenEXR fixture out of one of the reference's environment maps (data, not source): the first 32 scan
lines (= one PIZ chunk) of /root/reference/envmaps/envmap6.exr re-wrapped as a 1024 x 32 file, plus a ZIP/FLOAT one from
envmap3.exr (first 16 lines).  Runs only in the build container.

Expected values: no independent EXR decoder exists in this image (imageio / OpenEXR / cv2 are absent), so the stored
statistics come from robir_amd.exr itself (regression pin).  Independent evidence recorded in tests/golden/exr_expected.json:
the alpha plane of envmap6 decodes to exactly 1.0 everywhere, and the decoded full maps agree with the reference's own SG fits
of them (envmaps/envmap*/sg_128.npy): log-space correlation and mean energy, computed here.
"""
import hashlib
import json
import os
import struct
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from robir_amd import exr  # noqa: E402
from robir_oracle import sg as osg  # noqa: E402

REF = "/root/reference/envmaps"
GOLD = os.path.join(ROOT, "tests", "golden")


def cut(src, dst, n_chunks):
    b = open(src, "rb").read()
    attrs, p = exr._header(b)
    comp = attrs["compression"][1][0]
    per = exr._LINES[comp]
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    total = (y1 - y0 + per) // per
    offs = struct.unpack_from("<%dQ" % total, b, p)
    out = bytearray(b[:8])
    new_y1 = y0 + per * n_chunks - 1
    for name, (typ, val) in attrs.items():
        if name in ("dataWindow", "displayWindow"):
            val = struct.pack("<4i", x0, y0, x1, new_y1)
        out += name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(val)) + val
    out += b"\0"
    chunks = []
    for o in offs[:n_chunks]:
        y, size = struct.unpack_from("<ii", b, o)
        chunks.append(b[o:o + 8 + size])
    pos = len(out) + 8 * n_chunks
    for c in chunks:
        out += struct.pack("<Q", pos)
        pos += len(c)
    for c in chunks:
        out += c
    open(dst, "wb").write(bytes(out))


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<f4").tobytes()).hexdigest()[:16]


def main():
    exp = {}
    cut(os.path.join(REF, "envmap6.exr"), os.path.join(GOLD, "envmap6_rows0_31.exr"), 1)
    cut(os.path.join(REF, "envmap3.exr"), os.path.join(GOLD, "envmap3_rows0_15.exr"), 1)
    for name, full in (("envmap6_rows0_31", "envmap6"), ("envmap3_rows0_15", "envmap3")):
        part = exr.read_exr(os.path.join(GOLD, name + ".exr"))
        whole = exr.read_exr(os.path.join(REF, full + ".exr"))
        assert np.array_equal(part, whole[:part.shape[0]])
        exp[name] = {"shape": list(part.shape), "sha256_16": digest(part), "mean": [float(v) for v in part.reshape(-1, part.shape[-1]).mean(0)],
                     "max": float(part.max()), "min": float(part.min())}
    for n, (H, W, f) in {"envmap3": (50, 100, 5), "envmap6": (64, 128, 8), "envmap12": (64, 128, 8)}.items():
        im = exr.read_exr(os.path.join(REF, n + ".exr"))
        rgb = im[..., :3]
        sgs = torch.from_numpy(np.load(os.path.join(REF, n, "sg_128.npy"))).float()
        g = osg.envmap_grid(sgs, H, W).numpy()
        ds = rgb.reshape(H, f, W, f, 3).mean((1, 3))
        exp["evidence_" + n] = {"log_corr_with_reference_sg_fit": float(np.corrcoef(np.log1p(ds).ravel(), np.log1p(g).ravel())[0, 1]),
                                "mean_exr": float(ds.mean()), "mean_sg_fit": float(g.mean()), "shape": list(im.shape),
                                "alpha_all_one": bool(im.shape[-1] == 4 and (im[..., 3] == 1.0).all())}
    json.dump(exp, open(os.path.join(GOLD, "exr_expected.json"), "w"), indent=1)
    print(json.dumps(exp, indent=1))


if __name__ == "__main__":
    main()
