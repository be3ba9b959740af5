"""Minimal OpenEXR reader for the relighting environment maps (`EnvmapMaterialNetwork.load_light`,
model/sg_envmap_material.py:266-268, reads `<light>.exr` through imageio/FreeImage, which this image does not have).

Single-part scan-line files, HALF / FLOAT / UINT channels, compression NONE, ZIPS, ZIP (zlib + byte predictor +
interleave, decoded here with numpy) and PIZ (decoded by `rb_exr_piz_decode` in librobir_hip.so: a host routine, see
csrc/exr_piz.hip).  That covers the reference's shipped maps (envmap3: ZIP/FLOAT, envmap6 and envmap12: PIZ/HALF).
`read_exr(path)` returns float32 [H, W, C] with the channels in R, G, B(, A) order like imageio does.
"""
import ctypes
import struct
import zlib

import numpy as np

_LINES = {0: 1, 1: 1, 2: 1, 3: 16, 4: 32}      # scan lines per chunk: NONE, RLE, ZIPS, ZIP, PIZ
_NAMES = {0: "NONE", 1: "RLE", 2: "ZIPS", 3: "ZIP", 4: "PIZ", 5: "PXR24", 6: "B44", 7: "B44A", 8: "DWAA", 9: "DWAB"}
_SIZE = {0: 4, 1: 2, 2: 4}                      # bytes per sample: UINT, HALF, FLOAT
_DTYPE = {0: "<u4", 1: "<f2", 2: "<f4"}


class ExrError(ValueError):
    pass


def _header(b):
    if len(b) < 8 or struct.unpack_from("<I", b, 0)[0] != 20000630:
        raise ExrError("not an OpenEXR file")
    version = struct.unpack_from("<I", b, 4)[0]
    if version & 0x200 or version & 0x1800:
        raise ExrError("tiled / multi-part / deep OpenEXR files are not supported")
    p, attrs = 8, {}
    while True:
        e = b.index(b"\0", p)
        name = b[p:e].decode("latin1")
        p = e + 1
        if not name:
            return attrs, p
        e = b.index(b"\0", p)
        typ = b[p:e].decode("latin1")
        p = e + 1
        size = struct.unpack_from("<i", b, p)[0]
        p += 4
        if size < 0 or p + size > len(b):
            raise ExrError(f"header attribute {name!r} has an impossible size ({size})")
        attrs[name] = (typ, b[p:p + size])
        p += size


def _channels(raw):
    p, out = 0, []
    while raw[p] != 0:
        e = raw.index(b"\0", p)
        name = raw[p:e].decode("latin1")
        ptype, _, xs, ys = struct.unpack_from("<iB3xii", raw, e + 1)
        if xs != 1 or ys != 1:
            raise ExrError("sub-sampled channels are not supported")
        if ptype not in _SIZE:
            raise ExrError(f"unknown pixel type {ptype}")
        out.append((name, ptype))
        p = e + 1 + 16
    return out


def _unzip(data, n):
    t = np.frombuffer(zlib.decompress(data), dtype=np.uint8)
    if t.size != n:
        raise ExrError("ZIP chunk has the wrong size")
    t = (np.cumsum(t.astype(np.int64) - np.concatenate(([0], np.full(n - 1, 128)))) & 0xFF).astype(np.uint8)   # d[i] += d[i-1] - 128
    half = (n + 1) // 2
    out = np.empty(n, dtype=np.uint8)
    out[0::2] = t[:half]
    out[1::2] = t[half:]
    return out.tobytes()


def _unrle(data, n):
    out, p = bytearray(), 0
    while p < len(data):
        c = data[p] - 256 if data[p] > 127 else data[p]
        p += 1
        if c < 0:
            out += data[p:p - c]
            p -= c
        else:
            out += bytes([data[p]]) * (c + 1)
            p += 1
    if len(out) != n:
        raise ExrError("RLE chunk has the wrong size")
    t = np.frombuffer(bytes(out), dtype=np.uint8)
    t = (np.cumsum(t.astype(np.int64) - np.concatenate(([0], np.full(n - 1, 128)))) & 0xFF).astype(np.uint8)
    half = (n + 1) // 2
    o = np.empty(n, dtype=np.uint8)
    o[0::2] = t[:half]
    o[1::2] = t[half:]
    return o.tobytes()


def _unpiz(data, chans, width, lines):
    """-> the chunk in the plain layout ([line][channel][pixel]) as bytes."""
    from . import _lib
    spec = np.array([[width, lines, _SIZE[t] // 2] for _, t in chans], dtype=np.int32)
    n = int((spec[:, 0] * spec[:, 1] * spec[:, 2]).sum())
    out = np.empty(n, dtype=np.uint16)
    src = np.frombuffer(data, dtype=np.uint8)
    _lib.call("rb_exr_piz_decode", src.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(src.size),
              spec.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(len(chans)), out.ctypes.data_as(ctypes.c_void_p),
              ctypes.c_long(n))
    blocks, p = [], 0
    for (_, t), (nx, ny, sz) in zip(chans, spec):
        blocks.append(out[p:p + nx * ny * sz].reshape(ny, nx * sz))
        p += nx * ny * sz
    return np.concatenate(blocks, axis=1).tobytes()      # [line][channel][pixel words]


def read_exr(path):
    """float32 [H, W, C] (C in R, G, B, A order).  Anything malformed or unsupported raises ExrError."""
    try:
        return _read(path)
    except ExrError:
        raise
    except (struct.error, IndexError, zlib.error, ValueError, RuntimeError) as e:
        raise ExrError(f"{path}: malformed OpenEXR file ({type(e).__name__}: {e})") from e


def _read(path):
    b = open(path, "rb").read()
    attrs, p = _header(b)
    for need in ("channels", "compression", "dataWindow"):
        if need not in attrs:
            raise ExrError(f"missing header attribute {need}")
    chans = _channels(attrs["channels"][1])
    comp = attrs["compression"][1][0]
    if comp not in _LINES:
        raise ExrError(f"compression {_NAMES.get(comp, comp)} is not supported (NONE, RLE, ZIPS, ZIP, PIZ are)")
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    W, H = x1 - x0 + 1, y1 - y0 + 1
    if not (0 < W <= 65536 and 0 < H <= 65536):
        raise ExrError(f"implausible dataWindow ({x0},{y0})-({x1},{y1})")
    per = _LINES[comp]
    n_chunks = (H + per - 1) // per
    offsets = struct.unpack_from("<%dQ" % n_chunks, b, p)
    line_bytes = sum(_SIZE[t] for _, t in chans) * W
    planes = {name: np.empty((H, W), dtype=np.float32) for name, _ in chans}
    for off in offsets:
        if off + 8 > len(b):
            raise ExrError("chunk offset beyond the end of the file")
        y, size = struct.unpack_from("<ii", b, off)
        if not (y0 <= y <= y1) or (y - y0) % per != 0:
            raise ExrError(f"chunk starts at scan line {y}, outside the data window / chunk grid")
        if size < 0 or off + 8 + size > len(b):
            raise ExrError("chunk size runs past the end of the file")
        data = b[off + 8:off + 8 + size]
        lines = min(per, y1 + 1 - y)
        raw_n = line_bytes * lines
        if size == raw_n or comp == 0:
            raw = data
        elif comp in (2, 3):
            raw = _unzip(data, raw_n)
        elif comp == 1:
            raw = _unrle(data, raw_n)
        else:
            raw = _unpiz(data, chans, W, lines)
        q = 0
        for ln in range(lines):
            for name, t in chans:
                v = np.frombuffer(raw, dtype=_DTYPE[t], count=W, offset=q)
                planes[name][y - y0 + ln] = v.astype(np.float32)
                q += _SIZE[t] * W
    order = [c for c in ("R", "G", "B", "A") if c in planes] or [n for n, _ in chans]
    return np.stack([planes[c] for c in order], -1)
