"""State dict (reference key names) -> packed weight blobs for the HIP MLP kernels.

Layouts are defined by include/robir_hip.h ("Weight packing") and robir_amd/csrc/mlp_engine.h.  Packing itself
runs on the device (rb_pack_layer); this module only decides layer order, padding and input-column permutations.
Weight-norm (model/neus_model.py:378-379) is folded once here: W = g * v / |v|_row.
"""
import ctypes
import os

import torch

from . import _lib

SDF = "implicit_network.neus_model.sdf_network."
COL = "implicit_network.neus_model.color_network."
VIS = "visibility_network.vis_layer."
ILL = "indirect_illum_network."
MAT = "envmap_material_network."


def _pad16(n):
    return (n + 15) // 16 * 16


def _fold_wn(sd, prefix):
    v, g = sd[prefix + "weight_v"].float(), sd[prefix + "weight_g"].float()
    return (v * (g / v.norm(dim=1, keepdim=True))).contiguous()


def pack_layers(layers, device):
    """layers: list of dict(W[N,K], b[N] or None, n_pad, k_pad, perm (list[int]|None)).  Returns one fp32 blob."""
    L = _lib.lib()
    sizes = [int(L.rb_packed_layer_floats(l["n_pad"], l["k_pad"])) for l in layers]
    blob = torch.empty(sum(sizes), dtype=torch.float32, device=device)
    off = 0
    keep = []
    for l, sz in zip(layers, sizes):
        W = l["W"].to(device=device, dtype=torch.float32).contiguous()
        b = l["b"].to(device=device, dtype=torch.float32).contiguous() if l.get("b") is not None else None
        perm = None
        if l.get("perm") is not None:
            perm = torch.tensor(l["perm"], dtype=torch.int32, device=device)
            assert perm.numel() == l["k_pad"]
        keep += [W, b, perm]
        out = blob[off:off + sz]
        _lib.call("rb_pack_layer", _lib.ptr(W), _lib.ptr(b), ctypes.c_int(W.shape[0]), ctypes.c_int(W.shape[1]),
                  ctypes.c_int(l["n_pad"]), ctypes.c_int(l["k_pad"]), _lib.ptr(perm), ctypes.c_float(1.0),
                  ctypes.c_void_p(out.data_ptr()), _lib.stream_ptr())
        off += sz
    torch.cuda.current_stream().synchronize()   # W/b/perm temporaries must outlive the pack kernels
    return blob


H3_SCALE_LOG2 = 8      # weights * 2^8 before the hi/lo half split (keeps lo parts out of the f16 subnormal range)


def pack_layers_h3(layers, device, scale_log2=H3_SCALE_LOG2):
    """Split-precision (f16x3) packing of 256-wide hidden layers: list of dict(W, b, n_pad, k_pad)."""
    L = _lib.lib()
    sizes = [int(L.rb_packed_layer_floats(l["n_pad"], l["k_pad"])) for l in layers]
    blob = torch.empty(sum(sizes), dtype=torch.float32, device=device)
    off, keep = 0, []
    for l, sz in zip(layers, sizes):
        W = l["W"].to(device=device, dtype=torch.float32).contiguous()
        b = l["b"].to(device=device, dtype=torch.float32).contiguous() if l.get("b") is not None else None
        wmax = float(W.abs().max()) * 2.0 ** scale_log2
        if not wmax < 65504.0:       # the hi half would overflow to inf: fail here, not as NaNs in the image
            raise ValueError("split-precision packing: |w| * 2^%d = %.3g exceeds the f16 range; run this network with "
                             "ROBIR_MLP_PRECISION=fp32 / ROBIR_VIS_PRECISION=fp32" % (scale_log2, wmax))
        perm = None
        if l.get("perm") is not None:
            perm = torch.tensor(l["perm"], dtype=torch.int32, device=device)
            assert perm.numel() == l["k_pad"]
        keep += [W, b, perm]
        _lib.call("rb_pack_layer_h3", _lib.ptr(W), _lib.ptr(b), ctypes.c_int(W.shape[0]), ctypes.c_int(W.shape[1]),
                  ctypes.c_int(l["n_pad"]), ctypes.c_int(l["k_pad"]), _lib.ptr(perm), ctypes.c_int(scale_log2),
                  ctypes.c_void_p(blob[off:off + sz].data_ptr()), _lib.stream_ptr())
        off += sz
    torch.cuda.current_stream().synchronize()
    return blob


def pack_layers_x6(layers, device, scale_log2=H3_SCALE_LOG2):
    """Exact-operand (f16x6) packing of 256-wide hidden layers: every weight as three halves (rb_pack_layer_x6)."""
    L = _lib.lib()
    sizes = [int(L.rb_packed_layer_x6_floats(l["n_pad"], l["k_pad"])) for l in layers]
    blob = torch.empty(sum(sizes), dtype=torch.float32, device=device)
    off, keep = 0, []
    for l, sz in zip(layers, sizes):
        W = l["W"].to(device=device, dtype=torch.float32).contiguous()
        b = l["b"].to(device=device, dtype=torch.float32).contiguous() if l.get("b") is not None else None
        wmax = float(W.abs().max()) * 2.0 ** scale_log2
        if not wmax < 65504.0:
            raise ValueError("exact-operand packing: |w| * 2^%d = %.3g exceeds the f16 range; run this network with "
                             "ROBIR_VIS_PRECISION=fp32 (light-visibility kernel) / ROBIR_MLP_PRECISION=fp32 (SDF, colour, visibility, 512-wide and "
                             "CESR nets)" % (scale_log2, wmax))
        keep += [W, b]
        perm = None
        if l.get("perm") is not None:
            perm = torch.tensor(l["perm"], dtype=torch.int32, device=device)
            assert perm.numel() == l["k_pad"]
        keep.append(perm)
        _lib.call("rb_pack_layer_x6", _lib.ptr(W), _lib.ptr(b), ctypes.c_int(W.shape[0]), ctypes.c_int(W.shape[1]),
                  ctypes.c_int(l["n_pad"]), ctypes.c_int(l["k_pad"]), _lib.ptr(perm), ctypes.c_int(scale_log2),
                  ctypes.c_void_p(blob[off:off + sz].data_ptr()), _lib.stream_ptr())
        off += sz
    torch.cuda.current_stream().synchronize()
    return blob


def _t(sd, k):
    v = sd[k]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(v)


def pack_vis(sd, device):
    """[128->256, 256->256 x3, 256->16]  (VisNetwork, implicit_differentiable_renderer.py:241-248)."""
    ls = []
    for i in range(5):
        W, b = _t(sd, VIS + "%d.weight" % (2 * i)), _t(sd, VIS + "%d.bias" % (2 * i))
        ls.append(dict(W=W, b=b, n_pad=_pad16(W.shape[0]), k_pad=_pad16(W.shape[1])))
    return pack_layers(ls, device)


def pack_vis_h3(sd, device):
    """Same five layers in split-precision form (rb_vis_mlp_h3)."""
    ls = []
    for i in range(5):
        W, b = _t(sd, VIS + "%d.weight" % (2 * i)), _t(sd, VIS + "%d.bias" % (2 * i))
        ls.append(dict(W=W, b=b, n_pad=_pad16(W.shape[0]), k_pad=(W.shape[1] + 31) // 32 * 32))
    return pack_layers_h3(ls, device)


def pack_vis_x6(sd, device):
    """The five layers with every weight as three halves, scale 2^0 (rb_vis_x6_points, csrc/vis_x6.hip)."""
    ls = []
    for i in range(5):
        W, b = _t(sd, VIS + "%d.weight" % (2 * i)), _t(sd, VIS + "%d.bias" % (2 * i))
        ls.append(dict(W=W, b=b, n_pad=_pad16(W.shape[0]), k_pad=(W.shape[1] + 31) // 32 * 32))
    blob = pack_layers_x6(ls, device, scale_log2=0)
    return torch.cat([blob, torch.zeros(2048, device=device)])


def pack_vis_split(sd, device):
    """First visibility layer split into its point half (with bias) and direction half (no bias), each 64->256,
    plus the hidden stack [256->256 x3] and the 2x256 output layer kept in plain row-major form."""
    W0, b0 = _t(sd, VIS + "0.weight").float(), _t(sd, VIS + "0.bias").float()
    wp = pack_layers([dict(W=W0[:, :63].contiguous(), b=b0, n_pad=256, k_pad=64)], device)
    wd = pack_layers([dict(W=W0[:, 63:].contiguous(), b=None, n_pad=256, k_pad=64)], device)
    hl = [dict(W=_t(sd, VIS + "%d.weight" % (2 * i)), b=_t(sd, VIS + "%d.bias" % (2 * i)), n_pad=256, k_pad=256)
          for i in (1, 2, 3)]
    hid = pack_layers(hl, device)
    w_last = _t(sd, VIS + "8.weight").to(device=device, dtype=torch.float32).contiguous()      # [2,256]
    b_last = _t(sd, VIS + "8.bias").to(device=device, dtype=torch.float32).contiguous()        # [2]
    # 49 chunks: hidden stack + the output layer as one more 16-row chunk, every weight as three halves (rb_dvis_fused_x6t: exact fp32
    # operands on the f16 MFMA); the m / l pieces carry their own 2^11 / 2^22: no lift needed
    x6_scale = 0
    hid_x6_head = pack_layers_x6(hl + [dict(W=w_last, b=b_last, n_pad=16, k_pad=256)], device, scale_log2=x6_scale)
    return _VisSplit(dict(point=wp, dir=wd, hidden=hid, h3_scale_log2=H3_SCALE_LOG2, hidden_x6_head=hid_x6_head, x6_head_scale_log2=x6_scale,
                          h3_head_scale_log2=H3_SCALE_LOG2, w_last=w_last, b_last=b_last), hl, device)


def pack_vis_f16_head(x6_head):
    """The weights of the f16 THROUGHPUT mode's second-generation kernel (k_dvis_f16t2, csrc/vis_diffuse_f16t.hip) as a blob of their own:
    [49 chunks x 16 biases] then the h pieces (the weights rounded to f16) of the 49 chunks, 8 KB each ([k-block 8][lane 64] x 16 B),
    contiguous -- cut out of the exact-operand blob (a chunk there: 16 biases, then [k-block][piece h, m, l][lane]), so that the two
    generations multiply the same halves."""
    x = x6_head.view(-1)[:49 * 1540 * 4].view(49, 1540, 4)
    bias = x[:, :4].reshape(-1)
    h = x[:, 4:].reshape(49, 8, 3, 64, 4)[:, :, 0].reshape(-1)
    return torch.cat([bias, h]).contiguous()


class _VisSplit(dict):
    """pack_vis_split's blobs; the split-precision ones (`hidden_h3`: the hidden stack, `hidden_h3_head`: the 49 chunks of
    rb_dvis_fused_v2 / rb_dvis_stream) are packed by the LEGACY library on first use -- the default policy never asks for them."""

    def __init__(self, d, hl, device):
        super().__init__(d)
        self._hl, self._device = hl, device

    def __missing__(self, k):
        if k == "hidden_h3":
            v = pack_layers_h3(self._hl, self._device)
        elif k == "hidden_h3_head":
            v = pack_layers_h3(self._hl + [dict(W=self["w_last"], b=self["b_last"], n_pad=16, k_pad=256)], self._device)
        elif k == "hidden_f16_head":
            v = pack_vis_f16_head(self["hidden_x6_head"])
        elif k == "hidden_x6_head_fp8":      # the 49 chunks in the bf8 layout (k_dvis_x6t built with -DXT_FP8=1; format 8)
            v = repack_x6_chunks_fp8(self["hidden_x6_head"], self._device, ((256, 49),))
        else:
            raise KeyError(k)
        self[k] = v
        return v


def pack_sdf(sd, device, full=True):
    """[64->256, 256->256 x2, 256->208, 272->256, 256->256 x3, 256->272|16]  (SDFNetwork, neus_model.py:350-381)."""
    sdt = {k: _t(sd, k) for k in sd if k.startswith(SDF)}
    ls = []
    for l in range(9):
        W = _fold_wn(sdt, SDF + "lin%d." % l)
        b = sdt[SDF + "lin%d.bias" % l].float()
        n_pad, k_pad, perm = _pad16(W.shape[0]), _pad16(W.shape[1]), None
        if l == 4:   # input = cat[act(lin3) (193), PE (63)] -> packed order [208 slots | 64 slots]
            k_pad = 272
            perm = [k if k < 193 else -1 for k in range(208)] + [193 + j if j < 63 else -1 for j in range(64)]
        if l == 8 and not full:
            W, b, n_pad = W[:1].contiguous(), b[:1].contiguous(), 16
        ls.append(dict(W=W, b=b, n_pad=n_pad, k_pad=k_pad, perm=perm))
    return pack_layers(ls, device)


def _pad32(n):
    return (n + 31) // 32 * 32


def pack_sdf_h3(sd, device, full=True):
    """pack_sdf in split-precision form (rb_sdf_mlp_h3): K padded to multiples of 32, the skip layer to 288 slots."""
    sdt = {k: _t(sd, k) for k in sd if k.startswith(SDF)}
    ls = []
    for l in range(9):
        W = _fold_wn(sdt, SDF + "lin%d." % l)
        b = sdt[SDF + "lin%d.bias" % l].float()
        n_pad, k_pad, perm = _pad16(W.shape[0]), _pad32(W.shape[1]), None
        if l == 4:   # input = cat[act(lin3) (193), PE (63)] -> packed order [208 slots | 64 slots | 16 zero slots]
            k_pad = 288
            perm = [k if k < 193 else -1 for k in range(208)] + [193 + j if j < 63 else -1 for j in range(64)] + [-1] * 16
        if l == 8 and not full:
            W, b, n_pad = W[:1].contiguous(), b[:1].contiguous(), 16
        ls.append(dict(W=W, b=b, n_pad=n_pad, k_pad=k_pad, perm=perm))
    return pack_layers_h3(ls, device)


def pack_sdf_x6(sd, device, full=True):
    """pack_sdf_h3's layout with every weight as three halves, scale 2^0 (rb_sdf_x6_points, csrc/sdf_x6.hip)."""
    sdt = {k: _t(sd, k) for k in sd if k.startswith(SDF)}
    ls = []
    for l in range(9):
        W = _fold_wn(sdt, SDF + "lin%d." % l)
        b = sdt[SDF + "lin%d.bias" % l].float()
        n_pad, k_pad, perm = _pad16(W.shape[0]), _pad32(W.shape[1]), None
        if l == 4:
            k_pad = 288
            perm = [k if k < 193 else -1 for k in range(208)] + [193 + j if j < 63 else -1 for j in range(64)] + [-1] * 16
        if l == 8 and not full:
            W, b, n_pad = W[:1].contiguous(), b[:1].contiguous(), 16
        ls.append(dict(W=W, b=b, n_pad=n_pad, k_pad=k_pad, perm=perm))
    blob = pack_layers_x6(ls, device, scale_log2=0)
    return torch.cat([blob, torch.zeros(2048, device=device)])       # a wave's shifted-back copy span never leaves the blob; slack for the last chunk


def pack_sdf_back_h3(sd, device):
    """Transposed layers for the reverse-mode gradient (rb_sdf_value_grad, csrc/sdf_back.hip): W7^T, W6^T, W5^T, W4^T split
    into [193 continuing rows -> 208 | 63 skip-feature rows -> 64 | 48 zero rows], W3^T (K 193 -> 224), W2^T, W1^T, W0^T; and
    row 0 of layer 8.  -> (blob, w8row [256])."""
    sdt = {k: _t(sd, k) for k in sd if k.startswith(SDF)}
    W = [_fold_wn(sdt, SDF + "lin%d." % l) for l in range(9)]
    m4 = torch.zeros(320, 256)
    m4[:193] = W[4][:, :193].t()
    m4[208:271] = W[4][:, 193:256].t()
    mats = [W[7].t(), W[6].t(), W[5].t(), m4, W[3].t(), W[2].t(), W[1].t(), W[0].t()]
    ls = [dict(W=m.contiguous(), b=None, n_pad=_pad16(m.shape[0]), k_pad=_pad32(m.shape[1]), perm=None) for m in mats]
    assert [l["n_pad"] for l in ls] == [256, 256, 256, 320, 256, 256, 256, 64] and ls[4]["k_pad"] == 224
    blob = pack_layers_h3(ls, device)
    blob = torch.cat([blob, torch.zeros(1024, device=device)])      # the last DMA row of a K = 224 chunk reads 2 KB further
    return blob, W[8][0].contiguous().to(device)


def pack_sdf_back(sd, device):
    """pack_sdf_back_h3 for the f32-input MFMA (rb_sdf_value_grad_f32_points, k_sdf_back_f32 in csrc/mlp_kernels.hip): W7^T, W6^T, W5^T,
    W4^T as [193 continuing rows -> 208 | 63 skip-feature rows -> 64] (N = 272), W3^T (K 193 -> 208), W2^T, W1^T, W0^T (N 63 -> 64),
    no biases; and row 0 of layer 8.  -> (blob, w8row [256])."""
    sdt = {k: _t(sd, k) for k in sd if k.startswith(SDF)}
    W = [_fold_wn(sdt, SDF + "lin%d." % l) for l in range(9)]
    m4 = torch.zeros(272, 256)
    m4[:193] = W[4][:, :193].t()
    m4[208:271] = W[4][:, 193:256].t()
    mats = [W[7].t(), W[6].t(), W[5].t(), m4, W[3].t(), W[2].t(), W[1].t(), W[0].t()]
    ls = [dict(W=m.contiguous(), b=None, n_pad=_pad16(m.shape[0]), k_pad=_pad16(m.shape[1]), perm=None) for m in mats]
    assert [l["n_pad"] for l in ls] == [256, 256, 256, 272, 256, 256, 256, 64] and ls[4]["k_pad"] == 208
    return pack_layers(ls, device), W[8][0].contiguous().to(device)


def pack_sdf_back_x6(sd, device, two_tile=False):
    """pack_sdf_back with every weight as three halves, scale 2^0 (k_sdf_back_x6, csrc/sdf_back_x6.hip): W3^T's K 193 -> 224;
    two_tile: K -> 256, every chunk of the stream one shape (k_sdf_back_x6t, csrc/sdf_back_x6t.hip)."""
    sdt = {k: _t(sd, k) for k in sd if k.startswith(SDF)}
    W = [_fold_wn(sdt, SDF + "lin%d." % l) for l in range(9)]
    m4 = torch.zeros(272, 256)
    m4[:193] = W[4][:, :193].t()
    m4[208:271] = W[4][:, 193:256].t()
    mats = [W[7].t(), W[6].t(), W[5].t(), m4, W[3].t(), W[2].t(), W[1].t(), W[0].t()]
    ls = [dict(W=m.contiguous(), b=None, n_pad=_pad16(m.shape[0]), k_pad=_pad32(m.shape[1]), perm=None) for m in mats]
    assert [l["n_pad"] for l in ls] == [256, 256, 256, 272, 256, 256, 256, 64] and ls[4]["k_pad"] == 224
    if two_tile:
        ls[4]["k_pad"] = 256
    blob = pack_layers_x6(ls, device, scale_log2=0)
    return torch.cat([blob, torch.zeros(2048, device=device)]), W[8][0].contiguous().to(device)


def pack_color(sd, device):
    """[304->256 (cols permuted to [feat|x|PE4(view)|normal]), 256->256 x3, 256->16]  (neus_model.py:511-531)."""
    sdt = {k: _t(sd, k) for k in sd if k.startswith(COL)}
    ls = []
    for l in range(5):
        W = _fold_wn(sdt, COL + "lin%d." % l)
        b = sdt[COL + "lin%d.bias" % l].float()
        perm = None
        if l == 0:
            perm = [33 + k for k in range(256)] + list(range(33)) + [-1] * 15
        ls.append(dict(W=W, b=b, n_pad=_pad16(W.shape[0]), k_pad=304 if l == 0 else _pad16(W.shape[1]), perm=perm))
    return pack_layers(ls, device)


def pack_color_h3(sd, device):
    """pack_color in split-precision form (rb_color_mlp_h3): first layer padded to 320 slots."""
    sdt = {k: _t(sd, k) for k in sd if k.startswith(COL)}
    ls = []
    for l in range(5):
        W = _fold_wn(sdt, COL + "lin%d." % l)
        b = sdt[COL + "lin%d.bias" % l].float()
        perm = None
        if l == 0:
            perm = [33 + k for k in range(256)] + list(range(33)) + [-1] * 31
        ls.append(dict(W=W, b=b, n_pad=_pad16(W.shape[0]), k_pad=320 if l == 0 else _pad32(W.shape[1]), perm=perm))
    return pack_layers_h3(ls, device)


def pack_color_x6(sd, device):
    """pack_color_h3's layout with every weight as three halves, scale 2^0 (rb_color_x6_points, csrc/color_x6.hip)."""
    sdt = {k: _t(sd, k) for k in sd if k.startswith(COL)}
    ls = []
    for l in range(5):
        W = _fold_wn(sdt, COL + "lin%d." % l)
        b = sdt[COL + "lin%d.bias" % l].float()
        perm = None
        if l == 0:
            perm = [33 + k for k in range(256)] + list(range(33)) + [-1] * 31
        ls.append(dict(W=W, b=b, n_pad=_pad16(W.shape[0]), k_pad=320 if l == 0 else _pad32(W.shape[1]), perm=perm))
    blob = pack_layers_x6(ls, device, scale_log2=0)
    return torch.cat([blob, torch.zeros(2048, device=device)])


def pack_illum(sd, device):
    ls = []
    for i in range(5):
        W, b = _t(sd, ILL + "lobe_layer.%d.weight" % (2 * i)), _t(sd, ILL + "lobe_layer.%d.bias" % (2 * i))
        ls.append(dict(W=W, b=b, n_pad=_pad16(W.shape[0]), k_pad=_pad16(W.shape[1])))
    return pack_layers(ls, device)


def pack_illum_h3(sd, device):
    ls = []
    for i in range(5):
        W, b = _t(sd, ILL + "lobe_layer.%d.weight" % (2 * i)), _t(sd, ILL + "lobe_layer.%d.bias" % (2 * i))
        ls.append(dict(W=W, b=b, n_pad=_pad16(W.shape[0]), k_pad=_pad32(W.shape[1])))
    return pack_layers_h3(ls, device)


def pack_sparse_ae_encoder_h3(sd, prefix, device):
    """Encoder [64->512, 512->512 x3, 512->32] in split-precision form (rb_wide_mlp_h3); the small decoder stays fp32."""
    enc = []
    for i in range(5):
        W = _t(sd, prefix + ".brdf_encoder_layer.%d.weight" % (2 * i))
        b = _t(sd, prefix + ".brdf_encoder_layer.%d.bias" % (2 * i))
        enc.append(dict(W=W, b=b, n_pad=_pad16(W.shape[0]), k_pad=_pad32(W.shape[1])))
    return pack_layers_h3(enc, device)


def pack_wide_x6(layers, device):
    """Five layers [64->512, 512->512 x3, 512->32 | 144] (dicts W, b) with every weight as three halves, scale 2^0 (rb_wide_x6_points,
    csrc/wide_x6.hip)."""
    ls = [dict(W=l["W"], b=l["b"], n_pad=_pad16(l["W"].shape[0]), k_pad=_pad32(l["W"].shape[1])) for l in layers]
    assert [l["k_pad"] for l in ls] == [64, 512, 512, 512, 512] and ls[0]["n_pad"] == 512
    blob = pack_layers_x6(ls, device, scale_log2=0)
    return torch.cat([blob, torch.zeros(2048, device=device)])


def pack_sparse_ae_encoder_x6(sd, prefix, device):
    return pack_wide_x6([dict(W=_t(sd, prefix + ".brdf_encoder_layer.%d.weight" % (2 * i)), b=_t(sd, prefix + ".brdf_encoder_layer.%d.bias" % (2 * i)))
                         for i in range(5)], device)


def pack_illum_x6(sd, device):
    return pack_wide_x6([dict(W=_t(sd, ILL + "lobe_layer.%d.weight" % (2 * i)), b=_t(sd, ILL + "lobe_layer.%d.bias" % (2 * i))) for i in range(5)], device)


def pack_sparse_ae(sd, prefix, device):
    """-> (encoder blob [64->512, 512->512 x3, 512->32], decoder blob [32->128, 128->128, 128->16])."""
    enc, dec = [], []
    for i in range(5):
        W = _t(sd, prefix + ".brdf_encoder_layer.%d.weight" % (2 * i))
        b = _t(sd, prefix + ".brdf_encoder_layer.%d.bias" % (2 * i))
        enc.append(dict(W=W, b=b, n_pad=_pad16(W.shape[0]), k_pad=_pad16(W.shape[1])))
    for i in range(3):
        W = _t(sd, prefix + ".brdf_decoder_layer.%d.weight" % (2 * i))
        b = _t(sd, prefix + ".brdf_decoder_layer.%d.bias" % (2 * i))
        dec.append(dict(W=W, b=b, n_pad=_pad16(W.shape[0]), k_pad=_pad16(W.shape[1])))
    return pack_layers(enc, device), pack_layers(dec, device)


def pack_softplus512(sd, prefix, k_in, device):
    """512 x 8 SDFNetwork-style net of the CESR stage (multires 0, skip [4]); k_in = 63 (normal_net) or 191 (shadow_net).
    [K0P->512, 512->512 x2, 512->N3P, 528->512 (cols [lin3 | input]), 512->512 x3, 512->16]."""
    sdt = {k: _t(sd, k) for k in sd if k.startswith(prefix)}
    k0p = _pad16(k_in)
    n3 = 512 - k_in
    n3p = _pad16(n3)
    assert n3p + k0p == 528
    ls = []
    for l in range(9):
        W = _fold_wn(sdt, prefix + "lin%d." % l)
        b = sdt[prefix + "lin%d.bias" % l].float()
        n_pad, k_pad, perm = _pad16(W.shape[0]), _pad16(W.shape[1]), None
        if l == 4:
            k_pad = 528
            perm = [k if k < n3 else -1 for k in range(n3p)] + [n3 + j if j < k_in else -1 for j in range(k0p)]
        ls.append(dict(W=W, b=b, n_pad=n_pad, k_pad=k_pad, perm=perm))
    return pack_layers(ls, device)


def pack_softplus512_x6(sd, prefix, k_in, device):
    """pack_softplus512 with every weight as three halves, scale 2^0 (rb_cesr_net_x6_points, csrc/cesr_x6.hip): the skip layer padded to
    576 slots = [lin3 part | input part | 48 zero slots] (two half-chunks of K = 288)."""
    sdt = {k: _t(sd, k) for k in sd if k.startswith(prefix)}
    k0p = _pad16(k_in)
    n3 = 512 - k_in
    n3p = _pad16(n3)
    assert n3p + k0p == 528 and k0p % 32 == 0
    ls = []
    for l in range(9):
        W = _fold_wn(sdt, prefix + "lin%d." % l)
        b = sdt[prefix + "lin%d.bias" % l].float()
        n_pad, k_pad, perm = _pad16(W.shape[0]), _pad32(W.shape[1]), None
        if l == 4:
            k_pad = 576
            perm = ([k if k < n3 else -1 for k in range(n3p)] + [n3 + j if j < k_in else -1 for j in range(k0p)] + [-1] * 48)
        ls.append(dict(W=W, b=b, n_pad=n_pad, k_pad=k_pad, perm=perm))
    blob = pack_layers_x6(ls, device, scale_log2=0)
    return torch.cat([blob, torch.zeros(2048, device=device)])


def pack_softplus512_f16(sd, prefix, k_in, device):
    """The CESR nets for the f16 THROUGHPUT mode (rb_cesr_net_f16_points, csrc/cesr_f16.hip): the h pieces (the weights as halves) cut out
    of the exact-operand blob, so that both kernels multiply the same halves -- [16 biases per chunk, all chunks of the stream] then the
    chunks' fragments ([k-block][lane 64] x 16 B), contiguous.  Chunk K per layer: K0P, 512 x3, 576 (skip), 512 x4."""
    x6 = pack_softplus512_x6(sd, prefix, k_in, device).view(-1)
    k0p = _pad16(k_in)
    n3p = _pad16(512 - k_in)
    ks = [k0p, 512, 512, 512, 576, 512, 512, 512, 512]
    nch = [32, 32, 32, n3p // 16, 32, 32, 32, 32, 1]
    bias, frags, off = [], [], 0
    for K, n in zip(ks, nch):
        cf = 4 * (4 + 6 * K)                          # floats of a packed exact-operand chunk: 16 biases + K / 32 x 3 pieces x 64 lanes x 4
        lay = x6[off:off + n * cf].view(n, cf)
        bias.append(lay[:, :16].reshape(-1))
        frags.append(lay[:, 16:].reshape(n, K // 32, 3, 256)[:, :, 0].reshape(-1))
        off += n * cf
    return torch.cat(bias + frags + [torch.zeros(4096, device=x6.device)]).contiguous()


def pack_softplus512_h3(sd, prefix, k_in, device):
    """pack_softplus512 in split-precision form (rb_cesr_net_h3): the skip layer padded to 544 slots."""
    sdt = {k: _t(sd, k) for k in sd if k.startswith(prefix)}
    k0p = _pad16(k_in)
    n3 = 512 - k_in
    n3p = _pad16(n3)
    assert n3p + k0p == 528 and k0p % 32 == 0
    ls = []
    for l in range(9):
        W = _fold_wn(sdt, prefix + "lin%d." % l)
        b = sdt[prefix + "lin%d.bias" % l].float()
        n_pad, k_pad, perm = _pad16(W.shape[0]), _pad32(W.shape[1]), None
        if l == 4:
            k_pad = 544
            perm = ([k if k < n3 else -1 for k in range(n3p)] + [n3 + j if j < k_in else -1 for j in range(k0p)]
                    + [-1] * 16)
        ls.append(dict(W=W, b=b, n_pad=n_pad, k_pad=k_pad, perm=perm))
    return pack_layers_h3(ls, device)


def repack_vis_x6_fp8(blob, device):
    """EXPERIMENT (csrc/vis_x6.hip built with -DVX_FP8=1): the blob of pack_vis_x6 in the bf8 layout of repack_x6_chunks_fp8."""
    return repack_x6_chunks_fp8(blob, device, ((128, 16), (256, 16), (256, 16), (256, 16), (256, 1)))


def repack_softplus512_x6_fp8(blob, device, k_in):
    """EXPERIMENT (csrc/cesr_x6.hip built with -DQX_FP8=1): the blob of pack_softplus512_x6 with the layers of K = 512 in the bf8 layout of
    repack_x6_chunks_fp8; layer 0 (K = 64 / 192) and the skip layer (K = 576) keep rb_pack_layer_x6's layout."""
    k0p, n3p = _pad16(k_in), _pad16(512 - k_in)
    return repack_x6_chunks_fp8(blob, device, ((k0p, 32, False), (512, 32, True), (512, 32, True), (512, n3p // 16, True), (576, 32, False),
                                               (512, 32, True), (512, 32, True), (512, 32, True), (512, 1, True)))


def repack_x6_chunks_fp8(blob, device, layout):
    """A blob of exact-operand chunks (rb_pack_layer_x6: per chunk [16 bias floats][k-block][h | m | l][lane][8 halves]; `layout` = its (K,
    number of chunks[, convert]) runs) with the l pieces of every weight replaced by bf8 (e5m2) copies of the h and l pieces in the K = 128 order of
    v_mfma_f32_16x16x128_f8f6f4 (the products h.xl and l.xh of the 2^-22 class at twice the f16 rate).  Same size:
    per chunk [16 bias floats] + per group of 128 K: [k-block 0..3][h | m][lane][8 halves] (8 KB), [h8][2 planes][lane][16 bytes] (2 KB),
    [l8] (2 KB).  Byte 4 j + r of a lane's 32 is the K position of half 4 (j % 2) + r of k-block 4 G + j / 2 in the f16 planes: what
    the kernels' v_perm_b32 of the activations' f16 pieces produces.  Anything behind the chunks (padding) is kept."""
    import numpy as np
    src = blob.detach().cpu().numpy().view(np.uint16).copy()
    out = src.copy()
    pos = 0
    for run in layout:
        K, nch = run[0], run[1]
        cu16 = (16 + 24 * K) * 2                    # uint16 elements of a chunk: 16 bias floats + K/32 k-blocks x 3 planes x 64 lanes x 8 halves
        if len(run) > 2 and not run[2]:             # a run that keeps rb_pack_layer_x6's own layout (layers whose K is not a multiple of 128)
            pos += nch * cu16
            continue
        for _ in range(nch):
            body = src[pos + 32:pos + cu16].reshape(K // 32, 3, 64, 8)
            dst = out[pos + 32:pos + cu16].reshape(K // 128, 6144)      # 12 KB per group
            for G in range(K // 128):
                f16part = body[4 * G:4 * G + 4, 0:2]                                      # [kk][h|m][lane][8]
                dst[G, :4096] = f16part.reshape(-1)
                for which, plane in ((0, 0), (2, 1)):                                     # h -> h8, l -> l8
                    p = body[4 * G:4 * G + 4, which].astype(np.uint32)                    # [kk][lane][8] f16 patterns
                    b8 = ((p + 0x7F + ((p >> 8) & 1)) >> 8).astype(np.uint8)              # e5m2, round to nearest even
                    finite = (p & 0x7C00) != 0x7C00                                        # ... a finite half never rounds up to infinity: saturate
                    b8 = np.where(finite & ((b8 & 0x7F) >= 0x7C), (b8 & 0x80) | 0x7B, b8).astype(np.uint8)
                    # byte (j8, r) of a lane <- k-block j8 // 2, half 4 (j8 % 2) + r
                    lanes = b8.transpose(1, 0, 2).reshape(64, 4, 2, 4).reshape(64, 8, 4)  # [lane][j8 = 2 kk + half-block][r]
                    planes = lanes.reshape(64, 2, 16).transpose(1, 0, 2)                  # [plane = j8 // 4][lane][16 bytes]
                    dst[G, 4096 + plane * 1024:4096 + (plane + 1) * 1024] = np.ascontiguousarray(planes).reshape(-1).view(np.uint16)
            pos += cu16
    assert pos <= src.size
    return torch.from_numpy(out.view(np.float32)).to(device)
