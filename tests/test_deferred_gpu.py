"""Deferred chunk forwards on the GPU (robir_amd/deferred.py): the reference runners' per-chunk evaluation loop, unchanged, at
the batched rate -- utils.general.split_input(n_pixels=1024) -> model(s, trainstage='Material') per chunk -> the loop's own
detach / tone-map / add -> utils.general.merge_output (training/train_pbr.py:248-281, utils/general.py:27-38,55-69) -- and
bit-equal to IDRNetwork.render_chunks on the same chunks."""
import time
import types

import pytest
import torch

from conftest import record_metric
from test_runner_hooks_gpu import make_pbr_runner_hook, overlay_model_pkg  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def model(dev):
    from robir_amd import renderer
    return renderer.build_synthetic_model(dev, seed=0, variance=0.3)


@pytest.fixture()
def deferring(model):
    model.deferred_chunks = 1024
    yield model
    model.flush()
    model.__dict__.pop("deferred_chunks", None)
    model.__dict__.pop("get_sg_render", None)


def _same(a, b):
    """Bit-for-bit (rays parallel to an axis carry NaN points in both)."""
    a, b = a.cpu().contiguous().reshape(-1), b.cpu().contiguous().reshape(-1)
    return a.shape == b.shape and a.dtype == b.dtype and torch.equal(a.view(torch.uint8), b.view(torch.uint8))


def _view(dev, h, w):
    from robir_amd import synth
    uv, pose, K = synth.synth_camera(h, w)
    total = h * w
    mi = {"uv": torch.from_numpy(uv).to(dev)[None], "pose": torch.from_numpy(pose).to(dev)[None],
          "intrinsics": torch.from_numpy(K).to(dev)[None], "object_mask": torch.ones(1, total, dtype=torch.bool, device=dev)}
    return mi, total


def split_input(model_input, total_pixels, n_pixels=1024):
    """A view cut into per-chunk input dicts the way the runners' helper does (utils/general.py:27-38): the per-pixel entries
    (uv, object_mask) are gathered per chunk with index_select -- the op the deferred placeholders have to cope with --, the
    camera entries are shared."""
    dev = model_input["uv"].device
    per_pixel = ("uv", "object_mask")
    chunks = []
    for first in range(0, total_pixels, n_pixels):
        rows = torch.arange(first, min(first + n_pixels, total_pixels), device=dev)
        chunks.append({k: (torch.index_select(v, 1, rows) if k in per_pixel else v) for k, v in model_input.items()})
    return chunks


def merge_output(res, total_pixels, batch_size=1):
    """Per-chunk result dicts -> whole-view tensors, with the tensor ops of utils/general.py:55-69 (reshape to [batch, pixels, width],
    cat along the pixel axis, flatten): 1-D entries come out [batch * pixels], the others [batch * pixels, width]."""
    def whole(key):
        flat = res[0][key].dim() == 1
        parts = [r[key].reshape(batch_size, -1, 1 if flat else r[key].shape[-1]) for r in res]
        joined = torch.cat(parts, 1)
        return joined.reshape(batch_size * total_pixels) if flat else joined.reshape(batch_size * total_pixels, -1)
    return {key: whole(key) for key in res[0]}


def plot_loop(model, split, total):
    """The body of PBRTrainRunner.plot_to_disk's loop, train_pbr.py:259-281."""
    res = []
    for s in split:
        s["hdr_shift"] = model.gamma.hdr_shift.as_input().expand(s["uv"].shape[1], 1)
        out = model(s, trainstage="Material", lin_diff=False, fun_spec=False, train_spec=True)
        indir_rgb = out["indir_rgb"]
        roughness = out["roughness"][..., 0:1]
        diffuse_albedo = out["diffuse_albedo"]
        sg_rgb = out["sg_rgb"]
        pred_rgb = sg_rgb + indir_rgb
        sg_rgb = model.gamma.hdr_shift.hdr2ldr(sg_rgb)
        indir_rgb = model.gamma.hdr_shift.hdr2ldr(indir_rgb)
        pred_rgb = model.gamma.hdr_shift.hdr2ldr(pred_rgb)
        res.append({"roughness": roughness.detach().expand(diffuse_albedo.shape), "diffuse_albedo": diffuse_albedo.detach(),
                    "indir_rgb": indir_rgb.detach(), "sg_rgb": sg_rgb.detach(), "pred_rgb": pred_rgb.detach(),
                    "vis_shadow": out["vis_shadow"].detach(), "mask": out["network_object_mask"].detach()})
    return merge_output(res, total)


def _batched(model, mi, total, seed, per_pass):
    """render_chunks on the consecutive blocks of chunks a recorded loop runs as passes, one seed in front."""
    tm = model.gamma.hdr_shift
    torch.manual_seed(seed)
    hdr = tm.as_input().expand(total, 1).contiguous()
    # the passes of an uninterrupted loop: 16, 32, 64, ... per_pass chunks (robir_amd.deferred.pass_sizes)
    from robir_amd import deferred
    sizes = deferred.pass_sizes((total + 1023) // 1024, per_pass)
    starts = [sum(sizes[:i]) * 1024 for i in range(len(sizes))]
    parts = [model.render_chunks(mi["uv"][0, a:a + n * 1024], mi["pose"][0], mi["intrinsics"][0],
                                 hdr[a:a + n * 1024], chunk=1024, trainstage="Material")
             for a, n in zip(starts, sizes)]
    o = {k: torch.cat([p[k] for p in parts]) for k in ("roughness", "diffuse_albedo", "indir_rgb", "sg_rgb", "vis_shadow",
                                                       "network_object_mask")}
    return {"roughness": o["roughness"][..., 0:1].expand(total, 3), "diffuse_albedo": o["diffuse_albedo"],
            "indir_rgb": tm.hdr2ldr(o["indir_rgb"]), "sg_rgb": tm.hdr2ldr(o["sg_rgb"]),
            "pred_rgb": tm.hdr2ldr(o["sg_rgb"] + o["indir_rgb"]), "vis_shadow": o["vis_shadow"], "mask": o["network_object_mask"]}


@pytest.mark.parametrize("hook,per_pass", [("native", 1024), ("runner", 1024), ("runner", 128)])
def test_runner_loop_800x800_deferred_is_render_chunks(dev, deferring, overlay_model_pkg, hook, per_pass):
    """per_pass = model.deferred_chunks: 1024 holds the whole view (one pass when the image is read); 128 runs a pass every
    128 chunks, which the GPU works through while the loop records the next ones."""
    from robir_amd import deferred
    model = deferring
    model.deferred_chunks = per_pass
    mi, total = _view(dev, 800, 800)
    if hook == "runner":                         # the hook the unchanged runner installs (train_pbr.py:413)
        model.get_sg_render = make_pbr_runner_hook(types.SimpleNamespace(model=model, train_spec=True, no_normal=False,
                                                                         is_training=False))
    want = _batched(model, mi, total, 5, per_pass)
    best = 1e9
    for rep in range(3):
        split = split_input(mi, total)
        torch.cuda.synchronize()
        torch.manual_seed(5)
        t0 = time.time()
        merged = plot_loop(model, split, total)
        assert all(deferred.is_deferred(v) for v in merged.values())      # the merged image is still a placeholder
        got = {k: v.cpu() for k, v in merged.items()}                     # the plotting code reads the numbers
        torch.cuda.synchronize()
        best = min(best, time.time() - t0)
    for k in want:
        assert torch.equal(got[k], want[k].cpu()), k
    rate = total / best
    record_metric(f"deferred/runner_loop_800x800/{hook}_hook/{per_pass}_per_pass", rays_per_s=rate, seconds=best)
    # floors at ~70 % of the measured rates: split precision 7.1e5 / 5.4e5, the default exact-operand arithmetic 3.4e5 / 3.0e5
    from robir_amd import precision
    floor = {"split": (5.5e5, 5.0e5), "exact": (2.6e5, 2.3e5), "f16": (5.5e5, 5.0e5)}[precision.policy()]
    assert rate >= (floor[0] if per_pass == 128 else floor[1]), rate
    assert model.__dict__.get("_pending") is None


def test_deferred_single_chunk_equals_immediate_forward(dev, deferring):
    model = deferring
    mi, total = _view(dev, 64, 64)
    s = split_input(mi, total)[1]
    s["hdr_shift"] = torch.full((1024, 1), 0.5, device=dev)
    torch.manual_seed(3)
    lazy = model(s, trainstage="Material")
    model.flush()
    model.deferred_chunks = 0
    torch.manual_seed(3)
    now = model(s, trainstage="Material")
    assert set(lazy) == set(now)
    for k in now:
        assert _same(lazy[k], now[k]), k


def test_short_last_chunk_stage_change_and_limit(dev, deferring):
    from robir_amd import deferred
    model = deferring
    mi, total = _view(dev, 100, 100)             # 9 full chunks + 784 rays
    split = split_input(mi, total)
    assert split[-1]["uv"].shape[1] == 784
    hdr = lambda s: torch.full((s["uv"].shape[1], 1), 0.5, device=dev)
    torch.manual_seed(9)
    outs = []
    for s in split:
        s["hdr_shift"] = hdr(s)
        outs.append(model(s, trainstage="Material"))
    assert model._pending.closed and model._pending.result is None   # the short chunk closed the pass; it runs at the first read (or the next chunk)
    albedo = torch.cat([o["diffuse_albedo"] for o in outs]).cpu()
    torch.manual_seed(9)
    ref = model.render_chunks(mi["uv"][0], mi["pose"][0], mi["intrinsics"][0], torch.full((total, 1), 0.5, device=dev))
    assert torch.equal(albedo, ref["diffuse_albedo"].cpu())
    assert torch.equal(torch.cat([o["sg_rgb"] for o in outs]).cpu(), ref["sg_rgb"].cpu())
    # a different stage starts its own pass and runs the pending one first; so does reaching the limit
    model.deferred_chunks = 4
    a = model(split[0], trainstage="Material")
    q0 = model._pending
    b = model(split[1], trainstage="Illum")
    assert q0.result is not None and model._pending is not q0 and "indirect_sgs" in b and "sg_rgb" not in b
    assert b["indirect_sgs"].shape == (1024, model.indirect_illum_network.num_lgt_sgs, 7)
    tr = model.trace_radiance(b, nsamp=4)                 # reads the placeholders through this library's kernels
    assert tr["trace_radiance"].shape == (1024, 4, 3) and bool(torch.isfinite(tr["gt_integral"]).all())
    model.flush()
    many = [model(s, trainstage="Material") for s in split[:9]]
    assert [o._q for o in many].count(many[0]._q) == 4 and many[8]._q is model._pending and many[4]._q.result is not None
    assert torch.equal(torch.cat([o["diffuse_albedo"] for o in many]).cpu(), ref["diffuse_albedo"][:9 * 1024].cpu())
    assert torch.equal(a["network_object_mask"].cpu(), ref["network_object_mask"][:1024].cpu())
    assert not deferred._LIVE or all(q.result is None for q in deferred._LIVE)


def test_pending_chunks_and_changed_weights(dev, deferring):
    model = deferring
    mi, total = _view(dev, 64, 64)
    split = split_input(mi, total)
    for s in split:
        s["hdr_shift"] = torch.full((1024, 1), 0.5, device=dev)
    out = model(split[0], trainstage="Material")
    p = model.envmap_material_network.lgtSGs
    keep = p.detach().clone()
    p.mul_(1.0)                                  # an in-place update of a parameter while a chunk is pending
    with pytest.raises(RuntimeError, match="parameters changed"):
        out["sg_rgb"].cpu()
    p.copy_(keep)
    out = model(split[0], trainstage="Material")
    model.train()                                # train() / eval() run what is pending
    assert out._q.result is not None
    model.eval()
    out = model(split[1], trainstage="Material")
    q = out._q
    torch.manual_seed(77)                        # re-seeding in between does not change what the recorded pass draws ...
    v1 = out["vis_shadow"].cpu()
    assert q.result is not None
    torch.manual_seed(77)
    w1 = torch.rand(4, device=dev)               # ... and the pass does not consume the new stream
    torch.manual_seed(77)
    assert torch.equal(torch.rand(4, device=dev), w1)
    assert bool(torch.isfinite(v1).all())


def test_reseeding_per_chunk_under_default_settings_gives_immediate_results(dev, model):
    """deferral is ON by default (ROBIR_DEFER_CHUNKS = 128) and draws a pass's random numbers from the generator state at its
    first recorded chunk -- a caller that re-seeds before EVERY chunk forward (tests/test_runner_hooks_gpu.py does, to replay a run) must
    still get, chunk for chunk, what immediate execution gives under those seeds.  forward() notices that the generator left the state the
    pending pass was recorded under, runs the pending chunks from their own state and the new chunk at once.  Also: load_state_dict() while
    chunks are pending runs them first (with the OLD weights) instead of raising at the next read."""
    from robir_amd import deferred
    model.__dict__.pop("deferred_chunks", None)            # the DEFAULT, whatever ROBIR_DEFER_CHUNKS says (0 in an A/B run: nothing to test)
    if not deferred.DEFAULT_CHUNKS:
        pytest.skip("ROBIR_DEFER_CHUNKS=0 in this environment")
    mi, total = _view(dev, 64, 64)
    split = split_input(mi, total)
    for s in split:
        s["hdr_shift"] = torch.full((1024, 1), 0.5, device=dev)
    keys = ("sg_rgb", "indir_rgb", "vis_shadow", "diffuse_albedo", "network_object_mask")
    lazy = []
    for i, s in enumerate(split):
        torch.manual_seed(100 + i)                         # the caller's per-chunk seed
        lazy.append(model(s, trainstage="Material"))
    got = [{k: o[k].cpu() for k in keys} for o in lazy]
    model.flush()
    model.deferred_chunks = 0
    try:
        for i, s in enumerate(split):
            torch.manual_seed(100 + i)
            now = model(s, trainstage="Material")
            for k in keys:
                assert _same(got[i][k], now[k]), (i, k)
        # a loop that does NOT touch the generator is still recorded as one pass (the point of the default) ...
        model.__dict__.pop("deferred_chunks", None)
        torch.manual_seed(5)
        outs = [model(s, trainstage="Material") for s in split]
        assert all(o._q is outs[0]._q for o in outs) and outs[0]._q.result is None
        # ... and replacing the weights runs it first, with the weights it was recorded for
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        model.load_state_dict(sd)
        assert outs[0]._q.result is not None and model.__dict__.get("_pending") is None
        torch.manual_seed(5)
        ref = model.render_chunks(mi["uv"][0], mi["pose"][0], mi["intrinsics"][0], torch.full((total, 1), 0.5, device=dev))
        assert _same(torch.cat([o["sg_rgb"] for o in outs]), ref["sg_rgb"])
    finally:
        model.flush()
        model.__dict__.pop("deferred_chunks", None)


def test_constant_seed_per_chunk_gives_immediate_results(dev, model):
    """a loop that calls torch.manual_seed(0) before EVERY chunk leaves the generator in exactly the state the pending pass was
    recorded under -- comparing states cannot see it.  The seeding entry points bump deferred.seed_epoch(); forward() treats a new epoch
    like a changed state: chunk k draws from the seed, not from the stream advanced by chunks 0..k-1.  torch.cuda.manual_seed_all and
    torch.cuda.set_rng_state count too."""
    from robir_amd import deferred
    model.__dict__.pop("deferred_chunks", None)
    if not deferred.DEFAULT_CHUNKS:
        pytest.skip("ROBIR_DEFER_CHUNKS=0 in this environment")
    mi, total = _view(dev, 64, 64)
    split = split_input(mi, total)
    for s in split:
        s["hdr_shift"] = torch.full((1024, 1), 0.5, device=dev)
    keys = ("sg_rgb", "indir_rgb", "vis_shadow")
    e0 = deferred.seed_epoch()
    torch.manual_seed(0)
    assert deferred.seed_epoch() > e0            # (torch.manual_seed seeds the CUDA generators through torch.cuda.manual_seed_all: two bumps)
    state0 = torch.cuda.get_rng_state(dev)
    seeders = (lambda: torch.manual_seed(0), lambda: torch.cuda.manual_seed_all(0), lambda: torch.cuda.set_rng_state(state0, dev),
               lambda: torch.manual_seed(0))
    try:
        lazy = []
        for s, seed in zip(split, seeders):
            seed()
            lazy.append(model(s, trainstage="Material"))
        got = [{k: o[k].cpu() for k in keys} for o in lazy]
        model.flush()
        model.deferred_chunks = 0
        for i, s in enumerate(split):
            torch.manual_seed(0)
            now = model(s, trainstage="Material")
            for k in keys:
                assert _same(got[i][k], now[k]), (i, k)
        # the draws matter: without the per-chunk seed chunk 1 differs
        torch.manual_seed(0)
        model(split[0], trainstage="Material")
        other = model(split[1], trainstage="Material")
        assert not _same(got[1]["vis_shadow"], other["vis_shadow"])
    finally:
        model.flush()
        model.__dict__.pop("deferred_chunks", None)


@pytest.mark.parametrize("stage", ["Material", "Illum"])
def test_recorded_trace_radiance_per_chunk_equals_the_immediate_calls(dev, deferring, stage):
    """The CESR / visibility runners call `trace_radiance(out, nsamp=8)` right after every chunk forward and only reduce its results
    (training/train_cesr.py:321-326: max / mean of pred_vis).  On the outputs of a RECORDED chunk the call is recorded too and runs behind
    the pass as ONE grouped call: every chunk its own lock-step batch (the > 100 000-ray step size must not kick in), the two
    torch.rand(n_hit * nsamp) of every call taken from the CPU generator chunk by chunk.  Equal to render_chunks + one immediate
    trace_radiance per chunk under the same seed -- directions, hit masks and predicted visibilities bit for bit, the borrowed radiance to
    fp32 summation order; a short last chunk and the chunk that fills the pass included."""
    model = deferring
    model.deferred_chunks = 4
    mi, total = _view(dev, 72, 100)              # 7 full chunks + 32 rays: passes of 4 + 4 (the second ended by the short chunk)
    split = split_input(mi, total)
    hdr_all = torch.full((total, 1), 0.5, device=dev)
    for i, s in enumerate(split):
        s["hdr_shift"] = hdr_all[i * 1024:(i + 1) * 1024]
    torch.manual_seed(21)
    red, outs, trs = [], [], []
    for s in split:
        out = model(s, trainstage=stage)
        tr = model.trace_radiance(out, nsamp=8)
        assert isinstance(tr, dict) and "pred_vis" in tr and len(tr) == 6
        _, pv = torch.max(tr["pred_vis"].detach(), dim=-1)
        red.append(torch.mean(pv.float(), axis=1))       # the runner's reduction: still a placeholder
        outs.append(out)
        trs.append(tr)
    q_first, q_last = outs[0]._q, outs[-1]._q
    assert q_first is not q_last and q_first.result is not None and q_last.result is None and len(q_last.trace) == 4
    got_red = torch.cat(red).cpu()                       # first read: runs the second pass and its four traces
    got = {k: torch.cat([t[k] for t in trs]).cpu() for k in ("trace_radiance", "gt_vis", "pred_vis", "indir_mask", "gt_integral", "sample_dirs")}
    model.flush()
    model.deferred_chunks = 0
    uv, pose, K = mi["uv"][0], mi["pose"][0], mi["intrinsics"][0]
    ref = {k: [] for k in got}
    torch.manual_seed(21)
    for a, b in ((0, 4), (4, 8)):                        # the two passes: render_chunks draws like a recorded pass
        lo, hi = a * 1024, min(b * 1024, total)
        o = model.render_chunks(uv[lo:hi], pose, K, hdr_all[lo:hi], chunk=1024, trainstage=stage)
        for c in range(a, b):
            r0, r1 = c * 1024 - lo, min((c + 1) * 1024, total) - lo
            one = {k: o[k][r0:r1] for k in ("points", "network_object_mask", "normals")}
            one["hdr_shift"] = hdr_all[lo + r0:lo + r1]
            t = model.trace_radiance(one, nsamp=8)       # immediate, ONE lock-step batch, its own two CPU draws
            for k in ref:
                ref[k].append(t[k])
    for k in got:
        r = torch.cat(ref[k]).cpu()
        if k in ("trace_radiance", "gt_integral"):
            # the borrowed colours go through the SDF / colour kernels, whose form (one or two tiles per wave, ops.sdf_two_tile) follows the
            # batch size: a pass's grouped call and a chunk's own call agree to fp32 summation order, not always bit for bit
            assert got[k].shape == r.shape and float((got[k] - r).abs().max()) <= 5e-6 * max(1.0, float(r.abs().max())), k
        else:
            assert _same(got[k], r), k
    _, pv = torch.max(torch.cat(ref["pred_vis"]), dim=-1)
    assert _same(got_red, torch.mean(pv.float(), axis=1))
    assert int(torch.cat(ref["gt_vis"]).sum()) > 0 or stage == "Material"
    # a modified output dict, explicit draws or a test direction are not recorded
    model.deferred_chunks = 4
    out = model(split[0], trainstage=stage)
    out["hdr_shift"] = hdr_all[:1024] * 1.0
    tr = model.trace_radiance(out, nsamp=4)
    assert not hasattr(tr, "_q") and tr["trace_radiance"].shape == (1024, 4, 3)
    model.flush()
