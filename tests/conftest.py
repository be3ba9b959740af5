import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _legacy_missing(excinfo):
    """A test reached a retired kernel generation (split precision, first-generation kernels: robir_amd/librobir_hip_legacy.so) on a
    tree whose legacy library was not built -- `make -C robir_amd/csrc legacy` / ROBIR_BUILD_LEGACY=1 (round 6: no longer on
    __graft_entry__.build()'s default path).  Such a test is SKIPPED, not failed: the default policy never needs that library
    (tests/test_default_library_gpu.py proves it with the loader made to fail)."""
    if excinfo is None:
        return False
    e = excinfo[1]
    return type(e).__name__ == "RobirHipError" and "librobir_hip_legacy.so not found" in str(e)


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_setup(item):
    outcome = yield
    if _legacy_missing(outcome.excinfo):
        outcome.force_exception(pytest.skip.Exception("needs robir_amd/librobir_hip_legacy.so (ROBIR_BUILD_LEGACY=1)"))


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_call(item):
    outcome = yield
    if _legacy_missing(outcome.excinfo):
        outcome.force_exception(pytest.skip.Exception("needs robir_amd/librobir_hip_legacy.so (ROBIR_BUILD_LEGACY=1)"))


@pytest.fixture(autouse=True)
def _inference_mode_like_the_plot_path():
    """The kernels are forward-only and guard against training-mode calls (robir_amd.nets.forward_only_guard); tests render
    the way `--plot_only` / relight do: grad disabled.  The guard's own test re-enables grad explicitly."""
    with torch.no_grad():
        yield


def record_metric(name, **values):
    """Append measured distances to gpurun_out/test_metrics.jsonl (travels back from the GPU box): the evidence behind the
    tolerance bounds asserted in the GPU tests.  Best effort -- never fails a test."""
    import json
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "test_metrics.jsonl"), "a") as f:
            f.write(json.dumps({"name": name, **{k: (float(v) if isinstance(v, (int, float, np.floating)) else v)
                                                  for k, v in values.items()}}) + "\n")
    except OSError:
        pass


def err_entries(a, b):
    """Per-entry |a-b| / (|b| + mean|b|) as a flat float64 tensor (the repo's tolerance convention), NaN==NaN -> 0."""
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    same = (a == b) | (torch.isnan(a) & torch.isnan(b))
    fin = torch.isfinite(b)
    scale = b[fin].abs().mean() if fin.any() else torch.tensor(1.0, dtype=torch.float64)
    e = (a - b).abs() / (b.abs() + scale + 1e-30)
    return torch.nan_to_num(torch.where(same, torch.zeros_like(e), e), nan=float("inf")).reshape(-1)


def rel_err(a, b):
    """max |a-b| / (|b| + mean|b|) over the tensor; NaN==NaN and inf==inf count as equal.
    The mean-|b| floor is the tolerance convention of this repo ('Parity tolerances')."""
    a = torch.as_tensor(np.asarray(a) if not isinstance(a, torch.Tensor) else a).double().cpu()
    b = torch.as_tensor(np.asarray(b) if not isinstance(b, torch.Tensor) else b).double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.numel() == 0:
        return 0.0
    same = (a == b) | (torch.isnan(a) & torch.isnan(b))
    fin = torch.isfinite(b)
    scale = b[fin].abs().mean() if fin.any() else torch.tensor(1.0, dtype=torch.float64)
    e = (a - b).abs() / (b.abs() + scale + 1e-30)
    e = torch.where(same, torch.zeros_like(e), e)
    return float(torch.nan_to_num(e, nan=float("inf")).max())


def rel_err_plain(a, b):
    """max |a-b| / |b| over the entries with |b| > mean|b| -- the plain relative error where it is well defined, reported next to the
    floored figure of rel_err (0.0 if there is no such entry)."""
    a = torch.as_tensor(np.asarray(a) if not isinstance(a, torch.Tensor) else a).double().cpu()
    b = torch.as_tensor(np.asarray(b) if not isinstance(b, torch.Tensor) else b).double().cpu()
    fin = torch.isfinite(b) & torch.isfinite(a)
    if not fin.any():
        return 0.0
    big = fin & (b.abs() > b[fin].abs().mean())
    return float(((a - b).abs()[big] / b.abs()[big]).max()) if big.any() else 0.0


def bad_frac(a, b, tol):
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    if a.numel() == 0:
        return 0.0
    same = (a == b) | (torch.isnan(a) & torch.isnan(b))
    fin = torch.isfinite(b)
    scale = b[fin].abs().mean() if fin.any() else torch.tensor(1.0, dtype=torch.float64)
    e = (a - b).abs() / (b.abs() + scale + 1e-30)
    e = torch.where(same, torch.zeros_like(e), e)
    return float((torch.nan_to_num(e, nan=float("inf")) > tol).double().mean())


_CAPS = None


def bounded(name, a, b, tol, frac):
    """The repo's two-part bound for chained / end-to-end comparisons: at most `frac` of the entries beyond `tol`, and NO
    entry beyond the cap recorded for `name` in tests/golden/measured_caps.json (= twice the maximum measured on an MI355X,
    written by tools/update_caps.py from gpurun_out/test_metrics.jsonl; DESIGN.md 'Parity': the outliers are rays on a hit-mask
    or cull threshold -- bounded, never arbitrary).  A name without a recorded cap fails: every comparison has a max bound."""
    global _CAPS
    import json
    if _CAPS is None:
        path = os.path.join(GOLD, "measured_caps.json")
        _CAPS = json.load(open(path)) if os.path.exists(path) else {}
    f, m = bad_frac(a, b, tol), rel_err(a, b)
    cap = _CAPS.get(name)
    record_metric("bounded/" + name, tol=tol, frac=f, max=m, max_plain_rel_above_mean=rel_err_plain(a, b), frac_limit=frac,
                  cap=cap if cap is not None else -1.0)
    assert f <= frac, (name, "fraction beyond", tol, "is", f, "limit", frac)
    if os.environ.get("ROBIR_RECORD_CAPS") == "1":
        return
    assert cap is not None, f"no cap recorded for {name}: run the GPU tests with ROBIR_RECORD_CAPS=1, then tools/update_caps.py"
    assert m <= cap, (name, "max", m, "cap", cap)


def load_golden(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def synth_weights():
    from robir_amd import synth
    sd = synth.synth_state_dict(0, variance=0.3)
    return sd


@pytest.fixture(scope="session")
def oracle_sd(synth_weights):
    from robir_oracle import nets
    return nets.as_torch(synth_weights)


@pytest.fixture(scope="session")
def oracle_octree(oracle_sd):
    """Octree built by the oracle from the synthetic SDF (about 5-10 s on 8 cores)."""
    from robir_oracle import nets, octree
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    return octree.build(lambda x: nets.implicit_forward(oracle_sd, x)[:, 0],
                        lambda x: nets.implicit_gradient(oracle_sd, x), [-1.0] * 3, [1.0] * 3)


def oracle_tables_from_device(Td):
    """The oracle's OctreeTables from a device-built octree (same cells: lets oracle and kernels trace identical geometry
    for weight sets that have no CPU-built fixture)."""
    import torch
    from robir_oracle import octree as ooct
    node = Td.node.cpu()
    T = ooct.OctreeTables()
    T.root_min, T.root_size = torch.from_numpy(Td.root_min.copy()), torch.from_numpy(Td.root_size.copy())
    T.box_min, T.box_size, T.sdf_val = node[:, 0:3].contiguous(), node[:, 4:7].contiguous(), node[:, 7].contiguous()
    fc = node[:, 3].contiguous().view(torch.int32).long()
    T.is_split = fc >= 0
    T.child = torch.where(T.is_split[:, None], fc[:, None] + torch.arange(8)[None, :], torch.full((1, 8), -1))
    res = [int(v) for v in Td.res]
    T.base_index = torch.arange(res[0] * res[1] * res[2]).reshape(*res)
    T.sdf_nrm, T.centre = Td.nrm.cpu(), T.box_min + T.box_size * 0.5
    T.hit, T.min_step = T.sdf_val <= 1e-4, Td.min_step
    return T


def cull_marked_points(lgt_sgs, u_theta, u_phi, n_oracle, n_kernel, thr=1.0, ulps=4):
    """Which surface points have a sampled light direction ON the reference's `n.d > 1e-6` cull (model/sg_render.py:155)?

    lgt_sgs [L,7] the light (first row's light is used for every point, sg_render.py:388-390), u_theta / u_phi [L,nsamp] the recorded
    draws, n_oracle / n_kernel [n,3] the shading normal of the two evaluations.  The sampled directions depend on the light and the draws
    only, so they are rebuilt here with the oracle's own functions.  A (point, direction) pair is MARKED if the two normals put it on
    different sides of the threshold, or if the oracle's cosine lies within the rounding of an fp32 dot product of the threshold:
    |n.d - 1e-6| <= ulps * 2^-24 * sum_i |n_i d_i|  (the products are O(1) and cancel to 1e-6: the sum's absolute rounding error is set
    by the terms, not by the result).  Returns (marked_points bool [n], marked_pairs int): a marked point may differ between two
    fp32-accurate evaluations by one sample of a lobe's 32 moving in or out of the average -- O(1e-4 .. 1e-3) of a lobe, by construction
    not an arithmetic error; every UNMARKED point took the same cull decisions in both."""
    from robir_oracle import sg as osg
    lobe = lgt_sgs[:, :3] / (lgt_sgs[:, :3].norm(dim=-1, keepdim=True) + osg.TINY)
    lam = lgt_sgs[:, 3:4].abs()
    axis = osg.unit_eps(lobe.unsqueeze(-2))
    sharp = lam.unsqueeze(-2)[:, :, 0].clamp(min=1e-4)
    rng = sharp.min().clamp(max=thr)
    dirs = osg._cone_dirs(axis, u_theta, u_phi, torch.arccos((-0.95 * rng) / sharp + 1)).reshape(-1, 3).double()     # [L*nsamp,3]
    no, nk = n_oracle.double(), n_kernel.double()
    marked = torch.zeros(no.shape[0], dtype=torch.bool)
    pairs = 0
    for a in range(0, no.shape[0], 256):                       # [256, L*nsamp] blocks
        o, k = no[a:a + 256] @ dirs.T, nk[a:a + 256] @ dirs.T
        band = ulps * 2.0 ** -24 * (no[a:a + 256].abs() @ dirs.abs().T)
        m = ((o > osg.TINY) != (k > osg.TINY)) | ((o - osg.TINY).abs() <= band)
        marked[a:a + 256] = m.any(-1)
        pairs += int(m.sum())
    return marked, pairs
