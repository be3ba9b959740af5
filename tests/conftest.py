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
