"""Placeholders of deferred chunk forwards (robir_amd/deferred.py): recorded torch functions, what forces the numbers, the
output dict.  The queue is a stand-in here (a real pass needs the GPU: tests/test_deferred_gpu.py)."""
import numpy as np
import pytest
import torch

from robir_amd import deferred
from robir_amd.deferred import DeferredTensor, ChunkOutputs, materialize, is_deferred


class FakeQueue:
    def __init__(self, n):
        self.spec = deferred.output_spec("Material", 24, True)
        self.runs = 0
        self.n = n

    def get(self, slot, name):
        self.runs += 1
        s = self.spec[name]
        if s is None:
            return torch.zeros(())
        g = torch.Generator().manual_seed(slot * 1000 + sorted(self.spec).index(name))
        v = torch.rand((self.n,) + s[0], generator=g)
        return v > 0.5 if s[1] == torch.bool else v


def _leaf(value, counter):
    def thunk():
        counter.append(1)
        return value
    return DeferredTensor(torch.empty(value.shape, dtype=value.dtype, device="meta"), value.device, thunk)


def test_metadata_needs_no_numbers():
    calls = []
    t = _leaf(torch.arange(12.0).reshape(4, 3), calls)
    assert isinstance(t, torch.Tensor) and is_deferred(t)
    assert t.shape == (4, 3) and len(t) == 4 and t.dim() == 2 and t.ndim == 2 and t.dtype == torch.float32
    assert t.device.type == "cpu" and t.numel() == 12 and t.size(1) == 3 and not t.requires_grad
    assert t.detach() is t
    assert calls == []


def test_recorded_functions_run_once_when_numbers_are_needed():
    calls = []
    v = torch.arange(12.0).reshape(4, 3)
    a, b = _leaf(v, calls), _leaf(2 * v, calls)
    s = a + b
    r = a[..., 0:1].detach().expand(b.shape)
    c = torch.cat([x.reshape(1, -1, x.shape[-1]) for x in (s, r)], 1).reshape(8, -1)
    f = (s * 2.0).float().sum(-1, keepdim=True)
    for x, shape in ((s, (4, 3)), (r, (4, 3)), (c, (8, 3)), (f, (4, 1))):
        assert is_deferred(x) and x.shape == shape
    assert calls == []
    assert torch.equal(c.cpu(), torch.cat([3 * v, v[:, 0:1].expand(4, 3)], 0))
    assert len(calls) == 2                       # each leaf once
    assert torch.equal(materialize(f), (6 * v).sum(-1, keepdim=True))
    assert np.array_equal(s.numpy(), (3 * v).numpy()) and float(f[0, 0]) == 18.0
    assert len(calls) == 2
    mixed = torch.ones(4, 3) + a                 # ordinary tensor first: still recorded
    assert is_deferred(mixed) and torch.equal(materialize(mixed), v + 1)
    # arithmetic: same-shape float32 / Python numbers take the shortcut, broadcasting and type promotion the meta tensors
    for r, want in ((a * 2.0, v * 2), (1.0 - a, 1 - v), (a / torch.full((4, 3), 2.0), v / 2), (a + torch.ones(3), v + 1),
                    (a + torch.ones(4, 3, dtype=torch.float64), v.double() + 1), (torch.maximum(a, b), 2 * v),
                    (a * torch.tensor(3.0), v * 3), (a > 4.0, v > 4)):
        assert is_deferred(r) and r.shape == want.shape and r.dtype == want.dtype
        got = materialize(r)
        assert got.dtype == want.dtype and torch.equal(got, want)


def test_what_forces_the_numbers():
    for force in (lambda t: t.cpu(), lambda t: t.numpy(), lambda t: t.tolist(), lambda t: bool((t > -1).all()),
                  lambda t: repr(t), lambda t: t.nonzero(), lambda t: t.add_(0.0), lambda t: t.data_ptr(),
                  lambda t: torch.zeros(4, 3).copy_(t), lambda t: torch.equal(t, t)):
        calls = []
        t = _leaf(torch.arange(12.0).reshape(4, 3), calls)
        force(t)
        assert calls == [1], force
    calls = []
    t = _leaf(torch.arange(12.0).reshape(4, 3), calls)
    mx, arg = torch.max(t, dim=-1)               # several outputs share one evaluation
    assert is_deferred(mx) and is_deferred(arg) and calls == []
    assert arg.tolist() == [2, 2, 2, 2] and mx.tolist() == [2.0, 5.0, 8.0, 11.0] and calls == [1]
    dst = torch.zeros(4, 3)
    dst[:] = _leaf(torch.ones(4, 3), calls)      # __setitem__ on an ordinary tensor
    assert float(dst.sum()) == 12.0


def test_chunk_outputs_dict():
    q = FakeQueue(5)
    given = {"object_mask": torch.ones(5, dtype=torch.bool), "hdr_shift": torch.full((5, 1), 0.5)}
    out = ChunkOutputs(q, 3, 5, torch.device("cpu"), given)
    assert "sg_rgb" in out and "nope" not in out and out.get("nope") is None and out.get("hdr_shift") is given["hdr_shift"]
    with pytest.raises(KeyError):
        out["nope"]
    a = out["sg_rgb"]
    assert out["sg_rgb"] is a and a.shape == (5, 3) and out["network_object_mask"].dtype == torch.bool
    assert out["metallic"].shape == (5, 1) and out["gradient_error"].shape == ()
    assert q.runs == 0
    assert set(out) == set(q.spec) | set(given) and len(out) == len(q.spec) + 2
    assert q.runs == 0
    res = {"pred": (out["sg_rgb"] + out["indir_rgb"]).detach(), "mask": out["network_object_mask"].detach()}
    assert q.runs == 0
    assert res["pred"].cpu().shape == (5, 3) and q.runs == 2
    assert dict(out.items())["acc"].shape == (5, 1)
    illum = deferred.output_spec("Illum", 24, True)
    assert illum["indirect_sgs"] == ((24, 7), torch.float32) and "sg_rgb" not in illum
