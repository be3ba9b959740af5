"""Deferred chunk forwards: the unchanged runners' loop at the batched rate.

The reference's evaluation loops (training/train_pbr.py:248-281, training/train_cesr.py:310-350, scripts/relight.py:42-58) cut a
view into 1024-pixel chunks (utils/general.py:27-38), call `model(s, trainstage=...)` once per chunk, keep a few detached /
tone-mapped fields of every result and only look at the numbers after utils.merge_output (utils/general.py:55-69).  One
1024-ray call cannot fill 256 CUs (4 ms of dependent launches), the same chunks as ONE pass of
IDRNetwork.render_chunks run at the bench rate.  With `model.deferred_chunks = N` (env ROBIR_DEFER_CHUNKS) an eval-mode
`forward()` therefore only RECORDS its chunk and returns placeholders:

  * ChunkQueue     the recorded chunks of one pass (same stage / flags / camera / hook), copied into staging buffers;
  * ChunkOutputs   the dict `forward()` returns; its values are DeferredTensors created on first access;
  * DeferredTensor a tensor without storage that knows its shape, dtype and device.  torch functions applied to it are recorded
                   too (output shapes come from running the function on meta tensors), so `a + b`, `x[..., 0:1]`, `.expand`,
                   `.reshape`, `torch.cat`, `hdr2ldr` ... stay placeholders; anything that needs the numbers (`.cpu()`,
                   `.item()`, `.numpy()`, `bool()`, printing, an in-place write, a kernel of this library taking its address)
                   first runs the queue it descends from -- as one render_chunks-shaped pass -- and then the recorded functions.

What a recorded pass computes is exactly IDRNetwork._render on the concatenated chunks, i.e. what render_chunks returns for them
(every chunk keeps its own lock-step trace, its own sample tables and its own specular minimum).  Differences from immediate
execution, all stated in INTEGRATION.md: random numbers are drawn when the pass runs, starting from the generator state at the
time of its first recorded chunk -- a caller that re-seeds or draws BETWEEN two chunk forwards is detected at the next forward()
(the generator no longer is in the state the pass was recorded under): the pending chunks then run from their own state, the new
chunk runs at once, i.e. such a loop gets immediate-execution results chunk for chunk (tests/test_deferred_gpu.py) -- also when it sets
the SAME seed every time (the seeding functions of torch bump `seed_epoch()`; seeding through a Generator object's own methods is the one
case that needs deferred_chunks = 0); weights must not
change in place while chunks are pending (checked: RuntimeError); train(), eval(), load_state_dict(), load_light() and flush() run
what is pending.  A pass holds the outputs of up to `deferred_chunks` chunks (128 x 1024 rays x 240 B = 31 MB) plus the pass's scratch
(the direction tables of its chunks, 4.2 MB each: 0.5 GB for 128) -- sized for a 288 GB part, set ROBIR_DEFER_CHUNKS lower elsewhere.
"""
import os
import weakref

import torch

DEFAULT_CHUNKS = int(os.environ.get("ROBIR_DEFER_CHUNKS", "128") or 0)      # round 4: on by default (0 = every forward() runs at once)
# ROBIR_DEFER_RAMP=n: the first pass of a loop is n chunks and the following ones double up to the limit (16, 32, 64, 128, 128, ...), so
# that the GPU has work after n recorded chunks instead of idling while `limit` are recorded.  MEASURED NEUTRAL on the 800 x 800 plot loop
# (one box, tools/prof_deferred.py: 1.522 / 1.539 / 1.516 s for n = 0 / 16 / 64 at 128 chunks per pass; the 80 ms over one
# render_chunks pass are the loop's own per-chunk tone-mapping / merge kernels, not idle time): off by default (0 = every pass `limit` chunks).
RAMP_START = int(os.environ.get("ROBIR_DEFER_RAMP", "0") or 0)


def pass_sizes(n_chunks, limit, start=None):
    """Chunks per pass of an uninterrupted loop over n_chunks full chunks (what IDRNetwork._record_chunk does; tests build their reference
    partition from it)."""
    start = RAMP_START if start is None else start
    out, cur = [], (min(limit, start) if start else limit)
    while n_chunks > 0:
        k = min(cur, n_chunks)
        out.append(k)
        n_chunks -= k
        cur = min(limit, 2 * cur)
    return out


_NO_TF = torch._C.DisableTorchFunctionSubclass
_LIVE = weakref.WeakSet()            # queues with recorded chunks that have not run

# Seeding epoch: a caller that re-seeds with the SAME seed before every chunk leaves the generator in the state the pending
# pass was recorded under, so comparing states cannot see it.  Every seeding / state-setting entry point of torch that is a Python
# function is wrapped to bump a counter; a pass remembers the count at its first chunk and a different count at the next forward() is
# treated like a changed state (the pending chunks run from their own state, the new chunk runs at once: immediate-execution results).
# Not visible from here: methods called on a Generator OBJECT (gen.manual_seed / gen.set_state on torch.cuda.default_generators[i] or
# torch.default_generator with an unchanged resulting state) -- C methods of an extension type -- and a seeding function a caller bound
# by name (`from torch import manual_seed`) BEFORE this module was imported; such loops set deferred_chunks = 0.
_SEED_EPOCH = [0]


def seed_epoch():
    return _SEED_EPOCH[0]


def _wrap_seeding(mod, name):
    fn = getattr(mod, name, None)
    if fn is None or getattr(fn, "_rb_seed_hook", False):
        return
    import functools

    @functools.wraps(fn)
    def hooked(*a, **k):
        _SEED_EPOCH[0] += 1
        return fn(*a, **k)
    hooked._rb_seed_hook = True
    setattr(mod, name, hooked)


for _mod, _names in ((torch, ("manual_seed", "seed", "set_rng_state")), (torch.random, ("manual_seed", "seed", "set_rng_state")),
                     (torch.cuda, ("manual_seed", "manual_seed_all", "seed", "seed_all", "set_rng_state", "set_rng_state_all")),
                     (torch.cuda.random, ("manual_seed", "manual_seed_all", "seed", "seed_all", "set_rng_state", "set_rng_state_all"))):
    for _n in _names:
        _wrap_seeding(_mod, _n)

# need the numbers (or write into them): run what is pending, then call the function on ordinary tensors
_FORCE = frozenset((
    "cpu", "cuda", "to", "numpy", "item", "tolist", "data_ptr", "is_contiguous", "storage", "untyped_storage", "copy_", "set_",
    "record_stream", "backward", "pin_memory", "share_memory_", "__bool__", "__float__", "__int__", "__index__", "__complex__",
    "__repr__", "__str__", "__format__", "__reduce_ex__", "__reduce__", "__deepcopy__", "__array__", "__array_wrap__",
    "__setitem__", "__contains__", "__dlpack__", "__dlpack_device__", "__cuda_array_interface__", "__iter__", "__hash__",
    "equal", "allclose", "is_nonzero", "nonzero", "unique", "masked_select"))
_ELEMENTWISE = frozenset(("add", "sub", "mul", "div", "true_divide", "__add__", "__radd__", "__sub__", "__rsub__", "__mul__",
                          "__rmul__", "__truediv__", "__rtruediv__", "maximum", "minimum"))
_NOT_INPLACE = frozenset(("__index__", "__int__", "__invert__", "__iter__", "__init__", "__init_subclass__"))


def _inplace(name):
    if name.startswith("__"):
        return name.startswith("__i") and name not in _NOT_INPLACE
    return name.endswith("_")


class DeferredTensor(torch.Tensor):
    @staticmethod
    def __new__(cls, meta, device, thunk):
        t = torch.Tensor._make_wrapper_subclass(cls, meta.shape, dtype=meta.dtype, device=device)
        t._rb_meta, t._rb_thunk, t._rb_value, t._rb_dev = meta, thunk, None, device
        return t

    def __init__(self, *a, **k):
        pass

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, "__name__", "")
        if name == "__get__":                      # .shape .dtype .device .ndim .is_cuda .requires_grad ...: the wrapper knows
            prop = getattr(func.__self__, "__name__", "")
            if prop == "shape":
                return args[0]._rb_meta.shape
            if prop == "device":
                return args[0]._rb_dev
            if prop == "dtype":
                return args[0]._rb_meta.dtype
            with _NO_TF():
                return func(*args, **kwargs)
        if name == "detach" and len(args) == 1:    # results carry no graph
            return args[0]
        if name == "__hash__":
            return id(args[0])
        if name in _FORCE or _inplace(name) or kwargs.get("out") is not None:
            return func(*_plain(args), **_plain(kwargs))
        if name in _ELEMENTWISE and len(args) == 2 and not kwargs:
            # same-shape float32 operands (or a Python number): the result looks like the operand; meta tensors would take the
            # Python decomposition of the operator, 140 us
            a, b = args
            ma = a._rb_meta if type(a) is DeferredTensor else a
            mb = b._rb_meta if type(b) is DeferredTensor else b
            ta, tb = isinstance(ma, torch.Tensor), isinstance(mb, torch.Tensor)
            if ((ta and ma.dtype == torch.float32) or isinstance(ma, (int, float))) and \
                    ((tb and mb.dtype == torch.float32) or isinstance(mb, (int, float))) and \
                    (not (ta and tb) or ma.shape == mb.shape):
                lead = a if type(a) is DeferredTensor else b
                return DeferredTensor(lead._rb_meta, lead._rb_dev, lambda: func(*_plain(args)))
        try:
            with _NO_TF():
                mout = func(*_meta(args), **_meta(kwargs))
        except Exception:                          # data-dependent output, or no meta kernel: compute now
            return func(*_plain(args), **_plain(kwargs))
        dev = _device_of(args) or _device_of(tuple(kwargs.values()))
        if isinstance(mout, torch.Tensor):
            return DeferredTensor(mout, dev, lambda: func(*_plain(args), **_plain(kwargs)))
        if isinstance(mout, (tuple, list)) and mout and all(isinstance(m, torch.Tensor) for m in mout):
            cell = []

            def whole():
                if not cell:
                    cell.append(func(*_plain(args), **_plain(kwargs)))
                return cell[0]
            outs = [DeferredTensor(m, dev, (lambda i: (lambda: whole()[i]))(i)) for i, m in enumerate(mout)]
            try:
                return type(mout)(outs)
            except TypeError:
                return tuple(outs)
        return mout                                # shape arithmetic (sizes, bools, ints): no numbers involved

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        # reached only by calls that bypass __torch_function__ (C++ callers): compute on ordinary tensors
        return func(*_plain(args), **_plain(kwargs or {}))


def is_deferred(x):
    return type(x) is DeferredTensor


def materialize(t):
    """The ordinary tensor behind a placeholder (runs the recorded pass / functions on first use)."""
    v = t._rb_value
    if v is None:
        v = t._rb_thunk()
        if is_deferred(v):
            v = materialize(v)
        t._rb_value, t._rb_thunk = v, None
    return v


def plain(x):
    return materialize(x) if type(x) is DeferredTensor else x


def _plain(x):
    if type(x) is DeferredTensor:
        return materialize(x)
    if isinstance(x, (list, tuple)):
        return type(x)(_plain(v) for v in x)
    if isinstance(x, dict):
        return {k: _plain(v) for k, v in x.items()}
    return x


def _meta(x):
    if type(x) is DeferredTensor:
        return x._rb_meta
    if isinstance(x, torch.Tensor):
        return torch.empty_strided(x.shape, x.stride(), dtype=x.dtype, device="meta")
    if isinstance(x, (list, tuple)):
        return type(x)(_meta(v) for v in x)
    if isinstance(x, dict):
        return {k: _meta(v) for k, v in x.items()}
    return x


def _device_of(xs):
    for x in xs:
        if type(x) is DeferredTensor:
            return x._rb_dev
        if isinstance(x, (list, tuple)):
            d = _device_of(x)
            if d is not None:
                return d
    return None


def lazy_like(x, fn):
    """Placeholder with x's shape (float32) for fn(ordinary tensor of x) -- for element-wise kernels of this library."""
    meta = x._rb_meta if x._rb_meta.dtype == torch.float32 else torch.empty(x._rb_meta.shape, device="meta")
    return DeferredTensor(meta, x._rb_dev, lambda: fn(materialize(x)))


def flush_all():
    """Run every recorded pass (called before anything that changes what a pass would compute)."""
    for q in list(_LIVE):
        q.flush()


# per-ray trailing shapes of what IDRNetwork._shade returns (renderer.py); checked against the real result when a pass has run
_F3 = ("points", "ray_dirs", "bg_rgb", "sg_rgb", "indir_rgb", "sg_diffuse_rgb", "sg_specular_rgb", "indir_diffuse_rgb",
       "indir_specular_rgb", "normals", "diffuse_albedo", "roughness", "normal_map", "vis_shadow", "random_xi_roughness",
       "random_xi_diffuse_albedo")
_F1 = ("sdf_output", "metallic", "random_xi_metallic", "acc", "final_t")
_B0 = ("network_object_mask", "surface_mask")


def output_spec(trainstage, n_indirect_sgs, has_hdr):
    """name -> (trailing shape, dtype) of the per-ray outputs; None for the pass-wide scalar."""
    if trainstage == "Illum":
        spec = {"points": ((3,), torch.float32), "sdf_output": ((1,), torch.float32), "ray_dirs": ((3,), torch.float32),
                "network_object_mask": ((), torch.bool), "indirect_sgs": ((n_indirect_sgs, 7), torch.float32),
                "indir_integral": ((3,), torch.float32), "normals": ((3,), torch.float32)}
    else:
        spec = {k: ((3,), torch.float32) for k in _F3}
        spec.update({k: ((1,), torch.float32) for k in _F1})
        spec.update({k: ((), torch.bool) for k in _B0})
        spec["gradient_error"] = None
    return spec


class ChunkQueue:
    def __init__(self, model, sig, spec, chunk, limit, pose, K, has_hdr, device, render_args):
        self.model = weakref.ref(model)
        self.sig, self.spec, self.chunk, self.limit = sig, spec, chunk, limit
        self.pose, self.K = pose.clone(), K.clone()
        self.render_args = render_args
        cap = chunk * limit
        self.uv = torch.empty(cap, 2, device=device)
        self.mask = torch.empty(cap, dtype=torch.bool, device=device)
        self.hdr = torch.empty(cap, 1, device=device) if has_hdr else None
        self.rays, self.slots, self.closed = 0, [], False
        self.result = None
        self.running = False
        self.versions = _versions(model)
        gen = torch.cuda.default_generators[device.index or 0] if device.type == "cuda" else torch.default_generator
        self.gen, self.gen_state = gen, gen.get_state()
        self.seed_epoch = seed_epoch()
        # recorded trace_radiance calls on this pass's chunks (slot -> nsamp; IDRNetwork.trace_radiance): they run behind the pass as ONE
        # grouped call, every chunk its own lock-step batch, the CPU generator's draws taken chunk by chunk in slot order
        self.trace, self.trace_nsamp, self.cpu_state, self.trace_result = {}, None, None, None
        _LIVE.add(self)

    def add(self, uv, mask, hdr):
        n = uv.shape[0]
        a = self.rays
        self.uv[a:a + n].copy_(uv)
        self.mask[a:a + n].copy_(mask)
        if self.hdr is not None:
            self.hdr[a:a + n].copy_(hdr)
        self.slots.append((a, n))
        self.rays = a + n
        if n < self.chunk or len(self.slots) == self.limit:
            self.closed = True                     # a short chunk can only be the last one of a pass
        return len(self.slots) - 1

    def flush(self):
        if self.result is not None or self.running:
            return
        model = self.model()
        _LIVE.discard(self)
        if model is None:
            raise RuntimeError("deferred chunks outlived their model")
        if model.__dict__.get("_pending") is self:
            model._pending = None
        if not self.closed or len(self.slots) < self.limit:       # the loop was interrupted (a read, a short last chunk): the next one starts short again
            model.__dict__.pop("_defer_ramp", None)
        if _versions(model) != self.versions:
            raise RuntimeError("model parameters changed while chunk forwards were pending (deferred_chunks > 0): call "
                               "model.flush() before modifying weights")
        self.running = True
        try:
            now = self.gen.get_state()
            # "untouched": nobody seeded or drew since the first chunk was recorded -- then the pass consuming the stream is what immediate
            # execution would have done.  A re-seed with the SAME seed leaves an equal state but is a touch (seed_epoch): the caller's
            # state is put back afterwards, so that the chunk it seeded for draws from the seed
            untouched = torch.equal(now, self.gen_state) and self.seed_epoch == seed_epoch()
            self.gen.set_state(self.gen_state)       # draw as if the pass had run when its first chunk was recorded
            n = self.rays
            hdr = self.hdr[:n] if self.hdr is not None else None
            self.result = model._render(self.uv[:n], self.pose, self.K, self.mask[:n], hdr, self.chunk, *self.render_args)
            if not untouched:                      # somebody re-seeded / drew in between: leave their stream alone
                self.gen.set_state(now)
        finally:
            self.running = False
        hdr_rows = self.hdr[:self.rays] if self.hdr is not None else None
        self.uv = self.mask = self.hdr = None
        keys = set(self.result) - {"object_mask", "hdr_shift"}
        if keys != set(self.spec):
            raise RuntimeError(f"deferred pass returned {sorted(keys ^ set(self.spec))} unexpectedly")
        if self.trace:
            try:
                self._run_traces(model, hdr_rows)
            except BaseException as e:          # keep the failure: get_trace re-raises it instead of unpacking a missing result
                self.trace_error = e
                raise

    # ------------------------------------------------------------------ recorded trace_radiance calls
    def record_trace(self, slot, nsamp):
        """IDRNetwork.trace_radiance on the (unmodified) outputs of recorded chunk `slot` with default arguments: recorded, to run behind
        the pass with every chunk its own lock-step batch.  None = cannot be recorded (the pass has run; another nsamp; a second trace of
        the same chunk; the caller drew from the CPU generator since the first recorded trace): the caller runs it at once."""
        if self.result is not None or self.running or slot in self.trace or self.hdr is None:
            return None
        if self.trace and (nsamp != self.trace_nsamp or not torch.equal(torch.default_generator.get_state(), self.cpu_state)):
            return None
        if not self.trace:
            self.trace_nsamp, self.cpu_state = int(nsamp), torch.default_generator.get_state()
        self.trace[slot] = int(nsamp)
        return TraceOutputs(self, slot, self.slots[slot][1], int(nsamp))

    def _run_traces(self, model, hdr_rows):
        """One grouped trace_radiance over the traced chunks of the pass that has just run.  Draws: the reference's trace_radiance takes
        two torch.rand(n_hit * nsamp) from the CPU generator per call (implicit_differentiable_renderer.py:583-589) -- drawn here chunk by
        chunk in slot order from the generator state of the first recorded trace, i.e. the numbers the immediate per-chunk calls draw."""
        slots = sorted(self.trace)
        res, ns = self.result, self.trace_nsamp
        whole = slots == list(range(len(self.slots)))
        rows = None if whole else torch.cat([torch.arange(a, a + n, device=res["points"].device) for a, n in (self.slots[k] for k in slots)])
        pick = (lambda t: t) if whole else (lambda t: t.index_select(0, rows))
        inp = {"points": pick(res["points"]), "hdr_shift": pick(hdr_rows), "network_object_mask": pick(res["network_object_mask"]),
               "normals": pick(res["normals"])}
        hits = inp["network_object_mask"]
        sizes = [self.slots[k][1] for k in slots]
        counts = [int(c) for c in torch.stack([h.sum() for h in hits.split(sizes)]).cpu()]       # one host read per pass
        now = torch.default_generator.get_state()
        untouched = torch.equal(now, self.cpu_state)
        torch.default_generator.set_state(self.cpu_state)
        u1, u2 = [], []
        for c in counts:                             # torch.rand(n * nsamp) twice per chunk, in the reference's order
            u1.append(torch.rand(c * ns))
            u2.append(torch.rand(c * ns))
        if not untouched:
            torch.default_generator.set_state(now)
        dev = res["points"].device
        out = model.trace_radiance(inp, nsamp=ns, draws=(torch.cat(u1).to(dev), torch.cat(u2).to(dev)), chunk=self.chunk)
        row0, hit0, acc_r, acc_h = {}, {}, 0, 0
        for k, n, c in zip(slots, sizes, counts):
            row0[k], hit0[k] = (acc_r, n), (acc_h, c)
            acc_r, acc_h = acc_r + n, acc_h + c
        self.trace_result = (out, row0, hit0)

    def get_trace(self, slot, name):
        self.flush()
        if self.trace_result is None:
            err = getattr(self, "trace_error", None)
            raise RuntimeError("the recorded trace_radiance calls of this pass failed" + (f": {err!r}" if err is not None else "")) from err
        out, row0, hit0 = self.trace_result
        a, n = hit0[slot] if name == "sample_dirs" else row0[slot]
        return out[name][a:a + n]

    def get(self, slot, name):
        self.flush()
        v = self.result[name]
        if self.spec[name] is None:
            return v
        a, n = self.slots[slot]
        return v[a:a + n]


def _versions(model):
    return tuple(p._version for p in model.parameters())


class ChunkOutputs(dict):
    """What a recorded forward() returns: placeholders made on first access (the runners read 6-10 of the 25 fields)."""

    def __init__(self, queue, slot, n, device, given):
        super().__init__(given)
        self._q, self._slot, self._n, self._dev = queue, slot, n, device
        self._dirty = False                # the caller replaced / removed an entry: no longer "the outputs of the recorded chunk"

    def __setitem__(self, k, v):
        self._dirty = True
        dict.__setitem__(self, k, v)

    def __delitem__(self, k):
        self._dirty = True
        dict.__delitem__(self, k)

    def update(self, *a, **k):
        self._dirty = True
        dict.update(self, *a, **k)

    def pop(self, *a):
        self._dirty = True
        return dict.pop(self, *a)

    def popitem(self):
        self._dirty = True
        return dict.popitem(self)

    def clear(self):
        self._dirty = True
        dict.clear(self)

    def setdefault(self, k, default=None):
        if dict.__contains__(self, k) or k in self._q.spec:       # an output of the chunk: setdefault returns it, nothing changes
            return self[k]
        self._dirty = True
        return dict.setdefault(self, k, default)

    def __ior__(self, other):
        self.update(other)
        return self

    def __missing__(self, k):
        q = self._q
        if k not in q.spec:
            raise KeyError(k)
        s = q.spec[k]
        meta = (torch.empty((), device="meta") if s is None
                else torch.empty((self._n,) + s[0], dtype=s[1], device="meta"))
        slot = self._slot
        t = DeferredTensor(meta, self._dev, lambda: q.get(slot, k))
        dict.__setitem__(self, k, t)
        return t

    def _fill(self):
        for k in self._q.spec:
            if not dict.__contains__(self, k):
                self.__missing__(k)

    def get(self, k, default=None):
        if dict.__contains__(self, k) or k in self._q.spec:
            return self[k]
        return default

    def __contains__(self, k):
        return dict.__contains__(self, k) or k in self._q.spec

    def __iter__(self):
        self._fill()
        return dict.__iter__(self)

    def __len__(self):
        self._fill()
        return dict.__len__(self)

    def keys(self):
        self._fill()
        return dict.keys(self)

    def items(self):
        self._fill()
        return dict.items(self)

    def values(self):
        self._fill()
        return dict.values(self)

    def copy(self):
        self._fill()
        return dict(self)

    def __repr__(self):
        self._fill()
        return dict.__repr__(self)

    def __eq__(self, other):
        self._fill()
        return dict.__eq__(self, other)

    __hash__ = None


class TraceOutputs(dict):
    """What a RECORDED trace_radiance returns (IDRNetwork.trace_radiance on the outputs of a recorded chunk): placeholders for the five
    per-ray results; `sample_dirs` has one row per HIT ray (a data-dependent shape), so reading it runs the pass."""

    _SPEC = {"trace_radiance": (lambda n, s: (n, s, 3), torch.float32), "gt_vis": (lambda n, s: (n, s, 1), torch.bool),
             "pred_vis": (lambda n, s: (n, s, 2), torch.float32), "indir_mask": (lambda n, s: (n, s), torch.bool),
             "gt_integral": (lambda n, s: (n, 3), torch.float32)}
    _KEYS = ("trace_radiance", "sample_dirs", "gt_vis", "pred_vis", "indir_mask", "gt_integral")

    def __init__(self, queue, slot, n, nsamp):
        super().__init__()
        self._q, self._slot, self._n, self._ns = queue, slot, n, nsamp

    def __missing__(self, k):
        q, slot = self._q, self._slot
        if k == "sample_dirs":
            t = q.get_trace(slot, k)
        elif k in self._SPEC:
            shape, dt = self._SPEC[k]
            t = DeferredTensor(torch.empty(shape(self._n, self._ns), dtype=dt, device="meta"), q.pose.device, lambda: q.get_trace(slot, k))
        else:
            raise KeyError(k)
        dict.__setitem__(self, k, t)
        return t

    def _fill(self):
        for k in self._KEYS:
            if not dict.__contains__(self, k):
                self.__missing__(k)

    def get(self, k, default=None):
        return self[k] if k in self._KEYS else default

    def __contains__(self, k):
        return k in self._KEYS

    def __iter__(self):
        return iter(self._KEYS)

    def __len__(self):
        return len(self._KEYS)

    def keys(self):
        self._fill()
        return dict.keys(self)

    def items(self):
        self._fill()
        return dict.items(self)

    def values(self):
        self._fill()
        return dict.values(self)

    __hash__ = None
