"""Import shim that lets the *reference* (ingra14m/RobIR, mounted read-only at
/root/reference) run on CPU inside the build container.

TEST INFRASTRUCTURE ONLY.  Used by oracle/gen_golden.py to record golden vectors
and oracle/gen_golden_r2.py to record golden vectors and, in the same runs, to pin the CPU restatement in
oracle/robir_oracle (reports: oracle/PINNING.json, oracle/PINNING_r2.json).
Nothing in robir_amd/, bench.py or the gpu tests imports this file and
/root/reference does not exist on the GPU box.

What it does:
  * stub modules the reference imports but the image lacks (gin, pyhocon, imageio, cv2,
    ipdb, torchvision, tensorboardX, trimesh, xatlas, glfw, GPUtil, OpenGL rasteriser),
  * stub `datasets` (site-packages holds HuggingFace `datasets`, which would win),
  * torch_scatter.scatter_min -> scatter_reduce('amin'),
  * a TorchFunctionMode that turns `.cuda()` into identity and device='cuda' into 'cpu'.
"""
import os
import sys
import types

import torch
from torch.overrides import TorchFunctionMode

REF_ROOT = os.environ.get("ROBIR_REFERENCE", "/root/reference")


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__dict__.setdefault("__getattr__", lambda k: _Anything())      # PEP 562: any other attribute is a sink
    m.__dict__.setdefault("__file__", None)                          # ... except what `inspect` probes on every module
    sys.modules[name] = m
    return m


def _passthrough_decorator(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda f: f


class _Anything:
    """Attribute sink: any attribute access / call returns another sink."""

    def __getattr__(self, k):
        return _Anything()

    def __call__(self, *a, **k):
        return _Anything()


def _scatter_min(src, index, dim=-1, out=None, dim_size=None):
    if src.numel() == 0:
        e = torch.zeros(0, dtype=src.dtype, device=src.device)
        return e, e.long()
    n = int(index.max().item()) + 1 if dim_size is None else dim_size
    res = torch.full((n,), torch.iinfo(src.dtype).max if not src.is_floating_point() else float("inf"),
                     dtype=src.dtype, device=src.device)
    res = res.scatter_reduce(0, index, src, reduce="amin", include_self=True)
    return res, None


class CpuMode(TorchFunctionMode):
    """Make hard-coded `.cuda()` / device='cuda' land on the CPU."""

    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        name = getattr(func, "__name__", "")
        if name == "cuda" and args and isinstance(args[0], torch.Tensor):
            return args[0]
        dev = kwargs.get("device", None)
        if dev is not None and "cuda" in str(dev):
            kwargs["device"] = "cpu"
        if name == "to" and len(args) >= 2 and isinstance(args[1], (str, torch.device)) and "cuda" in str(args[1]):
            args = (args[0], "cpu") + tuple(args[2:])
        return func(*args, **kwargs)


_installed = False


def install():
    """Idempotent: install stubs, put the reference first on sys.path."""
    global _installed
    if _installed:
        return
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError(f"reference tree not found at {REF_ROOT}; the shim only works in the build container")

    _mod("gin", register=_passthrough_decorator, configurable=_passthrough_decorator,
         REQUIRED=None, constant=lambda *a, **k: None)
    fi = _mod("imageio.plugins.freeimage", download=lambda *a, **k: None)
    pl = _mod("imageio.plugins", freeimage=fi)
    _mod("imageio", plugins=pl, imread=_Anything(), imwrite=_Anything())
    for name in ("cv2", "ipdb", "tensorboardX", "trimesh", "xatlas", "glfw", "GPUtil", "pyhocon"):
        m = _mod(name)
        m.__getattr__ = lambda k: _Anything()  # type: ignore
    tvu = _mod("torchvision.utils", save_image=lambda *a, **k: None)
    _mod("torchvision", utils=tvu)
    _mod("model.rasterizor", Rasterizor=_Anything())
    sd = _mod("datasets.syn_dataset", SynDataset=_Anything)
    ds = _mod("datasets", syn_dataset=sd)
    ds.__path__ = []  # mark as package
    _mod("torch_scatter", scatter_min=_scatter_min)

    # torch.load in the reference passes a storage.cuda() map_location lambda
    _orig_load = torch.load

    def _load(f, *a, **k):
        k["map_location"] = "cpu"
        k["weights_only"] = False
        return _orig_load(f, **k)

    torch.load = _load

    # nn.Module.cuda() is a python method that maps t.cuda() over params -> handled by CpuMode,
    # but make it a no-op outright so it also works outside the mode.
    torch.nn.Module.cuda = lambda self, *a, **k: self

    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    _installed = True


class DictConf:
    """Tiny stand-in for the pyhocon accessors the reference uses on `conf.model`."""

    def __init__(self, d):
        self.d = d

    def _get(self, key):
        cur = self.d
        for part in key.split("."):
            cur = cur[part]
        return cur

    def get_bool(self, k):
        return bool(self._get(k))

    def get_int(self, k):
        return int(self._get(k))

    def get_float(self, k):
        return float(self._get(k))

    def get_config(self, k):
        return DictConf(self._get(k))

    def get_list(self, k):
        return list(self._get(k))

    # `**conf.get_config(...)` support
    def keys(self):
        return self.d.keys()

    def __getitem__(self, k):
        return self.d[k]


def hotdog_model_conf():
    """Values of confs_sg/hotdog.conf:65-123 (model section)."""
    return DictConf({
        "gamma": 1.0, "hdr_mode": 0, "use_neus": True, "use_octree": True, "feature_vector_size": 256,
        "implicit_network": {"d_in": 3, "d_out": 1, "dims": [512] * 8, "geometric_init": True, "bias": 0.6,
                             "skip_in": [4], "weight_norm": True, "multires": 6},
        "rendering_network": {"mode": "idr", "d_in": 9, "d_out": 3, "dims": [512] * 4, "weight_norm": True,
                              "multires_view": 4},
        "indirect_illum_network": {"multires": 10, "dims": [512] * 4, "num_lgt_sgs": 24},
        "visibility_network": {"points_multires": 10, "dirs_multires": 10, "dims": [256] * 4},
        "envmap_material_network": {"multires": 10, "brdf_encoder_dims": [512] * 4, "brdf_decoder_dims": [128, 128],
                                    "num_lgt_sgs": 128, "upper_hemi": False, "specular_albedo": 0.05,
                                    "latent_dim": 32},
        "ray_tracer": {"object_bounding_sphere": 1.0, "sdf_threshold": 5.0e-5, "line_search_step": 0.5,
                       "line_step_iters": 3, "sphere_tracing_iters": 10, "n_steps": 100, "n_rootfind_steps": 32},
    })
