"""ctypes loader for librobir_hip.so (the C-ABI of include/robir_hip.h).

The product path has no CPU fallback: if the shared library is missing or a call fails, we raise.
torch is imported first so that the process uses ONE HIP runtime (torch bundles libamdhip64.so.7; our library
resolves the same soname against the already-loaded copy).
"""
import ctypes
import os
import subprocess

import torch  # noqa: F401  (must precede the CDLL: pins the HIP runtime instance)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librobir_hip.so")
_lib = None

c_fp = ctypes.c_void_p
c_long = ctypes.c_long
c_int = ctypes.c_int
c_float = ctypes.c_float


class RobirHipError(RuntimeError):
    pass


LEGACY_PATH = os.path.join(_HERE, "librobir_hip_legacy.so")
_legacy = None
ABI_VERSION = 8


def build(verbose=False, legacy=True):
    """Compile every HIP translation unit for gfx950 and link, in-tree, librobir_hip.so (the default library) and -- legacy=True --
    librobir_hip_legacy.so (the superset with the retired kernel generations, csrc/Makefile)."""
    jobs = str(min(8, os.cpu_count() or 1))
    r = subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-j", jobs, "all" if legacy else "default"],
                       capture_output=not verbose, text=True)
    if r.returncode != 0:
        raise RobirHipError("building librobir_hip.so failed:\n" + (r.stdout or "") + (r.stderr or ""))
    return LIB_PATH


def _load(path, what):
    if not os.path.exists(path):
        raise RobirHipError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` " + what)
    L = ctypes.CDLL(path)
    L.rb_last_error.restype = ctypes.c_char_p
    for name in ("rb_packed_layer_floats", "rb_packed_layer_x6_floats", "rb_sdf_value_grad_scratch_floats",
                 "rb_sdf_value_grad_f32_scratch_floats"):
        if hasattr(L, name):
            getattr(L, name).restype = ctypes.c_long
    if L.rb_abi_version() != ABI_VERSION:
        raise RobirHipError(f"{os.path.basename(path)} ABI version mismatch")
    return L


def lib():
    """The default library (include/robir_hip.h): everything the `exact` / `f16` policies and the fp32 override run."""
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH, "(robir_amd has no CPU fallback)")
    return _lib


def legacy():
    """The legacy library (include/robir_hip_legacy.h; `make -C robir_amd/csrc legacy`): loaded the first time a retired entry point is
    called -- ROBIR_PRECISION=split, ROBIR_SDF_FUSED_PE=0, the tests that compare kernel generations.  Its own copy of the process-wide
    state (range sentinel block, device caches)."""
    global _legacy
    if _legacy is None:
        _legacy = _load(LEGACY_PATH, "(the split-precision family and the other retired kernel generations live in the LEGACY library: build it "
                                     "with `make -C robir_amd/csrc legacy` or ROBIR_BUILD_LEGACY=1.  ROBIR_PRECISION=split needs it, and so does "
                                     "ROBIR_PRECISION=f16 for every net but the light-visibility MLP -- ROBIR_PRECISION=f16-vis and the default "
                                     "policy ROBIR_PRECISION=exact do not; current policy: " + os.environ.get("ROBIR_PRECISION", "exact") + ")")
        if os.environ.get("ROBIR_SDF_RING_WAVES") in ("4", "8"):      # value rows of the split SDF net: csrc/sdf_ring8.hip | sdf_ring.hip
            _legacy.rb_sdf_ring_waves(int(os.environ["ROBIR_SDF_RING_WAVES"]))
    return _legacy


def legacy_loaded():
    return _legacy is not None


def resolve(name):
    """(library, function) of an entry point: the default library if it exports `name`, else the legacy one."""
    L = lib()
    fn = getattr(L, name, None)
    if fn is None:
        L = legacy()
        fn = getattr(L, name, None)
        if fn is None:
            raise RobirHipError(f"{name} is exported by neither librobir_hip.so nor librobir_hip_legacy.so")
    return L, fn


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    if t is None:
        return ctypes.c_void_p(0)
    assert t.is_cuda and t.is_contiguous(), "robir_amd kernels need contiguous device tensors"
    return ctypes.c_void_p(t.data_ptr())


def call_legacy(name, *args):
    """An entry point of the LEGACY library even where the default one exports the same name (rb_dvis_fused with precision 5)."""
    L = legacy()
    rc = getattr(L, name)(*args)
    if rc != 0:
        raise RobirHipError(f"{name} failed ({rc}): {L.rb_last_error().decode()}")


def call(name, *args):
    L, fn = resolve(name)
    rc = fn(*args)
    if rc != 0:
        raise RobirHipError(f"{name} failed ({rc}): {L.rb_last_error().decode()}")
