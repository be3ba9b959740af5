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


def build(verbose=False):
    """Compile every HIP translation unit for gfx950 and link librobir_hip.so in-tree."""
    jobs = str(min(8, os.cpu_count() or 1))
    r = subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-j", jobs], capture_output=not verbose, text=True)
    if r.returncode != 0:
        raise RobirHipError("building librobir_hip.so failed:\n" + (r.stdout or "") + (r.stderr or ""))
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RobirHipError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(robir_amd has no CPU fallback)")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.rb_last_error.restype = ctypes.c_char_p
        _lib.rb_packed_layer_floats.restype = ctypes.c_long
        _lib.rb_packed_layer_x6_floats.restype = ctypes.c_long
        _lib.rb_sdf_value_grad_scratch_floats.restype = ctypes.c_long
        _lib.rb_sdf_value_grad_f32_scratch_floats.restype = ctypes.c_long
        if _lib.rb_abi_version() != 5:
            raise RobirHipError("librobir_hip.so ABI version mismatch")
        if os.environ.get("ROBIR_SDF_RING_WAVES") in ("4", "8"):      # value rows of the SDF net: csrc/sdf_ring8.hip | sdf_ring.hip
            _lib.rb_sdf_ring_waves(int(os.environ["ROBIR_SDF_RING_WAVES"]))
    return _lib


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    if t is None:
        return ctypes.c_void_p(0)
    assert t.is_cuda and t.is_contiguous(), "robir_amd kernels need contiguous device tensors"
    return ctypes.c_void_p(t.data_ptr())


def call(name, *args):
    fn = getattr(lib(), name)
    rc = fn(*args)
    if rc != 0:
        raise RobirHipError(f"{name} failed ({rc}): {lib().rb_last_error().decode()}")
