"""robir_amd: MI355X-native (gfx950) forward renderer for RobIR's per-ray hot path.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed only); all arithmetic of the path
runs in hand-written HIP kernels reached through the C-ABI in include/robir_hip.h (librobir_hip.so).
There is no CPU fallback: importing the operator modules without the built library raises.
"""
from . import synth  # noqa: F401  (pure numpy; usable without the HIP library)

__all__ = ["synth"]
