"""Overlay for the reference's model/color_correction.py (hdr_mode 0)."""
from robir_amd.nets import GammaCorrect, ACESToneMapping  # noqa: F401
