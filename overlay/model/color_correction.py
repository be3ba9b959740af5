"""Overlay for the reference's model/color_correction.py: the tone-mapping objects and the nine curve functions (:31-73)."""
from robir_amd.color_correction import (GammaCorrect, ACESToneMapping, aces_fn, aces_inv, warp_aces_inv, warp_aces_fn,  # noqa: F401
                                        scale_aces_inv, scale_aces_fn, identity_fn, ln_space_fn, ln_space_inv)
