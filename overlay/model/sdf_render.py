"""Overlay for the reference's model/sdf_render.py."""
from robir_amd.sdf_render import (Rays, IComp, ISDF, sample_pdf, up_sample, cat_z_vals, render_core_outside, render_core,  # noqa: F401
                                  render_neus, wrap_renderer)
