"""Overlay for the reference's model/sdf_render.py."""
from robir_amd.sdf_render import Rays, render_neus  # noqa: F401
