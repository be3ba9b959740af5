"""Overlay for the reference's model/sg_render.py (PEP 420 namespace overlay): same public names."""
from robir_amd.sg_render import (TINY_NUMBER, compute_envmap, render_envmap_sg, render_envmap, norm_axis,  # noqa: F401
                                 hemisphere_int, lambda_trick, get_diffuse_visibility, get_specular_visibility,
                                 render_with_all_sg, render_with_sg)
