"""Overlay for the reference's model/sg_render.py (PEP 420 namespace overlay): same public names."""
from robir_amd.sg_render import (TINY_NUMBER, compute_envmap, render_envmap_sg, render_envmap, norm_axis,  # noqa: F401
                                 get_diffuse_visibility, get_specular_visibility, render_with_all_sg, render_with_sg)


def hemisphere_int(lambda_val, cos_beta):
    raise NotImplementedError("hemisphere_int is fused into rb_sg_shade; call render_with_sg / render_with_all_sg")


def lambda_trick(lobe1, lambda1, mu1, lobe2, lambda2, mu2):
    raise NotImplementedError("lambda_trick is fused into rb_sg_shade; call render_with_sg / render_with_all_sg")
