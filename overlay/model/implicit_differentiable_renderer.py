"""Overlay for the reference's model/implicit_differentiable_renderer.py: IDRNetwork and its sub-networks on HIP (the legacy IDR
ImplicitNetwork / RenderingNetwork of use_neus=False are present and raise: out of scope)."""
from robir_amd.renderer import IDRNetwork, ImplicitNetwork, RenderingNetwork, TINY_NUMBER  # noqa: F401
from robir_amd.nets import IndirctIllumNetwork, VisNetwork  # noqa: F401
