"""Overlay for the reference's model/implicit_differentiable_renderer.py: IDRNetwork and its sub-networks on HIP."""
from robir_amd.renderer import IDRNetwork, TINY_NUMBER  # noqa: F401
from robir_amd.nets import IndirctIllumNetwork, VisNetwork  # noqa: F401
