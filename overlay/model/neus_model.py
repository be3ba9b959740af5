"""Overlay for the reference's model/neus_model.py (the names the stage-2 path uses)."""
from robir_amd.nets import (SDFNetwork, RenderingNetwork, SingleVarianceNetwork, NeuSModel, ImplicitNetworkMy)  # noqa: F401
from robir_amd.embedder import get_embedder, PE, IPE, isotropic_cov  # noqa: F401
