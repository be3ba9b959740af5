"""Overlay for the reference's model/neus_model.py: every public name of the module (the tinycudann / NeRF++ / IPE-SDF variants are
present and raise NotImplementedError: out of scope)."""
from robir_amd.nets import (SDFNetwork, HashSDFNetwork, RenderingNetwork, NeRF, SingleVarianceNetwork, auto_flatten,  # noqa: F401
                            auto_flatten2, NeuSModel, ImplicitNetworkMy)
from robir_amd.embedder import (expected_sin, integrated_pos_enc, isotropic_cov, IPE, TCNNLinear, tcnn_encoding, PE, Hash,  # noqa: F401
                                get_embedder_neus as get_embedder)
