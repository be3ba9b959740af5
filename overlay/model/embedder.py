"""Overlay for the reference's model/embedder.py."""
from robir_amd.embedder import get_embedder, ipe_embedder, Embedder, IPE, isotropic_cov  # noqa: F401
