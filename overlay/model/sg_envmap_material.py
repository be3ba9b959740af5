"""Overlay for the reference's model/sg_envmap_material.py."""
from robir_amd.nets import SparseAE, EnvmapMaterialNetwork, fibonacci_sphere, compute_energy  # noqa: F401
