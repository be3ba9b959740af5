"""model.ray_tracing overlay: RayTracing (model/ray_tracing.py:6)."""
from robir_amd.ray_tracing import RayTracing  # noqa: F401
