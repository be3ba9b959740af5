"""Overlay for the reference's model/octree_tracing.py."""
from robir_amd.octree_tracing import OctreeTracing, OctreeVisModel, OctreeSDF  # noqa: F401
