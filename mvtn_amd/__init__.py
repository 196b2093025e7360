"""mvtn_amd — MI355X-native (gfx950) differentiable multi-view renderer behind MVTN's
`MVRenderer` interface (reference models/renderer.py).  See DESIGN.md."""
from .regularize import regualarize_rendered_views, regularize_rendered_views  # noqa: F401
from .renderer import MVRenderer  # noqa: F401
from .structures import Cameras, Meshes, PackedMeshes  # noqa: F401

__all__ = ["MVRenderer", "Meshes", "PackedMeshes", "Cameras", "regualarize_rendered_views", "regularize_rendered_views"]
