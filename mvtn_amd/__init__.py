"""mvtn_amd — MI355X-native (gfx950) differentiable multi-view renderer behind MVTN's
`MVRenderer` interface (reference models/renderer.py).  See DESIGN.md."""
from .renderer import MVRenderer  # noqa: F401
from .structures import Cameras, Meshes, PackedMeshes  # noqa: F401

__all__ = ["MVRenderer", "Meshes", "PackedMeshes", "Cameras"]
