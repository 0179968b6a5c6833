"""Same names as the reference's `style_blocks` module (reference style_blocks.py:27-160), backed by the HIP engine: see building_blocks.py."""

from .building_blocks import StyleResampleBlock3D, StyleResNetBlock3D  # noqa: F401

__all__ = ["StyleResampleBlock3D", "StyleResNetBlock3D"]
