"""Same names as the reference's `style_blocks_vel` module (reference style_blocks_vel.py:31-166), backed by the HIP engine: see building_blocks.py."""

from .building_blocks import StyleResampleBlock3DVel, StyleResNetBlock3DVel  # noqa: F401

__all__ = ["StyleResampleBlock3DVel", "StyleResNetBlock3DVel"]
