"""Same names as the reference's `style_layers_vel` module (reference style_layers_vel.py:20-281), backed by the HIP engine: see building_blocks.py."""

from .building_blocks import StyleConvBase3DVel, StyleTransposeBase3DVel, StyleConv3DVel, StyleSkip3DVel, StyleDownSample3DVel, StyleUpSample3DVel  # noqa: F401

__all__ = ["StyleConvBase3DVel", "StyleTransposeBase3DVel", "StyleConv3DVel", "StyleSkip3DVel", "StyleDownSample3DVel", "StyleUpSample3DVel"]
