"""Same names as the reference's `style_layers` module (reference style_layers.py:19-197), backed by the HIP engine: see building_blocks.py."""

from .building_blocks import StyleConvBase3D, StyleConvTransposeBase3D, StyleConv3D, StyleSkip3D, StyleDownSample3D, StyleUpSample3D  # noqa: F401

__all__ = ["StyleConvBase3D", "StyleConvTransposeBase3D", "StyleConv3D", "StyleSkip3D", "StyleDownSample3D", "StyleUpSample3D"]
