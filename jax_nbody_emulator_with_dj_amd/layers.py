"""Same names as the reference's `layers` module (reference layers.py:19-133), backed by the HIP engine: see building_blocks.py."""

from .building_blocks import ConvBase3D, ConvTransposeBase3D, LeakyReLU, Conv3D, Skip3D, DownSample3D, UpSample3D  # noqa: F401

__all__ = ["ConvBase3D", "ConvTransposeBase3D", "LeakyReLU", "Conv3D", "Skip3D", "DownSample3D", "UpSample3D"]
