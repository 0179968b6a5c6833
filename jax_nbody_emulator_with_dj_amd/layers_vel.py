"""Same names as the reference's `layers_vel` module (reference layers_vel.py:20-192), backed by the HIP engine: see building_blocks.py."""

from .building_blocks import ConvBase3DVel, ConvTransposeBase3DVel, LeakyReLUVel, Conv3DVel, Skip3DVel, DownSample3DVel, UpSample3DVel  # noqa: F401

__all__ = ["ConvBase3DVel", "ConvTransposeBase3DVel", "LeakyReLUVel", "Conv3DVel", "Skip3DVel", "DownSample3DVel", "UpSample3DVel"]
