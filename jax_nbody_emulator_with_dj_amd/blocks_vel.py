"""Same names as the reference's `blocks_vel` module (reference blocks_vel.py:30-159), backed by the HIP engine: see building_blocks.py."""

from .building_blocks import ResampleBlock3DVel, ResNetBlock3DVel  # noqa: F401

__all__ = ["ResampleBlock3DVel", "ResNetBlock3DVel"]
