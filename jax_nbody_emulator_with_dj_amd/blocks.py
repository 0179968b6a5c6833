"""Same names as the reference's `blocks` module (reference blocks.py:26-153), backed by the HIP engine: see building_blocks.py."""

from .building_blocks import ResampleBlock3D, ResNetBlock3D  # noqa: F401

__all__ = ["ResampleBlock3D", "ResNetBlock3D"]
