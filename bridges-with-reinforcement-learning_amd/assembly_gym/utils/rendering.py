"""render_blocks_2d (assembly_gym/assembly_gym/utils/rendering.py:105-113 of the reference) on the HIP
rasteriser.  The matplotlib / pybullet plotting helpers of the reference are visualisation only and not part of
the path."""
import numpy as np

from bridges_hip import ops


def render_blocks_2d_bits(blocks, xlim, ylim):
    """Device bit raster (int64 [64]) of the union of the blocks."""
    return ops.bits_or(ops.raster_bits(list(blocks), xlim, ylim))


def render_blocks_2d(blocks, xlim, ylim, img_size=(512, 512)):
    if tuple(img_size) != (64, 64):
        raise NotImplementedError("the HIP rasteriser renders 64x64 images (successor_dqn.py:585 default)")
    bits = render_blocks_2d_bits(blocks, xlim, ylim)
    return ops.bits_to_f32(bits)[0].cpu().numpy().astype(bool)
