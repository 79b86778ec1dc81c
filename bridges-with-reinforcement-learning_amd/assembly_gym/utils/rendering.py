"""render_blocks_2d (assembly_gym/assembly_gym/utils/rendering.py:105-113 of the reference) on the HIP
rasteriser.  The matplotlib / pybullet plotting helpers of the reference are visualisation only and not part of
the path."""
import numpy as np

from bridges_hip import ops


def render_blocks_2d_bits(blocks, xlim, ylim, img_size=(64, 64)):
    """Device bit raster (int64 [64]) of the union of the blocks."""
    return ops.bits_or(ops.raster_bits(list(blocks), xlim, ylim, img_size))


def render_blocks_2d(blocks, xlim, ylim, img_size=(512, 512)):
    """Square images of up to 64 pixels (the training loop's default is 64x64, successor_dqn.py:585); the
    reference's own default of 512x512 is only used by its plotting helpers and raises NotImplementedError here."""
    bits = render_blocks_2d_bits(blocks, xlim, ylim, img_size)
    return ops.crop(ops.bits_to_f32(bits), img_size)[0].cpu().numpy().astype(bool)
