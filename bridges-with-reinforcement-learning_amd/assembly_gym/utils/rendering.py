"""render_blocks_2d (assembly_gym/assembly_gym/utils/rendering.py:105-113 of the reference) on the HIP
rasteriser.  The matplotlib / pybullet plotting helpers of the reference are visualisation only and not part of
the path."""
import numpy as np

from bridges_hip import ops


def render_blocks_2d_bits(blocks, xlim, ylim, img_size=(64, 64)):
    """Device bit raster (int64 [64]) of the union of the blocks."""
    return ops.bits_or(ops.raster_bits(list(blocks), xlim, ylim, img_size))


def render_blocks_2d(blocks, xlim, ylim, img_size=(512, 512)):
    """bool image of the union of the blocks, any size (rendering.py:105-113; default 512 x 512 like the reference).  Square
    images of up to 64 pixels -- the training loop's 64x64, successor_dqn.py:585 -- come from the bit rasteriser, every other
    size from the per-pixel operator (bridges_render_blocks); both apply the same pixel test.  The reference builds its grid
    with meshgrid(linspace(xlim, img_size[0]), linspace(ylim, img_size[1])) -- an [img_size[1], img_size[0]] array of rows top to
    bottom -- and then reshapes it to img_size: for a non-square size that is a re-interpretation of the row-major buffer, not
    a transpose, and it is reproduced as is."""
    w, h = int(img_size[0]), int(img_size[1])
    if w == h and 2 <= w <= 64:
        bits = render_blocks_2d_bits(blocks, xlim, ylim, img_size)
        return ops.crop(ops.bits_to_f32(bits), img_size)[0].cpu().numpy().astype(bool)
    return ops.render_blocks(list(blocks), xlim, ylim, (w, h)).cpu().numpy().astype(bool).reshape(tuple(img_size))
