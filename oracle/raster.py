"""Oracle: boolean rasteriser and feature maps.

Restates ``render_blocks_2d`` (assembly_gym/assembly_gym/utils/rendering.py:105-113),
``Shape.contains_2d`` (assembly_gym/assembly_gym/envs/assembly_env.py:126-137),
``gaussian_kernel`` / ``convolve_with_gaussian`` (robotoddler/utils/utils.py:93-115)
and ``get_task_features`` (robotoddler/training/successor_dqn.py:67-85).

Arithmetic contract: pixel (row r, col q) samples the point
(X[q], Y[r]) with X = linspace(xlim0, xlim1, W), Y = linspace(ylim1, ylim0, H)
(row 0 = top).  It is inside a block iff for every 2-D face
    ((X[q] - c.x) * n.x) + ((Y[r] - c.z) * n.z) <= 0
with the four operations rounded separately (no FMA).  The reference evaluates
``np.dot(points - offset, normal)`` whose BLAS kernel may fuse; pixels within
one ulp of an edge are "parity unpinned" (SURVEY.md §7).
"""
import numpy as np


def pixel_grid(xlim, ylim, img_size=(64, 64)):
    X = np.linspace(xlim[0], xlim[1], img_size[0])
    Y = np.linspace(ylim[1], ylim[0], img_size[1])
    return X, Y


def contains_2d(block, X, Y):
    """bool[H, W] (assembly_env.py:126-137)."""
    inside = np.ones((len(Y), len(X)), dtype=bool)
    for (c, _t, n) in block.frames:
        tx = (X - c[0]) * n[0]          # [W]
        tz = (Y - c[1]) * n[1]          # [H]
        d = tx[None, :] + tz[:, None]
        inside &= d <= 0
    return inside


def render_blocks_2d(blocks, xlim, ylim, img_size=(64, 64)):
    """rendering.py:105-113 (square images only, as in the reference)."""
    X, Y = pixel_grid(xlim, ylim, img_size)
    image = np.zeros((len(Y), len(X)), dtype=bool)
    for b in blocks:
        image |= contains_2d(b, X, Y)
    return image


def gaussian_kernel_1d(kernel_size, sigma):
    """utils.py:95-100 in float32, as torch computes it."""
    coords = (np.arange(kernel_size) - kernel_size // 2).astype(np.float32)
    k1 = np.exp(-(coords ** 2) / np.float32(2 * sigma ** 2)).astype(np.float32)
    return (k1 / k1.sum(dtype=np.float32)).astype(np.float32)


def gaussian_kernel(kernel_size, sigma):
    """utils.py:93-106."""
    k1 = gaussian_kernel_1d(kernel_size, sigma)
    return (k1[None, :] * k1[:, None]).astype(np.float32)


def convolve_with_gaussian(img, kernel_size=101, sigma=16):
    """utils.py:107-115: zero-padded 'same' cross-correlation.  Evaluated
    separably in float64 (k2d = k1 k1^T up to one float32 rounding); the fixture
    test compares with torch's float32 conv2d to 1e-5."""
    k1 = gaussian_kernel_1d(kernel_size, sigma).astype(np.float64)
    H, W = img.shape
    p = kernel_size // 2
    pad = np.zeros((H + 2 * p, W + 2 * p))
    pad[p:p + H, p:p + W] = img
    tmp = np.zeros((H + 2 * p, W))
    for j in range(kernel_size):
        tmp += k1[j] * pad[:, j:j + W]
    out = np.zeros((H, W))
    for i in range(kernel_size):
        out += k1[i] * tmp[i:i + H, :]
    return out.astype(np.float32)
