"""Host-side arithmetic of the multi-GPU tile split (mirrors mvrt_pt_set_tile / mvrt_pt_assemble_tiles).

The frame is cut into the reference's own 256-pixel blocks (RENDER_NUMBER_OF_THREAD, renderCommon.hpp:13);
block b belongs to rank b % n.  Each rank stores its pixels compactly in block order, padded so that
every rank holds the same number of blocks (equal chunks for one all-gather).
"""
import numpy as np

TILE = 256


def owned_pixels(width, height, n_ranks):
    """padded pixels per rank"""
    n_blocks = (width * height + TILE - 1) // TILE
    return (n_blocks + n_ranks - 1) // n_ranks * TILE


def global_pixel_index(width, height, rank, n_ranks):
    """int64 array[owned]: global pixel index of each local pixel, -1 for padding"""
    owned = owned_pixels(width, height, n_ranks)
    local = np.arange(owned, dtype=np.int64)
    g = ((local // TILE) * n_ranks + rank) * TILE + local % TILE
    g[g >= width * height] = -1
    return g


def assemble(gathered, width, height):
    """gathered: array [n_ranks, owned, C] -> frame [width*height, C] (numpy mirror of kAssembleTiles)"""
    n_ranks = gathered.shape[0]
    out = np.zeros((width * height,) + gathered.shape[2:], gathered.dtype)
    for r in range(n_ranks):
        g = global_pixel_index(width, height, r, n_ranks)
        ok = g >= 0
        out[g[ok]] = gathered[r][ok]
    return out
