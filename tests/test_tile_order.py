"""The block -> tile maps of the 1-D batch kernels (csrc/sg_k1d.hpp, sg1d_tile_body: chunks of 2^s blocks dealt to the XCDs round robin) and of the 2-D tile
kernel (csrc/sg_2d_roll.hip, sg2d_rolling_kernel: chunks of `chunk` blocks) restated in Python and checked to be PERMUTATIONS of the launch's blocks for
every grid size and chunk -- a block that maps twice or not at all would drop or double a tile, and only on grids the GPU tests may never launch.
Also: what the order is for (an XCD = blockIdx % 8 sweeps whole chunks; the eight chunks of a span are neighbours)."""
import numpy as np
import pytest


def order_1d(nblocks, s):
    """csrc/sg_k1d.hpp, sg1d_tile_body: nb8 = gridDim.x >> 3; s == 0 contiguous eighths, 1..31 chunks of 2^s blocks, >= 32 launch order"""
    nb8 = nblocks >> 3
    out = np.arange(nblocks, dtype=np.int64)
    for blk in range(nblocks):
        b = blk
        if blk < nb8 * 8:
            if s == 0:
                b = (blk & 7) * nb8 + (blk >> 3)
            elif s < 32:
                span, q = 8 << s, blk >> (s + 3)
                if (q + 1) * span <= nb8 * 8:
                    r = blk & (span - 1)
                    b = (((q << 3) + (r & 7)) << s) + (r >> 3)
        out[blk] = b
    return out


def order_2d(nblk, chunk):
    """csrc/sg_2d_roll.hip, sg2d_rolling_kernel: chunk == 0 contiguous eighths, else chunks of `chunk` blocks; the last partial span keeps launch order"""
    out = np.arange(nblk, dtype=np.int64)
    for blk in range(nblk):
        b = blk
        if chunk == 0:
            b = (blk & 7) * (nblk >> 3) + (blk >> 3)
        else:
            span = 8 * chunk
            q = blk // span
            if (q + 1) * span <= nblk:
                r = blk - q * span
                b = (q * 8 + (r & 7)) * chunk + (r >> 3)
        out[blk] = b
    return out


@pytest.mark.parametrize("s", [0, 1, 3, 6, 8, 12, 40])
def test_1d_order_is_a_permutation(s):
    rng = np.random.default_rng(s)
    for nblocks in [8, 16, 24, 512, 520, 1024, 4104] + [int(v) * 8 for v in rng.integers(1, 3000, 12)]:
        o = order_1d(nblocks, s)
        assert np.array_equal(np.sort(o), np.arange(nblocks)), (s, nblocks)


@pytest.mark.parametrize("chunk", [0, 1, 2, 37, 64, 1845, 3690])
def test_2d_order_is_a_permutation(chunk):
    rng = np.random.default_rng(chunk)
    for nblk in [8, 64, 8 * 1845, 8 * 3690, 8 * 3690 + 8] + [int(v) * 8 for v in rng.integers(1, 6000, 12)]:
        o = order_2d(nblk, chunk)
        assert np.array_equal(np.sort(o), np.arange(nblk)), (chunk, nblk)


def test_an_xcd_sweeps_whole_chunks_next_to_the_other_seven():
    s, nblocks = 6, 8 * 64 * 5
    o = order_1d(nblocks, s)
    for xcd in range(8):
        mine = o[xcd::8]                                   # the blocks the dispatcher hands XCD `xcd` (blockIdx % 8), in time order
        chunks = mine.reshape(-1, 64)
        assert np.all(np.diff(chunks, axis=1) == 1)        # 64 consecutive tiles-of-four at a time
        assert np.array_equal(chunks[:, 0] // 64, xcd + 8 * np.arange(chunks.shape[0]))   # span q: chunks 8q .. 8q+7, one per XCD
