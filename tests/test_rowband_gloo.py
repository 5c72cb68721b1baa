"""2-D row-band split with the neighbour halo exchange, on CPU: `gloo`, world sizes 2 and 3.

The exchange (savitzky-golay-filter_amd/rowband.py) is the only communication step of the hot path; on the GPU box
it runs over RCCL.  Here the per-band filter is the CPU oracle (tests may use it), so what is checked is the
partition, the halo rows and the boundary handling at real vs artificial band edges, for all three 2-D modes:
the stitched bands must equal the whole-frame result bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILL = -777.0


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _frames(images, rows, cols):
    rng = np.random.default_rng(21)
    return rng.normal(0, 1, (images, rows, cols)).astype(np.float32)


def _oracle_apply(sgo, n, boundary):
    f = sgo.Filter2D(n, n, 3, 1, 0, 0.5, 1.0)

    def fn(frames):                                   # frames: torch [images, R, cols] -> same shape
        a = frames.numpy()
        out = np.stack([f.apply(a[k], a.shape[2], boundary, out=np.full(a[k].shape, FILL, np.float32)) for k in range(a.shape[0])])
        return torch.from_numpy(out)
    return fn


def _worker(rank, world, port, images, rows, cols, n, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    from oracle import sgo
    load_package()
    rowband = importlib.import_module("savgol_amd.rowband")
    band = rowband.RowBand(rows, n)
    x = _frames(images, rows, cols)
    local = torch.from_numpy(x[:, band.lo:band.hi].copy())
    ext = band.exchange(local)
    for b in range(3):
        own = band.apply(ext, _oracle_apply(sgo, n, b))
        np.save(os.path.join(out_dir, f"b{b}_r{rank}.npy"), own.numpy())
        # the overlapped form (exchange in flight while the band is filtered, edge strips redone afterwards)
        own2 = band.apply_overlapped(local, _oracle_apply(sgo, n, b))
        np.save(os.path.join(out_dir, f"o{b}_r{rank}.npy"), own2.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_row_band_halo_exchange_matches_whole_frame(tmp_path, sgo, world):
    images, rows, cols, n = 3, 53, 40, 4
    mp.spawn(_worker, args=(world, _free_port(), images, rows, cols, n, str(tmp_path)), nprocs=world, join=True)
    x = _frames(images, rows, cols)
    for b in range(3):
        whole = _oracle_apply(sgo, n, b)(torch.from_numpy(x)).numpy()
        got = np.concatenate([np.load(tmp_path / f"b{b}_r{r}.npy") for r in range(world)], axis=1)
        assert got.shape == whole.shape
        assert np.array_equal(got, whole), f"boundary {b}: max diff {np.max(np.abs(got - whole))}"
        got2 = np.concatenate([np.load(tmp_path / f"o{b}_r{r}.npy") for r in range(world)], axis=1)
        assert np.array_equal(got2, whole), f"overlapped, boundary {b}: max diff {np.max(np.abs(got2 - whole))}"


def test_thin_bands_take_the_simple_form_instead_of_truncating(tmp_path, sgo):
    """apply_overlapped rebuilds an artificial edge from the band's own 2*ny rows; with bands of ny .. 2*ny - 1 rows the
    slice used to truncate silently and the strip's middle rows were computed with boundary handling instead of real neighbour data.
    Such bands (here 3 ranks x 6-7 rows at ny = 4) now take exchange() + apply(); the stitched result is still the whole frame's."""
    images, rows, cols, n, world = 2, 20, 33, 4, 3
    mp.spawn(_worker, args=(world, _free_port(), images, rows, cols, n, str(tmp_path)), nprocs=world, join=True)
    x = _frames(images, rows, cols)
    for b in range(3):
        whole = _oracle_apply(sgo, n, b)(torch.from_numpy(x)).numpy()
        for tag in ("b", "o"):
            got = np.concatenate([np.load(tmp_path / f"{tag}{b}_r{r}.npy") for r in range(world)], axis=1)
            assert np.array_equal(got, whole), (tag, b)


def test_row_band_refuses_bands_thinner_than_the_window(sg):
    import importlib
    rowband = importlib.import_module("savgol_amd.rowband")
    with pytest.raises(ValueError):
        rowband.RowBand(20, 7, rank=0, world_size=4)
    b = rowband.RowBand(100, 7, rank=0, world_size=1)
    assert (b.lo, b.hi, b.top, b.bottom) == (0, 100, 0, 0)
    assert rowband.RowBand(40, 7, rank=1, world_size=4).thin and not rowband.RowBand(56, 7, rank=1, world_size=4).thin
    # the C ABI's plan agrees with the Python split and refuses bands below 2 x ny rows (no GPU needed: host arithmetic)
    import ctypes as C
    L = sg.lib()
    lo, hi, up, dn = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    for world in (1, 2, 3, 5):
        for rank in range(world):
            assert L.savgol2d_rowband_plan(173, 7, rank, world, C.byref(lo), C.byref(hi), C.byref(up), C.byref(dn)) == 0
            rb = rowband.RowBand(173, 7, rank=rank, world_size=world)
            assert (lo.value, hi.value, up.value, dn.value) == (rb.lo, rb.hi, rb.top, rb.bottom)
    assert L.savgol2d_rowband_plan(27, 7, 0, 2, C.byref(lo), C.byref(hi), None, None) == -1
