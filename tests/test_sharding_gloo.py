"""Multi-GPU path on CPU: world_size-2 `gloo` processes, each owning its slice of the channels.

The 1-D path shards by independent channels with no data-path collective; what has to
be right is the host logic: slice arithmetic (`shard_range`), the synthetic workload being a function of
the GLOBAL channel index, and the barrier / max-over-ranks timing reduce bench.py uses.  The GPU kernel is
replaced here by the CPU oracle (tests may use it), so the test runs without a GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, channels, length, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from __graft_entry__ import load_package
    from oracle import sgo
    sg = load_package()
    lo, hi = sg.shard_range(channels, world, rank)
    x = sgo.synth_f32(lo, hi - lo, length)                    # global channel index lo.., as bench.py does on the GPU
    y = sgo.Filter(5, 3).apply(x) if hi > lo else np.zeros((0, length), np.float32)
    np.save(os.path.join(out_dir, f"y{rank}.npy"), y)
    # bench.py's timing reduce: barrier, MAX over ranks
    dist.barrier()
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    counts = torch.tensor([hi - lo], dtype=torch.int64)
    dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    if rank == 0:
        np.save(os.path.join(out_dir, "meta.npy"), np.array([t.item(), counts.item()]))
    dist.destroy_process_group()


@pytest.mark.parametrize("channels", [7, 8])
def test_two_rank_channel_sharding_matches_single_process(tmp_path, sgo, channels):
    world, length = 2, 300
    mp.spawn(_worker, args=(world, _free_port(), channels, length, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"y{r}.npy") for r in range(world)]
    whole = sgo.Filter(5, 3).apply(sgo.synth_f32(0, channels, length))
    assert np.array_equal(np.concatenate(parts, axis=0), whole)
    tmax, total = np.load(tmp_path / "meta.npy")
    assert tmax == 2.0 and total == channels


def test_shard_range_partitions_exactly(sg):
    import ctypes as C
    L = sg.lib()
    for total in (0, 1, 7, 8, 4096, 32768):
        for world in (1, 2, 3, 8):
            spans = [sg.shard_range(total, world, r) for r in range(world)]
            for r in range(world):                             # the C-ABI twin (savgol_hip_shard_range) agrees
                lo, hi = C.c_size_t(), C.c_size_t()
                assert L.savgol_hip_shard_range(total, world, r, C.byref(lo), C.byref(hi)) == 0
                assert (lo.value, hi.value) == spans[r]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sg.shard_range(10, 2, 2)
