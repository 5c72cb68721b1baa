"""Length split of the 1-D path across GPUs: every rank owns a contiguous segment of EVERY channel and trades the `n` samples
next to each cut with its neighbours ("same pattern (n samples) if one 1-D channel is ever split by
length").

Channels are independent, so the default multi-GPU layout is "each rank filters its own channels" and needs no communication
(`shard_range`).  A split along the length is for the other case: fewer channels than GPUs, or channels too long for one GPU's
memory.  Like the 2-D row bands (rowband.py) it is point to point -- one `isend` / `irecv` pair per neighbour for the whole batch
(RCCL, `backend="nccl"`, one xGMI link per pair) -- and there is no collective.

    seg = LengthSplit(length, n, rank, world)         # which samples [lo, hi) of every channel this rank owns
    ext = seg.exchange(local, periodic)               # local [channels, hi-lo] -> + n samples from each neighbour
    out = seg.apply(ext, apply_full, apply_valid)     # -> [channels, hi-lo], this rank's share of savgol_apply's output

`apply_full(x)` is the batch filter with the caller's boundary mode on rows x (`Filter.apply_tensor` on the GPU), `apply_valid(x)`
its valid-only form (`Filter.apply_tensor(x, valid=True)`: len - 2n outputs).  What each rank computes
(reference src/savgolFilter.c:743-804; centre :763-766, edges :773-804):
  * a segment with neighbours on both sides: its outputs are centre outputs only -- `apply_valid` of [halo | own | halo];
  * the first / last segment of a POLYNOMIAL, REFLECT or CONSTANT signal: `apply_full` of [own | halo] (resp. [halo | own]) handles
    the real end with the boundary rule -- those outputs only see the 2n+1 (polynomial rows) or 3n (padded modes) samples next to the
    end, so a segment of at least 2n+1 samples has them all -- and the n outputs next to the CUT, computed with a boundary rule that
    does not belong there, are exactly the halo's and are dropped;
  * PERIODIC: the signal is a ring, so the halo exchange wraps around (rank 0 <-> rank W-1) and every rank runs `apply_valid`.
Every output is the same dot product over the same samples as in the unsplit call, so the stitched result equals it bit for bit
for any kernel whose per-output arithmetic does not depend on the tile position (the reference-order kernels and the CPU oracle;
the FMA kernels to fp32 rounding only where block moments are used: their blocks are tile-relative).
"""
import torch
import torch.distributed as dist

from . import shard_range


class LengthSplit:
    def __init__(self, length, half_window, rank=None, world_size=None):
        self.world = dist.get_world_size() if world_size is None else world_size
        self.rank = dist.get_rank() if rank is None else rank
        self.length, self.n = int(length), int(half_window)
        self.lo, self.hi = shard_range(self.length, self.world, self.rank)
        spans = [shard_range(self.length, self.world, r) for r in range(self.world)]
        if self.world > 1 and min(hi - lo for lo, hi in spans) < 2 * self.n + 1:
            raise ValueError(f"segments shorter than the window ({2 * self.n + 1}): use fewer ranks")

    def halos(self, periodic):
        """(left, right): samples received from the previous / next rank"""
        if self.world == 1:
            return 0, 0
        if periodic:
            return self.n, self.n
        return (self.n if self.rank > 0 else 0), (self.n if self.rank < self.world - 1 else 0)

    def exchange(self, local, periodic=False, comm=None):
        """local: [channels, hi-lo] tensor.  Returns [channels, left + own + right].
        comm: an rccl.Comm -- the halos then travel through savgol_lengthsplit_exchange_rccl (csrc/sg_rowband_rccl.hip), enqueued on
        the current stream; without it through torch.distributed's batch_isend_irecv."""
        channels, own = local.shape
        assert own == self.hi - self.lo
        left, right = self.halos(periodic)
        ext = torch.empty((channels, left + own + right), dtype=local.dtype, device=local.device)
        ext[:, left:left + own] = local
        if self.world == 1:
            return ext
        prev, nxt = (self.rank - 1) % self.world, (self.rank + 1) % self.world
        if comm is not None:
            local = local.contiguous()
            halo_l = torch.empty((channels, self.n), dtype=local.dtype, device=local.device) if left else None
            halo_r = torch.empty((channels, self.n), dtype=local.dtype, device=local.device) if right else None
            scratch = torch.empty((2, channels, self.n), dtype=local.dtype, device=local.device)
            comm.lengthsplit_exchange(local, self.n, halo_l, halo_r, scratch, (prev if left else -1, nxt if right else -1))
            if left:
                ext[:, :left] = halo_l
            if right:
                ext[:, left + own:] = halo_r
            return ext
        ops, keep = [], []
        send_l = send_r = None
        if left:                                              # my first n samples go left, the neighbour's last n come from there
            send_l = local[:, :self.n].contiguous()
            recv_l = torch.empty_like(send_l)
            keep.append((recv_l, slice(0, left)))
        if right:
            send_r = local[:, own - self.n:].contiguous()
            recv_r = torch.empty_like(send_r)
            keep.append((recv_r, slice(left + own, left + own + right)))
        # (two ranks on a ring: both halos travel between the same pair -- post the sends and receives in matching order)
        if left:
            ops += [dist.P2POp(dist.isend, send_l, prev), dist.P2POp(dist.irecv, recv_l, prev)]
        if right:
            ops += [dist.P2POp(dist.isend, send_r, nxt), dist.P2POp(dist.irecv, recv_r, nxt)]
        if self.world == 2 and left and right:
            # rank 0 receives its LEFT halo from rank 1's right-going send and vice versa: order the receives by what the peer sends first
            ops = [dist.P2POp(dist.isend, send_l, prev), dist.P2POp(dist.isend, send_r, nxt),
                   dist.P2POp(dist.irecv, recv_r, nxt), dist.P2POp(dist.irecv, recv_l, prev)]
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        for buf, where in keep:
            ext[:, where] = buf
        return ext

    def apply(self, ext, apply_full, apply_valid, periodic=False):
        """ext from exchange(); returns this rank's [channels, hi-lo] outputs."""
        own = self.hi - self.lo
        left, right = self.halos(periodic)
        if left and right:
            out = apply_valid(ext)
            assert out.shape[1] == own
            return out
        full = apply_full(ext)
        return full[:, left:left + own]
