"""Row-band split of the 2-D path across GPUs, with a nearest-neighbour halo exchange.

Images are independent, so the default multi-GPU layout is "each rank filters its own images" and needs no
communication.  When a frame stack must instead be split *inside* each frame (frames too large for one GPU,
or fewer frames than GPUs), every rank owns a horizontal band of rows and needs `ny` rows from the bands
above and below before it can filter: that is the only exchange step anywhere on the hot path
(SURVEY.md section 8e).  It is point-to-point -- one `isend`/`irecv` pair per neighbour, batched for the whole
stack -- so on MI355X it rides one xGMI link per neighbour pair (RCCL, `backend="nccl"`); there is no
all-reduce.

    band = RowBand(rows, ny, rank, world)            # which rows this rank owns
    out  = band.apply_overlapped(local, apply_fn)    # halo exchange in flight while the band itself is filtered
  or, in two steps (one extra copy of the band, nothing overlapped):
    ext  = band.exchange(local)                      # local [images, own_rows, cols] -> + halo rows from neighbours
    out  = band.apply(ext, apply_fn)                 # apply_fn(frames[images, R, cols]) -> same shape; returns own rows

`apply_fn` is the 2-D batch filter with the caller's boundary mode (`Filter2D.apply_batch` on the GPU).  The band
buffer is filtered as if it were a frame: where its edge is the real frame edge the boundary mode applies there,
as the reference does (src/savgol2d.c:428-445); where the edge is artificial the rows it taints are exactly the
halo rows, which are dropped.  VALID leaves the frame's own top/bottom `ny` rows untouched, like the reference.
"""
import torch
import torch.distributed as dist

from . import shard_range


class RowBand:
    def __init__(self, rows, ny, rank=None, world_size=None):
        self.world = dist.get_world_size() if world_size is None else world_size
        self.rank = dist.get_rank() if rank is None else rank
        self.rows, self.ny = rows, ny
        self.lo, self.hi = shard_range(rows, self.world, self.rank)
        spans = [shard_range(rows, self.world, r) for r in range(self.world)]
        if self.world > 1 and min(hi - lo for lo, hi in spans) < ny:
            raise ValueError(f"row bands thinner than the half window ({ny}): use fewer ranks")
        # apply_overlapped / apply_c rebuild an artificial edge from the band's own 2*ny rows next to it
        self.thin = self.world > 1 and min(hi - lo for lo, hi in spans) < 2 * ny
        self.top = ny if self.rank > 0 else 0                 # halo rows received from above / below
        self.bottom = ny if self.rank < self.world - 1 else 0

    def exchange(self, local):
        """local: [images, hi-lo, cols] tensor of this rank's rows.  Returns [images, top + own + bottom, cols]."""
        images, own, cols = local.shape
        assert own == self.hi - self.lo
        ext = torch.empty((images, self.top + own + self.bottom, cols), dtype=local.dtype, device=local.device)
        ext[:, self.top:self.top + own] = local
        if self.world == 1:
            return ext
        ops, keep = [], []
        if self.top:                                          # my first ny rows go up, the neighbour's last ny come down
            send_up = local[:, :self.ny].contiguous()
            recv_up = torch.empty_like(send_up)
            ops += [dist.P2POp(dist.isend, send_up, self.rank - 1), dist.P2POp(dist.irecv, recv_up, self.rank - 1)]
            keep.append((recv_up, slice(0, self.top)))
        if self.bottom:
            send_dn = local[:, own - self.ny:].contiguous()
            recv_dn = torch.empty_like(send_dn)
            ops += [dist.P2POp(dist.isend, send_dn, self.rank + 1), dist.P2POp(dist.irecv, recv_dn, self.rank + 1)]
            keep.append((recv_dn, slice(self.top + own, self.top + own + self.bottom)))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        for buf, rows in keep:
            ext[:, rows] = buf
        return ext

    def apply(self, ext, apply_fn):
        """Filter the extended band and return this rank's own rows [images, hi-lo, cols]."""
        out = apply_fn(ext)
        return out[:, self.top:self.top + (self.hi - self.lo)]

    # ---- the same result with the exchange hidden behind the interior ----
    def start_exchange(self, local):
        """Post the halo sends/receives (non-blocking) and return a handle for finish_exchange()."""
        images, own, cols = local.shape
        assert own == self.hi - self.lo
        ops, bufs = [], {}
        if self.top:
            send_up = local[:, :self.ny].contiguous()
            bufs["up"] = torch.empty_like(send_up)
            ops += [dist.P2POp(dist.isend, send_up, self.rank - 1), dist.P2POp(dist.irecv, bufs["up"], self.rank - 1)]
            bufs["_keep_up"] = send_up
        if self.bottom:
            send_dn = local[:, own - self.ny:].contiguous()
            bufs["dn"] = torch.empty_like(send_dn)
            ops += [dist.P2POp(dist.isend, send_dn, self.rank + 1), dist.P2POp(dist.irecv, bufs["dn"], self.rank + 1)]
            bufs["_keep_dn"] = send_dn
        reqs = dist.batch_isend_irecv(ops) if ops else []
        return reqs, bufs

    @staticmethod
    def finish_exchange(handle):
        reqs, bufs = handle
        for r in reqs:
            r.wait()
        return bufs.get("up"), bufs.get("dn")

    def apply_c(self, filter2d, local, boundary=1, method=0, stream=None, comm=None):
        """The same through the C ABI (csrc/sg_2d_rowband.hip), exchange overlapped with the band: post the halo sends / receives,
        enqueue the band itself (savgol2d_apply_batch_f32: it reads no halo), then the edge strips: with the C exchange they are gathered and
        filtered on the exchange's stream, beside the band, and only the copy of their finished rows waits for the band
        (savgol2d_apply_rowband_edges_streams_f32); with torch.distributed's exchange, after it (savgol2d_apply_rowband_edges_f32).
        `local`: [images, own, cols] fp32 on the GPU.
        comm: an rccl.Comm -- the halos then travel through savgol2d_rowband_exchange_rccl (csrc/sg_rowband_rccl.hip: one pack launch and
        one ncclSend / ncclRecv pair per neighbour) on a side stream of its own, no torch.distributed call on the data path; without it
        the exchange is torch.distributed's batch_isend_irecv."""
        from . import lib, last_error, _addr, _stream
        images, own, cols = local.shape
        if self.thin:
            raise ValueError(f"row bands thinner than 2 x half window ({2 * self.ny}): use fewer ranks")
        local = local.contiguous()
        out = torch.empty_like(local)
        L = lib()
        up = dn = None
        done = None
        if comm is not None and self.world > 1:
            cur = stream or torch.cuda.current_stream()
            if getattr(self, "_xstream", None) is None:
                self._xstream = torch.cuda.Stream()
            up = torch.empty((images, self.ny, cols), dtype=local.dtype, device=local.device) if self.top else None
            dn = torch.empty((images, self.ny, cols), dtype=local.dtype, device=local.device) if self.bottom else None
            scratch = torch.empty((2, images, self.ny, cols), dtype=local.dtype, device=local.device)
            # up / dn / scratch come from torch's caching allocator on ITS current stream, but are written on the exchange stream and read
            # on `cur`: tell the allocator, or it may hand the blocks to someone else while the pack kernel, ncclSend / ncclRecv or the edge
            # strips still use them
            for buf in (up, dn, scratch, local, out):
                if buf is not None:
                    buf.record_stream(self._xstream)
                    buf.record_stream(cur)
            ready = torch.cuda.Event()
            ready.record(cur)
            self._xstream.wait_event(ready)                   # the band's rows are there before they are packed
            comm.rowband_exchange(local, self.ny, up, dn, scratch, stream=self._xstream)
            done = torch.cuda.Event()
            done.record(self._xstream)
            handle = None
        else:
            handle = self.start_exchange(local)
        if not (boundary == 0 and own - 2 * self.ny <= 0):
            # With the C exchange the band goes out as TWO launches, a short head (1 / 16 of the frames) and the rest: RCCL's send / recv kernel finds no
            # registers beside a band launch that fills every wave slot and would sit until it has left the chip; in the gap between the two launches it
            # becomes resident, and then runs -- and finishes -- beside the second one (rocprofv3 timeline: profiles/r06_rowband_timeline.txt, R6.11)
            head = max(1, images // 16) if (done is not None and images >= 8) else 0
            for first, count in ((0, head), (head, images - head)):
                if count <= 0:
                    continue
                if L.savgol2d_apply_batch_f32(filter2d.ptr, _addr(local[first:]), own, cols, cols, own * cols, _addr(out[first:]), cols, own * cols, count, boundary,
                                              method, _stream(stream)) != 0:
                    raise RuntimeError(last_error())
        if done is not None:
            rc = L.savgol2d_apply_rowband_edges_streams_f32(filter2d.ptr, _addr(local), own, cols, cols, own * cols,
                                                            _addr(up) if up is not None else None, _addr(dn) if dn is not None else None,
                                                            cols, self.ny * cols, _addr(out), cols, own * cols, images, boundary, method,
                                                            self._xstream.cuda_stream, _stream(stream))
        else:
            up, dn = self.finish_exchange(handle)
            rc = L.savgol2d_apply_rowband_edges_f32(filter2d.ptr, _addr(local), own, cols, cols, own * cols,
                                                    _addr(up) if up is not None else None, _addr(dn) if dn is not None else None,
                                                    cols, self.ny * cols, _addr(out), cols, own * cols, images, boundary, method, _stream(stream))
        if rc != 0:
            raise RuntimeError(last_error())
        return out

    def apply_overlapped(self, local, apply_fn):
        """Filter this rank's band with the halo exchange in flight.

        1. the `ny` boundary rows go out / come in (point to point, non-blocking);
        2. meanwhile the whole local band is filtered as if it were a frame: every output row except the `ny` next to an
           artificial edge is already final (a row r needs input rows r-ny..r+ny, all local);
        3. when the halos have landed, each artificial edge is redone on a 3*ny-row strip (halo + the band's first / last
           2*ny rows): its middle `ny` output rows see only real data and replace the tainted ones.
        No extended copy of the band is made; the strips are 3*ny/own of the work (0.5 % for 4096-row bands, ny = 7)."""
        images, own, cols = local.shape
        ny = self.ny
        if self.thin:
            # bands of ny .. 2*ny - 1 rows: the 3*ny-row strips would reach past the band -- the simple form handles them
            return self.apply(self.exchange(local), apply_fn)
        handle = self.start_exchange(local)
        out = apply_fn(local)
        up, dn = self.finish_exchange(handle)
        if up is not None:
            strip = torch.cat([up, local[:, :2 * ny]], dim=1)
            out[:, :ny] = apply_fn(strip)[:, ny:2 * ny]
        if dn is not None:
            strip = torch.cat([local[:, own - 2 * ny:], dn], dim=1)
            out[:, own - ny:] = apply_fn(strip)[:, ny:2 * ny]
        return out
