"""The C halo exchanges (csrc/sg_rowband_rccl.hip -> lib/libsavgol_hip_rccl.so) EXECUTED on one GPU (VERDICT r03 missing #1 / next #4).

RCCL refuses two ranks on one device, but a ONE-rank communicator may send to itself (served as a local copy, matched in posting
order), and a ring of one rank is a well-defined case of both exchanges: the frame / signal is periodic, so the halo above the band is
the band's own LAST rows and the halo below its FIRST rows.  That runs every line of the C functions -- the pack kernel, ncclGroupStart,
ncclSend, ncclRecv, ncclGroupEnd -- with real bytes, and the result feeds savgol2d_apply_rowband_f32 / the 1-D valid kernel, whose output
must then equal the periodic extension filtered by the oracle.  Reference loops served: src/savgol2d.c:417-453, src/savgolFilter.c:763-766."""
import numpy as np
import pytest

from tests._util import normwise

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def comm(sg):
    import importlib
    import torch
    assert torch.cuda.is_available()
    rccl = importlib.import_module("savgol_amd.rccl")
    if not rccl.available():
        pytest.skip("librccl / libsavgol_hip_rccl.so not loadable")
    c = rccl.Comm(1, 0, rccl.unique_id())
    yield c
    c.close()


def test_rowband_exchange_moves_the_boundary_rows(sg, sgo, comm):
    import torch
    images, rows, cols, ny = 5, 40, 300, 7
    stride = cols + 4                                              # a row pitch larger than the row
    store = torch.empty((images, rows + 3, stride), dtype=torch.float32, device="cuda").uniform_(-1, 1)
    band = store[:, :rows, :cols]                                  # image pitch and row stride both differ from the dense layout
    up = torch.full((images, ny, cols), float("nan"), dtype=torch.float32, device="cuda")
    dn = torch.full_like(up, float("nan"))
    scratch = torch.empty((2, images, ny, cols), dtype=torch.float32, device="cuda")
    comm.rowband_exchange(band, ny, up, dn, scratch, peers=(0, 0))
    torch.cuda.synchronize()
    assert torch.equal(scratch[0], band[:, :ny]) and torch.equal(scratch[1], band[:, rows - ny:])       # the pack launches
    assert torch.equal(up, band[:, rows - ny:]) and torch.equal(dn, band[:, :ny])                       # the ring of one
    # no neighbour on one side: that buffer is neither packed nor touched
    up.fill_(float("nan")); dn.fill_(float("nan"))
    comm.rowband_exchange(band, ny, None, dn, scratch, peers=(-1, 0))
    torch.cuda.synchronize()
    assert torch.equal(dn, band[:, rows - ny:])                    # the only message of the pair: my last rows come back as the lower halo
    # the received halos drive the band filter: a vertically periodic frame, compared with the oracle on the periodic extension
    dense = band.contiguous()
    comm.rowband_exchange(dense, ny, up, dn, scratch, peers=(0, 0))
    f2 = sg.Filter2D(ny, ny, 3)
    out = torch.empty_like(dense)
    import ctypes as C
    L = sg.lib()
    rc = L.savgol2d_apply_rowband_f32(f2.ptr, dense.data_ptr(), rows, cols, cols, rows * cols, up.data_ptr(), dn.data_ptr(), cols, ny * cols,
                                      out.data_ptr(), cols, rows * cols, images, 1, 1, None)
    assert rc == 0, sg.last_error()
    torch.cuda.synchronize()
    xh = dense.cpu().numpy()
    o2 = sgo.Filter2D(ny, ny, 3)
    for k in (0, images - 1):
        ext = np.concatenate([xh[k, rows - ny:], xh[k], xh[k, :ny]], axis=0)
        want = o2.apply(ext, cols, 1)[ny:ny + rows]               # CONSTANT at the left / right frame edge, periodic above / below
        assert np.array_equal(out[k].cpu().numpy().view(np.uint32), want.view(np.uint32)), k


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_lengthsplit_exchange_moves_the_samples_next_to_the_cut(sg, sgo, comm, dtype):
    import torch
    tdt = torch.float32 if dtype == "f32" else torch.float64
    channels, own, ld, n = 37, 5000, 5008, 32
    store = torch.empty((channels, ld), dtype=tdt, device="cuda").uniform_(-1, 1)
    seg = store[:, :own]
    prev = torch.full((channels, n), float("nan"), dtype=tdt, device="cuda")
    nxt = torch.full_like(prev, float("nan"))
    scratch = torch.empty((2, channels, n), dtype=tdt, device="cuda")
    comm.lengthsplit_exchange(seg, n, prev, nxt, scratch, (0, 0))
    torch.cuda.synchronize()
    assert torch.equal(prev, seg[:, own - n:]) and torch.equal(nxt, seg[:, :n])                         # PERIODIC on one rank
    # [halo | own | halo] through the valid kernel == the PERIODIC whole-channel call (reference get_padded_sample :465-468)
    ext = torch.cat([prev, seg, nxt], dim=1).contiguous()
    f = sg.Filter(n, 4, 0, 1.0, 2)
    got = f.apply_tensor(ext, valid=True)
    whole = f.apply_tensor(seg.contiguous())
    assert got.shape == whole.shape
    ref = sgo.Filter(n, 4, 0, 1.0, 2).apply_f64(seg.cpu().numpy().astype(np.float64))
    tol = 1e-6 if dtype == "f32" else 1e-12
    assert normwise(got.cpu().numpy(), ref) < tol and normwise(whole.cpu().numpy(), ref) < tol
    # one-sided: only the message towards `next`
    prev.fill_(float("nan")); nxt.fill_(float("nan"))
    comm.lengthsplit_exchange(seg, n, None, nxt, scratch, (-1, 0))
    torch.cuda.synchronize()
    assert torch.equal(nxt, seg[:, own - n:])
