"""Device-resident strided (array-of-structs) batch call: time and bytes moved for a few record sizes (tools, not product).
   Records of `rec` floats, field 1 filtered in place into field 2 of a second array of the same shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
sg = load_package()
import torch
L = sg.lib()
f = sg.Filter(int(sys.argv[1]) if len(sys.argv) > 1 else 32, 4, 0, 1.0, 0)
for rec, ch, count in ((2, 2048, 1 << 19), (4, 2048, 1 << 18), (8, 1024, 1 << 18), (16, 1024, 1 << 17)):
    src = torch.randn((ch, count, rec), device="cuda"); dst = torch.zeros_like(src)
    run = lambda: L.savgol_apply_strided_batch_f32(f.ptr, src.data_ptr(), rec * 4, 4, count * rec * 4, dst.data_ptr(), rec * 4, 8 if rec > 2 else 4, count * rec * 4, ch, count, None)
    assert run() == 0, sg.last_error(); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[len(ts) // 2]; n = ch * count
    print(f"records of {rec:2d} floats, {ch} x {count}: {ms:7.3f} ms  {n / ms / 1e6:7.1f} Gsamples/s   field bytes {8 * n / ms / 1e6:7.0f} GB/s   whole-record bytes (read in, write out) {2 * rec * 4 * n / ms / 1e6:7.0f} GB/s", flush=True)

# round 4: two fields of the SAME records (field 0 -> field 1): whole-record stores (SAVGOL_HIP_STRIDED_WHOLE_RECORDS=0: 4-byte stores)
for rec, ch, count in ((2, 2048, 1 << 19), (4, 2048, 1 << 18)):
    aos = torch.randn((ch, count, rec), device="cuda")
    run = lambda: L.savgol_apply_strided_batch_f32(f.ptr, aos.data_ptr(), rec * 4, 0, count * rec * 4, aos.data_ptr(), rec * 4, 4, count * rec * 4, ch, count, None)
    assert run() == 0, sg.last_error(); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[len(ts) // 2]; n = ch * count
    print(f"same array, records of {rec:2d} floats, field 0 -> field 1, {ch} x {count}: {ms:7.3f} ms  {n / ms / 1e6:7.1f} Gsamples/s   record bytes (read + write) {2 * 4 * rec * n / ms / 1e6:7.0f} GB/s")
