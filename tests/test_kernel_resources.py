"""No kernel a BASELINE config dispatches may carry a private segment (scratch) or spill registers (config 4's tile kernel shipped
with 2 spilled VGPRs and 12 B of scratch -- in its scalar path, which the launch never ran -- and DESIGN.md said otherwise).  Read from the shipped
library's code-object metadata (tools/kernel_resources.py); no GPU needed."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "savitzky-golay-filter_amd", "lib", "libsavgol_hip.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"

# what bench.py's five configs run (kernel names as DESIGN.md section 4 / bench.py's roofline objects give them)
BASELINE_KERNELS = [
    r"sg1d_center_momenth_kernel<32, 5>",                                   # config 2 (the headline)
    r"sg1d_center_kernel<float, 5,",                                        # config 1
    r"sg_bank_dma_kernel<16, true, 32, 8, 16, 1, 2, sg::MomTaps<16, 2>",    # config 3, fused bank
    r"sg_bank_dma_kernel<16, false, 32, 4, 12,",                            # config 3, bit-exact bank
    r"sg_bank_tick_n_kernel<16,",                                           # config 3, per tick
    r"sg2d_rolling_kernel<7, 2, 1, true, false, 20>",                       # config 4
    r"sg1d_center_moment64_kernel<32, 5>",                                  # config 5 (savgol_apply_batch_f64_tol, rel_tol 1e-6)
    r"sg1d_center_kernel<double, 32,",                                      # config 5, 1e-12 path
]


@pytest.fixture(scope="module")
def kernels():
    if not (os.path.exists(LIB) and os.path.exists(READELF)):
        pytest.skip("library or llvm-readelf not present")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py"), LIB], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = []
    for line in out.stdout.splitlines():
        m = re.match(r"vgpr\s+(\d+) sgpr\s+(\d+) scratch\s+(\d+) spill\s+(\d+) lds\s+(\d+)\s+(.*)", line)
        if m:
            rows.append({"vgpr": int(m.group(1)), "scratch": int(m.group(3)), "spill": int(m.group(4)), "name": m.group(6)})
    assert len(rows) > 200, len(rows)
    return rows


def test_baseline_kernels_have_no_private_segment(kernels):
    for pat in BASELINE_KERNELS:
        hit = [k for k in kernels if pat in k["name"]]
        assert hit, f"no kernel named like {pat!r} in the library (renamed? update this list and DESIGN.md)"
        for k in hit:
            assert k["scratch"] == 0 and k["spill"] == 0, k


def test_no_kernel_in_the_library_uses_scratch(kernels):
    bad = [k for k in kernels if k["scratch"] or k["spill"]]
    assert not bad, bad[:10]
