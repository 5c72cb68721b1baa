"""The host-side code of the library (weight tables, the wide-window moment fit, the export format) under AddressSanitizer +
UndefinedBehaviorSanitizer, on the CPU build (GPU sanitizers are not available on this pool).  tools/sanitize_host.cpp walks
every valid 1-D configuration (n <= 32, m <= 10, d <= 4) and a grid of 2-D ones."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "savitzky-golay-filter_amd", "csrc")


@pytest.mark.skipif(shutil.which("gcc") is None or shutil.which("g++") is None, reason="needs gcc/g++")
def test_host_code_is_clean_under_asan_and_ubsan(tmp_path):
    san = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__"]
    wobj = str(tmp_path / "w.o")
    r = subprocess.run(["gcc", "-std=gnu11", "-ffp-contract=off", *san, *inc, "-c", os.path.join(CSRC, "sg_weights.c"), "-o", wobj],
                       capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("no sanitizer runtime in this image")
    assert r.returncode == 0, r.stderr
    exe = str(tmp_path / "san")
    r = subprocess.run(["g++", "-std=c++17", *san, *inc, os.path.join(ROOT, "tools", "sanitize_host.cpp"),
                        os.path.join(CSRC, "sg_k1d_moment_fit.cpp"), os.path.join(CSRC, "sg_export.cpp"), wobj, "-o", exe, "-lm"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "sanitizer pass" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
