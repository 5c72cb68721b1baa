"""Disassembly of ONE gfx950 kernel inside a host object or shared library:
   python tools/kernel_isa.py savitzky-golay-filter_amd/build/sg_2d_roll_g1.o 'sg2d_rolling_kernel<7, 2, 1, true, false, 20>' > /tmp/k.s
(the demangled name must contain every given substring)."""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_resources import code_objects

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def main():
    path, filters = sys.argv[1], sys.argv[2:]
    for co in code_objects(open(path, "rb").read()):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(co)
        txt = subprocess.run([OBJDUMP, "-d", "--demangle", f.name], capture_output=True, text=True).stdout
        os.unlink(f.name)
        for block in re.split(r"\n(?=[0-9a-f]{16} <)", txt):
            head = block.split("\n", 1)[0]
            if all(s in head for s in filters) and ">:" in head:
                print(block)


if __name__ == "__main__":
    main()
