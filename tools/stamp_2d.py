"""Phase stamps of ONE wave of the 2-D rolling kernel, tile form and strip walk ("name what the walk waits on").
Needs a diagnostic build:   tools/build_variant.sh stamps sg_2d_roll_g1.o "-USEP_ROLL_MIN_N -DSEP_ROLL_MIN_N=7 -DSG_STAMPS2D"
   python tools/stamp_2d.py tools/ab/lib_stamps.so
The stamps (s_memtime + a wait each) serialise the wave's own phases: read the SHARES, not the totals; the other waves of the chip run the
normal instruction stream apart from the stamps."""
import ctypes as C, os, sys
import numpy as np, torch
lib = os.path.abspath(sys.argv[1])
os.environ["SAVGOL_HIP_LIB"] = lib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
form = "walk" if os.environ.get("SAVGOL_HIP_ROLL_TILE") == "0" else "tile"
sg = load_package()
L = sg.lib()
L.savgol_hip_debug_set_stamps2d.argtypes = [C.c_void_p]
images, size, n = 256, 4096, 7
x = torch.randn((images, size, size), device="cuda"); y = torch.empty_like(x)
st = torch.zeros(512, dtype=torch.int64, device="cuda")
assert L.savgol_hip_debug_set_stamps2d(st.data_ptr()) == 0
f = sg.Filter2D(n, n, 3)
for _ in range(3):
    f.apply_batch(x, y, size, size, images, boundary=1, method=2)
torch.cuda.synchronize()
allst = st.cpu().numpy().astype(np.int64).reshape(4, 128)
cands = [c for c in range(4) if allst[c, 0] and allst[c, 4]]
assert cands, "none of the four candidate waves was an interior strip"
s = allst[cands[0]]
rows = 16 if form == "tile" else 60
print(f"## {form}: one interior wave, cycles (s_memtime ticks)")
print(f"issue of the first loads            {s[1] - s[0]:8d}")
print(f"wait for rows 0..14 + vertical row 0 {s[2] - s[1]:8d}")
ver, hor = [], []
prev = s[2]
for r in range(rows):
    a, b = s[3 + 2 * r], s[4 + 2 * r]
    if a == 0 or b == 0: break
    ver.append(a - prev); hor.append(b - a); prev = b
print("per row: (wait for the next input row + vertical pass) / (LDS round trip + horizontal pass + store)")
print("  vertical+wait:", " ".join(f"{v:5d}" for v in ver))
print("  horizontal   :", " ".join(f"{v:5d}" for v in hor))
print(f"rows stamped {len(ver)}; median vertical+wait {int(np.median(ver))}, median horizontal {int(np.median(hor))}; wave alive {prev - s[0]} cycles for {len(ver)} rows")
