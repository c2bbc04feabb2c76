"""Phase cycles of the one-kernel path; needs a -DSTEIN_STAMPS build of the library at scratch/libsteinhip_stamps.so
(hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSTEIN_STAMPS -Iinclude stein_amd/csrc/*.hip -o scratch/libsteinhip_stamps.so)."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsteinhip_stamps.so")
from stein_amd.engine import SvgdEngine
lib = _lib.load()
names = ["distances + norms + level 0", "median level 0 (locate)", "median level 1", "median level 2", "bandwidth", "K + rowsum", "stage + phi"]
for n, d in ((20, 10), (100, 10), (128, 1), (128, 128)):
    T = torch.randn(n, d, device="cuda"); G = torch.randn(n, d, device="cuda")
    eng = SvgdEngine(n, d, device="cuda")
    for _ in range(3): eng.compute_phi(T, G)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    lib.stein_debug_small(buf)
    v = list(buf)[:7]
    print("n=%d d=%d total %d cycles: " % (n, d, sum(v)) + ", ".join("%s %d" % (a, b) for a, b in zip(names, v)))
