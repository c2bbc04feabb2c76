import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/stein_amd") else os.getcwd())
from stein_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("STAMPLIB", "libsteinhip_stamps.so"))
from stein_amd.engine import SvgdEngine
lib = _lib.load()
n, d = int(sys.argv[1]), int(sys.argv[2])
T = torch.randn(n, d, device="cuda"); G = torch.randn(n, d, device="cuda")
eng = SvgdEngine(n, d, device="cuda", x3=True)
eng.compute_phi(T, G); torch.cuda.synchronize()
buf = (ctypes.c_uint64 * 8)()
lib.stein_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
def mark(label):        # the staged calls: reset the counters right before the contraction, read them right after
    if label == "contract":
        torch.cuda.synchronize(); lib.stein_debug_stamps(buf, 1)
    elif label == "finish":
        torch.cuda.synchronize(); lib.stein_debug_stamps(buf, 0)
eng.compute_phi(T, G, mark=mark); torch.cuda.synchronize()
v = np.array(list(buf), dtype=np.float64)
nb = v[7]
jt = (n + 31) // 32
names = ["P:produce", "P:issue", "P:barrier", "C4:mma", "C4:barrier", "C8:total", "C8:bwait"]
print("blocks", nb, "split", eng.split, "ktiles/block", jt / eng.split)
for k, nm in enumerate(names):
    print("%-15s %8.1f ticks per k-tile (100 MHz ticks? s_memtime = shader clock)" % (nm, v[k] / nb / (jt / eng.split)))

