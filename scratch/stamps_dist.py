"""Phase cycles of k_distance_x3 (wave 0 of every workgroup; -DSTEIN_STAMPS build)."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("STAMPLIB", "libsteinhip_stamps.so"))
from stein_amd.engine import SvgdEngine
lib = _lib.load()
n, d = int(sys.argv[1]), int(sys.argv[2])
T = torch.randn(n, d, device="cuda"); G = torch.randn(n, d, device="cuda")
eng = SvgdEngine(n, d, device="cuda", x3=True)
st = eng.stages
lib.stein_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
buf = (ctypes.c_uint64 * 8)()
st.rownorms(T, n, d, eng.rownorm); st.x3_prepare(T, G, n, d, eng.planes)
for rep in range(2):
    st.median_begin(eng.hist, eng.select_state, n * n)
    torch.cuda.synchronize(); lib.stein_debug_stamps(buf, 1)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    st.distance_block(T, eng.rownorm, n, d, 0, n, eng.dist, eng.ld_dist, hist0=eng.hist[0], symmetric=True, planes=eng.planes)
    e1.record(); torch.cuda.synchronize(); lib.stein_debug_stamps(buf, 0)
v = np.array(list(buf), dtype=np.float64); nb = v[7]
print("kernel ms", e0.elapsed_time(e1), "workgroups", nb)
for k, nm in enumerate(["wait loads + LDS stores", "barriers", "issue + frags + MFMA", "epilogue"]):
    print("%-26s %9.0f cycles per workgroup" % (nm, v[k] / nb))
print("sum %9.0f" % (v[:4].sum() / nb))
