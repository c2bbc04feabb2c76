"""Phase stamps of k_spec_select (s_memtime, thread 0): scratch/build_variant.py selstamps -DSTEIN_SEL_STAMPS; usage: sel_stamps.py n d [bf16]"""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib_selstamps.so")
from stein_amd.engine import SvgdEngine
from stein_amd.optimizers import AdagradGradientDescent
lib = _lib.load()
lib.stein_debug_sel_stamps.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
n, d = int(sys.argv[1]), int(sys.argv[2]); dt = torch.bfloat16 if len(sys.argv) > 3 else torch.float32
torch.manual_seed(0)
T = torch.randn(n, d, device="cuda"); G = torch.randn(n, d, device="cuda").to(dt)
eng = SvgdEngine(n, d, device="cuda", dtype=dt); gd = AdagradGradientDescent(learning_rate=1e-3)
acc = []
for step in range(30):
    phi = eng.compute_phi(T.to(dt), G); gd.apply_(T, phi, eng.sqnorm)
    torch.cuda.synchronize()
    buf = (ctypes.c_uint64 * 16)(); lib.stein_debug_sel_stamps(buf)
    a = np.array(list(buf)[:8], dtype=np.int64)
    if step >= 5: acc.append(np.diff(a))
m = np.median(np.array(acc), axis=0)
print("n=%d d=%d %s: median cycles between stamps (100 MHz ticks x ?): state loads %d | slots + barriers %d | pass 1 %d | locate x2 %d | pass 2 %d | locate x2 %d | result + predictor %d | total %d"
      % ((n, d, dt) + tuple(int(x) for x in m) + (int(m.sum()),)))
