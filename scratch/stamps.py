"""Per-phase cycles of the split contraction kernel (k_phi_x3fs) and the clock the chip holds inside it.
-DSTEIN_STAMPS build (scratch/build_variant.py stamps -DSTEIN_STAMPS); usage: STAMPLIB=lib_stamps.so python scratch/stamps.py n d
The clock is read after ~2 s of back-to-back steps on random data (MI355X_MICROARCH.md, DVFS give-back item 6)."""
import ctypes, os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("STAMPLIB", "lib_stamps.so"))
from stein_amd.engine import SvgdEngine
if "noupper" in _lib.LIB_PATH: SvgdEngine._full_distance_image = True   # -DSTEIN_NO_UPPER builds
lib = _lib.load()
lib.stein_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
lib.stein_debug_clock.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
n, d = int(sys.argv[1]), int(sys.argv[2])
T = torch.randn(n, d, device="cuda"); G = torch.randn(n, d, device="cuda")
eng = SvgdEngine(n, d, device="cuda", x3=True)
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(50): eng.compute_phi(T, G)
    torch.cuda.synchronize()
lib.stein_debug_waves.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
buf = (ctypes.c_uint64 * 8)(); clk = (ctypes.c_uint64 * 2)(); wv = (ctypes.c_uint64 * 24)()
acc = np.zeros(8); cacc = np.zeros(2); wacc = np.zeros(24)
def mark(label):        # the staged calls: reset the counters right before the contraction, read them right after
    if label == "contract":
        torch.cuda.synchronize(); lib.stein_debug_stamps(buf, 1); lib.stein_debug_clock(clk, 1); lib.stein_debug_waves(wv, 1)
    elif label == "finish":
        torch.cuda.synchronize(); lib.stein_debug_stamps(buf, 0); lib.stein_debug_clock(clk, 0); lib.stein_debug_waves(wv, 0)
        acc[:] += np.array(list(buf), dtype=np.float64); cacc[:] += np.array(list(clk), dtype=np.float64)
        wacc[:] += np.array(list(wv), dtype=np.float64)
reps = 8
lib.stein_debug_wg.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
wgbuf = (ctypes.c_uint64 * 3072)()
kev = []
def mark2(label):
    mark(label)
    if label in ("contract", "finish"):
        e = torch.cuda.Event(enable_timing=True); e.record(); kev.append(e)
kms = 0.0
for _ in range(reps):
    for _ in range(30): eng.compute_phi(T, G)          # keep the chip loaded between the measured launches
    kev.clear()
    eng.compute_phi(T, G, mark=mark2)
    torch.cuda.synchronize()
    kms += kev[0].elapsed_time(kev[1])
    lib.stein_debug_wg(wgbuf, 0)
torch.cuda.synchronize()
print("contraction launch, event-timed (this stamped build): %.4f ms" % (kms / reps))
v = acc; clk = cacc
nb = v[7]
jt = (n + 31) // 32
names = ["P:exp/split/LDS", "P:wait D loads", "P:barrier", "C4:mma", "C4:barrier", "C8:total", "C8:bwait"]
print("workgroups", nb / reps, "split", eng.split, "k tiles per workgroup", jt / eng.split)
for k, nm in enumerate(names):
    print("%-17s %8.1f shader cycles per k tile" % (nm, v[k] / nb / (jt / eng.split)))
if clk[1]:
    print("main loop of a matrix wave (first stage barrier .. last): %.4f ms real time per workgroup" % (clk[1] / nb * 1e-5))
    print("in-kernel clock of the contraction (delta s_memtime / delta s_memrealtime x 100 MHz): %.3f GHz" % (clk[0] / clk[1] * 0.1))
per = nb * (jt / eng.split)
print("per wave (cycles per k tile): work | at the stage barrier")
for w in range(12):
    print("  wave %2d (%s): %7.1f | %7.1f" % (w, "producer" if w < 4 else "matrix", wacc[2 * w] / per, wacc[2 * w + 1] / per))
w = np.array(list(wgbuf), dtype=np.float64).reshape(1024, 3)
w = w[w[:, 2] > 0]
t0 = w[:, 0].min()
ent, beg, end = (w[:, 0] - t0) * 1e-2, (w[:, 1] - t0) * 1e-2, (w[:, 2] - t0) * 1e-2      # microseconds
print("last launch, %d workgroups (us from the first workgroup's entry): entry %.1f..%.1f  main loop starts %.1f..%.1f (mean %.1f)  ends %.1f..%.1f (mean %.1f)"
      % (len(w), ent.min(), ent.max(), beg.min(), beg.max(), beg.mean(), end.min(), end.max(), end.mean()))
dur = end - beg
print("main-loop duration per workgroup: min %.1f  mean %.1f  max %.1f us; percentiles 10/50/90/99: %s" % (dur.min(), dur.mean(), dur.max(), np.percentile(dur, [10, 50, 90, 99]).round(1)))
xcd = np.arange(len(w)) % 8
for x in range(8): print("  XCD %d: mean main loop %.1f us, mean end %.1f" % (x, dur[xcd == x].mean(), end[xcd == x].mean()))
