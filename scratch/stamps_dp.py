"""Phase cycles of k_distance_panel per wave and strip: -DSTEIN_STAMPS build (scratch/build_variant.py stamps -DSTEIN_STAMPS).
usage: STAMPLIB=lib_stamps.so python scratch/stamps_dp.py n d [world]   (world > 1: rank 0's row block, no window)"""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("STAMPLIB", "lib_stamps.so"))
from stein_amd.engine import SvgdEngine, HipStages
lib = _lib.load()
lib.stein_debug_dp.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
lib.stein_debug_dp_wg.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
n, d = int(sys.argv[1]), int(sys.argv[2]); world = int(sys.argv[3]) if len(sys.argv) > 3 else 1   # 0: the whole matrix through the staged call (SYM=1: symmetric)
buf = (ctypes.c_uint64 * 12)()
dev = "cuda"
def report(tag, ms):
    a = np.array(list(buf), dtype=np.float64)
    strips, waves = a[5], a[6]
    tot = a[:5].sum()
    print("%s: %.3f ms  strips %d waves %d | cycles per strip and wave: wait %6.0f  reads+MFMA %6.0f  requests %5.0f  epilogue %6.0f  switches %5.0f  total %6.0f | per wave %.0f cycles"
          % (tag, ms, strips, waves, a[0] / strips, a[1] / strips, a[2] / strips, a[3] / strips, a[4] / strips, tot / strips, tot / waves), flush=True)
    wg = (ctypes.c_uint64 * 1536)()
    lib.stein_debug_dp_wg(wg)
    w = np.array(list(wg), dtype=np.float64).reshape(256, 6)
    t0 = w[:, 0].min()
    life = (w[:, 1] - w[:, 0]) * 0.01
    print("   last launch, by logical workgroup id (16 per line): own segments done > end (us) / wave 0's own + stolen strips")
    for k in range(0, 256, 16):
        print("   %3d: " % k + " ".join("%4.0f>%-4.0f/%2d+%-2d" % ((w[j, 4] - t0) * 0.01, (w[j, 1] - t0) * 0.01, w[j, 5], w[j, 2] - w[j, 5]) for j in range(k, k + 16)))
    clk = w[:, 3] / np.maximum(w[:, 1] - w[:, 0], 1) * 0.1
    print("   per XCD (32 logical ids each): mean end %s us | mean clock of wave 0 %s GHz" %
          ([int(round(((w[32 * x:32 * x + 32, 1] - t0) * 0.01).mean())) for x in range(8)], [round(float(clk[32 * x:32 * x + 32].mean()), 3) for x in range(8)]))
    lib.stein_debug_dp_slow.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
    sl = (ctypes.c_uint64 * (8 * 2048))()
    lib.stein_debug_dp_slow(sl)
    S = np.array(list(sl), dtype=np.float64).reshape(2048, 8)
    order = np.argsort(-S[:, 0])[:24]
    print("   slowest strips (one per wave): wg.wave  cycles total = wait + mfma + requests + epilogue | segment strip | at us")
    for o in order:
        print("     %3d.%d  %8.0f = %7.0f + %7.0f + %7.0f + %7.0f | seg %3d strip %3d | %5.0f" % (o // 8, o % 8, S[o, 0], S[o, 1], S[o, 2], S[o, 3], S[o, 4], S[o, 5], S[o, 6], (S[o, 7] - t0) * 0.01))
    print("   median over waves of the slowest strip: %.0f cycles" % np.median(S[S[:, 0] > 0, 0]))
    print("   diagonal strips: %d, mean epilogue %.0f cycles" % (a[10], a[9] / max(a[10], 1)))
    print("   in-kernel clock %.3f GHz; mean wave lifetime %.1f us" % (a[7] / a[8] * 0.1, a[8] / waves * 0.01), flush=True)
if world == 1:
    T = torch.randn(n, d, device=dev); G = torch.randn(n, d, device=dev)
    eng = SvgdEngine(n, d, device=dev)
    for _ in range(8):
        eng.compute_phi(T, G); T = T + 1e-4 * eng.phi
    torch.cuda.synchronize(); lib.stein_debug_dp(buf, 1)
    _lib.timing_reserve(5)
    for _ in range(5):
        eng.compute_phi(T, G, timing=True); T = T + 1e-4 * eng.phi
    torch.cuda.synchronize(); lib.stein_debug_dp(buf, 0)
    per = _lib.timing_read(5)
    report("fused n=%d d=%d (5 launches, window %s)" % (n, d, eng.window_stats()), sum(c["distance"] for c in per) / 5)
else:
    nl = n // max(world, 1)
    st = HipStages()
    total, offs, extra = st.workspace_layout(nl, n, d, _lib.FLAG_X3)
    ws = torch.empty(total, dtype=torch.uint8, device=dev)
    ld = extra[_lib.WSX_LD_DIST]
    r = ws[offs[_lib.WS_ROWNORM]:offs[_lib.WS_ROWNORM] + n * 4].view(torch.float32)
    D = ws[offs[_lib.WS_DIST]:offs[_lib.WS_DIST] + nl * ld * 4].view(torch.float32).view(nl, ld)
    planes = ws[offs[_lib.WS_PLANES]:total]
    T = torch.randn(n, d, device=dev); G = torch.randn(n, d, device=dev)
    st.rownorms(T, n, d, r); st.x3_prepare(T, G, n, d, planes)
    sym = os.environ.get("SYM", "0") == "1"
    for rep in range(3):
        torch.cuda.synchronize(); lib.stein_debug_dp(buf, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); st.distance_block(T, r, n, d, 0, nl, D, ld, planes=planes, kernel=_lib.STAGE_PANEL, symmetric=sym); e1.record()
        torch.cuda.synchronize(); lib.stein_debug_dp(buf, 0)
    report("row block %d x %d d=%d sym=%s (plain epilogue)" % (nl, n, d, sym), e0.elapsed_time(e1))
