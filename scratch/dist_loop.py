"""A plain loop of staged distance passes (panel kernel, symmetric C3 block, no window, no histogram) under a given library
build, for rocprofv3: python3 scratch/dist_loop.py <lib|shipped> [reps] [waves|0] [window]
"window": the pass runs as the fused step's does on a hit -- with a median window in SpecState (written by hand around the median
of a sample of pairs; the epilogue counts the entries below it and queues those inside it), which is the specialised epilogue."""
import os, sys, torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from stein_amd import _lib
if sys.argv[1] != "shipped":
    _lib.LIB_PATH = os.path.join(HERE, sys.argv[1])
from stein_amd.engine import SvgdEngine
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
if len(sys.argv) > 3 and int(sys.argv[3]): _lib.call("stein_debug_dpanel_waves", int(sys.argv[3]))
WINDOW = len(sys.argv) > 4 and sys.argv[4] == "window"
n, d = 16384, 256
torch.manual_seed(0)
T = torch.randn(n, d, device="cuda"); G = torch.randn(n, d, device="cuda")
eng = SvgdEngine(n, d, device="cuda", small=False)
st = eng.stages
st.rownorms(T, n, d, eng.rownorm); st.x3_prepare(T, G, n, d, eng.planes)
ev = []
if WINDOW:
    i, j = torch.randint(0, n, (2, 20000), device="cuda")
    med = ((T[i] - T[j]) ** 2).sum(1).median().item()
    key = lambda x: int(torch.tensor(x, dtype=torch.float32).view(torch.int32).item())
    lo, hi = key(med * (1 - 5e-6)), key(med * (1 + 5e-6))     # a few thousand entries inside, as in a real step
    sel, spec, hist = eng.select_state, eng.spec_section, eng.hist
    base = _lib.SPEC_HIT_OFFSET - 28                       # SpecState: lo_key at +12, width at +16 (stein_common.h)
    win = torch.tensor([lo, hi - lo], dtype=torch.int32, device="cuda").view(torch.uint8)
    print("window: median of a sample %.3f, keys [%#x, +%d]" % (med, lo, hi - lo))
for _ in range(reps):
    if WINDOW:
        st.spec_begin(hist, sel, spec, n * n)
        sel[base + 12:base + 20].copy_(win)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    if WINDOW:
        st.distance_block_spec(T, eng.rownorm, n, d, 0, n, eng.dist, eng.ld_dist, hist[0], sel, spec, planes=eng.planes,
                               kernel=_lib.STAGE_PANEL, symmetric=True)
    else:
        st.distance_block(T, eng.rownorm, n, d, 0, n, eng.dist, eng.ld_dist, symmetric=True, planes=eng.planes, kernel=_lib.STAGE_PANEL)
    b.record(); ev.append((a, b))
torch.cuda.synchronize()
ts = sorted(a.elapsed_time(b) for a, b in ev[2:])
print("distance pass alone (%s%s): median %.4f ms, min %.4f" % (sys.argv[1], ", window" if WINDOW else "", ts[len(ts) // 2], ts[0]))
if WINDOW:
    sp = sel[base:base + 64].view(torch.int32).cpu()
    print("   SpecState after the last pass: count %d overflow %d" % (int(sp[5]), int(sp[6])))
