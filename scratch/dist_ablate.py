"""Distance-pass ablations on ONE box: kernel time of the symmetric split-path distance pass ("plain": no histogram, no
window) for the shipped library and the diagnostic builds of scratch/build_variant.py, interleaved over rounds.
usage: python scratch/dist_ablate.py n d [lib names ...]   (names as in scratch/lib_<name>.so; 'base' = shipped)"""
import os, subprocess, sys, json, statistics
HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import os, sys, json, numpy as np, torch
sys.path.insert(0, os.path.dirname(HERE))
from stein_amd import _lib
if LIBPATH: _lib.LIB_PATH = LIBPATH
from stein_amd.engine import SvgdEngine
n, d = N, D
torch.manual_seed(0)
T = torch.randn(n, d, device="cuda"); G = torch.randn(n, d, device="cuda")
eng = SvgdEngine(n, d, device="cuda", x3=True, small=False)
st = eng.stages
st.rownorms(T, n, d, eng.rownorm); st.x3_prepare(T, G, n, d, eng.planes)
out = {}
for sym in (True, False):
    ts = []
    for rep in range(14):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        st.distance_block(T, eng.rownorm, n, d, 0, n, eng.dist, eng.ld_dist, hist0=None, symmetric=sym, planes=eng.planes)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    out["sym" if sym else "full"] = float(np.median(ts[3:]))
print(json.dumps(out))
'''
def run(lib, n, d):
    path = "" if lib == "base" else os.path.join(HERE, "lib_%s.so" % lib)
    code = CHILD.replace("HERE", repr(HERE)).replace("LIBPATH", repr(path)).replace("N, D", "%d, %d" % (n, d))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    if out.returncode: raise SystemExit(out.stderr[-2000:])
    return json.loads(out.stdout.strip().splitlines()[-1])
if __name__ == "__main__":
    n, d = int(sys.argv[1]), int(sys.argv[2]); libs = sys.argv[3:] or ["base"]
    acc = {l: [] for l in libs}
    for r in range(2):
        for l in libs: acc[l].append(run(l, n, d))
    for l in libs:
        print("%-12s" % l, {k: round(statistics.median(x[k] for x in acc[l]), 4) for k in acc[l][0]}, flush=True)
