"""A/B timing of library builds on ONE box (devices differ by up to ~12 % on MFMA-dense kernels, so numbers from
different gpurun calls are not comparable).  usage: python scratch/ab.py n d libA.so libB.so ...   (paths relative to scratch/)
Each build runs in its own subprocess, interleaved over `rounds` rounds; prints contract / step ms (median)."""
import os, subprocess, sys, json, statistics
HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import os, sys, torch, json
sys.path.insert(0, os.path.dirname(HERE))
from stein_amd import _lib
_lib.LIB_PATH = LIBPATH
from stein_amd.engine import SvgdEngine
if "noupper" in _lib.LIB_PATH: SvgdEngine._full_distance_image = True   # -DSTEIN_NO_UPPER builds
n, d = N, D
torch.manual_seed(0)
T = torch.randn(n, d, device="cuda"); G = torch.randn(n, d, device="cuda")
eng = SvgdEngine(n, d, device="cuda", x3=True)
for _ in range(3): eng.compute_phi(T, G)
torch.cuda.synchronize()
ev = {}
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); ev.setdefault(name, []).append(e)
res = {}
for _ in range(10):
    ev.clear()
    mark("start"); eng.compute_phi(T, G, mark=mark); mark("end")
    torch.cuda.synchronize()
    names = list(ev.keys())
    for a, b in zip(names[:-1], names[1:]):
        res.setdefault(b, []).append(ev[a][0].elapsed_time(ev[b][0]))
    res.setdefault("total", []).append(ev["start"][0].elapsed_time(ev["end"][0]))
import statistics
print(json.dumps({k: statistics.median(v) for k, v in res.items()}))
'''
def run(lib, n, d):
    code = CHILD.replace("HERE", repr(HERE)).replace("LIBPATH", repr(os.path.join(HERE, lib))).replace("N, D", "%d, %d" % (n, d))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    if out.returncode: raise SystemExit(out.stderr[-2000:])
    return json.loads(out.stdout.strip().splitlines()[-1])
if __name__ == "__main__":
    n, d = int(sys.argv[1]), int(sys.argv[2]); libs = sys.argv[3:]
    acc = {l: [] for l in libs}
    for r in range(3):
        for l in libs: acc[l].append(run(l, n, d))
    for l in libs:
        keys = acc[l][0].keys()
        print(l, {k: round(statistics.median(x[k] for x in acc[l]), 4) for k in keys})
