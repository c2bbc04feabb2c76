"""D of the fused call is bitwise symmetric and equals the row-block (non-symmetric) distance pass where that is exact.
usage: python scratch/check_sym.py [lib_<name>.so]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd import _lib
if len(sys.argv) > 1 and sys.argv[1]:
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), sys.argv[1])
from stein_amd.engine import SvgdEngine
ok = True
for n, d in [(1000, 130), (1536, 256), (4096, 128), (2500, 77)]:
    g = torch.Generator().manual_seed(n)
    T = torch.randn(n, d, generator=g).cuda(); G = torch.randn(n, d, generator=g).cuda()
    eng = SvgdEngine(n, d, device="cuda", small=False)
    for _ in range(4): eng.compute_phi(T, G)
    D = eng.dist_matrix()
    T64 = T.double(); r = (T64 * T64).sum(1); ref = r[:, None] + r[None, :] - 2 * T64 @ T64.T
    sym = bool(torch.equal(D, D.T)); err = float((D.double() - ref).abs().max() / ref.abs().max())
    print((n, d), "symmetric", sym, "max err / max", "%.2e" % err, flush=True)
    ok = ok and sym and err < 4e-6
print("OK" if ok else "FAILED")
