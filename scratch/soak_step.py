"""Determinism soak of the whole fused step at full size (C3, C4) under memory pressure from a second stream: phi, the
bandwidth and |phi|^2 bit-identical over all repetitions.  usage: soak_step.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd.engine import SvgdEngine
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = "cuda"
side = torch.cuda.Stream(device=dev)
a = torch.empty(128 << 20, dtype=torch.float32, device=dev); b = torch.empty_like(a)
for name, n, d in (("C3", 16384, 256), ("C4", 8192, 2001)):
    T = torch.randn(n, d, device=dev); G = torch.randn(n, d, device=dev)
    eng = SvgdEngine(n, d, device=dev)
    first = eng.compute_phi(T, G).clone(); h2, sq = float(eng.h2), float(eng.sqnorm)
    bad = 0
    for rep in range(reps):
        with torch.cuda.stream(side):
            for _ in range(rep % 5):
                b.copy_(a)
        phi = eng.compute_phi(T, G); torch.cuda.synchronize()
        if not (torch.equal(phi, first) and float(eng.h2) == h2 and float(eng.sqnorm) == sq):
            bad += 1
            print("   %s: repetition %d differs (%d entries)" % (name, rep, int((phi != first).sum())), flush=True)
    print("%s fused step (n=%d d=%d): %d repetitions, %d differed; window %s" % (name, n, d, reps, bad, eng.window_stats()), flush=True)
    del eng, T, G, first
    torch.cuda.empty_cache()
