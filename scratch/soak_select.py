"""Soak of the one-launch radix select (k_hist_all) at full size with the window disabled -- every step takes its in-launch
level barriers -- under memory pressure from a second stream and, every fourth repetition, with a forced grid (fewer
workgroups than virtual ones, or more than the chip holds): bandwidth, phi and |phi|^2 bit-identical over all repetitions
and equal to the staged calls' (separate k_hist / k_resolve launches).  usage: soak_select.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd import _lib
from stein_amd.engine import SvgdEngine
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
dev = "cuda"
side = torch.cuda.Stream(device=dev)
a = torch.empty(128 << 20, dtype=torch.float32, device=dev); b = torch.empty_like(a)
for name, n, d, dt in (("C3", 16384, 256, torch.float32), ("C2", 4096, 128, torch.bfloat16), ("odd", 3001, 17, torch.float32)):
    T = torch.randn(n, d, device=dev).to(dt); G = torch.randn(n, d, device=dev).to(dt)
    eng = SvgdEngine(n, d, device=dev, dtype=dt, window=False, small=False)
    ref = SvgdEngine(n, d, device=dev, dtype=dt, small=False)
    want = ref.compute_phi(T, G, mark=lambda s: None).clone(); h2, sq = float(ref.h2), float(ref.sqnorm)
    bad = 0
    for rep in range(reps):
        grid = 0 if rep % 4 else (5, 300, 5000)[(rep // 4) % 3]
        _lib.call("stein_debug_hist_all_grid", grid)
        with torch.cuda.stream(side):
            for _ in range(rep % 5):
                b.copy_(a)
        phi = eng.compute_phi(T, G); torch.cuda.synchronize()
        if not (torch.equal(phi, want) and float(eng.h2) == h2 and float(eng.sqnorm) == sq):
            bad += 1
            print("   %s: repetition %d (grid %d) differs: h2 %r vs %r" % (name, rep, grid, float(eng.h2), h2), flush=True)
    _lib.call("stein_debug_hist_all_grid", 0)
    _lib.call("stein_take_device_error")
    print("%s (n=%d d=%d %s): %d repetitions of the miss path, %d differed from the staged calls" % (name, n, d, str(dt).split(".")[1], reps, bad), flush=True)
    del eng, ref, T, G, want
    torch.cuda.empty_cache()
