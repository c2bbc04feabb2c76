"""Per-tile vs panel distance kernel on small symmetric blocks (is the panel kernel's size threshold right?):
staged stein_distance_block, torch events over 50 launches.  usage: dist_small_ab.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd import _lib
from stein_amd.engine import SvgdEngine
dev = "cuda"
for n, d, dtype in ((4096, 512, torch.bfloat16), (4096, 1024, torch.bfloat16), (8192, 128, torch.bfloat16), (8192, 512, torch.bfloat16), (4096, 320, torch.float32), (4096, 2001, torch.float32), (1024, 128, torch.float32), (2048, 128, torch.float32), (4096, 128, torch.bfloat16), (4096, 128, torch.float32),
                    (4096, 256, torch.float32), (8192, 128, torch.float32), (8192, 256, torch.float32), (8192, 40, torch.float32),
                    (4096, 1000, torch.float32), (2048, 2001, torch.float32)):
    eng = SvgdEngine(n, d, device=dev, x3=True, dtype=dtype, small=False)
    st = eng.stages
    T = torch.randn(n, d, device=dev).to(dtype); G = torch.randn(n, d, device=dev).to(dtype)
    st.rownorms(T, n, d, eng.rownorm); st.x3_prepare(T, G, n, d, eng.planes)
    out = {}
    for name, kernel in (("tiles", _lib.STAGE_TILES), ("panel", _lib.STAGE_PANEL)):
        for _ in range(5):
            st.distance_block(T, eng.rownorm, n, d, 0, n, eng.dist, eng.ld_dist, symmetric=True, planes=eng.planes, kernel=kernel)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            st.distance_block(T, eng.rownorm, n, d, 0, n, eng.dist, eng.ld_dist, symmetric=True, planes=eng.planes, kernel=kernel)
        e1.record(); torch.cuda.synchronize()
        out[name] = e0.elapsed_time(e1) / 50 * 1e3
    print("n=%5d d=%4d %-8s  tiles %7.1f us   panel %7.1f us   strips/wave %.1f" % (n, d, str(dtype).split(".")[1], out["tiles"], out["panel"], (n // 128) * (n // 32 + 4) / 2 / 2048), flush=True)
    del eng
