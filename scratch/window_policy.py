"""Which window-sizing rule would have missed how often?  Records the per-step lower-median key of real runs (the centre of
the next window depends on the last two medians only, never on the sizing rule), then replays sizing rules offline: a rule hits
step t iff |key_t - center_t| <= halfwidth_t.  Cost proxy: mean halfwidth (entries collected grow linearly with it).
usage: python scratch/window_policy.py [steps]  -> one block per scenario"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stein_amd import _lib
from stein_amd.engine import SvgdEngine
from stein_amd.optimizers import AdagradGradientDescent, AdamGradientDescent
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 300
HW_MAX = 32768


def f32_key(x):      # order-preserving key of a positive float (stein_common.h)
    return int(np.float32(x).view(np.uint32)) | 0x80000000
def key_f32(k):
    return float(np.uint32(k & 0x7fffffff).view(np.float32))


def record(n, d, dtype, make_gd, score):
    torch.manual_seed(0)
    theta = torch.randn(n, d, device="cuda"); G0 = torch.randn(n, d, device="cuda")
    eng = SvgdEngine(n, d, device="cuda", dtype=dtype); gd = make_gd()
    _, offs, _ = _lib.workspace_layout(n, n, d, flags=eng.flags)
    o = offs[_lib.WS_SELECT]
    keys, hits, hws, cnts = [], [], [], []
    for step in range(STEPS):
        G = score(theta, G0, step)
        phi = eng.compute_phi(theta.to(dtype), G.to(dtype)); gd.apply_(theta, phi, eng.sqnorm)
        raw = eng.ws[o:o + 128].cpu().numpy()
        lo = raw[0:64].view(np.float32)     # SelState: find `lo` by value below
        u = raw[64:128].view(np.uint32)
        keys.append(int(u[12]))             # SpecState.last_key = key of this step's lower median target (after the update)
        hits.append(int(u[7])); hws.append(int(u[2])); cnts.append(int(u[5]))
    return np.array(keys, dtype=np.int64), np.array(hits), np.array(hws), np.array(cnts)


def replay(keys, rule):
    """rule(err_history, hw_prev, hit_prev) -> halfwidth for the next step"""
    miss, widths = 0, []
    center, hw, errs = None, None, []
    last = None
    for t, key in enumerate(keys):
        if center is not None and hw is not None:
            err = abs(int(key) - int(center))
            hit = err <= hw
            miss += 0 if hit else 1
            widths.append(hw)
            errs.append(err)
        else:
            hit = False
        if last is not None:
            pred = 2.0 * key_f32(key) - key_f32(last)
            center = f32_key(pred)
            hw = min(HW_MAX, 4096 if len(errs) == 0 else rule(errs, hw if len(errs) > 1 else 0, hit))   # (the 4096-key first window is no floor)
        last = key
    return miss, float(np.mean(widths)) if widths else 0.0, float(np.median(widths)) if widths else 0.0


RULES = {
    "round 3: 4 err + 48": lambda e, hw, hit: HW_MAX if e[-1] > HW_MAX // 4 else 4 * e[-1] + 48,
    "shipped: max(4 err + 48, 3/4 previous)": lambda e, hw, hit: max(HW_MAX if e[-1] > HW_MAX // 4 else 4 * e[-1] + 48, hw - hw // 4),
    "max(4 err + 48, 1/2 previous)": lambda e, hw, hit: max(4 * e[-1] + 48, hw // 2),
    "4 max(last 2 err) + 48": lambda e, hw, hit: 4 * max(e[-2:]) + 48,
    "4 max(last 4 err) + 48": lambda e, hw, hit: 4 * max(e[-4:]) + 48,
    "3 max(last 4 err) + 48": lambda e, hw, hit: 3 * max(e[-4:]) + 48,
    "4 max(last 8 err) + 48": lambda e, hw, hit: 4 * max(e[-8:]) + 48,
    "8 err + 48": lambda e, hw, hit: 8 * e[-1] + 48,
}

SCEN = [
    ("C2 n=4096 d=128 bf16, adagrad 1e-3, fixed score", 4096, 128, torch.bfloat16, lambda: AdagradGradientDescent(learning_rate=1e-3), lambda th, G0, s: G0),
    ("C2 shape, fp32, adagrad 1e-3, fixed score", 4096, 128, torch.float32, lambda: AdagradGradientDescent(learning_rate=1e-3), lambda th, G0, s: G0),
    ("C2 shape, fp32, adam 1e-2, score = -theta + 30% noise", 4096, 128, torch.float32, lambda: AdamGradientDescent(learning_rate=1e-2), lambda th, G0, s: -th + 0.3 * torch.randn_like(th)),
    ("C3 n=16384 d=256 fp32, adagrad 1e-3, fixed score", 16384, 256, torch.float32, lambda: AdagradGradientDescent(learning_rate=1e-3), lambda th, G0, s: G0),
    ("C3 shape, adam 1e-2, score = -theta + 30% noise", 16384, 256, torch.float32, lambda: AdamGradientDescent(learning_rate=1e-2), lambda th, G0, s: -th + 0.3 * torch.randn_like(th)),
    ("n=2048 d=32 fp32, adam 1e-2, score = -theta + 30% noise", 2048, 32, torch.float32, lambda: AdamGradientDescent(learning_rate=1e-2), lambda th, G0, s: -th + 0.3 * torch.randn_like(th)),
]
for name, n, d, dt, mk, score in SCEN:
    keys, hits, hws, cnts = record(n, d, dt, mk, score)
    print("== %s: %d steps, the library hit %d (after step 2: %d misses), median halfwidth %d, median entries %d" %
          (name, STEPS, hits.sum(), int((1 - hits[2:]).sum()), int(np.median(hws[2:])), int(np.median(cnts[2:]))), flush=True)
    for rn, rule in RULES.items():
        m, mean_w, med_w = replay(keys, rule)
        print("   %-40s misses %3d   mean halfwidth %8.0f   median %6.0f" % (rn, m, mean_w, med_w))
