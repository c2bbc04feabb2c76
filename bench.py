#!/usr/bin/env python3
"""bench.py -- SVGD particle-updates/s on MI355X, with the MFMA roofline of the dominant kernel.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one full particle update given the score matrix G (resident in HBM): the fused C call
stein_svgd_phi (row norms, operand scales and split planes -> distance pass on the 16-bit matrix cores with the
speculative median window -> exact median -> fused exp + split-precision MFMA contraction K.[G|theta] -> phi,
|phi|^2) followed by clip + Adagrad apply (lr 1e-3, alpha 0.9); particles move every step.  Workload:
BASELINE.json's roofline config "n=16384, d=256, fp32" (C3, the configuration the north star's contraction
target is quoted on); with N ranks the same n is sharded by rows (strong scaling): all-gathers of the theta and G
rows, the median collectives (stein_amd/engine.py) and one scalar all-reduce per step over RCCL.  The secondary
entry times C5 (n=131072) the same way; a one-GPU run also times the other BASELINE configs (C1, C2 bf16, C4) for 20
steps each and reports them under `other_configs` (they are parity-test cases, not the headline value).

value = n * K / (max-over-ranks wall time of the K timed steps), bracketed by barrier + synchronize.

roofline (the K.[G|theta] contraction kernel, k_phi_x3fs; k_phi_partial with x3=False): duration from HIP events
recorded around each of its launches on the launching stream inside the timed region (by the library at one rank:
STEIN_FLAG_TIMING).  `achieved` = the ALGORITHMIC flops per launch of SURVEY 8(d) (4*n_local*n*d: 2*n^2*d for K.G plus
2*n^2*d for K.theta) per second, `peak` = 2.5 PFLOP/s dense fp16/bf16 MFMA (MI355X_MICROARCH.md: the pipe the kernel runs
on), `frac` = achieved / peak (= `frac_algorithmic`).  The split path executes 3 fp16 products per fp32 operand pair:
`frac_executed` (3x) is the utilisation of that pipe, `mfma_busy_frac` the counter's reading of the same, and
`frac_vs_fp32_mfma_peak` the algorithmic flops against the 157.3 TFLOP/s fp32-input MFMA peak (what an fp32 GEMM could
reach at most); the strict fp32-input MFMA kernels are timed under `fp32_path`.

Headline timing (one rank): `ms_per_step` / `value` come from a timed loop of exactly K steps that carries two HIP events per
step (they bracket the contraction: `roofline.ms_per_launch`, measured live); `stage_ms`, `events_ms`, `wall_minus_events_ms` and
`instrumented_ms_per_step` come from a second loop of K steps with an event at every stage boundary.  An event between two kernels
costs the step ~3 us of GPU time (scratch/event_cost.py): eight of them are 2.5 % of a C3 step and a quarter of a C2 step.
Behind the W warm-up steps the same loop keeps running untimed until 80 ms of load have passed (`settle_steps` says how many
steps that took): the chip's clock governor settles under this load in ~50 ms, and the 20 steps behind a 5-step warm-up alone
are 8 % slower than every later block of 20 (scratch/step_trend.py); `first_block_ms_per_step` is that block, timed on the side.

wall_minus_events_ms (every timed entry): wall time per step minus the GPU time the HIP events of the same steps bracket;
`host_stall_suspected` when the gap exceeds max(0.03 ms, 4 %) (a healthy C3 loop shows 0.012 ms).  A stalled timed loop
(headline or variant) is timed again, at most twice; the line carries the run with the smallest wall time and `retimed` lists
what the other runs measured -- every one of them is a complete timed loop of the requested number of steps.

window: the headline step uses the speculative median window (exact, stein_common.h); `window` reports how many of the
timed steps it delivered the median, and `miss_path` times the same steps with the window disabled (every step pays the
radix-select passes over D) -- the cost of a miss, driver-timed.

parity_sample_relerr: after the timed region, phi of one more (untimed) call is compared on sampled rows with an fp64
evaluation of stein/samplers/abstract_stein_sampler.py:100-105 + stein/kernels/squared_exponential_kernel.py:22,32 over
all n columns (torch fp64 on the device, using the bandwidth the kernels produced).
cpu_baseline: the NumPy oracle (a port: the reference's TF-1.12 kernel graph cannot run anywhere here) timed
on the host on a bounded row block of the same workload.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA = 157.3e12      # MI355X_MICROARCH.md, chip-level parameters (dense fp32-input MFMA)
PEAK_16BIT_MFMA = 2.5e15       # same table: dense bf16 / fp16 MFMA
WORKLOADS = {
    "c3": dict(n=16384, d=256, name="C3 bayesian-logreg-shaped synthetic block n=16384 d=256 fp32 (roofline config)"),
    "c5": dict(n=131072, d=256, name="C5 n=131072 d=256 fp32, rows sharded over the ranks"),
    "c2": dict(n=4096, d=128, bf16=True, name="C2 n=4096 d=128, theta and score fed to the kernels in bf16 (fp32 master copy)"),
    "c2f32": dict(n=4096, d=128, name="C2-shaped n=4096 d=128 run in fp32"),
    "c4": dict(n=8192, d=2001, name="C4 BNN-shaped n=8192 d=2001 (H=666) fp32"),
    "c1": dict(n=100, d=10, name="C1 n=100 d=10"),
}


def make_inputs(n, d, row0, n_local, device, torch):
    # T ~ N(0,1) seed 0, G ~ N(0,1) seed 1, generated in fp64 then cast
    T = np.random.default_rng(0).normal(size=(n, d))
    G = np.random.default_rng(1).normal(size=(n, d))
    Tl = torch.tensor(T[row0:row0 + n_local], dtype=torch.float32, device=device).contiguous()
    Gl = torch.tensor(G[row0:row0 + n_local], dtype=torch.float32, device=device).contiguous()
    return T, G, Tl, Gl


class StageClock:
    """Records a HIP event (torch.cuda.Event on the launching = current stream) at every stage mark."""

    def __init__(self, torch):
        self.torch = torch
        self.steps = []
        self.cur = None

    def begin_step(self):
        self.cur = []
        self.steps.append(self.cur)

    def mark(self, label):
        ev = self.torch.cuda.Event(enable_timing=True)
        ev.record()
        self.cur.append((label, ev))

    def summary(self):
        """-> {label: mean ms between this mark and the next one}"""
        acc = {}
        for marks in self.steps:
            for (la, ea), (_, eb) in zip(marks[:-1], marks[1:]):
                acc.setdefault(la, []).append(ea.elapsed_time(eb))
        return {k: float(np.mean(v)) for k, v in acc.items()}


SETTLE_MS = 80.0    # untimed load in front of a light run's timed loop (see run_workload)
SETTLE_STEPS_MULTI = 60   # the same for a sharded run, as a step count that is equal on every rank


def gap_fields(wall_ms, events_ms):
    """Wall time per step against the GPU time the HIP events of the same steps bracket (the library's stage events plus the
    apply kernel's pair; they do not cover the host's time BETWEEN two steps): a gap means the host, not the GPU, set the
    pace -- a stall that would otherwise hide inside ms_per_step."""
    if events_ms is None:
        return {"events_ms": None, "wall_minus_events_ms": None, "host_stall_suspected": None}
    gap = wall_ms - events_ms
    return {"events_ms": round(events_ms, 4), "wall_minus_events_ms": round(gap, 4),
            "host_stall_suspected": bool(gap > max(0.03, 0.04 * events_ms))}


def headline_gap(ms_per_step, res, steps):
    """The gap fields of the headline entry.  One rank: the timed loop carries only the contraction's two events, so the stage
    split, `events_ms` and `wall_minus_events_ms` describe the SECOND loop (every stage event; `instrumented_ms_per_step` is its
    wall time per step), and `host_stall_suspected` says whether the timed loop was slower than that one allows."""
    if res.get("instr_elapsed") is None:
        g = gap_fields(ms_per_step, res["events_ms"])
        g["settle_steps"] = res.get("settle_steps")
        return g
    instr_ms = res["instr_elapsed"] / steps * 1e3
    g = gap_fields(instr_ms, res["events_ms"])
    ref_ms = min(instr_ms, res["events_ms"])
    g["host_stall_suspected"] = bool(g["host_stall_suspected"] or ms_per_step > ref_ms + max(0.03, 0.04 * res["events_ms"]))
    g["instrumented_ms_per_step"] = round(instr_ms, 4)
    g["settle_steps"] = res.get("settle_steps")
    g["first_block_ms_per_step"] = round(res["first_block_ms"], 4) if res.get("first_block_ms") else None
    g["timing_note"] = ("ms_per_step / value: the timed loop, which carries two HIP events per step (around the contraction: "
                        "roofline.ms_per_launch).  stage_ms / events_ms / wall_minus_events_ms / instrumented_ms_per_step: a second "
                        "loop of the same steps with an event at every stage boundary -- each event between two kernels costs "
                        "~3 us of GPU time, so that loop is slower than the one it explains.  settle_steps: untimed steps of the same "
                        "loop run behind the W warm-up steps until %g ms of load have passed since the first one (the clock governor "
                        "settles under this load in ~50 ms; the 20 steps behind a 5-step warm-up alone are 8 %% slower than every "
                        "later block of 20, scratch/step_trend.py).  first_block_ms_per_step: the first `steps` of those settle steps, "
                        "timed on the side without any event: what a measurement right behind the W warm-up steps reads" % SETTLE_MS)
    return g


def parity_sample(torch, T_all, G_all, phi_local, row0, h2, n_rows=48):
    """Relative Frobenius error of `n_rows` sampled rows of phi_local (rows row0.. of the global matrix) against the
    fp64 formulae over all n columns, with bandwidth^2 = h2; rows from the first, a middle and the last row tile."""
    n, nl = T_all.shape[0], phi_local.shape[0]
    rng = np.random.default_rng(12345)
    tiles = sorted({0, (nl // 128) // 2, (nl - 1) // 128})
    rows = sorted({int(r) for t in tiles for r in rng.integers(t * 128, min(nl, t * 128 + 128), size=n_rows // 3)})
    loc = torch.as_tensor(rows, device=T_all.device)
    Ta, Ga = T_all.double(), G_all.double()
    Ti = Ta[loc + row0]
    ra = (Ta * Ta).sum(1)
    D = ra[loc + row0][:, None] + ra[None, :] - 2.0 * (Ti @ Ta.T)
    K = torch.exp(-D / h2 / 2.0)
    ref = (K @ Ga + (K.sum(1)[:, None] * Ti - K @ Ta) / h2) / n
    diff = phi_local[loc].double() - ref
    return float((diff.norm() / ref.norm()).item()), len(rows)


def run_workload(torch, dist, wl, device, rank, world, group, steps, warmup, clock_stages=True, x3=None, window=True,
                 comm="torch", tile_distance=False, light=False, settle=False):
    """light (one rank): the timed loop carries only the two HIP events that bracket the contraction (the roofline's kernel time,
    measured live); the full stage split comes from a second loop of the same number of steps right behind it, with every stage
    event and the apply kernel's pair.  A HIP event between two kernels costs the step ~3 us of GPU time (scratch/event_cost.py:
    eight of them are 2.5 % of a C3 step and a quarter of a C2 step), so the loop whose wall time is the headline should not
    be the one that carries them."""
    from stein_amd import _lib
    from stein_amd.engine import SvgdEngine
    from stein_amd.optimizers import AdagradGradientDescent
    n, d = wl["n"], wl["d"]
    n_local = n // world
    row0 = rank * n_local
    T64, G64, theta, G = make_inputs(n, d, row0, n_local, device, torch)
    bf16 = bool(wl.get("bf16"))      # BASELINE config 2: the kernels see bf16 theta / score; theta's master copy stays fp32
    ekw = dict(device=device, group=group, dtype=torch.bfloat16 if bf16 else torch.float32, x3=x3, window=window,
               tile_distance=tile_distance)
    comm_note = None
    if world > 1:
        # who issues the collectives: torch.distributed over RCCL (default: the path every multi-rank test has run) or the
        # library's own RCCL communicator (--comm native; its set-up is collective-safe: it fails on every rank or on none)
        try:
            eng = SvgdEngine(n, d, comm=comm, **ekw)
        except RuntimeError as exc:
            if comm != "native":
                raise
            eng = SvgdEngine(n, d, comm="torch", **ekw)
            comm_note = "library communicator unavailable (%s): torch.distributed collectives" % exc
    else:
        eng = SvgdEngine(n, d, **ekw)
    gd = AdagradGradientDescent(learning_rate=1e-3, alpha=0.9)
    if bf16:
        G = G.to(torch.bfloat16)
    feed = (lambda: theta.to(torch.bfloat16)) if bf16 else (lambda: theta)
    clock = StageClock(torch)

    # The step is what SteinSampler.update_particles runs: the fused C call on a single rank, the rank-step segments with
    # the collectives between them on several.  Either way the library records HIP events on the launching stream
    # (STEIN_FLAG_TIMING): at every stage boundary of the fused call, around the contraction and the finish pass of a
    # sharded step (the other stages of a sharded step are separated by collectives and are not split here).
    fused = True
    light = bool(light and world == 1 and clock_stages)
    apply_events = []
    if fused and clock_stages:
        _lib.timing_reserve(steps)
    full_events = [not light]          # (rebound for the second, fully instrumented loop of a light run)

    def step(timed):
        if fused:
            phi = eng.compute_phi(feed(), G, timing=(True if full_events[0] else "contract") if (timed and clock_stages) else False)
            if timed and clock_stages and full_events[0]:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                gd.apply_(theta, phi, eng.sqnorm)
                e1.record()
                apply_events.append((e0, e1))
            else:
                gd.apply_(theta, phi, eng.sqnorm)
        elif timed and clock_stages:
            clock.begin_step()
            phi = eng.compute_phi(feed(), G, mark=clock.mark)
            gd.apply_(theta, phi, eng.sqnorm)
            clock.mark("apply_end")
        else:
            phi = eng.compute_phi(feed(), G, mark=(lambda s: None) if clock_stages else None)
            gd.apply_(theta, phi, eng.sqnorm)

    t_warm = time.perf_counter()
    for _ in range(warmup):
        step(False)
    # light runs: the chip's clock governor needs ~50 ms of THIS load to settle -- the 20 steps that follow a 5-step warm-up
    # are 8 % slower than every later block of 20 (contraction 0.65 -> 0.61 ms; scratch/step_trend.py), and a generic load
    # beforehand settles only half of it.  More untimed steps of the same loop follow the W warm-up steps until 80 ms have
    # passed since the first one; how many is reported (`settle_steps`).
    settle_steps = 0
    first_block_ms = None
    if light or (settle and world == 1):
        # (the first `steps` of them are timed on the side: what a measurement right behind the W warm-up steps reads)
        torch.cuda.synchronize(device)
        tb = time.perf_counter()
        for _ in range(steps):
            step(False)
        torch.cuda.synchronize(device)
        first_block_ms = (time.perf_counter() - tb) / steps * 1e3
        settle_steps = steps
        while (time.perf_counter() - t_warm) * 1e3 < SETTLE_MS and settle_steps < 2000:
            for _ in range(10):
                step(False)
            settle_steps += 10
            torch.cuda.synchronize(device)
    if world > 1 and clock_stages:
        # several ranks: a FIXED number of settle steps (every rank must take the same number of steps: each holds collectives)
        for _ in range(SETTLE_STEPS_MULTI):
            step(False)
        settle_steps = SETTLE_STEPS_MULTI
    if world > 1:
        dist.barrier(group=group)
    torch.cuda.synchronize(device)
    has_window = world == 1 and window and n > 160       # the one-kernel path (n <= 160) keeps no window state
    stats0 = eng.window_stats() if has_window else (0, 0)
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier(group=group)
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(te, op=dist.ReduceOp.MAX, group=group)
        elapsed = float(te.item())
    stats1 = eng.window_stats() if has_window else (0, 0)
    contract_live = instr_elapsed = None
    if light:
        per_call = _lib.timing_read(steps)
        contract_live = sum(c["contract"] for c in per_call) / max(1, len(per_call))
        # the second loop: the same steps with every stage event (not part of `elapsed`)
        full_events[0] = True
        _lib.timing_reserve(steps)
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        for _ in range(steps):
            step(True)
        torch.cuda.synchronize(device)
        instr_elapsed = time.perf_counter() - t1
    if fused and clock_stages:
        per_call = _lib.timing_read(steps)
        stages = {k: round(sum(c[k] for c in per_call) / max(1, len(per_call)), 4) for k in _lib.T_STAGES
                  if world == 1 or k in ("contract", "finish")}
        stages["apply"] = round(sum(a.elapsed_time(b) for a, b in apply_events) / max(1, len(apply_events)), 4)
    else:
        stages = clock.summary() if clock_stages else {}
        if "end" in stages:            # interval end -> apply_end is the optimizer apply kernel
            stages["apply"] = stages.pop("end")
    # the GPU time the stage events account for, per step (one rank: every stage of the fused call + the apply kernel)
    events_ms = float(sum(stages.values())) if (world == 1 and clock_stages and stages) else None
    finite = bool(torch.isfinite(theta).all().item())
    # parity on sampled rows: one more call, untimed, no apply (theta and phi then belong together)
    cur = feed()
    phi = eng.compute_phi(cur, G)
    torch.cuda.synchronize(device)
    if world > 1:
        T_all, G_all = eng.T_all.float(), eng.G_all.float()
    else:
        T_all, G_all = cur.float(), G.float()
    relerr, nrows = parity_sample(torch, T_all, G_all, phi, row0, float(eng.h2.item()))
    crosscheck = None
    return dict(comm=eng.comm, comm_note=comm_note, comm_crosscheck=crosscheck, n=n, d=d, n_local=n_local, elapsed=elapsed, stages=stages, events_ms=events_ms, finite=finite, split=eng.split,
                contract_live_ms=contract_live, instr_elapsed=instr_elapsed, settle_steps=settle_steps, first_block_ms=first_block_ms,
                ws_bytes=eng.ws_bytes, T64=T64, G64=G64, x3=eng.x3, parity_relerr=relerr, parity_rows=nrows,
                window=dict(timed_steps=stats1[0] - stats0[0], hits=stats1[1] - stats0[1]) if has_window else None)


def run_stable(torch, dist, wl, device, rank, world, group, steps, warmup, **kw):
    """run_workload, timed again (at most twice) when the host, not the GPU, set the pace of the timed loop (one rank: wall time
    per step beyond the event-bracketed GPU time by more than max(0.03 ms, 4 %): a launch thread that lost its core on a shared
    box).  Keeps the run with the smallest wall time.  -> (result, None | what the other runs measured)"""
    res = run_workload(torch, dist, wl, device, rank, world, group, steps, warmup, **kw)

    def stalled(r):
        if r.get("instr_elapsed") is None:
            return gap_fields(r["elapsed"] / steps * 1e3, r["events_ms"])["host_stall_suspected"]
        # a light run: its timed loop carries two events per step; it must not be slower than the loop with all of them
        # behind it, nor than that loop's event-bracketed GPU time by more than the usual margin
        instr = gap_fields(r["instr_elapsed"] / steps * 1e3, r["events_ms"])
        ref_ms = min(r["instr_elapsed"] / steps * 1e3, r["events_ms"])
        return bool(instr["host_stall_suspected"] or r["elapsed"] / steps * 1e3 > ref_ms + max(0.03, 0.04 * r["events_ms"]))

    if world != 1 or not stalled(res):
        return res, None
    others = []
    for _ in range(2):
        again = run_workload(torch, dist, wl, device, rank, world, group, steps, warmup, **kw)
        keep, drop = (again, res) if again["elapsed"] < res["elapsed"] else (res, again)
        others.append({"ms_per_step": drop["elapsed"] / steps * 1e3, "events_ms": drop["events_ms"]})
        res = keep
        del drop, again
        torch.cuda.empty_cache()
        if not stalled(res):
            break
    return res, {"why": "wall time per step exceeded the event-bracketed GPU time by more than max(0.03 ms, 4 %): host stall; "
                        "timed again, the run with the smallest wall time is reported", "other_runs": others}


def crosscheck_comms(torch, dist, wl, device, rank, world, group, steps=4):
    """Several ranks: the same steps with torch.distributed issuing the collectives between the rank segments and with the
    library's own RCCL communicator issuing them inside stein_rank_step -- same kernels, same data, so phi and the bandwidth
    must come out bit-identical at every step (|phi|^2 to the order of its fp64 rank sum).  Also reads back, through
    stein_comm_info, how many ranks the library's RCCL communicator really has."""
    import ctypes
    from stein_amd import _lib
    from stein_amd.engine import SvgdEngine
    from stein_amd.optimizers import AdagradGradientDescent
    n, d = wl["n"], wl["d"]
    nl = n // world
    _, _, theta0, G = make_inputs(n, d, rank * nl, nl, device, torch)
    res = {}
    try:
        native = SvgdEngine(n, d, device=device, group=group, comm="native")
    except Exception as exc:      # collective: raised on every rank or on none
        return {"native_available": False, "why": "%s: %s" % (type(exc).__name__, exc)}
    nr, rk = ctypes.c_int(-1), ctypes.c_int(-1)
    _lib.call("stein_comm_info", native._comm, ctypes.byref(nr), ctypes.byref(rk))
    res.update(native_available=True, rccl_nranks=int(nr.value), rccl_rank=int(rk.value))
    other = SvgdEngine(n, d, device=device, group=group, comm="torch")
    th = {"native": theta0.clone(), "torch": theta0.clone()}
    gd = {k: AdagradGradientDescent(learning_rate=1e-3, alpha=0.9) for k in th}
    worst_phi, h2_equal, worst_sq, hits = 0.0, True, 0.0, []
    for _ in range(steps):
        pn = native.compute_phi(th["native"], G).clone()
        pt = other.compute_phi(th["torch"], G).clone()
        torch.cuda.synchronize(device)
        worst_phi = max(worst_phi, float((pn - pt).abs().max().item()))
        h2_equal = h2_equal and float(native.h2.item()) == float(other.h2.item())
        sq = float(other.sqnorm.item())
        worst_sq = max(worst_sq, abs(float(native.sqnorm.item()) - sq) / max(sq, 1e-300))
        hits.append([native.window_hit, other.window_hit])
        gd["native"].apply_(th["native"], pn, native.sqnorm)
        gd["torch"].apply_(th["torch"], pt, other.sqnorm)
    res.update(steps=steps, phi_max_abs_diff=worst_phi, h2_equal=h2_equal, sqnorm_rel_diff=worst_sq, window_hits=hits,
               theta_equal=bool(torch.equal(th["native"], th["torch"])))
    native.close()
    return res


def train_on_batch_entry(torch, device, steps, warmup):
    """One GPU, C3 shape: SteinSampler.train_on_batch with the device score producer of the logistic-regression example
    (stein_amd.scores.GlmScore: 255 weights + log alpha, minibatch 50 of 16000 points) -- the score is RECOMPUTED from theta
    every iteration, so the median moves as it does in a real run and the window has to follow it."""
    from stein_amd.optimizers import AdagradGradientDescent
    from stein_amd.samplers import SteinSampler
    from stein_amd.scores import GlmScore
    n, nf, batch, ntrain = 16384, 255, 50, 16000
    g = torch.Generator(device=device).manual_seed(7)
    X = torch.randn(batch, nf, device=device, generator=g)
    y = (torch.rand(batch, device=device, generator=g) < 0.5).float()
    theta0 = 0.1 * torch.randn(n, nf + 1, device=device, generator=g)
    s = SteinSampler(n, None, AdagradGradientDescent(learning_rate=1e-3, alpha=0.9), theta=theta0,
                     score=GlmScore("logistic", nf, w_col=1, alpha_col=0, n_train=ntrain), device=device)
    feed = {"X": X, "y": y}
    for _ in range(warmup):
        s.train_on_batch(feed)
    torch.cuda.synchronize(device)
    st0 = s.engine.window_stats()
    pairs = []
    t0 = time.perf_counter()
    for _ in range(steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        s.train_on_batch(feed)
        e1.record()
        pairs.append((e0, e1))
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    st1 = s.engine.window_stats()
    ev_ms = sum(a.elapsed_time(b) for a, b in pairs) / steps     # GPU time bracketed per iteration (not between iterations)
    return {**gap_fields(dt / steps * 1e3, ev_ms),
            "what": "SteinSampler.train_on_batch at the C3 shape (n=16384, d=256): logistic-regression score recomputed from theta "
                    "on the device every iteration (GlmScore, minibatch 50), fused SVGD step, Adagrad apply",
            "steps": steps, "ms_per_step": dt / steps * 1e3, "value": n * steps / dt, "unit": "particle-updates/s",
            "window": {"timed_steps": st1[0] - st0[0], "hits": st1[1] - st0[1]},
            "finite": bool(torch.isfinite(s.theta_matrix).all().item())}


def host_cores():
    """(logical CPUs this process may run on, physical cores among them) -- from the affinity mask and /proc/cpuinfo"""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cpus = list(range(os.cpu_count() or 1))
    phys = set()
    try:
        cur = {}
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, v = [x.strip() for x in line.split(":", 1)]
                cur[k] = v
            elif cur:
                if int(cur.get("processor", -1)) in cpus:
                    phys.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
                cur = {}
        if cur and int(cur.get("processor", -1)) in cpus:
            phys.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
    except Exception:
        phys = set()
    return len(cpus), (len(phys) or None)


def cpu_baseline(wl, T64, G64, rows, one_thread_rows=0):
    """The NumPy oracle (the reference's dtype flow: fp32 kernel, fp64 contraction) timed on the host on a bounded row block
    of the workload, with every BLAS thread the box gives and -- one_thread_rows > 0 -- once more on one thread
    (SURVEY 8(d)); `cores` = the BLAS threads actually used, the physical core count is reported beside it."""
    from oracle import svgd_oracle as orc
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        info = threadpool_info()
        threads = max([i.get("num_threads", 1) for i in info] or [1])
        blas = ",".join(sorted({"%s %s" % (i.get("internal_api"), i.get("version")) for i in info}))
    except Exception:
        threadpool_limits, threads, blas = None, os.cpu_count() or 1, "unknown"
    logical, physical = host_cores()
    n, d = wl["n"], wl["d"]
    rows = min(rows, n)
    gd = orc.AdagradState(learning_rate=1e-3, alpha=0.9)
    t0 = time.perf_counter()
    orc.svgd_step_rows(T64, G64, 0, rows, gd, np.float32)
    dt = time.perf_counter() - t0
    out = dict(value=rows / dt, unit="particle-updates/s", cores=int(threads), kind="port",
               blas_threads=int(threads), logical_cpus=logical, physical_cores=physical,
               sample="one step of the NumPy oracle (fp32 kernel, fp64 contraction, %s) on rows [0,%d) of the same "
                      "n=%d d=%d inputs: %.1f s of CPU work; the reference's TF-1.12 graph cannot run here" %
                      (blas, rows, n, d, dt),
               seconds=dt)
    if one_thread_rows and threadpool_limits is not None:
        r1 = min(one_thread_rows, n)
        with threadpool_limits(limits=1):
            t0 = time.perf_counter()
            orc.svgd_step_rows(T64, G64, 0, r1, orc.AdagradState(learning_rate=1e-3, alpha=0.9), np.float32)
            d1 = time.perf_counter() - t0
        out["one_thread"] = dict(value=r1 / d1, unit="particle-updates/s", cores=1,
                                 sample="the same on ONE thread, rows [0,%d): %.1f s" % (r1, d1), seconds=d1)
    return out


def cpu_extrapolations(base, n0, d0):
    """C4 / C5 host numbers by the n^2 d law from a measured full step ('C4/C5 extrapolated by the n^2 d law and
    labelled as such'): seconds per step scale with n^2 d, particle-updates/s with 1 / (n d)."""
    out = {}
    for key, (n, d) in (("c4", (8192, 2001)), ("c5", (131072, 256))):
        f = (float(n) * n * d) / (float(n0) * n0 * d0)
        ent = {"n": n, "d": d, "label": "EXTRAPOLATED from the measured n=%d d=%d step by the n^2 d law, not measured" % (n0, d0),
               "seconds_per_step": base["seconds"] * f, "value": n / (base["seconds"] * f), "unit": "particle-updates/s",
               "cores": base["cores"]}
        if "one_thread" in base:
            s1 = base["one_thread"]["seconds"] * (float(n0) / one_rows(base)) * f
            ent["one_thread"] = {"seconds_per_step": s1, "value": n / s1, "cores": 1}
        out[key] = ent
    return out


def one_rows(base):
    """rows of the one-thread sample (parsed back from its own record)"""
    return int(base["one_thread"]["sample"].split("rows [0,")[1].split(")")[0])


def pmc_traffic(workload_key, x3):
    """HBM bytes per launch of the dominant kernel as measured with rocprofv3 --pmc (separate passes, FETCH_SIZE doubled
    as MI355X_MICROARCH.md prescribes for gfx950) and committed under profiles/.  Counters cannot be read from inside
    this process; the entry says which file and which commit the number comes from."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(p) as f:
            rec = json.load(f)
        ent = rec.get(workload_key, {})
        val = ent.get("k_phi_x3fs_hbm_bytes" if x3 else "k_phi_partial_hbm_bytes")
        src = "profiles/pmc_traffic.json <- %s, measured at commit %s" % (rec.get("source", "?"), rec.get("commit", "?"))
        return val, (src if val is not None else None), (ent.get("k_phi_x3fs_mfma_busy_frac") if x3 else None), ent
    except Exception:
        return None, None, None, {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--n", type=int, default=0, help="override the workload's particle count (experiments)")
    ap.add_argument("--d", type=int, default=0, help="override the workload's parameter count (experiments)")
    ap.add_argument("--cpu-rows", type=int, default=16384, help="rows of the bounded CPU-baseline sample")
    ap.add_argument("--cpu-rows-one-thread", type=int, default=2048, help="rows of the one-thread CPU sample (0: skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--comm", default="torch", choices=["torch", "native"],
                    help="several ranks: who issues the timed steps' collectives (torch.distributed over RCCL, or the "
                         "library's own RCCL communicator); the other way is run once afterwards as a cross-check")
    ap.add_argument("--no-train-on-batch", action="store_true", help="skip the SteinSampler.train_on_batch entry (one GPU)")
    ap.add_argument("--secondary", default="c5", help="also time this workload briefly (extra key); 'none' to skip")
    ap.add_argument("--secondary-steps", type=int, default=10)
    ap.add_argument("--no-other-configs", action="store_true", help="skip the brief C1 / C2 / C4 timings (one GPU)")
    ap.add_argument("--fp32-mfma", action="store_true", help="headline on the strict fp32-input MFMA kernels (x3=False)")
    ap.add_argument("--no-variants", action="store_true", help="skip the miss-path and fp32-path timings (one GPU)")
    args = ap.parse_args()

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path in stein_amd)")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs one process per GPU: launch with python -m torch.distributed.run "
                             "--nproc-per-node %d ..." % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if os.environ.get("STEIN_REHEARSAL_ONE_GPU"):   # rehearse the multi-rank path with all ranks on one card
        local_rank = 0
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    dist, group = None, None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("STEIN_DIST_BACKEND", "nccl")   # RCCL; "gloo" only for one-GPU rehearsals
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)
        group = dist.group.WORLD

    import __graft_entry__ as ge
    ge.build()

    wl = dict(WORKLOADS[args.workload])
    if args.n or args.d:
        wl["n"], wl["d"] = args.n or wl["n"], args.d or wl["d"]
        wl["name"] = "custom n=%d d=%d fp32" % (wl["n"], wl["d"])
    env_x3 = not args.fp32_mfma
    hx3 = False if args.fp32_mfma else None
    # (a stalled timed loop -- a launch thread that lost its core on a shared box -- is timed once more; `retimed` keeps the
    # first run's numbers)
    res, retimed = run_stable(torch, dist, wl, device, rank, world, group, args.steps, args.warmup, comm=args.comm, x3=hx3,
                              light=True)
    n, d, nl = res["n"], res["d"], res["n_local"]
    ms_per_step = res["elapsed"] / args.steps * 1e3
    value = n * args.steps / res["elapsed"]
    nprod = (1 if wl.get("bf16") else 3) if res["x3"] else 1     # 16-bit MFMA products per operand pair
    # the contraction's mean launch time: from the two events that bracket it inside the timed loop (one rank), else from
    # the stage events
    k_ms = res["contract_live_ms"] if res.get("contract_live_ms") else res["stages"].get("contract")
    flops = 4.0 * nl * n * d
    alg = flops / (k_ms * 1e-3) if k_ms else None
    traffic, traffic_src, mfma_busy, pmc_ent = pmc_traffic(args.workload, res["x3"]) if world == 1 else (None, None, None, {})
    if res["x3"]:
        # achieved = ALGORITHMIC flops per launch (4 n_local n d) / the kernel's mean duration, against the peak
        # of the pipe the kernel runs on (dense fp16/bf16 MFMA).  The split path executes `nprod` 16-bit products per fp32
        # operand pair, so the pipe utilisation is nprod x that; it is reported beside it, never as `frac`.
        roof = {
            "kernel": "k_phi_x3fs (exp + split-precision MFMA K.[G|theta] contraction, %d 16-bit products per operand pair)" % nprod,
            "bound": "mfma", "achieved": alg / 1e12 if alg else None, "peak": PEAK_16BIT_MFMA / 1e12,
            "unit": "TFLOP/s", "frac": alg / PEAK_16BIT_MFMA if alg else None,
            "frac_algorithmic": alg / PEAK_16BIT_MFMA if alg else None,
            "frac_executed": nprod * alg / PEAK_16BIT_MFMA if alg else None,
            "executed_tflops": nprod * alg / 1e12 if alg else None,
            "mfma_busy_frac": mfma_busy,
            "mfma_busy_measured_at_ms_per_launch": (round(pmc_ent["k_phi_x3fs_avg_ns_rocprofv3"] / 1e6, 4)
                                                    if pmc_ent.get("k_phi_x3fs_avg_ns_rocprofv3") else None),
            "frac_vs_fp32_mfma_peak": alg / PEAK_FP32_MFMA if alg else None,
            "note": "achieved / frac = frac_algorithmic = the ALGORITHMIC 4 n_local n d flops of SURVEY 8(d) / mean kernel time / the "
                    "dense fp16/bf16 MFMA peak (the pipe the kernel runs on).  frac_executed = the 16-bit MFMA flops the kernel "
                    "EXECUTES (%d products per fp32 operand pair x the algorithmic flops) against the same peak: the utilisation "
                    "of that pipe.  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x 2.4 GHz x kernel time) from the "
                    "committed rocprofv3 --pmc pass (profiles/pmc_traffic.json), an independent reading of frac_executed AT THAT PASS'S launch time (mfma_busy_measured_at_ms_per_launch: the profiler's launches "
                    "are slower than the settled timed loop's, so it reads lower than frac_executed here).  "
                    "frac_vs_fp32_mfma_peak = the algorithmic flops against the 157.3 TFLOP/s fp32-input MFMA peak (SURVEY 7: a "
                    "split emulation is also reported against the fp32 peak; > 1 means faster than any fp32-MFMA GEMM)" % nprod,
            "fp32_equivalent_tflops": alg / 1e12 if alg else None,
            "fp32_equivalent_vs_fp32_mfma_peak": alg / PEAK_FP32_MFMA if alg else None,
        }
    else:
        roof = {
            "kernel": "k_phi_partial (exp + fp32-input MFMA K.[G|theta] contraction)",
            "bound": "mfma", "achieved": alg / 1e12 if alg else None, "peak": PEAK_FP32_MFMA / 1e12,
            "unit": "TFLOP/s", "frac": alg / PEAK_FP32_MFMA if alg else None,
            "frac_algorithmic": alg / PEAK_FP32_MFMA if alg else None,
            "frac_executed": alg / PEAK_FP32_MFMA if alg else None,
            "note": "fp32-input MFMA kernel: algorithmic = executed flops",
        }
    roof.update({"algorithmic_flops_per_launch": flops, "executed_flops_per_launch": nprod * flops, "ms_per_launch": k_ms,
                 "traffic": traffic, "traffic_source": traffic_src})
    out = {
        "metric": "SVGD particle-updates/sec",
        "value": value,
        "unit": "particle-updates/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "scaling_note": ("`value` is BASELINE's roofline config (n=16384, d=256) sharded by rows over the ranks: at 8 ranks a rank holds "
                         "2048 rows, ~0.3 ms of kernels next to four latency-bound collectives, so this series saturates near 2-3x; the "
                         "configuration the north star's >= 6x at 8 GPUs refers to (n=131072, d=256) is timed in the same run under "
                         "`secondary`, one entry per N") if world > 1 else None,
        "vs_baseline": None,
        "dtype": "bf16" if wl.get("bf16") else "f32",
        "data": "synthetic",
        "config": {"workload": wl["name"], "n": n, "d": d, "rows_per_rank": nl,
                   "optimizer": "adagrad lr=1e-3 alpha=0.9",
                   "parallelism": "rows sharded x%d, all-gather(theta,G) + median (3 histogram all-reduces, or one window-table "
                                  "all-reduce when the local block has >= 2^24 entries) + 1 scalar all-reduce" % world
                   if world > 1 else "single GPU"},
        "element_updates_per_s": value * d,
        "pair_interactions_per_s": value * n,
        "roofline": roof,
        "gemm_path": ({1: "bf16 inputs (1 product)", 3: "split fp16 x 2 (3 products)"}[nprod]) if res["x3"] else "fp32 mfma",
        "stage_ms": {k: round(v, 4) for k, v in res["stages"].items()},
        **headline_gap(ms_per_step, res, args.steps),
        "retimed": retimed,
        "window": res["window"],
        "full_step_tflops": 6.0 * nl * n * d / (ms_per_step * 1e-3) / 1e12,
        "finite": res["finite"],
        "parity_sample_relerr": res["parity_relerr"],
        "parity_sample_rows": res["parity_rows"],
    }
    # the distance pass (north star: "achieved HBM GB/s on the distance pass"): one rank stores the 128 x 128 tiles on and
    # above the diagonal, a row block all of its tiles; 2 n_local n d algorithmic flops (half with the symmetry), x nprod executed
    d_ms = res["stages"].get("distance")
    if d_ms and res["x3"]:
        tiles = (n // 128) * (n // 128 + 1) // 2 if world == 1 else (nl // 128) * (n // 128)
        stored = tiles * 128 * 128 * 4.0 if n % 128 == 0 and nl % 128 == 0 else None
        dflop = 2.0 * nl * n * d * (0.5 if world == 1 else 1.0)
        out["distance_pass"] = {
            "kernel": ("k_distance_x3 (one 128 x 128 tile per workgroup)" if not (n % 128 == 0 and nl % 128 == 0 and ((nl // 128) * (n // 32) >= 16384 or
                       ((nl // 128) * (n // 32) >= 4096 and (1 if wl.get("bf16") else 2) * ((d + 31) // 32) >= 16)))   # stein_dpanel_ok
                       else "k_distance_panel (operand panel in LDS, strips streamed from L2)" if d <= (512 if wl.get("bf16") else 256)
                       else "k_distance_panel_deep (the panel in LDS a chunk of K at a time, strips streamed from L2)"),
            "ms_per_launch": d_ms, "bound": "hbm",
            "stored_bytes": stored, "achieved_write_GBps": stored / (d_ms * 1e-3) / 1e9 if stored else None,
            "peak_GBps": 8000.0, "frac_of_hbm_peak": stored / (d_ms * 1e-3) / 8e12 if stored else None,
            "executed_mfma_tflops": nprod * dflop / (d_ms * 1e-3) / 1e12,
            "frac_of_16bit_mfma_peak": nprod * dflop / (d_ms * 1e-3) / PEAK_16BIT_MFMA,
            "hbm_bytes_measured": pmc_ent.get("k_distance_hbm_bytes") if world == 1 else None,
            "note": "the pass writes D once (algorithmic bytes = stored bytes) and executes %d x n^2 d MFMA flops with the "
                    "symmetry; at C3 a plain fill writes these bytes in 0.083 ms and the MFMAs need 0.11 ms at the sustained clock" % nprod}
    if world > 1:
        out["collectives"] = {"issued_by": {"native": "libsteinhip (own RCCL communicator, stein_rank_step: one C call per step)",
                                            "torch": "torch.distributed (backend %s) between the rank segments" % dist.get_backend(group)}[res["comm"]],
                              "note": res["comm_note"], "backend": str(dist.get_backend(group)),
                              "torch_world_size": int(dist.get_world_size(group))}
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(wl, res["T64"], res["G64"], args.cpu_rows, one_thread_rows=args.cpu_rows_one_thread)
        if args.workload == "c3" and args.cpu_rows >= n:
            out["cpu_baseline"]["extrapolated"] = cpu_extrapolations(out["cpu_baseline"], n, d)
    del res

    # one GPU: what the headline does not show -- a step whose window misses, and the strict fp32-input MFMA kernels
    if world == 1 and not args.no_variants and not wl.get("bf16") and env_x3:
        torch.cuda.empty_cache()
        rm, rt_m = run_stable(torch, dist, wl, device, rank, world, group, 10, 3, window=False)
        out["miss_path"] = {"retimed": rt_m, "what": "the same step with the speculative median window disabled: every step runs the "
                                    "radix-select passes over D (what a window miss costs)",
                            "steps": 10, "ms_per_step": rm["elapsed"] / 10 * 1e3,
                            "stage_ms": {k: round(v, 4) for k, v in rm["stages"].items()},
                            **gap_fields(rm["elapsed"] / 10 * 1e3, rm["events_ms"]),
                            "parity_sample_relerr": rm["parity_relerr"]}
        del rm
        torch.cuda.empty_cache()
        rf, rt_f = run_stable(torch, dist, wl, device, rank, world, group, 5, 2, x3=False)
        kf = rf["stages"].get("contract")
        out["fp32_path"] = {"retimed": rt_f, "what": "the same step on the fp32-input MFMA kernels (x3=False): exact k-ordered fmaf chains",
                            "steps": 5, "ms_per_step": rf["elapsed"] / 5 * 1e3,
                            "stage_ms": {k: round(v, 4) for k, v in rf["stages"].items()},
                            **gap_fields(rf["elapsed"] / 5 * 1e3, rf["events_ms"]),
                            "contract_ms": kf, "contract_tflops": flops / (kf * 1e-3) / 1e12 if kf else None,
                            "contract_frac_of_fp32_mfma_peak": flops / (kf * 1e-3) / PEAK_FP32_MFMA if kf else None,
                            "parity_sample_relerr": rf["parity_relerr"]}
        del rf

    if world == 1 and not args.no_variants and not wl.get("bf16") and env_x3:
        torch.cuda.empty_cache()
        rt, rt_t = run_stable(torch, dist, wl, device, rank, world, group, 10, 3, tile_distance=True)
        out["tile_distance_path"] = {"retimed": rt_t, "what": "the same step with the distance pass on the per-tile kernel (k_distance_x3, round 2's) "
                                             "instead of the panel-resident one: same-box A/B",
                                     "steps": 10, "ms_per_step": rt["elapsed"] / 10 * 1e3,
                                     "stage_ms": {k: round(v, 4) for k, v in rt["stages"].items()},
                                     **gap_fields(rt["elapsed"] / 10 * 1e3, rt["events_ms"])}
        del rt
    if world == 1 and args.workload == "c3" and not args.no_train_on_batch:
        torch.cuda.empty_cache()
        out["train_on_batch"] = train_on_batch_entry(torch, device, args.steps, max(args.warmup, 3))

    if args.secondary != "none" and args.secondary != args.workload:
        wl2 = WORKLOADS[args.secondary]
        need = (wl2["n"] // world) * wl2["n"] * 4 * 1.15 + 4 * wl2["n"] * wl2["d"] * 8
        free = torch.cuda.mem_get_info(device)[0]
        if need < free * 0.9:
            torch.cuda.empty_cache()
            r2, rt_2 = run_stable(torch, dist, wl2, device, rank, world, group, args.secondary_steps, 3, comm=args.comm)   # 3 warm-up steps: the median predictor needs two medians of history
            k2 = r2["stages"].get("contract")
            f2 = 4.0 * r2["n_local"] * r2["n"] * r2["d"]
            np2 = (1 if wl2.get("bf16") else 3) if r2["x3"] else 1     # 16-bit products per operand pair (1: the fp32-input MFMA kernel)
            # the configuration the north star's ">= 6x at 8 GPUs" refers to: its per-N value also at the top level of the line
            out["secondary_value"] = r2["n"] * args.secondary_steps / r2["elapsed"]
            out["secondary_config"] = "%s: n=%d d=%d, %d ranks, %d timed steps" % (args.secondary, r2["n"], r2["d"], world, args.secondary_steps)
            out["secondary"] = {
                "retimed": rt_2, "workload": wl2["name"], "n": r2["n"], "d": r2["d"], "steps": args.secondary_steps,
                "ms_per_step": r2["elapsed"] / args.secondary_steps * 1e3,
                "value": r2["n"] * args.secondary_steps / r2["elapsed"], "unit": "particle-updates/s",
                "scaling": "strong (fixed n=%d)" % r2["n"],
                "contract_products_per_pair": np2,
                "contract_executed_tflops": np2 * f2 / (k2 * 1e-3) / 1e12 if k2 else None,
                ("contract_frac_of_16bit_mfma_peak" if r2["x3"] else "contract_frac_of_fp32_mfma_peak"):
                    np2 * f2 / (k2 * 1e-3) / (PEAK_16BIT_MFMA if r2["x3"] else PEAK_FP32_MFMA) if k2 else None,
                "stage_ms": {k: round(v, 4) for k, v in r2["stages"].items()}, "finite": r2["finite"],
                **gap_fields(r2["elapsed"] / args.secondary_steps * 1e3, r2["events_ms"]),
                "window": r2["window"], "parity_sample_relerr": r2["parity_relerr"],
            }
            del r2
    # one GPU: the remaining BASELINE configs (parity-test cases, timed briefly for the record; not the headline value)
    if world == 1 and args.workload == "c3" and not args.no_other_configs:
        others = {}
        for key in ("c1", "c2", "c4"):
            torch.cuda.empty_cache()
            # the step time without stage events in the loop (at these sizes the event records are not free), then a
            # short run with them for the stage split
            ro = run_workload(torch, dist, dict(WORKLOADS[key]), device, rank, world, group, 20, 3, clock_stages=False, settle=True)
            rs = run_workload(torch, dist, dict(WORKLOADS[key]), device, rank, world, group, 8, 3)
            others[key] = {"workload": WORKLOADS[key]["name"], "n": ro["n"], "d": ro["d"], "steps": 20,
                           "ms_per_step": ro["elapsed"] / 20 * 1e3, "value": ro["n"] * 20 / ro["elapsed"],
                           "unit": "particle-updates/s", "finite": ro["finite"],
                           "settle_steps": ro["settle_steps"], "first_block_ms_per_step": round(ro["first_block_ms"], 4),
                           "stage_ms": {k: round(v, 4) for k, v in rs["stages"].items()},
                           "staged_run_ms_per_step": rs["elapsed"] / 8 * 1e3,   # the 8-step run that carried the stage events
                           **gap_fields(rs["elapsed"] / 8 * 1e3, rs["events_ms"]),
                           "window": ro["window"], "parity_sample_relerr": ro["parity_relerr"]}
            del rs
            if not args.no_cpu_baseline:     # the NumPy oracle on the same inputs (C4: a bounded row block)
                others[key]["cpu_baseline"] = cpu_baseline(WORKLOADS[key], ro["T64"], ro["G64"],
                                                           2048 if key == "c4" else ro["n"])
            del ro
        out["other_configs"] = others
    # the commit the working tree was at: from git where there is one, else from the stamp build() leaves beside the library
    # (the GPU boxes get a snapshot without .git)
    commit = None
    try:
        r = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=10)
        commit = r.stdout.strip() or None if r.returncode == 0 else None
    except Exception:
        commit = None
    if commit is None:
        try:
            commit = open(os.path.join(ROOT, "stein_amd", ".build_commit")).read().strip() or None
        except OSError:
            commit = None
    out["commit"] = commit
    if world > 1 and str(dist.get_backend(group)) == "nccl" and not os.environ.get("STEIN_NO_CROSSCHECK"):
        # Last, and guarded: the library's own communicator has only ever run on one-rank groups before this machine.  If the
        # native-vs-torch cross-check hangs (a mismatched collective has no timeout in RCCL), a watchdog prints the line that
        # is already complete -- the timed numbers above do not depend on it -- and ends the process.
        import threading
        done = threading.Event()

        def watchdog():
            if not done.wait(float(os.environ.get("STEIN_CROSSCHECK_TIMEOUT", "120"))):
                out["collectives"]["crosscheck_native_vs_torch"] = {"timed_out": True}
                if rank == 0:
                    print(json.dumps(out), flush=True)
                os._exit(3)      # a hang is a failure: never exit 0 from here
        threading.Thread(target=watchdog, daemon=True).start()
        try:
            out["collectives"]["crosscheck_native_vs_torch"] = crosscheck_comms(torch, dist, wl, device, rank, world, group)
        except Exception as exc:   # noqa: BLE001 -- reported, never silent
            out["collectives"]["crosscheck_native_vs_torch"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        done.set()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier(group=group)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
