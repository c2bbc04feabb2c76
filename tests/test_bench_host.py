"""CPU: the host-side helpers of bench.py that decide what the bench line says about a run (no GPU needed)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_gap_fields_flag_a_host_stall():
    ok = bench.gap_fields(0.994, 0.981)                       # round 3's headline: 13 us between wall and events
    assert ok["wall_minus_events_ms"] == 0.013 and ok["host_stall_suspected"] is False
    stalled = bench.gap_fields(7.20, 1.14)                    # the driver's tile_distance_path of round 3
    assert stalled["host_stall_suspected"] is True and abs(stalled["wall_minus_events_ms"] - 6.06) < 1e-9
    small = bench.gap_fields(0.029, 0.025)                    # C1: a few microseconds of launch gaps are not a stall
    assert small["host_stall_suspected"] is False
    hiccup = bench.gap_fields(1.0824, 0.9747)                 # round 4: one ~2 ms hiccup of the launch thread in a 20-step loop
    assert hiccup["host_stall_suspected"] is True
    assert bench.gap_fields(0.45, None) == {"events_ms": None, "wall_minus_events_ms": None, "host_stall_suspected": None}


def test_cpu_extrapolations_follow_the_n2d_law():
    base = {"seconds": 6.8, "cores": 128, "one_thread": {"seconds": 0.68, "sample": "the same on ONE thread, rows [0,2048): 0.7 s"}}
    ex = bench.cpu_extrapolations(base, 16384, 256)
    f5 = (131072.0 ** 2 * 256) / (16384.0 ** 2 * 256)
    assert abs(ex["c5"]["seconds_per_step"] - 6.8 * f5) < 1e-9 and "EXTRAPOLATED" in ex["c5"]["label"]
    assert abs(ex["c5"]["one_thread"]["seconds_per_step"] - 0.68 * 8 * f5) < 1e-9      # 2048 of 16384 rows -> x 8 for a full step
    assert abs(ex["c4"]["value"] - 8192 / ex["c4"]["seconds_per_step"]) < 1e-12


def test_workloads_are_the_baseline_configs():
    assert (bench.WORKLOADS["c3"]["n"], bench.WORKLOADS["c3"]["d"]) == (16384, 256)
    assert (bench.WORKLOADS["c2"]["n"], bench.WORKLOADS["c2"]["d"], bench.WORKLOADS["c2"]["bf16"]) == (4096, 128, True)
    assert (bench.WORKLOADS["c4"]["n"], bench.WORKLOADS["c4"]["d"]) == (8192, 2001)
    assert (bench.WORKLOADS["c5"]["n"], bench.WORKLOADS["c5"]["d"]) == (131072, 256)
    assert bench.PEAK_16BIT_MFMA == 2.5e15 and bench.PEAK_FP32_MFMA == 157.3e12


def test_headline_gap_describes_the_instrumented_loop_and_flags_a_slow_timed_loop():
    # a light run: the timed loop (two events per step) must not be slower than the fully instrumented loop behind it
    res = dict(instr_elapsed=20 * 0.9209e-3, events_ms=0.9084, settle_steps=80, first_block_ms=0.9838)
    ok = bench.headline_gap(0.9029, res, 20)
    assert ok["host_stall_suspected"] is False and ok["instrumented_ms_per_step"] == 0.9209
    assert abs(ok["wall_minus_events_ms"] - (0.9209 - 0.9084)) < 1e-9 and ok["settle_steps"] == 80
    assert ok["first_block_ms_per_step"] == 0.9838 and "timing_note" in ok
    slow = bench.headline_gap(0.9738, res, 20)                 # round 4: the timed loop ran in the unsettled first block
    assert slow["host_stall_suspected"] is True
    # several ranks (no second loop): the plain gap fields plus the settle count
    multi = bench.headline_gap(0.30, dict(instr_elapsed=None, events_ms=None, settle_steps=60), 20)
    assert multi["settle_steps"] == 60 and multi["host_stall_suspected"] is None
