#!/bin/bash
# One GPU call of round 4: the -m gpu tests, the default bench line, the C2 line over 200 steps (window misses included).
# usage (from the repo root on the GPU box): bash scratch/r4_round.sh <tag>
tag=${1:-r4}
out=gpurun_out/$tag
mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/gputest.log 2>&1; echo "pytest rc=$?" >> $out/gputest.log
tail -4 $out/gputest.log
python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python bench.py --workload c2 --steps 200 --secondary none --no-other-configs --no-variants --no-cpu-baseline > $out/bench_c2_200.json 2> $out/bench_c2.err; echo "bench c2 rc=$?"
python - <<PY
import json
b = json.load(open("$out/bench.json"))
print("C3 ms/step %.4f" % b["ms_per_step"], b["stage_ms"], "gap", b["wall_minus_events_ms"])
print("roofline frac %.4f executed %.4f" % (b["roofline"]["frac"], b["roofline"]["frac_executed"]))
for k in ("miss_path", "tile_distance_path", "train_on_batch"):
    e = b.get(k)
    print(k, "%.4f" % e["ms_per_step"], e.get("stage_ms"), e.get("wall_minus_events_ms"))
print("secondary", b["secondary"]["ms_per_step"], b["secondary"]["stage_ms"])
for k, e in b["other_configs"].items():
    print(k, "%.4f" % e["ms_per_step"], e["stage_ms"], e["window"])
c = json.load(open("$out/bench_c2_200.json"))
print("C2 x200 %.4f" % c["ms_per_step"], c["stage_ms"], c["window"])
PY
