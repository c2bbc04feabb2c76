#!/bin/bash
# per-kernel average durations of the fused C3 step for a given library: bash scratch/kstat.sh <lib|shipped> [n d]
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/kstat_$1
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- python3 $GRAFT_REPO_ROOT/scratch/fused_loop.py $1 ${2:-16384} ${3:-256} 20 > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/k_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if r["Name"].split("(")[0].replace("void ", "").startswith("k_"):
        print("%-44s calls %4s  avg %9.1f us  min %9.1f" % (r["Name"].split("(")[0][:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
rm -rf $OUT
