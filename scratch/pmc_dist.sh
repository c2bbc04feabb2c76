#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / L2 hit counters of the distance kernel under library builds: bash scratch/pmc_dist.sh <tag> lib1 lib2 ...
set -e -o pipefail
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    N=$(echo $C | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/${L}_$N -o x -- python3 $GRAFT_REPO_ROOT/scratch/fused_loop.py $L 16384 256 8 > $OUT/${L}_$N.log 2>&1
  done
done
python3 - <<PY
import csv, glob, os, collections
out = "$OUT"
for d in sorted(glob.glob(out + "/*/")):
    p = os.path.join(d, "x_counter_collection.csv")
    if not os.path.exists(p): continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if "k_distance" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(os.path.basename(d.rstrip("/")), {k: round(sum(v[3:]) / max(1, len(v[3:])), 1) for k, v in acc.items()})
PY
