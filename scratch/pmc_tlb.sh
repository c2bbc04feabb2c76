#!/bin/bash
# Address-translation counters of the distance and contraction kernels (are the store stalls of k_distance_panel TLB misses?)
# usage (gpurun): bash scratch/pmc_tlb.sh <tag>
set -e -o pipefail
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/avail.txt 2>&1 || true
grep -o "UTCL[A-Za-z0-9_]*\|TCP_[A-Z0-9_]*STALL[A-Z0-9_]*\|TA_[A-Z_]*STALL[A-Z_]*\|TCP_PENDING[A-Z_]*" $OUT/avail.txt | sort -u > $OUT/names.txt || true
for C in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$N -o x -- python3 $GRAFT_REPO_ROOT/scratch/fused_loop.py shipped 16384 256 8 > $OUT/$N.log 2>&1 || echo "counter set $N refused" >&2
done
python3 - <<PY
import csv, glob, os, collections
out = "$OUT"
for d in sorted(glob.glob(out + "/*/")):
    p = os.path.join(d, "x_counter_collection.csv")
    if not os.path.exists(p): continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("k_distance") or k.startswith("k_phi_x3fs"):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(k, {c: round(sum(v[3:]) / max(1, len(v[3:])), 1) for c, v in cs.items()})
PY
