#!/bin/bash
# Where do the vector-memory instructions of k_distance_panel wait?  L1 (TCP/TA) and L2 (TCC -> memory) stall counters,
# one rocprofv3 --pmc pass per group; the contraction is printed beside it for scale.  usage (gpurun): bash scratch/pmc_stall.sh <tag>
set -e -o pipefail
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_TAG_STALL_sum TCC_SRC_FIFO_FULL_sum" "TCC_BUSY_sum TCC_IB_STALL_sum" "TCC_WRITE_REQ_LATENCY_sum TCC_WRITE_sum" "TCP_TCR_TCP_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_BUSY_sum" "TCP_RFIFO_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_sum" "TCC_READ_REQ_LATENCY_sum TCC_READ_sum"; do
  N=$(echo $C | tr ' ' '+')
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
