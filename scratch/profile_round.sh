#!/bin/bash
# Round-end measurement on the GPU box: bench line, rocprofv3 kernel stats of the same command, PMC passes (separate runs).
# usage (from the repo root, via gpurun): bash scratch/profile_round.sh <tag>
set -e -o pipefail
TAG=${1:-final}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench done" >&2
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --secondary none --no-other-configs --no-variants --no-train-on-batch"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o c3 -- $BENCH > $OUT/stats.log 2>&1
echo "stats done" >&2
PMCB="python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 4 --no-cpu-baseline --secondary none --no-other-configs --no-variants --no-train-on-batch"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -o c3 -- $PMCB > $OUT/pmc_$C.log 2>&1
  echo "pmc $C done" >&2
done
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc_mfma -o c3 -- $PMCB > $OUT/pmc_mfma.log 2>&1 || echo "mfma counters unavailable" >&2
find $OUT -name "*.csv" | head -20 >&2
