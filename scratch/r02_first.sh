#!/bin/bash
# round-2 first GPU call: GPU tests, bench line, distance-pass modes and ablations
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02a
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout -k 10 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout -k 10 120 python scratch/dist_modes_time.py 16384 256 > $OUT/dist_modes.txt 2>&1; cat $OUT/dist_modes.txt
timeout -k 10 400 python scratch/dist_ablate.py 16384 256 base nomfma noload noepi nomirror norowstore nostore > $OUT/dist_ablate.txt 2>&1; cat $OUT/dist_ablate.txt
