#!/bin/bash
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -5 $OUT/pytest.log
STAMPLIB=lib_stamps.so timeout -k 10 120 python scratch/stamps_dist.py 16384 256 > $OUT/stamps_dist.txt 2>&1; cat $OUT/stamps_dist.txt
STAMPLIB=lib_stamps.so timeout -k 10 120 python scratch/stamps_dist_modes.py 16384 256 > $OUT/stamps_dist_modes.txt 2>&1; cat $OUT/stamps_dist_modes.txt
