#!/bin/bash
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02d
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python scratch/dist_ablate.py 16384 256 base stag1 stag2 stag3 > $OUT/dist_stag.txt 2>&1; cat $OUT/dist_stag.txt
