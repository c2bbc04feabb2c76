#!/bin/bash
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02e
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
STAMPLIB=lib_stamps.so timeout -k 10 120 python scratch/stamps.py 16384 256 > $OUT/stamps_contract.txt 2>&1; cat $OUT/stamps_contract.txt
