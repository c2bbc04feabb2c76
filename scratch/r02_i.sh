#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02i
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 100 ./scratch/probes/producer > $OUT/producer.txt 2>&1; cat $OUT/producer.txt
