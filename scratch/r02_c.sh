#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 200 ./scratch/probes/cu_ingest > $OUT/cu_ingest2.txt 2>&1; grep "workgroups of" $OUT/cu_ingest2.txt
