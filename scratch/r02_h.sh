#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02h
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for v in s_c_nomfma s_c_nolds s_c_novload s_c_nomfma_nolds s_c_nomfma_novload; do
  echo "== $v" | tee -a $OUT/stamps_abl5.txt
  STAMPLIB=lib_$v.so timeout -k 10 120 python scratch/stamps.py 16384 256 2>&1 | grep -v amdgpu.ids | tee -a $OUT/stamps_abl5.txt
done
