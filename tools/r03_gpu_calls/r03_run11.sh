#!/bin/bash
cd $GRAFT_REPO_ROOT
for m in 32768 16384 8192 4096; do
  echo "== GRAM_GEMM_PP_MINM=$m"
  GRAM_GEMM_PP_MINM=$m timeout -k 10 300 python tests/bench_small_batch.py --batches 16,32,64,128,200 --iters 8 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r03o_small_batch_pp_minm.txt
