#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in abl8 abl9 abl24; do
  echo "== $v"
  GRAM_LIB=$PWD/gram_amd/csrc/libgram_hip_$v.so timeout -k 10 200 python tests/bench_gemm_x3.py --iters 4 --only "enc qkv" 2>&1 | grep -v amdgpu.ids
  GRAM_LIB=$PWD/gram_amd/csrc/libgram_hip_$v.so timeout -k 10 200 python tests/bench_gemm_x3.py --iters 4 --only "enc wi" 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r03i_insl_ablation.txt 2>&1
cat gpurun_out/r03i_insl_ablation.txt
