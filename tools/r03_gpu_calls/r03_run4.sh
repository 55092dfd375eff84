#!/bin/bash
# Round 3, GPU call: ablations of the two-piece ping-pong GEMM (make ABL=n builds) + in-kernel clock, f16 vs bf16 pieces.
cd $GRAFT_REPO_ROOT
for v in abl8 abl9 abl11 abl15 bf16_abl8; do
  echo "== $v"
  GRAM_LIB=$PWD/gram_amd/csrc/libgram_hip_$v.so timeout -k 10 200 python tests/bench_gemm_x3.py --iters 4 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r03d_gemm_x3_ablation.txt 2>&1
cat gpurun_out/r03d_gemm_x3_ablation.txt
