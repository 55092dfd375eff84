#!/bin/bash
# Round 3, GPU call: start-stagger A/B of the two-piece ping-pong GEMM (epilogue bursts), then the GPU test suite.
cd $GRAFT_REPO_ROOT
for s in 0 1 2 4 8; do
  echo "== stagger $s"
  GRAM_GEMM_STAGGER=$s timeout -k 10 200 python tests/bench_gemm_x3.py --iters 4 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r03e_gemm_x3_stagger.txt 2>&1
cat gpurun_out/r03e_gemm_x3_stagger.txt
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03e_gputests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r03e_gputests.log
