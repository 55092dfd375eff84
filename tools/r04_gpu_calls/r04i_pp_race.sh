#!/bin/bash
# r04i: is the ping-pong GEMM racy when its diagnostic clock stamps are off (the product path since the hygiene commit)?
# tests/test_gpu_kernels.py::test_folded_layernorm_gemms[33000] (one piece, fp32-residual epilogue) fails intermittently without them.
mkdir -p gpurun_out/r04i
for stamps in 0 1; do
  echo "== one piece, stamps=$stamps"; STAMPS=$stamps timeout -k 10 100 python tests/stress_gemm_pp.py 60 2>&1 | grep -v amdgpu.ids | tail -2
  echo "== two pieces, stamps=$stamps"; STAMPS=$stamps timeout -k 10 100 python tests/stress_gemm_pp_x3.py 60 2>&1 | grep -v amdgpu.ids | tail -2
done | tee gpurun_out/r04i/stress.txt
