#!/bin/bash
# Round 3, GPU call: in-load-slot epilogue of the two-piece 16-bit outputs: bit-identity tests, race screen, timing.
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_split.py -x -q -k "persistent or epilogues" 2>&1 | tail -6
echo "tests rc=$?"
timeout -k 10 200 python tests/stress_gemm_pp_x3.py 100 2>&1 | grep -v amdgpu.ids | tail -5
echo "stress rc=$?"
timeout -k 10 200 python tests/bench_gemm_x3.py --iters 4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03h_gemm_x3_insl.txt
