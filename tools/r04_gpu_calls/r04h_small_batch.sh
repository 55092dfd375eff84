#!/bin/bash
# r04h: small-batch latency work: residual prefetch by LDS-DMA in the streaming GEMM's fp32 epilogue, sparse logits of a handful of users by
# their own kernel.  Bit-identity tests (stream vs tiled GEMM, sparse vs dense search, whole-path batch invariance), then B = 1 / 4 / 16 / 64.
set -o pipefail
mkdir -p gpurun_out/r04h
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_split.py tests/test_gpu_configs.py tests/test_gpu_path.py -x -q -m gpu -k "stream or beam or sparse or batch or generate or live or folded or row_factors" > gpurun_out/r04h/pytest.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -6 gpurun_out/r04h/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tests/bench_small_batch.py 2>/dev/null | tee gpurun_out/r04h/small_batch.txt
echo "--- GRAM_BEAM_PRE_MAXB=0 (the search step computes its own sparse logits)" | tee -a gpurun_out/r04h/small_batch.txt
GRAM_BEAM_PRE_MAXB=0 timeout -k 10 300 python tests/bench_small_batch.py 2>/dev/null | tee -a gpurun_out/r04h/small_batch.txt
