#!/bin/bash
# r04j: after the clock-flag race fix + the M-tail shfl fix + small-batch work: long stress screens (stamps off / on), the FULL GPU suite,
# the small-batch latencies, the bench line
set -o pipefail
mkdir -p gpurun_out/r04j
for st in 0 1; do
  echo "== one piece stamps=$st";  STAMPS=$st timeout -k 10 200 python tests/stress_gemm_pp.py 60 2>&1 | grep -v amdgpu.ids | tail -1
  echo "== two pieces stamps=$st"; STAMPS=$st timeout -k 10 200 python tests/stress_gemm_pp_x3.py 60 2>&1 | grep -v amdgpu.ids | tail -1
done | tee gpurun_out/r04j/stress.txt
grep -q MISMATCH gpurun_out/r04j/stress.txt && exit 1
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04j/pytest_all.log 2>&1; rc=$?; echo "full suite rc=$rc"; tail -6 gpurun_out/r04j/pytest_all.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tests/bench_small_batch.py 2>/dev/null | tee gpurun_out/r04j/small_batch.txt
echo "--- GRAM_BEAM_PRE_MAXB=0" | tee -a gpurun_out/r04j/small_batch.txt
GRAM_BEAM_PRE_MAXB=0 timeout -k 10 300 python tests/bench_small_batch.py --batches 1,4,16 2>/dev/null | tee -a gpurun_out/r04j/small_batch.txt
