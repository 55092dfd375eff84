#!/bin/bash
# Round 4: the chaos build (make CHAOS=1: waves sleep at random behind workgroup barriers, common.h gram_sync / pp_barrier).
#   1. does the test hook reproduce the round-4 prologue race on the kernel WITHOUT the barrier?            (nofix + ENTRY_DELAY=20)
#   2. does the chaos build find that race by itself?                                                        (nofix_chaos)
#   3. the product sources as a chaos build: both race screens, then every kernel / path parity test.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04r
step() {  # label lib seconds script [ENTRY_DELAY]
  GRAM_LIB=$2 ENTRY_DELAY=${5:-0} timeout -k 10 200 python tests/$4.py $3 > gpurun_out/r04r/$1_$4.log 2>&1; rc=$?
  echo "$1 $4 rc=$rc: $(tail -1 gpurun_out/r04r/$1_$4.log)"
  if [ $rc = 124 ] || [ $rc = 137 ]; then echo "timed out: stopping"; exit 1; fi
}
V=$PWD/gram_amd/csrc/variants
step nofix_delay20 $V/libgram_hip_nofix.so 20 stress_gemm_pp_x3 20
step nofix_plain $V/libgram_hip_nofix.so 30 stress_gemm_pp_x3 0
step nofix_chaos $V/libgram_hip_nofix_chaos.so 45 stress_gemm_pp_x3 0
step nofix_chaos $V/libgram_hip_nofix_chaos.so 30 stress_gemm_pp 0
CH=$PWD/gram_amd/csrc/libgram_hip_chaos.so
step chaos $CH 60 stress_gemm_pp_x3 0
step chaos $CH 60 stress_gemm_pp 0
GRAM_LIB=$CH timeout -k 10 700 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_split.py tests/test_gpu_path.py tests/test_gpu_configs.py -x -q -m gpu --durations=5 > gpurun_out/r04r/chaos_pytest.log 2>&1; rc=$?
tail -12 gpurun_out/r04r/chaos_pytest.log; exit $rc
