#!/bin/bash
# Round 4, final build: a longer soak of the randomized race screens -- product library with stamps off / on / a late wave group, and the chaos library.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04u
run() {  # label seconds script  (env from the caller)
  timeout -k 10 $(( $2 + 120 )) python tests/$3.py $2 > gpurun_out/r04u/$1_$3.log 2>&1; rc=$?
  echo "$1 $3 rc=$rc: $(tail -1 gpurun_out/r04u/$1_$3.log)"
  if [ $rc != 0 ]; then exit 1; fi
}
STAMPS=0 run stamps0 150 stress_gemm_pp_x3
STAMPS=0 run stamps0 100 stress_gemm_pp
STAMPS=1 run stamps1 100 stress_gemm_pp_x3
ENTRY_DELAY=20 run late_group 100 stress_gemm_pp_x3
ENTRY_DELAY=20 run late_group 60 stress_gemm_pp
export GRAM_LIB=$PWD/gram_amd/csrc/libgram_hip_chaos.so
run chaos 150 stress_gemm_pp_x3
run chaos 100 stress_gemm_pp
