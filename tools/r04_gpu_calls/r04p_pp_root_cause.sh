#!/bin/bash
# Round 4: the root cause of the "stamps behind a flag" build, proved both ways (tools/probes/pp_clock_variants/build.sh):
#   delay        = the round-4 kernel + group 1 held back ~5 us before its first W read  -> wrong first tiles
#   fixed_delay  = the same with the prologue barrier                                     -> clean
#   fixed_bad    = the conditional-stamp build with the prologue barrier                  -> clean
# then the product library: both race screens, stamps off and on.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04p
run() {  # name seconds [env...]
  local v=$1 secs=$2
  for s in stress_gemm_pp_x3 stress_gemm_pp; do
    timeout -k 10 200 python tests/$s.py $secs > gpurun_out/r04p/${v}_$s.log 2>&1; rc=$?
    echo "$v $s rc=$rc: $(tail -1 gpurun_out/r04p/${v}_$s.log)"
    if [ $rc = 124 ] || [ $rc = 137 ]; then echo "timed out: stopping"; exit 1; fi
  done
}
for v in delay fixed_delay fixed_bad fixed_bad_nop0; do
  export GRAM_LIB=$PWD/gram_amd/csrc/variants/libgram_hip_$v.so
  run $v 45
done
export GRAM_LIB=$PWD/gram_amd/csrc/variants/libgram_hip_delay.so
timeout -k 10 200 python tools/probes/pp_clock_variants/forensics.py 60 > gpurun_out/r04p/forensics_delay.log 2>&1; echo "forensics(delay) rc=$?"; tail -8 gpurun_out/r04p/forensics_delay.log
unset GRAM_LIB
STAMPS=0 run product_stamps0 60
STAMPS=1 run product_stamps1 60
