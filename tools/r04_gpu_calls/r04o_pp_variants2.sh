#!/bin/bash
# Round 4: localising the counted wait that the "stamps behind a flag" build outruns (r04n: every counted vmcnt a full drain = clean), and
# what a wrong tile holds (tools/probes/pp_clock_variants/forensics.py).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04o
export GRAM_LIB=$PWD/gram_amd/csrc/variants/libgram_hip_bad_nop0.so
timeout -k 10 200 python tools/probes/pp_clock_variants/forensics.py 90 > gpurun_out/r04o/forensics_bad_nop0.log 2>&1; rc=$?
echo "forensics rc=$rc"; tail -25 gpurun_out/r04o/forensics_bad_nop0.log
if [ $rc = 124 ] || [ $rc = 137 ]; then exit 1; fi
for v in bad_drain_entry bad_drain_slot bad_vm10 bad_tend_drain; do
  export GRAM_LIB=$PWD/gram_amd/csrc/variants/libgram_hip_$v.so
  for s in stress_gemm_pp_x3 stress_gemm_pp; do
    timeout -k 10 150 python tests/$s.py 45 > gpurun_out/r04o/${v}_$s.log 2>&1; rc=$?
    echo "$v $s rc=$rc: $(tail -1 gpurun_out/r04o/${v}_$s.log)"
    if [ $rc = 124 ] || [ $rc = 137 ]; then echo "timed out: stopping"; exit 1; fi
  done
done
