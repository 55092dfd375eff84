#!/bin/bash
# Round 4: the "stamps behind a flag" wrong-result build again, on today's tree (tools/probes/pp_clock_variants/build.sh made the libraries):
# does it still fail, does padding the DMA asm statements (VALU-written SGPR -> VMEM hazard) or draining every counted wait change it?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04n
for v in bad_nop0 bad bad_drain bad_asm good_nop0; do
  export GRAM_LIB=$PWD/gram_amd/csrc/variants/libgram_hip_$v.so
  for s in stress_gemm_pp_x3 stress_gemm_pp; do
    timeout -k 10 150 python tests/$s.py 45 > gpurun_out/r04n/${v}_$s.log 2>&1; rc=$?
    echo "$v $s rc=$rc: $(tail -1 gpurun_out/r04n/${v}_$s.log)"
    if [ $rc = 124 ] || [ $rc = 137 ]; then echo "timed out: stopping"; exit 1; fi
  done
done
