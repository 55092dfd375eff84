#!/bin/bash
# Round 3, GPU call: (1) tile-end epilogue without its global stores (ABL 24) vs the full kernel (ABL 8): is the epilogue bound by its
# stores or by its own latency chain?  (2) XCD-phase stagger.  (3) kernel trace of one-user generate() calls.
cd $GRAFT_REPO_ROOT
for v in abl8 abl24; do
  echo "== $v"
  GRAM_LIB=$PWD/gram_amd/csrc/libgram_hip_$v.so timeout -k 10 200 python tests/bench_gemm_x3.py --iters 4 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r03f_gemm_x3_nostore.txt 2>&1
cat gpurun_out/r03f_gemm_x3_nostore.txt
for s in 2 4 8; do
  echo "== XCD-phase stagger $s"
  GRAM_GEMM_STAGGER_XCD=1 GRAM_GEMM_STAGGER=$s timeout -k 10 200 python tests/bench_gemm_x3.py --iters 4 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r03f_gemm_x3_stagger_xcd.txt 2>&1
cat gpurun_out/r03f_gemm_x3_stagger_xcd.txt
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03f_b1 -o b1 -- python $R/tests/bench_small_batch.py --batches 1 --iters 20 > $R/gpurun_out/r03f_b1.json 2> $R/gpurun_out/r03f_b1.err
rm -f $R/gpurun_out/prof_r03f_b1/*kernel_trace.csv
cat $R/gpurun_out/r03f_b1.json
