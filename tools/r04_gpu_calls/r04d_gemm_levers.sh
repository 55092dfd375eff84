#!/bin/bash
# r04d: the two GEMM levers VERDICT r03 item 3 names, as decisive experiments on one box (tests/bench_gemm_x3.py, all in one call):
#  (a) what a residual stream kept as its two pieces only could save: the fp32-residual tile-end epilogue WITHOUT its fp32 store
#      (ablation build ABL=32: same loads, same 16-bit copy, same partials) against the product kernel, O / FFN-out shapes
#  (b) the KV-bank GEMM's tile order: groups of gm m-tiles per XCD round (GRAM_GEMM_GROUPM) from n-fastest to 12 -- time AND in-kernel clock
set -o pipefail
mkdir -p gpurun_out/r04d
out=gpurun_out/r04d/gemm_levers.txt
: > $out
for rep in 1 2; do
  for lib in base abl32; do
    if [ $lib = base ]; then unset GRAM_LIB; else export GRAM_LIB=$PWD/gram_amd/csrc/libgram_hip_abl32.so; fi
    for shape in "enc o" "enc wo" "dec wo"; do
      echo -n "[a] rep$rep $lib " >> $out
      timeout -k 10 120 python tests/bench_gemm_x3.py --only "$shape" --iters 8 2>&1 | tail -1 >> $out || exit 1
    done
  done
done
unset GRAM_LIB
for rep in 1 2; do
  for gm in 0 2 4 6 8 12; do
    echo -n "[b] rep$rep GROUPM=$gm " >> $out
    GRAM_GEMM_GROUPM=$gm timeout -k 10 200 python tests/bench_gemm_x3.py --only "kv bank" --iters 4 2>&1 | tail -1 >> $out || exit 1
  done
done
cat $out
