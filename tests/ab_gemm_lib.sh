#!/bin/bash
# microbench A/B of alternative library builds: ab_gemm_lib.sh BATCH VARIANTS lib1.so lib2.so ...
b=$1; v=$2; shift; shift
cp gram_amd/csrc/libgram_hip.so /tmp/lib_orig.so
for l in "$@"; do
  if [ "$l" = "base" ]; then cp /tmp/lib_orig.so gram_amd/csrc/libgram_hip.so; else cp gram_amd/csrc/$l gram_amd/csrc/libgram_hip.so; fi
  echo "== $l"; timeout -k 10 300 python tests/bench_gemm.py --batch $b --variants $v 2>&1 | grep -v amdgpu.ids | head -4
done
cp /tmp/lib_orig.so gram_amd/csrc/libgram_hip.so
