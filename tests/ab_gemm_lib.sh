#!/bin/bash
# microbench A/B of alternative library builds in ONE box: ab_gemm_lib.sh BATCH VARIANTS "shape;shape" lib1.so lib2.so ...
# (libraries live in gram_amd/csrc/, "base" = the library that was there)
b=$1; v=$2; sh=$3; shift; shift; shift
cp gram_amd/csrc/libgram_hip.so /tmp/lib_orig.so
for l in "$@"; do
  if [ "$l" = "base" ]; then cp /tmp/lib_orig.so gram_amd/csrc/libgram_hip.so; else cp gram_amd/csrc/$l gram_amd/csrc/libgram_hip.so; fi
  echo "== $l"; bash tests/ab_gemm_variants.sh $b $v "$sh" 2>&1 | grep -v amdgpu.ids
done
cp /tmp/lib_orig.so gram_amd/csrc/libgram_hip.so
