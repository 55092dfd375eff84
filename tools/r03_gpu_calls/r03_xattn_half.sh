#!/bin/bash
# Round 3: half-slot cross-attention variants (R = 0; A/B library built with -DGRAM_XA_AB=1): parity, then timing at the bench shape
# (12 layers of bank cycled like a decode step), and in the bench.
cd $GRAFT_REPO_ROOT
export GRAM_LIB=$PWD/gram_amd/csrc/libgram_hip_xa.so
for v in 20 10; do
  GRAM_XA_VARIANT=$v timeout -k 10 200 python -m pytest tests/test_gpu_split.py tests/test_gpu_kernels.py -x -q -k "cross_attn" > gpurun_out/t_xa_$v.log 2>&1; echo "variant $v tests rc=$? $(tail -1 gpurun_out/t_xa_$v.log)"
done
for v in "" 20 10 40 21; do
  if [ -z "$v" ]; then unset GRAM_XA_VARIANT; else export GRAM_XA_VARIANT=$v; fi
  timeout -k 10 200 python tests/bench_xattn.py 2 4096 12 384 20 12 2>&1 | grep -v amdgpu.ids | sed "s/^/variant ${v:-default} /"
done | tee gpurun_out/r03_xattn_half_slot.txt
