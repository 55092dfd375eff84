#!/bin/bash
# Round 3, GPU call: ablations of the encoder attention (make EABL=n builds).
cd $GRAFT_REPO_ROOT
echo "== product"; timeout -k 10 120 python tests/bench_enc_attn.py 2>&1 | grep -v amdgpu.ids
for v in 1 2 4 7 8 16 32 64 72 127; do
  echo "== eabl$v"
  GRAM_LIB=$PWD/gram_amd/csrc/libgram_hip_eabl$v.so timeout -k 10 120 python tests/bench_enc_attn.py 2>&1 | grep -v amdgpu.ids
done
