#!/bin/bash
# A/B alternative builds of libgram_hip.so inside one box: ab_lib.sh BATCH lib1.so lib2.so ...
b=$1; shift
cp gram_amd/csrc/libgram_hip.so /tmp/lib_orig.so
for l in "$@"; do
  cp gram_amd/csrc/$l gram_amd/csrc/libgram_hip.so
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-users 0 --batch $b 2>/dev/null > /tmp/ab.json || exit 1
  python -c "import json; d=json.load(open('/tmp/ab.json')); print('$l', round(d['value'],1), round(d['ms_per_step'],1), d['kernel_ms_per_step'], round(d['roofline_cross_attn']['achieved'],1))"
done
cp /tmp/lib_orig.so gram_amd/csrc/libgram_hip.so
