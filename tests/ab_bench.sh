#!/bin/bash
# A/B of the big-GEMM variant inside one box (device-to-device variance is ~6 %): usage ab_bench.sh BATCH V1 V2 ...
b=$1; shift
for v in "$@"; do
  GRAM_GEMM_BIG=$v timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-users 0 --batch $b 2>/dev/null > /tmp/ab.json || exit 1
  python -c "import json; d=json.load(open('/tmp/ab.json')); print('big=$v', round(d['value'],1), round(d['ms_per_step'],1), d['kernel_ms_per_step']['gemm'], round(d['roofline_gemm']['achieved'],1), round(d['roofline_cross_attn']['achieved'],1))"
done
