#!/bin/bash
# A/B an environment switch inside one box: ab_env.sh BATCH VAR v1 v2 ...
b=$1; var=$2; shift; shift
for v in "$@"; do
  env $var=$v timeout -k 10 400 python bench.py --steps 3 --warmup 1 --cpu-users 0 --batch $b 2>/dev/null > /tmp/ab.json || exit 1
  python -c "import json; d=json.load(open('/tmp/ab.json')); print('$var=$v', round(d['value'],1), round(d['ms_per_step'],1), d['kernel_ms_per_step'], round(d['roofline_gemm']['achieved'],1), round(d['roofline_cross_attn']['achieved'],1))"
done
