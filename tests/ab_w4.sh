#!/bin/bash
# microbench: 8-wave persistent (8) vs 4-wave 128x128-wave-tile persistent (13) + ablations
set -e
B=${1:-512}
for s in "enc qkv" "enc o" "enc wi" "enc wo"; do
  timeout -k 10 120 python tests/bench_gemm.py --batch $B --variants ${2:-8,13,15,16,10,11} --only "$s" --fullcheck
done
