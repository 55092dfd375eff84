#!/bin/bash
# microbench of the persistent GEMM variants (+ ablations) on the encoder shapes
#   ab_gemm_variants.sh [batch] [variants] [shapes separated by ';']
set -e
B=${1:-512}
V=${2:-8,22,23}
SH=${3:-"enc qkv;enc o;enc wi;enc wo"}
IFS=';' read -ra SHAPES <<< "$SH"
for s in "${SHAPES[@]}"; do
  ${DRY:+echo} timeout -k 10 120 python tests/bench_gemm.py --batch "$B" --variants "$V" --only "$s" --fullcheck
done
