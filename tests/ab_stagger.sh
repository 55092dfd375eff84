#!/bin/bash
# start-stagger sweep of the persistent kernel on the fp32-residual GEMMs (HBM-heavy epilogue)
set -e
for st in ${1:-0 8 16 24 32}; do
  echo "== stagger $st"
  timeout -k 10 120 python tests/bench_gemm.py --batch ${2:-2048} --variants=8 --only "enc o" --stagger $st --iters 10 ${3:-}
  timeout -k 10 120 python tests/bench_gemm.py --batch ${2:-2048} --variants=8 --only "enc wo" --stagger $st --iters 10 ${3:-}
done
