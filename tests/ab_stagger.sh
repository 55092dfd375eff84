#!/bin/bash
# start-stagger sweep of the ping-pong kernel on the fp32-residual GEMMs (bandwidth-heavy tile-end epilogue)
set -e
for st in ${1:-0 8 16 24}; do
  echo "== stagger $st"
  timeout -k 10 120 python tests/bench_gemm.py --batch ${2:-2048} --variants=22,22 --only "enc o" --stagger $st --iters 10
  timeout -k 10 120 python tests/bench_gemm.py --batch ${2:-2048} --variants=22,22 --only "enc wo" --stagger $st --iters 10
done
