#!/bin/bash
set -e
for st in 0 3 6 12; do
  echo "== stagger $st"
  timeout -k 10 120 python tests/bench_gemm.py --batch 512 --variants 22 --only "enc qkv" --stagger $st
  timeout -k 10 120 python tests/bench_gemm.py --batch 512 --variants 22 --only "enc wi" --stagger $st
done
