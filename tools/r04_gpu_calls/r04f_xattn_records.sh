#!/bin/bash
# r04f: cross-attention read patterns at the bench shape (two pieces, 12 layers cycled as in a decode step), one box, interleaved:
#   planar bank + per-item kernel (round 3), planar + persistent-wave kernel, step-major 16-KB records (read side only) + per-item kernel
set -o pipefail
mkdir -p gpurun_out/r04f
for rep in 1 2 3; do
  echo -n "rep$rep planar/per-item   "; XA_PW_MIN=0 python tests/bench_xattn.py 2 4096 12 384 20 12 || exit 1
  echo -n "rep$rep planar/persistent "; XA_PW_MIN=1 python tests/bench_xattn.py 2 4096 12 384 20 12 || exit 1
  echo -n "rep$rep records/per-item  "; XA_RECORDS=1 python tests/bench_xattn.py 2 4096 12 384 20 12 || exit 1
done 2>&1 | tee gpurun_out/r04f/xattn_layouts.txt
