#!/bin/bash
# Round 4: the whole GPU suite on the prologue-barrier build (new: late-group race screens, 48-token ids, max_length 50), then the
# two-piece GEMM micro-benchmark (did the extra barrier / the s_nop 3 padding cost anything?).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04q
timeout -k 10 800 python -m pytest tests -x -q -m gpu --durations=6 > gpurun_out/r04q/suite.log 2>&1; rc=$?; tail -12 gpurun_out/r04q/suite.log; [ $rc = 0 ] || exit $rc
timeout -k 10 300 python tests/bench_gemm_x3.py --iters 5 > gpurun_out/r04q/bench_gemm_x3.log 2>&1; rc=$?; tail -14 gpurun_out/r04q/bench_gemm_x3.log; exit $rc
