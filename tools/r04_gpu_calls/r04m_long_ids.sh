#!/bin/bash
# Round 4: GRAM_MAX_DEC_LEN 32 -> 64 (the "term" id type's max_length = 50): the new parity cases, the search-step kernel tests (its sequence /
# ancestor staging moved to dynamic LDS), then the whole GPU suite.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04m
timeout -k 10 300 python -m pytest tests/test_gpu_path.py -x -q -s -m gpu -k "long_ids or max_length_50" > gpurun_out/r04m/new_tests.log 2>&1; rc=$?; tail -5 gpurun_out/r04m/new_tests.log; [ $rc = 0 ] || exit $rc
timeout -k 10 300 python -m pytest tests/test_gpu_split.py tests/test_gpu_kernels.py -x -q -m gpu -k "dec_self_attn or beam or trie or sparse or greedy" > gpurun_out/r04m/kernels.log 2>&1; rc=$?; tail -3 gpurun_out/r04m/kernels.log; [ $rc = 0 ] || exit $rc
timeout -k 10 800 python -m pytest tests -x -q -m gpu --durations=8 > gpurun_out/r04m/suite.log 2>&1; rc=$?; tail -14 gpurun_out/r04m/suite.log; exit $rc
