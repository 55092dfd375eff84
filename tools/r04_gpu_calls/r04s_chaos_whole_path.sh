#!/bin/bash
# Round 4: the whole path at the bench's shape, product library vs chaos library, bit for bit (tests/chaos_equal.py), + the suite's new chaos test.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s
timeout -k 10 500 python tests/chaos_equal.py 2048 0 > gpurun_out/r04s/equal_2048.log 2>&1; rc=$?; tail -3 gpurun_out/r04s/equal_2048.log; [ $rc = 0 ] || exit $rc
timeout -k 10 400 python tests/chaos_equal.py 1024 1 > gpurun_out/r04s/equal_1024_ragged.log 2>&1; rc=$?; tail -3 gpurun_out/r04s/equal_1024_ragged.log; [ $rc = 0 ] || exit $rc
timeout -k 10 200 python tests/chaos_equal.py 4 0 > gpurun_out/r04s/equal_4.log 2>&1; rc=$?; tail -3 gpurun_out/r04s/equal_4.log; [ $rc = 0 ] || exit $rc
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "race_screen" > gpurun_out/r04s/race_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r04s/race_tests.log; exit $rc
