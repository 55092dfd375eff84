#!/bin/bash
# Round 3, GPU call 1: (1) kernel / split / path tests on the bf16 library after the out_scale plumbing, (2) the per-stage precision
# sensitivity sweep (bf16 pieces, rest at bf16x6) on both populations, (3) the f16 build's modes on both populations.
cd $GRAFT_REPO_ROOT
timeout -k 10 420 python -m pytest tests/test_gpu_split.py tests/test_gpu_path.py -x -q > gpurun_out/r03a_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r03a_tests.log
timeout -k 10 300 python tests/precision_population.py --users 2048 --sweep bf16x6 --out gpurun_out/r03a_sweep_bf16_plain.json > gpurun_out/r03a_sweep_plain.log 2>&1
echo "sweep plain rc=$?"; tail -2 gpurun_out/r03a_sweep_plain.log | cut -c1-300
timeout -k 10 300 python tests/precision_population.py --users 2048 --sweep bf16x6 --sharpen 4 --out gpurun_out/r03a_sweep_bf16_sharp4.json > gpurun_out/r03a_sweep_sharp.log 2>&1
echo "sweep sharp rc=$?"; tail -2 gpurun_out/r03a_sweep_sharp.log | cut -c1-300
export GRAM_LIB=$PWD/gram_amd/csrc/libgram_hip_f16.so
timeout -k 10 200 python tests/precision_population.py --users 4096 --modes f16,f16x3,f16x6 --out gpurun_out/r03a_f16_plain.json > gpurun_out/r03a_f16_plain.log 2>&1
echo "f16 plain rc=$?"; grep "^\[precision\] f16" gpurun_out/r03a_f16_plain.log | cut -c1-300
timeout -k 10 200 python tests/precision_population.py --users 4096 --modes f16,f16x3,f16x6 --sharpen 4 --out gpurun_out/r03a_f16_sharp4.json > gpurun_out/r03a_f16_sharp.log 2>&1
echo "f16 sharp rc=$?"; grep "^\[precision\] f16" gpurun_out/r03a_f16_sharp.log | cut -c1-300
