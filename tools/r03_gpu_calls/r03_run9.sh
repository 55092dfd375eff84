#!/bin/bash
# Round 3, GPU call: precision evidence in the headline mode (f16x3) at 16 384 users on the three populations, and the per-stage
# sensitivity of IEEE-half pieces (one stage at ONE piece, the rest at f16x3): profiles/r03k_*.
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tests/precision_population.py --users 16384 --modes f16x3 --out gpurun_out/r03k_precision_f16x3_16384users.json > gpurun_out/r03k_plain.log 2>&1
echo "plain rc=$?"; grep "^\[precision\]" gpurun_out/r03k_plain.log | cut -c1-400
timeout -k 10 300 python tests/precision_population.py --users 16384 --modes f16x3 --sharpen 4 --out gpurun_out/r03k_precision_f16x3_16384users_sharp4.json > gpurun_out/r03k_sharp.log 2>&1
echo "sharp rc=$?"; grep "^\[precision\]" gpurun_out/r03k_sharp.log | cut -c1-400
timeout -k 10 200 python tests/precision_population.py --users 4096 --modes f16x3 --ragged --out gpurun_out/r03k_precision_f16x3_4096users_ragged.json > gpurun_out/r03k_ragged.log 2>&1
echo "ragged rc=$?"; grep "^\[precision\]" gpurun_out/r03k_ragged.log | cut -c1-400
timeout -k 10 300 python tests/precision_population.py --users 2048 --sweep f16x3 --out gpurun_out/r03k_precision_sweep_f16_stages_2048users.json > gpurun_out/r03k_sweep_plain.log 2>&1
echo "sweep plain rc=$?"; tail -3 gpurun_out/r03k_sweep_plain.log | cut -c1-300
timeout -k 10 300 python tests/precision_population.py --users 2048 --sweep f16x3 --sharpen 4 --out gpurun_out/r03k_precision_sweep_f16_stages_2048users_sharp4.json > gpurun_out/r03k_sweep_sharp.log 2>&1
echo "sweep sharp rc=$?"; tail -3 gpurun_out/r03k_sweep_sharp.log | cut -c1-300
