#!/bin/bash
# Round 3, final verification: the GPU test suite, then the bench + rocprofv3 kernel stats of the same command.
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03z_gputests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r03z_gputests.log
bash tools/profile_round.sh r03z bench
echo "profile rc=$?"
tail -c 400 gpurun_out/r03z_bench.json
