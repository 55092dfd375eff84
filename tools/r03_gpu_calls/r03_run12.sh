#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03p_b64 -o b64 -- python $R/tests/bench_small_batch.py --batches 64 --iters 10 > $R/gpurun_out/r03p_b64.json 2> $R/gpurun_out/r03p_b64.err
rm -f $R/gpurun_out/prof_r03p_b64/*kernel_trace.csv
cat $R/gpurun_out/r03p_b64.json
