#!/bin/bash
# One profiling pass of the default bench on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r01e
# writes gpurun_out/<tag>_bench.json, gpurun_out/prof_<tag>/ (kernel-trace stats) and gpurun_out/pmc_<tag>_{fetch,write}/
# (HBM counters, one rocprofv3 pass each, never combined with a trace domain); tools/pmc_traffic.py turns the
# counter CSVs into profiles/<tag>_pmc_traffic.json (python tools/pmc_traffic.py gpurun_out/pmc_<tag>_fetch gpurun_out/pmc_<tag>_write 4096 bf16x3).
set -e
TAG=${1:-round}
WHAT=${2:-all}   # all | bench (bench + kernel stats) | pmc (the two counter passes)
R=$GRAFT_REPO_ROOT
cd $R
if [ $WHAT != pmc ]; then
timeout -k 10 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo BENCH_DONE
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG} -o ${TAG} -- python $R/bench.py --steps 3 --warmup 1 --cpu-users 0 --no-prof --no-extras > $R/gpurun_out/${TAG}_prof_bench.json 2> $R/gpurun_out/${TAG}_prof.err
echo STATS_DONE
fi
if [ $WHAT = bench ]; then exit 0; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_${TAG}_fetch -o ${TAG} -- python $R/bench.py --steps 1 --warmup 1 --cpu-users 0 --no-prof --no-extras > /dev/null 2> $R/gpurun_out/${TAG}_pmc_fetch.err
echo FETCH_DONE
timeout -k 10 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_${TAG}_write -o ${TAG} -- python $R/bench.py --steps 1 --warmup 1 --cpu-users 0 --no-prof --no-extras > /dev/null 2> $R/gpurun_out/${TAG}_pmc_write.err
echo WRITE_DONE
rm -f $R/gpurun_out/prof_${TAG}/*kernel_trace.csv   # (large; the stats CSV is what is kept)
ls $R/gpurun_out/pmc_${TAG}_fetch $R/gpurun_out/pmc_${TAG}_write
