set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python bench.py > gpurun_out/r01d_bench.json 2> gpurun_out/r01d_bench.err
echo BENCH_DONE
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01d -o r01d -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-users 0 --no-prof > $GRAFT_REPO_ROOT/gpurun_out/r01d_prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r01d_prof.err
echo PROF_DONE
ls -la $GRAFT_REPO_ROOT/gpurun_out/prof_r01d/* | head
