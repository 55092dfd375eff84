#!/bin/bash
# Round 4, final measurements (behind the green full suite of r04j / r04z_tests): the default bench line, rocprofv3 kernel stats of the same
# command, the two PMC passes (HBM traffic), the precision tests with their numbers, the other BASELINE configs at full shape, a B = 1 trace.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04z
bash tools/profile_round.sh r04z all > gpurun_out/r04z/profile_round.log 2>&1; echo "profile rc=$?"; tail -3 gpurun_out/r04z/profile_round.log
python tools/pmc_traffic.py gpurun_out/pmc_r04z_fetch gpurun_out/pmc_r04z_write 4096 f16x3 > gpurun_out/r04z/r04z_pmc_traffic.json 2> gpurun_out/r04z/pmc.err; echo "pmc rc=$?"
timeout -k 10 600 python -m pytest tests/test_gpu_precision.py -x -q -s -m gpu 2>&1 | grep "\[precision\]\|passed\|failed" > gpurun_out/r04z/r04z_precision.txt; tail -2 gpurun_out/r04z/r04z_precision.txt
timeout -k 10 300 python bench.py --dataset Toys --passages 21 --batch 512 --steps 3 --warmup 1 --cpu-users 0 --no-extras > gpurun_out/r04z/r04cfg3_toys_n21.json 2> gpurun_out/r04z/cfg3.err; echo "cfg3 rc=$?"
timeout -k 10 300 python bench.py --dataset Sports --passages 21 --ragged --batch 512 --steps 3 --warmup 1 --cpu-users 0 --no-extras > gpurun_out/r04z/r04cfg4_sports_ragged.json 2> gpurun_out/r04z/cfg4.err; echo "cfg4 rc=$?"
timeout -k 10 400 python bench.py --backbone t5-large --dataset Yelp --passages 21 --beams 50 --batch 192 --steps 3 --warmup 1 --cpu-users 0 --no-extras > gpurun_out/r04z/r04cfg5_t5large_yelp_k50.json 2> gpurun_out/r04z/cfg5.err; echo "cfg5 rc=$?"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r04z_b1 -o r04z_b1 -- python $R/tests/bench_small_batch.py --batches 1 --iters 20 > $R/gpurun_out/r04z/b1.json 2> $R/gpurun_out/r04z/b1.err; echo "b1 trace rc=$?"
rm -f $R/gpurun_out/prof_r04z_b1/*/*kernel_trace.csv $R/gpurun_out/prof_r04z_b1/*kernel_trace.csv
cd $R
python - <<'P'
import json
for f in ("r04cfg3_toys_n21","r04cfg4_sports_ragged","r04cfg5_t5large_yelp_k50"):
    try:
        d=json.loads(open("gpurun_out/r04z/%s.json"%f).read().strip().splitlines()[-1])
        print(f, round(d["value"],1), "users/s", round(d["ms_per_step"],1), "ms/step", d["dtype"], {k:round(v,1) for k,v in d["kernel_ms_per_step"].items()}, "xattn GB/s", round(d["roofline_cross_attn"]["achieved"]), d["output_check"])
    except Exception as e:
        print(f, "failed", e)
d=json.loads(open("gpurun_out/r04z_bench.json").read().strip().splitlines()[-1])
print("BENCH", round(d["value"],1), round(d["ms_per_step"],1), d["kernel_ms_per_step"], {k:d["roofline"][k] for k in ("achieved","frac","achieved_mfma_executed","clock_ghz_in_kernel","frac_mfma_executed_at_measured_clock")}, d["roofline_cross_attn"]["achieved"], d["roofline_cross_attn"]["frac"])
print("extras", {k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk!="runs"}) for k,v in d.get("extras",{}).items()})
print("cpu", d.get("cpu_baseline"))
P
