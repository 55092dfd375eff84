#!/bin/bash
# Round 3, GPU call 2: bench in the new default mode, then the GPU test suite.
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py --cpu-users 4 > gpurun_out/r03b_bench.json 2> gpurun_out/r03b_bench.err
echo "bench rc=$?"; python - <<'E'
import json
try:
    d=json.loads(open("gpurun_out/r03b_bench.json").read().strip().splitlines()[-1])
    print({k:d[k] for k in ("value","ms_per_step","dtype")}, d["config"].get("cross_attn_launches_per_generate"), d.get("kernel_ms_per_step"))
    print("gemm", {k:round(v,3) if isinstance(v,float) else v for k,v in d["roofline_gemm"].items() if k in ("achieved","achieved_mfma_executed","frac_mfma_executed")})
    print("xattn", {k:round(v,3) if isinstance(v,float) else v for k,v in d["roofline_cross_attn"].items() if k in ("achieved","frac","avg_launch_us")})
    print("extras", d.get("extras"))
except Exception as e:
    print("no bench line", e); print(open("gpurun_out/r03b_bench.err").read()[-3000:])
E
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -s > gpurun_out/r03b_gputests.log 2>&1
echo "tests rc=$?"; grep -E "passed|failed|error" gpurun_out/r03b_gputests.log | tail -5
