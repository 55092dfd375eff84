#!/bin/bash
# Round 3, GPU call: split2 helpers in the epilogues + encoder attention (bias as accumulator init, mask skip): tests, timing.
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_split.py tests/test_gpu_kernels.py tests/test_gpu_path.py -x -q 2>&1 | tail -6
echo "tests rc=$?"
timeout -k 10 200 python tests/bench_gemm_x3.py --iters 4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03l_gemm_x3_split2.txt
timeout -k 10 400 python bench.py --cpu-users 0 --no-extras > gpurun_out/r03l_bench.json 2> gpurun_out/r03l_bench.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/r03l_bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","dtype")}, d.get("kernel_ms_per_step"))
P
