#!/bin/bash
# r04c: per-row power-of-two factors of the 16-bit residual copy: the new tests first, then the whole GPU suite, then the bench line
set -o pipefail
mkdir -p gpurun_out/r04c
timeout -k 10 400 python -m pytest tests/test_gpu_split.py tests/test_gpu_path.py tests/test_gpu_kernels.py -x -q -m gpu -k "row_factors or large_residual or overflow or folded or embed or rowops" > gpurun_out/r04c/pytest_new.log 2>&1; rc=$?; echo "new tests rc=$rc"; tail -15 gpurun_out/r04c/pytest_new.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --cpu-users 0 --no-extras > gpurun_out/r04c/bench.json 2> gpurun_out/r04c/bench.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r04c/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['roofline_gemm']['clock_ghz_in_kernel'])
"
timeout -k 10 800 python -m pytest tests -x -q -m gpu > gpurun_out/r04c/pytest_all.log 2>&1; echo "full suite rc=$?"; tail -8 gpurun_out/r04c/pytest_all.log
