#!/bin/bash
# r04a: new GPU tests (item index, non-finite flags, rebatched runner) + the bench line with extras.e2e_runner
set -o pipefail
mkdir -p gpurun_out/r04a
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_runner.py -x -q -m gpu -k "nonfinite or item_index or beam_search or runner" > gpurun_out/r04a/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r04a/pytest.log
tail -5 gpurun_out/r04a/pytest.log
timeout -k 10 500 python bench.py --steps 5 --warmup 2 --cpu-users 0 > gpurun_out/r04a/bench.json 2> gpurun_out/r04a/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04a/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
print(json.dumps(d.get('extras',{}).get('e2e_runner'), indent=1))
PY
