#!/bin/bash
# r04b: passage harvest + lazy Trie: the runner/cache GPU tests, then the bench line with extras.e2e_runner
set -o pipefail
mkdir -p gpurun_out/r04b
timeout -k 10 700 python -m pytest tests/test_gpu_runner.py tests/test_gpu_configs.py -x -q -m gpu -k "runner or passage or flow or compaction" > gpurun_out/r04b/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r04b/pytest.log
tail -5 gpurun_out/r04b/pytest.log
timeout -k 10 500 python bench.py --steps 5 --warmup 2 --cpu-users 0 > gpurun_out/r04b/bench.json 2> gpurun_out/r04b/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04b/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
print(json.dumps(d.get('extras',{}).get('e2e_runner'), indent=1))
PY
