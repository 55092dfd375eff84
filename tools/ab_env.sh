#!/bin/bash
# In-bench A/B of an environment switch on ONE box:  bash tools/ab_env.sh VAR v1 v2 ...   (two repetitions, interleaved)
# prints: value  users/s  enc_attn  dec_self_attn  cross_attn  gemm  (ms per step)
VAR=$1; shift
for rep in 1 2; do for v in "$@"; do
  env $VAR=$v timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-users 0 --no-extras > gpurun_out/ab_env.json 2>/dev/null
  python - <<E
import json
d=json.loads(open("gpurun_out/ab_env.json").read().strip().splitlines()[-1])
k=d["kernel_ms_per_step"]
print("$VAR=$v", round(d["value"],1), k["enc_attn"], k["dec_self_attn"], k["cross_attn"], k["gemm"])
E
done; done
