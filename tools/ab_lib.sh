#!/bin/bash
# In-bench A/B of alternative builds of the library on ONE box:  bash tools/ab_lib.sh name1 name2 ...  ("base" = the in-tree library,
# otherwise gpurun_ab_<name>.so in the repo root, loaded through GRAM_LIB); two repetitions, interleaved
# prints: name  users/s  enc_attn  dec_self_attn  cross_attn  gemm  (ms per step)
for rep in 1 2; do for v in "$@"; do
  if [ $v = base ]; then unset GRAM_LIB; else export GRAM_LIB=$PWD/gpurun_ab_$v.so; fi
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-users 0 --no-extras > gpurun_out/ab_lib.json 2>/dev/null
  python - <<E
import json
d=json.loads(open("gpurun_out/ab_lib.json").read().strip().splitlines()[-1])
k=d["kernel_ms_per_step"]
print("$v", round(d["value"],1), k["enc_attn"], k["dec_self_attn"], k["cross_attn"], k["gemm"])
E
done; done
