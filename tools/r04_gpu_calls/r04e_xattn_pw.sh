#!/bin/bash
# r04e: the persistent-wave cross-attention: bit-identity tests, then the bench A/B (GRAM_XA_PW_MIN=0 = the per-item kernel) on one box
set -o pipefail
mkdir -p gpurun_out/r04e
timeout -k 10 500 python -m pytest tests/test_gpu_split.py tests/test_gpu_kernels.py -x -q -m gpu -k "cross_attn" > gpurun_out/r04e/pytest.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -12 gpurun_out/r04e/pytest.log
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do for v in 0 8192; do
  GRAM_XA_PW_MIN=$v timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-users 0 --no-extras > gpurun_out/r04e/b.json 2>gpurun_out/r04e/b.err || { tail -5 gpurun_out/r04e/b.err; exit 1; }
  python - <<E
import json
d=json.loads(open("gpurun_out/r04e/b.json").read().strip().splitlines()[-1])
k=d["kernel_ms_per_step"]; x=d["roofline_cross_attn"]
print("PW_MIN=$v", round(d["value"],1), "xattn ms", k["cross_attn"], "GB/s", round(x["achieved"],1), "frac", round(x["frac"],4), "probe", x.get("stream_read_gbs_this_box"), "gemm", k["gemm"])
E
done; done 2>&1 | tee gpurun_out/r04e/ab.txt
