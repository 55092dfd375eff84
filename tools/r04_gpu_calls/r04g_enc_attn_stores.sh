#!/bin/bash
# r04g: encoder attention with its output rows staged through the (dead) K tiles' LDS -> whole-line stores.  Parity tests, then the
# micro-benchmark and the bench against the previous kernel (gpurun_ab_encbase.so = this tree with round 3's enc_attn.hip), interleaved
set -o pipefail
mkdir -p gpurun_out/r04g
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_split.py tests/test_gpu_path.py -x -q -m gpu -k "enc_self_attn or encoder or generate" > gpurun_out/r04g/pytest.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/r04g/pytest.log
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
  echo -n "base: "; GRAM_LIB=$PWD/gpurun_ab_encbase.so python tests/bench_enc_attn.py 2>/dev/null | tail -1
  echo -n "new:  "; python tests/bench_enc_attn.py 2>/dev/null | tail -1
done | tee gpurun_out/r04g/enc_attn_ab.txt
bash tools/ab_lib.sh encbase base | tee -a gpurun_out/r04g/enc_attn_ab.txt
