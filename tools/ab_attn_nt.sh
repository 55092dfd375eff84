for rep in 1 2; do for v in base resnt; do if [ $v = base ]; then unset GRAM_LIB; else export GRAM_LIB=$PWD/gpurun_ab_$v.so; fi; timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-users 0 --no-extras > gpurun_out/ab_$v.json 2>/dev/null; python - <<E
import json
d=json.loads(open("gpurun_out/ab_$v.json").read().strip().splitlines()[-1])
k=d["kernel_ms_per_step"]
print("$v", round(d["value"],1), k["enc_attn"], k["dec_self_attn"], k["cross_attn"], k["gemm"])
E
done; done
