#!/bin/bash
# Round 3: the other BASELINE.json configs at full shape in f16x3 (single GPU; not the headline): config 3 (Toys, N = 21, S = 2 688, beam 20),
# config 4's shape (Sports Trie, ragged realistic N), config 5 (T5-large, Yelp Trie, N = 21, beam 50).
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --dataset Toys --passages 21 --batch 512 --steps 3 --warmup 1 --cpu-users 0 --no-extras > gpurun_out/r03cfg3_toys_n21.json 2> gpurun_out/r03cfg3.err; echo "cfg3 rc=$?"
timeout -k 10 300 python bench.py --dataset Sports --passages 21 --ragged --batch 512 --steps 3 --warmup 1 --cpu-users 0 --no-extras > gpurun_out/r03cfg4_sports_ragged.json 2> gpurun_out/r03cfg4.err; echo "cfg4 rc=$?"
timeout -k 10 400 python bench.py --backbone t5-large --dataset Yelp --passages 21 --beams 50 --batch 192 --steps 3 --warmup 1 --cpu-users 0 --no-extras > gpurun_out/r03cfg5_t5large_yelp_k50.json 2> gpurun_out/r03cfg5.err; echo "cfg5 rc=$?"
python - <<'P'
import json
for f in ("r03cfg3_toys_n21","r03cfg4_sports_ragged","r03cfg5_t5large_yelp_k50"):
    try:
        d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
        print(f, round(d["value"],1), "users/s", round(d["ms_per_step"],1), "ms/step", d["dtype"], {k:round(v,1) for k,v in d["kernel_ms_per_step"].items()}, "xattn GB/s", round(d["roofline_cross_attn"]["achieved"]), d["output_check"])
    except Exception as e:
        print(f, "failed", e)
P
