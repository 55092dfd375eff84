#!/bin/bash
# r04k: the precision instrument on a population with a trained T5's residual stream: outlier features x 300 / x 3 000 / x 30 000 (the last
# leaves the IEEE-half range by far), q x 4, 2 048 users each, f16x3 vs the fp32 reference on the GPU
mkdir -p gpurun_out/r04k
for F in 300 3000 30000; do
  timeout -k 10 500 python tests/precision_population.py --users 2048 --chunk 256 --modes f16x3 --sharpen 4 --outliers $F --out gpurun_out/r04k/precision_outliers_$F.json 2>&1 | grep "\[precision\] f16x3\|Error\|error" | tail -3
done
