#!/bin/bash
# Round 4: the precision instrument on TRAINED weights (tests/precision_population.py --train-steps: Adam on a synthetic retrieval task through the
# oracle's own functions, then the fp32 reference arithmetic vs the HIP path on users of that task), T5-base, Beauty Trie, beam 20.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04v
STEPS=${1:-1200}; USERS=${2:-16384}
timeout -k 10 1100 python tests/precision_population.py --users $USERS --chunk 256 --train-steps $STEPS --modes f16x3,f16 --out gpurun_out/r04v/r04v_precision_trained_t5base.json > gpurun_out/r04v/trained.log 2>&1; rc=$?
grep "training step\|\[precision\] f16" gpurun_out/r04v/trained.log | tail -14; exit $rc
