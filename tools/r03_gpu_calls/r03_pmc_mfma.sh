#!/bin/bash
# Round 3: MFMA-busy counters of the two-piece GEMMs (rocprofv3 --pmc, its own pass, kernel names from the counter CSV).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d $R/gpurun_out/pmc_r03_mfma -o mfma -- python $R/tests/bench_gemm_x3.py --iters 2 > $R/gpurun_out/r03_pmc_mfma.log 2> $R/gpurun_out/r03_pmc_mfma.err
echo rc=$?; tail -9 $R/gpurun_out/r03_pmc_mfma.log; ls $R/gpurun_out/pmc_r03_mfma
