#!/bin/bash
# round 4, profile batch at HEAD: rocprofv3 kernel trace of the default bench -> steady-state summary; PMC passes of the GEMM for the bench line's traffic
set -o pipefail
bash tools/prof_step.sh r04 > gpurun_out/r4prof_step.log 2>&1; echo "prof_step rc=$?"; head -12 gpurun_out/prof_step_r04.txt | cut -c1-200
bash tools/pmc_gemm.sh > gpurun_out/r4prof_pmc_gemm.log 2>&1; echo "pmc_gemm rc=$?"; tail -6 gpurun_out/r4prof_pmc_gemm.log | cut -c1-200
