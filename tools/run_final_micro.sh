# kernel micro-benchmarks + PMC passes for the roofline kernel (outputs under gpurun_out/, copied into profiles/ by hand)
cd /root/repo
timeout -k 10 500 python tools/bench_kernels.py all > gpurun_out/micro_final.log 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /root/repo/gpurun_out/pmc_gemm_fetch -o p --output-format csv -- python3 /root/repo/tools/gemm_only.py > /root/repo/gpurun_out/pmc_gemm_fetch.log 2>&1 || exit 2
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /root/repo/gpurun_out/pmc_gemm_write -o p --output-format csv -- python3 /root/repo/tools/gemm_only.py > /root/repo/gpurun_out/pmc_gemm_write.log 2>&1 || exit 3
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d /root/repo/gpurun_out/pmc_gemm_sq -o p --output-format csv -- python3 /root/repo/tools/gemm_only.py > /root/repo/gpurun_out/pmc_gemm_sq.log 2>&1 || exit 4
cat /root/repo/gpurun_out/micro_final.log
