# final verification on the GPU box: full GPU suite, smoke, then the default bench (with cpu_baseline) under rocprofv3 --kernel-trace --stats
cd /root/repo
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/final_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/final_tests.log
tail -4 gpurun_out/final_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final_smoke.log 2>&1; tail -2 gpurun_out/final_smoke.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_final -o fin --output-format csv -- python3 /root/repo/bench.py > /root/repo/gpurun_out/final_bench.json 2> /root/repo/gpurun_out/final_bench.err
cd /root/repo; tail -3 gpurun_out/final_bench.err; cat gpurun_out/final_bench.json
