# final verification without the profiler: full GPU suite, smoke, default bench (with cpu_baseline)
cd /root/repo
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/final_tests.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/final_tests.log
tail -4 gpurun_out/final_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final_smoke.log 2>&1 || { tail -5 gpurun_out/final_smoke.log; exit 1; }
tail -2 gpurun_out/final_smoke.log
timeout -k 10 600 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err || { tail -5 gpurun_out/final_bench.err; exit 1; }
tail -2 gpurun_out/final_bench.err; cat gpurun_out/final_bench.json
