cd /root/repo
timeout -k 10 500 python -m pytest tests -m gpu -q -k "cpam or full_model or selective_scan or vss or synchronis" > gpurun_out/t8.log 2>&1; tail -15 gpurun_out/t8.log
timeout -k 10 200 python tools/bench_kernels.py scan 2>&1 | grep scan
timeout -k 10 400 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/b16.log 2> gpurun_out/b16.err; tail -2 gpurun_out/b16.err; cut -c1-330 gpurun_out/b16.log
