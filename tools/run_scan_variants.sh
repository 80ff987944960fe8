# usage: bash tools/run_scan_variants.sh v1 v2 ...   (variants built by tools/build_scan_variant.sh); tests run on the default library
cd /root/repo
timeout -k 10 400 python -m pytest tests -m gpu -q -k "selective_scan or vss" > gpurun_out/t_scan.log 2>&1 || { tail -40 gpurun_out/t_scan.log; exit 1; }
tail -2 gpurun_out/t_scan.log
echo "== default"; timeout -k 10 200 python tools/bench_kernels.py scan 2>&1 | grep scan
for v in "$@"; do echo "== $v"; TAMTR_HIP_LIB=/root/repo/tam-tr_amd/csrc/variants/libtamtr_$v.so timeout -k 10 200 python tools/bench_kernels.py scan 2>&1 | grep scan; done
