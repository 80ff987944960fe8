set -e
cd /root/repo
timeout -k 10 300 python -m pytest tests -m gpu -q -k "selective_scan or vss" > gpurun_out/t7.log 2>&1 || { tail -30 gpurun_out/t7.log; exit 1; }
tail -2 gpurun_out/t7.log
echo "== default (STG4 w2)"; timeout -k 10 200 python tools/bench_kernels.py scan 2>&1 | grep scan
for v in g2w3 g8w2 g4w3; do echo "== $v"; TAMTR_HIP_LIB=/root/repo/tam-tr_amd/csrc/variants/libtamtr_$v.so timeout -k 10 200 python tools/bench_kernels.py scan 2>&1 | grep scan; done
TAMTR_HIP_LIB=/root/repo/tam-tr_amd/csrc/variants/libtamtr_g2w3.so timeout -k 10 300 python -m pytest tests -m gpu -q -k "selective_scan or vss" 2>&1 | tail -2
