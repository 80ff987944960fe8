cd $GRAFT_REPO_ROOT
for v in "$@"; do echo "== $v"; TAMTR_HIP_LIB=$GRAFT_REPO_ROOT/tam-tr_amd/csrc/variants/libtamtr_$v.so timeout -k 10 200 python tools/bench_kernels.py gemm 2>&1 | grep "linear_bf16"; done
