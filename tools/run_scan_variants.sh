# usage: bash tools/run_scan_variants.sh v1 v2 ...   (variants built by tools/build_scan_variant.sh)
cd $GRAFT_REPO_ROOT
for v in "$@"; do echo "== $v"; TAMTR_HIP_LIB=$GRAFT_REPO_ROOT/tam-tr_amd/csrc/variants/libtamtr_$v.so timeout -k 10 200 python tools/bench_kernels.py scan 2>&1 | grep scan; done
