// capi.hip - ABI version of libtamtr_hip.so (the kernels live in gate/msdeform/contrastive/selfattn/selscan/gemm_bf16/lsap/cpam/dwconv/ss2d_out/bn/conv3x3/imgaug.hip).
#include "common.h"

extern "C" int tamtr_abi_version(void) { return 24; }

