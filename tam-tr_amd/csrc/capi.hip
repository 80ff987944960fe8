// capi.hip - ABI version + entry points whose kernels are not built yet return TAMTR_EUNSUP (never a silent fallback).
#include "common.h"

extern "C" int tamtr_abi_version(void) { return 1; }

// ---- placeholders until the kernels land (loud: EUNSUP raises TamtrHipError on the Python side) ----
extern "C" int tamtr_linear_bf16(const void*, const void*, const float*, void*, int, int, int, void*) { return TAMTR_EUNSUP; }
extern "C" int tamtr_selfattn_fwd(const void*, const void*, const void*, const uint8_t*, void*, float*, int, int, int, int, int, void*) { return TAMTR_EUNSUP; }
extern "C" int tamtr_selfattn_bwd(const void*, const void*, const void*, const void*, const void*, const float*, const uint8_t*, void*, void*, void*, int, int, int, int, int, void*) { return TAMTR_EUNSUP; }
extern "C" int tamtr_selective_scan_chunk(void) { return 256; }
extern "C" int tamtr_selective_scan_fwd(const float*, const float*, const float*, const float*, const float*, const float*, const float*, float*, float*, int, int, int, int, int, void*) { return TAMTR_EUNSUP; }
extern "C" int tamtr_selective_scan_bwd(const float*, const float*, const float*, const float*, const float*, const float*, const float*, const float*, const float*, float*, float*, float*, float*, float*, float*, float*, int, int, int, int, int, void*) { return TAMTR_EUNSUP; }
