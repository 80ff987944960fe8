"""ctypes binding of libtamtr_hip.so (C ABI declared in include/tamtr_hip.h).

There is deliberately NO fallback: if the shared library is missing or a tensor is not resident on an AMD GPU the
call raises TamtrHipError.  (A CPU restatement exists under oracle/ but it is test infrastructure and is never
imported from this package.)
"""
import ctypes
import os
from ctypes import c_float, c_int, c_longlong, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('TAMTR_HIP_LIB') or os.path.join(_HERE, 'csrc', 'libtamtr_hip.so')  # env override: kernel A/B experiments
ABI_VERSION = 34

F32, BF16 = 0, 1
_ERR = {-1: 'TAMTR_EINVAL (bad argument)', -2: 'TAMTR_EUNSUP (shape/dtype outside what the kernels are built for)',
        -3: 'TAMTR_ELAUNCH (HIP launch error)'}


class TamtrHipError(RuntimeError):
    pass


_P, _I, _F, _LL = c_void_p, c_int, c_float, c_longlong
_SIGS = {
    'tamtr_abi_version': [],
    'tamtr_maxsigmoid_gate_fwd': [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P],
    'tamtr_maxsigmoid_gate_bwd': [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P],
    'tamtr_msdeform_attn_fwd': [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    'tamtr_msdeform_attn_bwd': [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    'tamtr_msdeform_attn_bwd_sorted': [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _LL, _I, _P],
    'tamtr_contrastive_logits_fwd': [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    'tamtr_contrastive_logits_bwd': [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    'tamtr_contrastive_bwd_slabs': [_I],
    'tamtr_linear_bf16': [_P, _P, _P, _P, _I, _I, _I, _P],
    'tamtr_selfattn_fwd': [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    'tamtr_selfattn_bwd': [_P] * 11 + [_I, _I, _I, _I, _I, _I, _I, _I, _P],
    'tamtr_selective_scan_chunk': [],
    'tamtr_selective_scan_fwd': [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    'tamtr_selective_scan_bwd_slabs': [_I],
    'tamtr_selective_scan_row_sums': [],
    'tamtr_selective_scan_bwd': [_P] * 15 + [_I, _I, _I, _I, _I, _I, _P],
    'tamtr_selective_scan_dtproj_fwd': [_P] * 10 + [_I] * 8 + [_P],
    'tamtr_selective_scan_dtproj_bwd': [_P] * 17 + [_I] * 8 + [_P],
    'tamtr_lsap_assign': [_P, _P, _I, _I, _I, _P, _P, _P, _P],
    'tamtr_img_augment_u8': [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    'tamtr_cpam_fwd': [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    'tamtr_cpam_cl_fwd': [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    'tamtr_cpam_bwd': [_P] * 8 + [_I, _I, _I, _I, _I, _P],
    'tamtr_cpam_cl_bwd': [_P] * 10 + [_I, _I, _I, _I, _I, _P],
    'tamtr_dwconv_silu_cross_fwd': [_P, _LL, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    'tamtr_dwconv_tiles': [_I, _I],
    'tamtr_cross_merge_fwd': [_P, _P, _I, _I, _I, _I, _I, _P],
    'tamtr_cross_merge_bwd': [_P, _P, _I, _I, _I, _I, _I, _P],
    'tamtr_ln_gate_blocks': [_LL],
    'tamtr_bn_slices': [_I, _I],
    'tamtr_maxpool_out': [_I, _I, _I, _I],
    'tamtr_maxpool_fwd': [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    'tamtr_maxpool_bwd': [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    'tamtr_sum_n': [_P, _I, _P, _LL, _I, _P],
    'tamtr_colsum_blocks': [_LL],
    'tamtr_colsum_bf16': [_P, _P, _LL, _I, _P],
    'tamtr_fold_add': [_P, _P, _P, _P, _I, _LL, _I, _P],
    'tamtr_slab_sum_rows': [_P, _P, _I, _LL, _I, _P],
    'tamtr_graph_capture_census': [_P, _P, _I],
    'tamtr_box_refine_fwd': [_P, _P, _P, _LL, _P],
    'tamtr_box_refine_bwd': [_P, _P, _P, _P, _P, _LL, _P],
    'tamtr_bncl_act_seg_fwd': [_P, _P, _P, _P, _P, _P, _LL, _LL, _P, _P, _LL, _I, _F, _F, _I, _I, _P],
    'tamtr_bncl_act_seg_bwd': [_P, _LL, _LL, _P, _P, _P, _P, _P, _P, _P, _P, _LL, _I, _I, _I, _P],
    'tamtr_detr_blocks': [_I],
    'tamtr_detr_layers_fwd': [_P] * 8 + [_I] * 5 + [_P] * 5 + [_F, _F, _F, _P, _P],
    'tamtr_detr_layers_bwd': [_P] * 10 + [_I] * 5 + [_F, _F, _F, _P, _P, _P],
    'tamtr_detr_match_cost': [_P, _P, _P, _P, _LL, _I, _I, _F, _F, _F, _F, _F, _P, _P],
    'tamtr_xproj_dw_slices': [_I],
    'tamtr_xproj_fwd': [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    'tamtr_xproj_bwd_dx': [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    'tamtr_xproj_bwd_dw': [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    'tamtr_optim_chunk': [],
    'tamtr_optim_step': [_P] * 11 + [_I, _I, _P, _P, _P, _P, _I, _F, _F, _F, _F, _F, _I, _P],
    'tamtr_resample2': [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    'tamtr_cat_rows': [_P, _P, _P, _I, _P, _LL, _LL, _I, _P],
    'tamtr_copy_rows': [_P, _LL, _P, _LL, _LL, _I, _I, _P],
    'tamtr_relayout': [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    'tamtr_bncl_blocks': [_LL, _I, _I],
    'tamtr_bncl_act_fwd': [_P, _P, _P, _P, _P, _P, _P, _P, _P, _LL, _I, _F, _F, _I, _I, _P],
    'tamtr_bncl_stats': [_P, _P, _P, _P, _P, _LL, _I, _F, _F, _I, _P],
    'tamtr_bn_finalize': [_P, _P, _P, _P, _I, _I, _F, _F, _P],
    'tamtr_conv3x3_tiles': [_I, _I, _I],
    'tamtr_conv3x3_pack_weight': [_P, _P, _I, _I, _I, _P],
    'tamtr_conv3x3_cl_stats_fwd': [_P, _LL, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _P],
    'tamtr_maxsigmoid_gate_cl_fwd': [_P, _LL, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P],
    'tamtr_bncl2_act_fwd': [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _LL, _I, _F, _F, _I, _I, _P],
    'tamtr_bncl2_act_bwd': [_P, _LL, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _LL, _I, _I, _I, _P],
    'tamtr_bncl_act_bwd': [_P, _LL, _P, _P, _P, _P, _P, _P, _P, _P, _LL, _I, _I, _I, _P],
    'tamtr_bn_act_fwd': [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _I, _I, _P],
    'tamtr_bn_act_bwd': [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    'tamtr_layernorm_fwd': [_P, _P, _P, _P, _P, _LL, _I, _F, _I, _P],
    'tamtr_layernorm_bwd': [_P, _P, _P, _P, _P, _P, _LL, _I, _I, _P],
    'tamtr_ln_gate_fwd': [_P, _P, _LL, _P, _P, _P, _P, _LL, _I, _F, _I, _P],
    'tamtr_ln_gate_bwd': [_P, _P, _P, _LL, _P, _P, _P, _P, _P, _P, _LL, _I, _I, _P],
    'tamtr_dwconv_silu_cross_bwd': [_P, _P, _LL, _P, _P, _P, _LL, _P, _I, _I, _I, _I, _I, _I, _P],
}
EXPORTS = tuple(_SIGS)
_lib = None


def lib():
    """Load (once) and return the ctypes handle; raise if the HIP library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TamtrHipError(f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                                '(hipcc --offload-arch=gfx950). There is no CPU fallback.')
        h = ctypes.CDLL(LIB_PATH)
        for name, args in _SIGS.items():
            fn = getattr(h, name)  # AttributeError here == header/library mismatch
            fn.argtypes = args
            fn.restype = c_int
        if h.tamtr_abi_version() != ABI_VERSION:
            raise TamtrHipError(f'libtamtr_hip.so ABI {h.tamtr_abi_version()} != expected {ABI_VERSION}: rebuild')
        _lib = h
    return _lib


def call(name, *args):
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise TamtrHipError(f'{name} failed: {_ERR.get(rc, rc)}')


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise TamtrHipError('TAM-TR HIP ops need tensors resident on an MI355X (got a CPU tensor); there is no CPU fallback')


def dtype_code(t):
    import torch
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TamtrHipError(f'unsupported activation dtype {t.dtype} (float32 and bfloat16 kernels are built)')


def stream_ptr():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)
