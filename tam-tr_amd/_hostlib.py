"""ctypes binding of libtamtr_host.so (C ABI declared in include/tamtr_host.h): the data path's 8-bit image kernels.

No fallback: a missing library raises (build it with `make -C tam-tr_amd/csrc` or __graft_entry__.build()).  ctypes releases
the GIL around each call, so DataLoader workers and threads run them in parallel.
"""
import ctypes
import os
from ctypes import c_int, c_longlong, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libtamtr_host.so')
ABI_VERSION = 1

_P, _I, _LL = c_void_p, c_int, c_longlong
_SIGS = {
    'tamtr_host_abi_version': [],
    'tamtr_resize_linear_u8': [_P, _I, _I, _I, _P, _I, _I],
    'tamtr_warp_affine_u8': [_P, _I, _I, _I, _P, _P, _I, _I, _I],
    'tamtr_hsv_lut_u8': [_P, _LL, _P, _P, _P],
}
EXPORTS = tuple(_SIGS)
_lib = None


class TamtrHostError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TamtrHostError(f'{LIB_PATH} is missing: build it with `make -C tam-tr_amd/csrc` (there is no Python fallback)')
        h = ctypes.CDLL(LIB_PATH)
        for name, args in _SIGS.items():
            fn = getattr(h, name)
            fn.argtypes, fn.restype = args, c_int
        if h.tamtr_host_abi_version() != ABI_VERSION:
            raise TamtrHostError(f'{LIB_PATH}: ABI {h.tamtr_host_abi_version()} != {ABI_VERSION}; rebuild')
        _lib = h
    return _lib


def check(status, what):
    if status != 0:
        raise TamtrHostError(f'{what} -> {status} (bad argument)')
