"""ctypes + autograd wrapper of oracle/selscan_ref.c (TEST INFRASTRUCTURE: imported only by tests/, smoke() and the
cpu_baseline leg of bench.py).  `scan` has the signature of tamtr_oracle.selective_scan and can be passed as its
`scan_fn` so that the CPU oracle reaches L = 25 600 (640^2) in seconds instead of hours."""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, 'libselscan_ref.so')
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            subprocess.run(['make', '-C', _HERE], check=True)
        _lib = ctypes.CDLL(_PATH)
    return _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


class _Scan(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, delta, A, Bm, Cm, D, bias):
        args = [t.detach().float().contiguous() for t in (u, delta, A, Bm, Cm, D, bias)]
        Bn, KD, L = args[0].shape
        K, N = args[3].shape[1], args[3].shape[2]
        y = torch.empty_like(args[0])
        lib().selscan_ref_fwd(*[_p(t) for t in args], _p(y), Bn, K, KD // K, N, L)
        ctx.save_for_backward(*args)
        return y

    @staticmethod
    def backward(ctx, gy):
        args = ctx.saved_tensors
        u, delta, A, Bm, Cm, D, bias = args
        Bn, KD, L = u.shape
        K, N = Bm.shape[1], Bm.shape[2]
        gy = gy.float().contiguous()
        gu, gd = torch.empty_like(u), torch.empty_like(u)
        gA, gB, gC, gD, gb = (torch.zeros_like(t) for t in (A, Bm, Cm, D, bias))
        lib().selscan_ref_bwd(_p(gy), *[_p(t) for t in args], _p(gu), _p(gd), _p(gA), _p(gB), _p(gC), _p(gD), _p(gb), Bn, K,
                              KD // K, N, L)
        return gu, gd, gA, gB, gC, gD, gb


def scan(u, delta, A, Bm, Cm, D, delta_bias, delta_softplus=True):
    assert delta_softplus
    return _Scan.apply(u, delta, A, Bm, Cm, D, delta_bias)
