"""torch.autograd.Function wrappers around the C-ABI kernels (libtamtr_hip.so).

Each op launches on torch.cuda.current_stream(); tensors are plain device buffers to the kernels (data_ptr + sizes).
No op has a CPU or eager-PyTorch fallback: CPU tensors or a missing library raise TamtrHipError.
"""
import ctypes
import math

import torch

from . import _lib
from ._lib import call, dtype_code, ptr, require_gpu, stream_ptr

_I, _F = ctypes.c_int, ctypes.c_float


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------------------------------------ a-1 text gate
class _MaxSigmoidGate(torch.autograd.Function):
    """out = v * sigmoid(max_n <x, gk_n> / sqrt(hc) + bias) * scale  (extra_modules/block.py:217-226)."""

    @staticmethod
    def forward(ctx, x, gk, bias, v, nh, scale):
        require_gpu(x, gk, bias, v)
        B, C, H, W = x.shape
        T = gk.shape[1]
        hc = C // nh
        x, v = _c(x), _c(v)
        gk32, b32 = _c(gk.float()), _c(bias.float())
        out = torch.empty_like(x)
        aw = torch.empty(B, nh, H * W, device=x.device, dtype=torch.float32)
        arg = torch.empty(B, nh, H * W, device=x.device, dtype=torch.int32)
        call('tamtr_maxsigmoid_gate_fwd', ptr(x), ptr(gk32), ptr(b32), ptr(v), ptr(out), ptr(aw), ptr(arg), B, nh, hc, H * W, T,
             _F(scale), dtype_code(x), stream_ptr())
        ctx.save_for_backward(x, gk32, v, aw, arg)
        ctx.cfg = (nh, hc, T, scale, gk.dtype, bias.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, gk32, v, aw, arg = ctx.saved_tensors
        nh, hc, T, scale, gk_dt, b_dt = ctx.cfg
        B, C, H, W = x.shape
        HW = H * W
        dout = _c(dout.to(x.dtype))
        dx, dv = torch.empty_like(x), torch.empty_like(x)
        dlogit = torch.empty(B, nh, HW, device=x.device, dtype=torch.float32)
        call('tamtr_maxsigmoid_gate_bwd', ptr(dout), ptr(x), ptr(gk32), ptr(v), ptr(aw), ptr(arg), ptr(dx), ptr(dv), ptr(dlogit),
             B, nh, hc, HW, T, _F(scale), dtype_code(x), stream_ptr())
        # text-side reductions: a [T x HW] x [HW x hc] batched GEMM per (image, head) - plain library GEMM
        sel = torch.zeros(B, nh, T, HW, device=x.device, dtype=torch.float32)
        sel.scatter_(2, arg.long().unsqueeze(2), dlogit.unsqueeze(2))
        dgk = torch.matmul(sel, x.view(B, nh, hc, HW).float().transpose(2, 3))  # [B,nh,T,hc]
        dgk = dgk.permute(0, 2, 1, 3).reshape(B, T, C)
        dbias = dlogit.sum((0, 2)) * math.sqrt(hc)
        return dx, dgk.to(gk_dt), dbias.to(b_dt), dv, None, None


def maxsigmoid_gate(x, gk, bias, v, nh, scale=1.0):
    """x, v: [B,C,H,W] (f32|bf16); gk: [B,T,C] guide after `gl`; bias: [nh]."""
    return _MaxSigmoidGate.apply(x, gk, bias, v, int(nh), float(scale))


# ------------------------------------------------------------------------------------------------ a-6 deformable core
class _MSDeformCore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, value, shapes, loc, aw):
        require_gpu(value, loc, aw)
        B, L, M, D = value.shape
        _, Q, _, nl, P, _ = loc.shape
        value = _c(value)
        loc32, aw32 = _c(loc.float()), _c(aw.float())
        sh = (ctypes.c_int32 * (2 * nl))(*[int(v) for hw in shapes for v in hw])
        out = torch.empty(B, Q, M * D, device=value.device, dtype=value.dtype)
        call('tamtr_msdeform_attn_fwd', ptr(value), ctypes.cast(sh, ctypes.c_void_p), ptr(loc32), ptr(aw32), ptr(out), B, L, M, D,
             Q, nl, P, dtype_code(value), stream_ptr())
        ctx.save_for_backward(value, loc32, aw32)
        ctx.cfg = (sh, nl, P, loc.dtype, aw.dtype)
        return out

    @staticmethod
    def backward(ctx, gout):
        value, loc32, aw32 = ctx.saved_tensors
        sh, nl, P, loc_dt, aw_dt = ctx.cfg
        B, L, M, D = value.shape
        Q = loc32.shape[1]
        gout = _c(gout.to(value.dtype))
        gvalue = torch.zeros(B, L, M, D, device=value.device, dtype=torch.float32)  # float-atomic accumulator
        gloc = torch.empty_like(loc32)
        gaw = torch.empty_like(aw32)
        call('tamtr_msdeform_attn_bwd', ptr(gout), ptr(value), ctypes.cast(sh, ctypes.c_void_p), ptr(loc32), ptr(aw32),
             ptr(gvalue), ptr(gloc), ptr(gaw), B, L, M, D, Q, nl, P, dtype_code(value), stream_ptr())
        return gvalue.to(value.dtype), None, gloc.to(loc_dt), gaw.to(aw_dt)


def ms_deform_attn_core(value, shapes, loc, aw):
    """value [B,L,M,D]; shapes [[H,W]]*nl; loc [B,Q,M,nl,P,2]; aw [B,Q,M,nl,P] -> [B,Q,M*D] (nn/modules/utils.py:42-89)."""
    return _MSDeformCore.apply(value, [tuple(int(v) for v in s) for s in shapes], loc, aw)


# ------------------------------------------------------------------------------------------------ a-8 contrastive head
class _ContrastiveLogits(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, logit_scale, bias):
        require_gpu(x, w, logit_scale, bias)
        B, Q, C = x.shape
        K = w.shape[1]
        x = _c(x)
        w32 = _c(w.float())
        ls, bi = _c(logit_scale.float().reshape(1)), _c(bias.float().reshape(1))
        logits = torch.empty(B, Q, K, device=x.device, dtype=torch.float32)
        xinv = torch.empty(B, Q, device=x.device, dtype=torch.float32)
        winv = torch.empty(B, K, device=x.device, dtype=torch.float32)
        call('tamtr_contrastive_logits_fwd', ptr(x), ptr(w32), ptr(ls), ptr(bi), ptr(logits), ptr(xinv), ptr(winv), B, Q, K, C,
             dtype_code(x), stream_ptr())
        ctx.save_for_backward(x, w32, ls, bi, logits, xinv, winv)
        ctx.cfg = (w.dtype, logit_scale.dtype, logit_scale.shape, bias.dtype, bias.shape)
        return logits

    @staticmethod
    def backward(ctx, g):
        x, w32, ls, bi, logits, xinv, winv = ctx.saved_tensors
        w_dt, ls_dt, ls_shape, b_dt, b_shape = ctx.cfg
        B, Q, C = x.shape
        K = w32.shape[1]
        g = _c(g.float())
        dx = torch.empty_like(x)
        dwhat = torch.zeros(B, K, C, device=x.device, dtype=torch.float32)
        call('tamtr_contrastive_logits_bwd', ptr(g), ptr(x), ptr(w32), ptr(ls), ptr(xinv), ptr(winv), ptr(dx), ptr(dwhat), B, Q, K,
             C, dtype_code(x), stream_ptr())
        what = w32 * winv.unsqueeze(-1)
        dw = winv.unsqueeze(-1) * (dwhat - (dwhat * what).sum(-1, keepdim=True) * what)
        dls = (g * (logits - bi)).sum().reshape(ls_shape).to(ls_dt)
        dbias = g.sum().reshape(b_shape).to(b_dt)
        return dx, dw.to(w_dt), dls, dbias


def contrastive_logits(x, w, logit_scale, bias):
    """x [B,Q,C] (f32|bf16), w [B,K,C] -> f32 logits [B,Q,K] (nn/modules/block.py:534-541)."""
    return _ContrastiveLogits.apply(x, w, logit_scale, bias)
