"""torch.autograd.Function wrappers around the C-ABI kernels (libtamtr_hip.so).

Each op launches on torch.cuda.current_stream(); tensors are plain device buffers to the kernels (data_ptr + sizes).
No op has a CPU or eager-PyTorch fallback: CPU tensors or a missing library raise TamtrHipError.
"""
import ctypes
import math

import torch

from . import _lib
from ._lib import call, dtype_code, ptr, require_gpu, stream_ptr
import os as _os

_PROJ_CONV_LIB = _os.environ.get('TAMTR_PROJ_CONV') == 'miopen'   # A/B switch: the gate's 3x3 value convolution on the library

_I, _F = ctypes.c_int, ctypes.c_float

# measurement hook (bench.py): KERNEL_EVENTS['tamtr_linear_bf16'] = [] makes the op bracket the C call itself (not the
# dtype casts around it) with a pair of events on the launch stream and append (start, end, algorithmic flops)
KERNEL_EVENTS = {}


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------------------------------------ layout edges
def _is_cl(t):
    return t.dim() == 4 and not t.is_contiguous() and t.is_contiguous(memory_format=torch.channels_last)


def is_cl(t):
    return _is_cl(t)


def _cl_pitch(t):
    """Pixel pitch ld if t [B,C,H,W] is a channels-last map or a channel slice of one (strides (H*W*ld, 1, W*ld, ld), ld >= C), else 0."""
    if t.dim() != 4:
        return 0
    B, C, H, W = t.shape
    sb, sc, sh, sw = t.stride()
    ok = sc == 1 and sw >= C and sh == W * sw and (B == 1 or sb == H * W * sw)
    return sw if ok else 0


class _Relayout(torch.autograd.Function):
    """Channels-last map -> NCHW-contiguous copy (to_nchw) or back, by the tiled transpose of csrc/layout.hip; values unchanged.
    The backward is the opposite repacking of the gradient."""

    @staticmethod
    def forward(ctx, x, to_nchw):
        require_gpu(x)
        ctx.to_nchw = to_nchw
        return _relayout(x, to_nchw)

    @staticmethod
    def backward(ctx, g):
        if ctx.to_nchw:   # gradient arrives NCHW-shaped; hand it back channels-last like the input was
            return (_relayout(g, False) if g.is_contiguous() else g.contiguous(memory_format=torch.channels_last)), None
        return (_relayout(g, True) if _cl_pitch(g) else g.contiguous()), None


def _relayout(x, to_nchw):
    B, C, H, W = x.shape
    if x.dtype not in (torch.float32, torch.bfloat16):
        return x.contiguous() if to_nchw else x.contiguous(memory_format=torch.channels_last)
    out = torch.empty(x.shape, dtype=x.dtype, device=x.device, memory_format=torch.contiguous_format if to_nchw else torch.channels_last)
    call('tamtr_relayout', ptr(x), ptr(out), B, C, H * W, _cl_pitch(x) if to_nchw else C, 0 if to_nchw else 1, dtype_code(x), stream_ptr())
    return out


def to_nchw(x):
    """NCHW-contiguous version of a feature map; a channels-last one (or a channel slice of one) goes through the transpose kernel."""
    if x.is_contiguous():
        return x
    return _Relayout.apply(x, True) if (x.is_cuda and _cl_pitch(x)) else x.contiguous()


class _CatChannels(torch.autograd.Function):
    """torch.cat(maps, 1) for channels-last maps (or channel slices of such): one 16-byte-vector row copy per input
    (csrc/layout.hip tamtr_copy_rows); the backward hands out channel slices of the gradient, no copies."""

    @staticmethod
    def forward(ctx, *maps):
        B, _, H, W = maps[0].shape
        widths = [m.shape[1] for m in maps]
        out = torch.empty((B, sum(widths), H, W), dtype=maps[0].dtype, device=maps[0].device, memory_format=torch.channels_last)
        ct, e, v = sum(widths), maps[0].element_size(), 16 // maps[0].element_size()
        N = B * H * W
        pitches = [_cl_pitch(m) for m in maps]
        if all(c % v == 0 and ld % v == 0 and m.data_ptr() % 16 == 0 for m, c, ld in zip(maps, widths, pitches)):
            for i in range(0, len(maps), 4):   # up to four inputs per launch
                grp = list(range(i, min(i + 4, len(maps))))
                n = len(grp)
                src = (ctypes.c_void_p * n)(*[maps[j].data_ptr() for j in grp])
                lds = (ctypes.c_longlong * n)(*[pitches[j] for j in grp])
                cs = (ctypes.c_int * n)(*[widths[j] for j in grp])
                call('tamtr_cat_rows', ctypes.cast(src, ctypes.c_void_p), ctypes.cast(lds, ctypes.c_void_p), ctypes.cast(cs, ctypes.c_void_p), n,
                     out.data_ptr() + sum(widths[:i]) * e, ct, N, dtype_code(maps[0]), stream_ptr())
        else:
            off = 0
            for m, c, ld in zip(maps, widths, pitches):
                call('tamtr_copy_rows', ptr(m), ld, out.data_ptr() + off * e, ct, N, c, dtype_code(m), stream_ptr())
                off += c
        ctx.widths = widths
        return out

    @staticmethod
    def backward(ctx, g):
        return tuple(g.split(ctx.widths, 1))


def cat_channels(maps):
    """torch.cat(maps, 1); channels-last CUDA maps of one dtype take the row-copy kernel."""
    maps = list(maps)
    if len(maps) > 1 and all(m.is_cuda and m.dim() == 4 and m.dtype == maps[0].dtype and m.dtype in (torch.float32, torch.bfloat16)
                             and m.shape[0] == maps[0].shape[0] and m.shape[2:] == maps[0].shape[2:] and _cl_pitch(m) and not m.is_contiguous()
                             for m in maps):
        return _CatChannels.apply(*maps)
    return torch.cat(maps, 1)


class _ChunkChannels(torch.autograd.Function):
    """x.chunk(2, 1) of a channels-last map with the backward on the row-copy kernel: autograd's own backward of `chunk` concatenates
    the two gradients - one of them a channel slice of a wider gradient - through torch's generic strided copy."""

    @staticmethod
    def forward(ctx, x):
        h = x.shape[1] // 2
        return x[:, :h], x[:, h:]

    @staticmethod
    def backward(ctx, g0, g1):
        return cat_channels([g0, g1])


def chunk2_channels(x):
    """x.chunk(2, 1); channels-last CUDA maps get the kernel-backed backward."""
    if x.is_cuda and _is_cl(x) and x.shape[1] % 2 == 0 and x.dtype in (torch.float32, torch.bfloat16) and x.requires_grad:
        return _ChunkChannels.apply(x)
    return x.chunk(2, 1)


class _PackChannels(torch.autograd.Function):
    """Packed channels-last copy of a channel slice of a channels-last map (what a convolution makes of `x.chunk(2, 1)[1]`)."""

    @staticmethod
    def forward(ctx, x):
        B, C, H, W = x.shape
        out = torch.empty((B, C, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        call('tamtr_copy_rows', ptr(x), _cl_pitch(x), ptr(out), C, B * H * W, C, dtype_code(x), stream_ptr())
        return out

    @staticmethod
    def backward(ctx, g):
        return g


def pack_channels(x):
    if x.is_cuda and x.dim() == 4 and not _is_cl(x) and not x.is_contiguous() and _cl_pitch(x) and x.dtype in (torch.float32, torch.bfloat16):
        return _PackChannels.apply(x)
    return x


class _Resample2(torch.autograd.Function):
    """nn.Upsample(scale_factor=2.0 | 0.5, mode='nearest') on a channels-last map - csrc/layout.hip tamtr_resample2."""

    @staticmethod
    def forward(ctx, x, up):
        B, C, H, W = x.shape
        ctx.cfg = (up, B, C, H, W)
        Ho, Wo = (2 * H, 2 * W) if up else (H // 2, W // 2)
        out = torch.empty((B, C, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        call('tamtr_resample2', ptr(x), ptr(out), B, H, W, C, 0 if up else 2, dtype_code(x), stream_ptr())
        return out

    @staticmethod
    def backward(ctx, g):
        up, B, C, H, W = ctx.cfg
        g = g if _is_cl(g) else (pack_channels(g) if _cl_pitch(g) else g.contiguous(memory_format=torch.channels_last))
        gx = torch.empty((B, C, H, W), dtype=g.dtype, device=g.device, memory_format=torch.channels_last)
        call('tamtr_resample2', ptr(g), ptr(gx), B, H, W, C, 1 if up else 3, dtype_code(g), stream_ptr())
        return gx, None


def resample2(x, up):
    """Nearest x2 (up) / x0.5 of a channels-last CUDA map whose channel count fills 16-byte vectors; None if the kernel does not apply."""
    v = 8 if x.dtype == torch.bfloat16 else 4
    if x.is_cuda and _is_cl(x) and x.dtype in (torch.float32, torch.bfloat16) and x.shape[1] % v == 0 and (up or (x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0)):
        return _Resample2.apply(x, bool(up))
    return None


def to_channels_last(x):
    if x.dim() != 4 or _is_cl(x):
        return x
    return _Relayout.apply(x, False) if (x.is_contiguous() and x.is_cuda) else x.contiguous(memory_format=torch.channels_last)


class _MaxPool(torch.autograd.Function):
    """nn.MaxPool2d(k, s, p) on an NCHW or channels-last map - csrc/pool.hip (one-byte winner codes, gather backward)."""

    @staticmethod
    def forward(ctx, x, k, s, p):
        require_gpu(x)
        nhwc = _is_cl(x)
        if not nhwc:
            x = _c(x)
        B, C, H, W = x.shape
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        fmt = torch.channels_last if nhwc else torch.contiguous_format
        y = torch.empty((B, C, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=fmt)
        code = torch.empty((B, C, Ho, Wo), dtype=torch.uint8, device=x.device, memory_format=fmt)
        call('tamtr_maxpool_fwd', ptr(x), ptr(y), ptr(code), B, C, H, W, k, s, p, int(nhwc), dtype_code(x), stream_ptr())
        ctx.save_for_backward(code)
        ctx.cfg = (B, C, H, W, k, s, p, nhwc, x.dtype)
        return y

    @staticmethod
    def backward(ctx, gy):
        code, = ctx.saved_tensors
        B, C, H, W, k, s, p, nhwc, dt = ctx.cfg
        gy = gy.to(dt)
        gy = (pack_channels(gy) if _cl_pitch(gy) else gy.contiguous(memory_format=torch.channels_last)) if nhwc else _c(gy)
        gx = torch.empty((B, C, H, W), dtype=dt, device=gy.device, memory_format=torch.channels_last if nhwc else torch.contiguous_format)
        call('tamtr_maxpool_bwd', ptr(gy), ptr(code), None, ptr(gx), B, C, H, W, k, s, p, int(nhwc), dtype_code(gy), stream_ptr())
        return gx, None, None, None


def max_pool2d(x, k, s, p):
    """F.max_pool2d(x, k, s, p) for fp32 / bf16 CUDA maps (NCHW or channels-last); anything else goes to torch."""
    if x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16) and k <= 15 and 2 * p <= k:
        return _MaxPool.apply(x, int(k), int(s), int(p))
    return torch.nn.functional.max_pool2d(x, k, s, p)


# ------------------------------------------------------------------------------------------------ a-1 text gate
class _MaxSigmoidGate(torch.autograd.Function):
    """out = v * sigmoid(max_n <x, gk_n> / sqrt(hc) + bias) * scale  (extra_modules/block.py:217-226)."""

    @staticmethod
    def forward(ctx, x, gk, bias, v, nh, scale):
        require_gpu(x, gk, bias, v)
        B, C, H, W = x.shape
        T = gk.shape[1]
        hc = C // nh
        x, v = to_nchw(x), to_nchw(v)   # the kernels read NCHW planes; channels-last maps (NHWC trunk) are repacked
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


def gate_cl_ok(x, C, nh):
    """Shapes / layouts tamtr_maxsigmoid_gate_cl_fwd takes: x a channels-last CUDA map (or a channel slice of one), C = nh * hc with
    C / 8 and hc / 8 powers of two, C <= 512."""
    hc = C // nh if nh else 0
    lpr, lph = C // 8, hc // 8
    return (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and _cl_pitch(x) and nh * hc == C and C % 8 == 0 and hc % 8 == 0
            and 0 < lpr <= 64 and lpr & (lpr - 1) == 0 and lph & (lph - 1) == 0 and _cl_pitch(x) % (16 // x.element_size()) == 0
            and x.data_ptr() % 16 == 0)


@torch.no_grad()
def bn_stats_cl(x2d, bn):
    """Batch statistics of a channels-last [N, C] map for a consumer that normalises in its own load: mean_rstd f32 [C, 2]; updates the
    BatchNorm's running statistics and counter like a training-mode forward."""
    require_gpu(x2d)
    x2d = _c(x2d)
    N, C = x2d.shape
    mom = _bn_tick(bn)
    mr = torch.empty(C, 2, device=x2d.device, dtype=torch.float32)
    part = torch.empty(C * _lib.lib().tamtr_bncl_blocks(N, C, dtype_code(x2d)) * 3, device=x2d.device, dtype=torch.float32)
    rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
    call('tamtr_bncl_stats', ptr(x2d), ptr(rm), ptr(rv), ptr(mr), ptr(part), N, C, float(bn.eps), float(mom), dtype_code(x2d), stream_ptr())
    return mr


def conv3x3_cl_ok(x, conv):
    """What tamtr_conv3x3_cl_stats_fwd takes: a bf16 channels-last map (or a channel slice of one) into a plain 3x3 / stride 1 / pad 1
    convolution without bias, C1 % 32 == 0, C2 % 64 == 0.  TAMTR_PROJ_CONV=miopen keeps the library convolution (A/B switch)."""
    if _PROJ_CONV_LIB or not (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4):
        return False
    ld = _cl_pitch(x)
    return (isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None and conv.padding_mode == 'zeros'
            and conv.in_channels == x.shape[1] and conv.in_channels % 32 == 0 and conv.out_channels % 64 == 0
            and ld and ld % 8 == 0 and x.data_ptr() % 16 == 0 and conv.weight.dtype in (torch.float32, torch.bfloat16))


@torch.no_grad()
def conv3x3_cl_stats(x, conv, bn):
    """proj_conv of the text gate (block.py:205,223) without its BatchNorm's apply pass: returns (v_raw, mean_rstd) - the raw 3x3
    convolution of the channels-last bf16 map x [B,C1,H,W] as [B,C2,H,W] channels-last bf16, and the BatchNorm's batch statistics
    f32 [C2, 2], taken in the convolution's epilogue; the running statistics and the counter of `bn` are updated like a training-mode
    forward.  Forward only (the discarded evaluation); feeds maxsigmoid_gate_cl."""
    require_gpu(x)
    B, C1, H, W = x.shape
    C2 = conv.out_channels
    w = _c(conv.weight)
    wpk = torch.empty(9 * C1 * C2, device=x.device, dtype=torch.bfloat16)
    call('tamtr_conv3x3_pack_weight', ptr(w), ptr(wpk), C1, C2, dtype_code(w), stream_ptr())
    y = torch.empty((B, C2, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    mom = _bn_tick(bn)
    mr = torch.empty(C2, 2, device=x.device, dtype=torch.float32)
    part = torch.empty(C2 * _lib.lib().tamtr_conv3x3_tiles(B, H, W) * 3, device=x.device, dtype=torch.float32)
    rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
    call('tamtr_conv3x3_cl_stats_fwd', ptr(x), _cl_pitch(x), ptr(wpk), ptr(y), ptr(rm), ptr(rv), ptr(mr), ptr(part), B, H, W, C1, C2,
         float(bn.eps), float(mom), stream_ptr())
    return y, mr


@torch.no_grad()
def maxsigmoid_gate_cl(e, gk, bias, v_raw, v_stats, v_bn, nh, scale=1.0):
    """Forward-only gate on channels-last operands with the value branch's BatchNorm (batch statistics v_stats from bn_stats_cl, affine
    of v_bn) applied inside the kernel: e [B,C,H,W] channels-last or a channel slice of such, v_raw [B,C,H,W] channels-last = the raw
    proj_conv output.  Returns out [B,C,H,W] channels-last.  (The differentiable path is maxsigmoid_gate.)"""
    require_gpu(e, gk, bias, v_raw)
    B, C, H, W = e.shape
    v_raw = v_raw if _is_cl(v_raw) else v_raw.contiguous(memory_format=torch.channels_last)
    out = torch.empty((B, C, H, W), dtype=e.dtype, device=e.device, memory_format=torch.channels_last)
    gk32, b32 = _c(gk.float()), _c(bias.float())
    ga, be = _c(v_bn.weight.float()), _c(v_bn.bias.float())
    call('tamtr_maxsigmoid_gate_cl_fwd', ptr(e), _cl_pitch(e), None, None, None, ptr(v_raw), ptr(v_stats), ptr(ga), ptr(be), ptr(gk32), ptr(b32),
         ptr(out), None, None, B, nh, C // nh, H * W, gk.shape[1], _F(scale), dtype_code(e), stream_ptr())
    return out


def maxsigmoid_gate(x, gk, bias, v, nh, scale=1.0):
    """x, v: [B,C,H,W] (f32|bf16); gk: [B,T,C] guide after `gl`; bias: [nh]."""
    return _MaxSigmoidGate.apply(x, gk, bias, v, int(nh), float(scale))


# ------------------------------------------------------------------------------------------------ a-6 deformable core
_MSDA_ATOMICS = _os.environ.get('TAMTR_MSDA_ATOMICS') == '1'   # A/B switch: the round-1/2 float-atomic scatter


def deterministic():
    """TAMTR_DETERMINISTIC=1 or torch.use_deterministic_algorithms(True): the reference's `deterministic: True`
    (cfg/default.yaml:26, utils/torch_utils.py:371-389).  Kernels of this package are order-fixed by construction; the few shapes that
    only an atomic kernel serves go through _atomic_fallback(), which follows torch's own convention for nondeterministic ops."""
    return _os.environ.get('TAMTR_DETERMINISTIC') == '1' or torch.are_deterministic_algorithms_enabled()


_WARNED = set()


def _atomic_fallback(what):
    """Called before a float-atomic kernel runs in deterministic mode.  Strict mode (use_deterministic_algorithms(True)) raises, like
    torch's own nondeterministic ops; warn-only mode - what the reference sets (utils/torch_utils.py:376: warn_only=True) and what
    tuning.use_deterministic_convolutions() therefore sets - warns once per shape class and lets the atomic kernel run."""
    if not deterministic():
        return
    if torch.are_deterministic_algorithms_enabled() and not torch.is_deterministic_algorithms_warn_only_enabled():
        raise _lib.TamtrHipError(f'deterministic mode: {what}')
    if what not in _WARNED:
        _WARNED.add(what)
        import warnings
        warnings.warn(f'{what}; running the float-atomic kernel, this step is not bitwise reproducible '
                      '(torch.use_deterministic_algorithms(True) without warn_only raises here instead)', UserWarning, stacklevel=3)


def msda_sorted_ok(Q, P, D):
    return Q * P * 4 <= 8192 and D % 8 == 0 and D <= 256


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
        gloc = torch.empty_like(loc32)
        gaw = torch.empty_like(aw32)
        if msda_sorted_ok(Q, P, D) and not _MSDA_ATOMICS:
            # ordered segmented sum (csrc/msdeform.hip): every element of the value gradient written once, in the value's dtype -
            # no 1.1 GB fp32 zero fill, no float atomics, no cast pass, and the same bits on every run
            gvalue = torch.empty(B, L, M, D, device=value.device, dtype=value.dtype)
            call('tamtr_msdeform_attn_bwd_sorted', ptr(gout), ptr(value), ctypes.cast(sh, ctypes.c_void_p), ptr(loc32), ptr(aw32),
                 ptr(gvalue), ptr(gloc), ptr(gaw), None, B, L, M, D, Q, nl, P, M * D, dtype_code(value), stream_ptr())
            return gvalue, None, gloc.to(loc_dt), gaw.to(aw_dt)
        _atomic_fallback(f'the deformable-attention backward has no atomics-free kernel for Q*P*4 = {Q * P * 4} > 8192 corners per level '
                         f'or D = {D} (csrc/msdeform.hip)')
        gvalue = torch.zeros(B, L, M, D, device=value.device, dtype=torch.float32)  # float-atomic accumulator
        call('tamtr_msdeform_attn_bwd', ptr(gout), ptr(value), ctypes.cast(sh, ctypes.c_void_p), ptr(loc32), ptr(aw32),
             ptr(gvalue), ptr(gloc), ptr(gaw), B, L, M, D, Q, nl, P, dtype_code(value), stream_ptr())
        return gvalue.to(value.dtype), None, gloc.to(loc_dt), gaw.to(aw_dt)


def ms_deform_attn_core(value, shapes, loc, aw):
    """value [B,L,M,D]; shapes [[H,W]]*nl; loc [B,Q,M,nl,P,2]; aw [B,Q,M,nl,P] -> [B,Q,M*D] (nn/modules/utils.py:42-89)."""
    return _MSDeformCore.apply(value, [tuple(int(v) for v in s) for s in shapes], loc, aw)


class _ValueProjMSDA(torch.autograd.Function):
    """MSDeformAttn's value projection and its sampling core as ONE node (reference nn/modules/transformer.py:273-311,
    nn/modules/utils.py:42-89): out = msda(x W^T + b, loc, aw).  Same kernels as linear_bf16 + ms_deform_attn_core in both directions;
    what the pairing buys is the BIAS gradient: db = column sums of d(value) over its B*L rows = sum_{b,q} gout[b,q,m,:] * colw[b,q,m],
    where colw is the weight an item put on the map (1 unless a corner falls off) - the backward kernel that computes d/d(loc), d/d(aw)
    emits it on the side, so the 550 MB pass over d(value) that colsum() made per decoder layer (111 us each) is a product of
    [B*Q, M, D] operands."""

    @staticmethod
    def forward(ctx, x, weight, bias, shapes, loc, aw, M):
        require_gpu(x, weight, bias, loc, aw)
        B, L, K = x.shape
        N = weight.shape[0]
        D = N // M
        _, Q, _, nl, P, _ = loc.shape
        x2 = _c(x.reshape(-1, K))
        w16, b32 = _c(bf16_of(weight)), _c(bias.float())
        value = torch.empty(B, L, M, D, device=x.device, dtype=torch.bfloat16)
        rec = KERNEL_EVENTS.get('tamtr_linear_bf16')
        if rec is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        call('tamtr_linear_bf16', ptr(x2), ptr(w16), ptr(b32), ptr(value), B * L, N, K, stream_ptr())
        if rec is not None:
            e1.record()
            rec.append((e0, e1, 2.0 * B * L * N * K))
        loc32, aw32 = _c(loc.float()), _c(aw.float())
        sh = (ctypes.c_int32 * (2 * nl))(*[int(v) for hw in shapes for v in hw])
        out = torch.empty(B, Q, N, device=x.device, dtype=torch.bfloat16)
        call('tamtr_msdeform_attn_fwd', ptr(value), ctypes.cast(sh, ctypes.c_void_p), ptr(loc32), ptr(aw32), ptr(out), B, L, M, D,
             Q, nl, P, dtype_code(value), stream_ptr())
        ctx.save_for_backward(x2, w16, value, loc32, aw32)
        ctx.cfg = (x.shape, sh, nl, P, M, weight.dtype, bias.dtype, loc.dtype, aw.dtype)
        return out

    @staticmethod
    def backward(ctx, gout):
        x2, w16, value, loc32, aw32 = ctx.saved_tensors
        (B, L, K), sh, nl, P, M, w_dt, b_dt, loc_dt, aw_dt = ctx.cfg
        N = w16.shape[0]
        D = N // M
        Q = loc32.shape[1]
        gout = _c(gout.to(torch.bfloat16))
        gloc, gaw = torch.empty_like(loc32), torch.empty_like(aw32)
        colw = torch.empty(B, Q, M, device=gout.device, dtype=torch.float32)
        gvalue = torch.empty(B * L, N, device=gout.device, dtype=torch.bfloat16)
        call('tamtr_msdeform_attn_bwd_sorted', ptr(gout), ptr(value), ctypes.cast(sh, ctypes.c_void_p), ptr(loc32), ptr(aw32),
             ptr(gvalue), ptr(gloc), ptr(gaw), ptr(colw), B, L, M, D, Q, nl, P, N, dtype_code(value), stream_ptr())
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty(B * L, K, device=gout.device, dtype=torch.bfloat16)
            call('tamtr_linear_bf16', ptr(gvalue), ptr(_c(w16.t())), None, ptr(gx), B * L, K, N, stream_ptr())
            gx = gx.view(B, L, K)
        gw = dw_splitk(gvalue, x2).to(w_dt) if ctx.needs_input_grad[1] else None
        gb = (gout.view(B * Q, M, D).float() * colw.view(B * Q, M, 1)).sum(0).view(N).to(b_dt) if ctx.needs_input_grad[2] else None
        return gx, gw, gb, None, gloc.to(loc_dt), gaw.to(aw_dt), None


def value_proj_msda_ok(x, lin, n_heads, Q, P):
    """The paired node serves the bf16 token memory on the W-stationary GEMM's shapes with the sorted (atomics-free) backward
    (TAMTR_VALUE_BIAS=colsum: linear_bf16 + ms_deform_attn_core as separate nodes, the bias gradient as a pass over d(value))."""
    K, N = lin.in_features, lin.out_features
    return (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 3 and lin.bias is not None and N % n_heads == 0 and K % 128 == 0 and N % 128 == 0
            and msda_sorted_ok(Q, P, N // n_heads) and not _MSDA_ATOMICS and _os.environ.get('TAMTR_VALUE_BIAS') != 'colsum')


def value_proj_msda(x, lin, n_heads, shapes, loc, aw):
    """msda(value_proj(x), loc, aw): x [B, L, K] bf16 -> [B, Q, N] bf16; see _ValueProjMSDA."""
    return _ValueProjMSDA.apply(x, lin.weight, lin.bias, [tuple(int(v) for v in s) for s in shapes], loc, aw, int(n_heads))


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
        if K > 16:
            _atomic_fallback(f'the contrastive head backward sums d(what) with LDS atomics for K = {K} > 16 prompts (csrc/contrastive.hip)')
        slabs = _lib.lib().tamtr_contrastive_bwd_slabs(Q)   # per-workgroup partials of d(what), added below in a fixed order
        dwhat = torch.empty(B, slabs, K, C, device=x.device, dtype=torch.float32)
        call('tamtr_contrastive_logits_bwd', ptr(g), ptr(x), ptr(w32), ptr(ls), ptr(xinv), ptr(winv), ptr(dx), ptr(dwhat), B, Q, K,
             C, dtype_code(x), stream_ptr())
        dwhat = dwhat.sum(1) if slabs > 1 else dwhat[:, 0]
        what = w32 * winv.unsqueeze(-1)
        dw = winv.unsqueeze(-1) * (dwhat - (dwhat * what).sum(-1, keepdim=True) * what)
        dls = (g * (logits - bi)).sum().reshape(ls_shape).to(ls_dt)
        dbias = g.sum().reshape(b_shape).to(b_dt)
        return dx, dw.to(w_dt), dls, dbias


def contrastive_logits(x, w, logit_scale, bias):
    """x [B,Q,C] (f32|bf16), w [B,K,C] -> f32 logits [B,Q,K] (nn/modules/block.py:534-541)."""
    return _ContrastiveLogits.apply(x, w, logit_scale, bias)


# ------------------------------------------------------------------------------------------------ a-5 value projection
_SPLIT_MAX = int(_os.environ.get('TAMTR_SPLITK_MAX', '64'))   # A/B knob: slices of the row-sliced weight-gradient products


def _split_count(M, min_rows=2048, max_split=None):
    max_split = _SPLIT_MAX if max_split is None else max_split
    S = 1
    while S < max_split and M % (2 * S) == 0 and M // (2 * S) >= min_rows:
        S *= 2
    return S


_BMM_F32_OUT = None  # does this torch build take bmm(..., out_dtype=float32) on the GPU?


def dw_splitk(g2, x2, min_rows=2048):
    """dW [N, K] = g2^T x2 for very tall operands (M = B*L rows >> N, K), fp32 result.
    A single M-reduction GEMM of this shape has only (N/64)*(K/128) = 8..32 output tiles: hipBLASLt ran it on that many
    workgroups (945 us for M = 537 600, N = K = 512: 0.3 PF/s).  Sliced into S row blocks it is ONE batched GEMM with S x
    the tiles, and the S partial products are summed in fp32."""
    M, N = g2.shape
    K = x2.shape[1]
    S = _split_count(M, min_rows)
    if S == 1:
        return (g2.t() @ x2).float()
    a, b = g2.view(S, M // S, N).transpose(1, 2), x2.view(S, M // S, K)
    global _BMM_F32_OUT
    if _BMM_F32_OUT is not False and N % 8 == 0 and K % 8 == 0:   # (skinny operands - a 4-wide box head - take the bf16 partials below)
        try:
            out = torch.bmm(a, b, out_dtype=torch.float32)
            _BMM_F32_OUT = True
            return slab_sum(out)
        except (NotImplementedError, RuntimeError, TypeError):
            if _BMM_F32_OUT:
                raise
            _BMM_F32_OUT = False
    return slab_sum(torch.bmm(a, b))


class _BoxRefine(torch.autograd.Function):
    """sigmoid(delta + inverse_sigmoid(ref)) - the decoder's box refinement (transformer.py:881-887) - as one kernel each way."""

    @staticmethod
    def forward(ctx, delta, ref):
        require_gpu(delta, ref)
        d, r = _c(delta.float()), _c(ref.float())
        out = torch.empty_like(d)
        call('tamtr_box_refine_fwd', ptr(d), ptr(r), ptr(out), d.numel(), stream_ptr())
        ctx.save_for_backward(out, r)
        ctx.cfg = (delta.dtype, ref.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        out, r = ctx.saved_tensors
        d_dt, r_dt = ctx.cfg
        g = _c(g.float())
        gd = torch.empty_like(out)
        gr = torch.empty_like(out) if ctx.needs_input_grad[1] else None
        call('tamtr_box_refine_bwd', ptr(g), ptr(out), ptr(r), ptr(gd), ptr(gr), out.numel(), stream_ptr())
        return gd.to(d_dt), None if gr is None else gr.to(r_dt)


def box_refine(delta, ref):
    """sigmoid(delta + inverse_sigmoid(ref)); delta, ref of the same shape."""
    if delta.is_cuda and delta.shape == ref.shape and delta.dtype in (torch.float32, torch.bfloat16) and _os.environ.get('TAMTR_BOX_REFINE') != 'torch':
        return _BoxRefine.apply(delta, ref)
    x = ref.clamp(min=0, max=1)
    return torch.sigmoid(delta + torch.log(x.clamp(min=1e-5) / (1 - x).clamp(min=1e-5)))


def bf16_shadow(p):
    """The bf16 copy of an fp32 parameter that the optimizer kernel keeps next to it (engine.FusedOptimStep(shadows=True)), if it is there and
    still describes the parameter's current value (same tensor version as when it was last derived); else None."""
    s = getattr(p, '_tamtr_bf16', None)
    if s is not None:
        base = getattr(p, '_tamtr_alias_of', p)   # (graphs.GraphedPart records through aliases of the parameters: the version is the real one's)
        if s._tamtr_version == base._version and s.device == p.device and s.shape == p.shape:
            return s
    return None


def bf16_of(p):
    """A weight as bf16 for this step's products: the optimizer's shadow copy when there is one (no kernel), else the cast that autocast does."""
    if p.dtype == torch.bfloat16:
        return p
    s = bf16_shadow(p)
    return s if s is not None else p.to(torch.bfloat16)


class _LinearMaster(torch.autograd.Function):
    """y = x W^T + b for the SHORT token-wise linears of the decoder side (M = B * Q rows; nn.Linear under bf16 autocast, transformer.py:
    539-558,869-889) with the gradients of W and b produced in fp32 FOR THE fp32 MASTERS: autocast's form casts W and b to bf16 per use, gets
    bf16 gradients for the copies (the bias gradient from a torch reduction behind a memset) and casts each back - per linear and step four
    cast kernels, a reduction and a memset around three GEMMs.  Here: the optimizer's shadow copies (ops.bf16_of), dW = dY^T X as one GEMM with
    fp32 output, db by the ordered column-sum kernels."""

    @staticmethod
    def forward(ctx, x, weight, bias, w16=None, b16=None):
        # (w16 / b16: the caller's bf16 copies, for a SLICE of a parameter - e.g. the q/k and v rows of a packed in_proj - whose shadow is
        # the same slice of the parameter's shadow)
        w16 = bf16_of(weight) if w16 is None else w16
        b16 = None if bias is None else (bf16_of(bias) if b16 is None else b16)
        x16 = x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)
        ctx.save_for_backward(x16, w16)
        ctx.cfg = (x.dtype, weight.dtype, None if bias is None else bias.dtype)
        return torch.nn.functional.linear(x16, w16, b16)

    @staticmethod
    def backward(ctx, gy):
        x16, w16 = ctx.saved_tensors
        x_dt, w_dt, b_dt = ctx.cfg
        N, K = w16.shape
        g2 = _c(gy.reshape(-1, N).to(torch.bfloat16))
        x2 = _c(x16.reshape(-1, K))
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.mm(g2, w16).view(x16.shape).to(x_dt)
        if ctx.needs_input_grad[1]:
            # M = B * Q = 4 672 rows against N x K <= 1 024 x 1 024 outputs: as ONE product the library runs it on 16 - 64 workgroups (35 us
            # whatever N and K: profiles/r03_gemm_census.txt); as 16 row slices it is a batched product on 16 x the tiles + the ordered sum (A/B: 77.5 -> 77.1 ms per step)
            wide = N % 8 == 0 and K % 8 == 0
            gw = (dw_splitk(g2, x2, _LM_MIN_ROWS) if _LM_MIN_ROWS else (_mm_f32(g2.t(), x2) if wide else torch.mm(g2.t(), x2))).to(w_dt)
        if b_dt is not None and ctx.needs_input_grad[2]:
            gb = colsum(g2).to(b_dt)
        return gx, gw, gb, None, None


_MM_F32_OUT = None
_LM_MIN_ROWS = int(_os.environ.get('TAMTR_LINEAR_MASTER_SLICE_ROWS', '256'))   # A/B knob: 0 = the weight gradient as one product


def _mm_f32(a, b):
    """a @ b for bf16 operands with the fp32 accumulator stored as it is (no rounding to bf16, no cast kernel) where this torch build's
    mm takes out_dtype; else the bf16 product widened."""
    global _MM_F32_OUT
    if _MM_F32_OUT is not False:
        try:
            out = torch.mm(a, b, out_dtype=torch.float32)
            _MM_F32_OUT = True
            return out
        except (NotImplementedError, RuntimeError, TypeError):
            if _MM_F32_OUT:
                raise
            _MM_F32_OUT = False
    return torch.mm(a, b).float()


def linear_master_ok(x, lin):
    return (x.is_cuda and torch.is_grad_enabled() and torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.bfloat16
            and isinstance(lin, torch.nn.Linear) and lin.weight.dtype == torch.float32 and x.dtype in (torch.float32, torch.bfloat16)
            and lin.in_features % 4 == 0 and lin.out_features % 4 == 0 and _os.environ.get('TAMTR_LINEAR_MASTER') != '0')


def linear(x, lin):
    """lin(x) for an nn.Linear of the decoder side: in bf16 training mode on the GPU through _LinearMaster, else the module itself."""
    if linear_master_ok(x, lin):
        return _LinearMaster.apply(x, lin.weight, lin.bias)
    return lin(x)


def shared_bf16(x, *lins):
    """x as the bf16 operand of SEVERAL ops.linear calls (one cast, and one cast of the summed input gradient, instead of one per consumer);
    x itself where ops.linear would not take the _LinearMaster path."""
    if x.dtype == torch.float32 and lins and all(linear_master_ok(x, lin) for lin in lins):
        return x.to(torch.bfloat16)
    return x


def linear_rows(x, weight, bias, lo, hi):
    """F.linear(x, weight[lo:hi], bias[lo:hi]) - a row block of a packed projection (nn.MultiheadAttention's in_proj) - the same way."""
    w, b = weight[lo:hi], bias[lo:hi]
    if (x.is_cuda and torch.is_grad_enabled() and torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.bfloat16
            and weight.dtype == torch.float32 and weight.shape[1] % 8 == 0 and (hi - lo) % 8 == 0 and _os.environ.get('TAMTR_LINEAR_MASTER') != '0'):
        return _LinearMaster.apply(x, w, b, bf16_of(weight)[lo:hi], bf16_of(bias)[lo:hi])
    return torch.nn.functional.linear(x, w, b)


class _LinearSplitK(torch.autograd.Function):
    """y = x W^T + b through the library GEMM (forward and dX); dW through dw_splitk.  For the tall-skinny linears of the
    VSS blocks (in_proj / out_proj / fc1 / fc2: M = B*H*W rows, 128..2048 features)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        bf = x.dtype == torch.bfloat16
        w16 = bf16_of(weight) if bf else weight.to(x.dtype)
        ctx.save_for_backward(x, w16)
        ctx.cfg = (weight.dtype, None if bias is None else bias.dtype)
        return torch.nn.functional.linear(x, w16, None if bias is None else (bf16_of(bias) if bf else bias.to(x.dtype)))

    @staticmethod
    def backward(ctx, gy):
        x, w16 = ctx.saved_tensors
        w_dt, b_dt = ctx.cfg
        g2 = _c(gy.reshape(-1, gy.shape[-1]).to(x.dtype))
        x2 = _c(x.reshape(-1, x.shape[-1]))
        gx = (g2 @ w16).view(x.shape) if ctx.needs_input_grad[0] else None
        gw = dw_splitk(g2, x2).to(w_dt) if ctx.needs_input_grad[1] else None
        gb = colsum(g2).to(b_dt) if (b_dt is not None and ctx.needs_input_grad[2]) else None
        return gx, gw, gb


def linear_splitk(x, weight, bias=None):
    return _LinearSplitK.apply(x, weight, bias)


class _LinearBF16(torch.autograd.Function):
    """Y = X W^T + b on the hand-written MFMA kernel (bf16 in/out, fp32 accumulate).  Backward: dX = dY W on the same kernel
    (against W^T); dW = dY^T X (a reduction over the M = B*L rows) is a batched library GEMM over row slices (dw_splitk)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        require_gpu(x, weight, bias)
        K = x.shape[-1]
        N = weight.shape[0]
        x2 = _c(x.reshape(-1, K))
        if x2.dtype != torch.bfloat16:
            raise _lib.TamtrHipError('linear_bf16 needs bf16 activations')
        w16 = _c(bf16_of(weight))
        b32 = _c(bias.float()) if bias is not None else None
        y = torch.empty(x2.shape[0], N, device=x.device, dtype=torch.bfloat16)
        rec = KERNEL_EVENTS.get('tamtr_linear_bf16')
        if rec is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        call('tamtr_linear_bf16', ptr(x2), ptr(w16), ptr(b32), ptr(y), x2.shape[0], N, K, stream_ptr())
        if rec is not None:
            e1.record()
            rec.append((e0, e1, 2.0 * x2.shape[0] * N * K))
        ctx.save_for_backward(x2, w16)
        ctx.cfg = (x.shape, weight.dtype, None if bias is None else bias.dtype)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, gy):
        x2, w16 = ctx.saved_tensors
        xshape, w_dt, b_dt = ctx.cfg
        g2 = _c(gy.reshape(-1, gy.shape[-1]).to(torch.bfloat16))
        gx = None
        if ctx.needs_input_grad[0]:
            N, K = w16.shape
            if K % 128 == 0 and N % 64 == 0:  # dX = dY W is the same "NT" GEMM against W^T (a 0.5 MB transpose): same MFMA kernel
                gx = torch.empty(g2.shape[0], K, device=g2.device, dtype=torch.bfloat16)
                call('tamtr_linear_bf16', ptr(g2), ptr(_c(w16.t())), None, ptr(gx), g2.shape[0], K, N, stream_ptr())
                gx = gx.view(xshape)
            else:
                gx = (g2 @ w16).view(xshape)
        gw = dw_splitk(g2, x2).to(w_dt) if ctx.needs_input_grad[1] else None
        gb = colsum(g2).to(b_dt) if (b_dt is not None and ctx.needs_input_grad[2]) else None
        return gx, gw, gb


def colsum(g2):
    """Column sums (fp32) of a [M, N] gradient: the bias gradient of a token-wise Linear.  Tall bf16 matrices take the streaming kernel of
    csrc/fold.hip (torch's generic reduction runs M = 537 600, N = 512 at 2.9 TB/s); everything else torch."""
    M, N = g2.shape
    if g2.is_cuda and g2.dtype == torch.bfloat16 and M >= 4096 and N % 8 == 0 and N <= 2048 and 256 % (N // 8) == 0 and g2.is_contiguous():
        nblk = _lib.lib().tamtr_colsum_blocks(M)
        part = torch.empty(nblk, N, device=g2.device, dtype=torch.float32)
        call('tamtr_colsum_bf16', ptr(g2), ptr(part), M, N, stream_ptr())
        return slab_sum(part)
    if g2.is_cuda and g2.dtype in (torch.float32, torch.bfloat16) and N % 4 == 0:
        return slab_sum(_c(g2))   # short or fp32 matrices: the ordered row sum directly (no torch reduction: see slab_sum)
    return g2.sum(0, dtype=torch.float32)


def slab_sum(t):
    """t [R, ...] (f32 | bf16, contiguous) -> f32 [...] = t.sum(0) in a fixed order in ONE kernel (csrc/fold.hip tamtr_slab_sum_rows): the
    last stage of the two-stage reductions (partial rows written by a kernel's workgroups, per-image rows, split-K slices).  torch's
    `t.sum(0)` of such a shape is a multi-workgroup reduction behind a memset node (its arrival semaphores), the node kind that breaks
    HIP-graph replays under AQL packet capture; it also needs `.float()` first for bf16 slices."""
    R = t.shape[0]
    C = t.numel() // max(R, 1)
    if not (t.is_cuda and t.dtype in (torch.float32, torch.bfloat16) and t.is_contiguous() and R > 0 and C % 4 == 0 and t.data_ptr() % 16 == 0):
        return t.float().sum(0)
    if R == 1:
        return t[0].float()
    out = torch.empty(t.shape[1:], device=t.device, dtype=torch.float32)
    call('tamtr_slab_sum_rows', ptr(t), ptr(out), R, C, dtype_code(t), stream_ptr())
    return out


class _LinearBF16ZeroRows(torch.autograd.Function):
    """y = (x with the rows `idx` of every image zeroed) W^T + b, without materialising the masked x: y = x W^T + b on the MFMA
    kernel, then the few masked rows of y are set to b.  Backward: dX = dY W with those rows zeroed, dW = dY^T X minus the masked
    rows' contribution (a small GEMM), db = column sums of dY over ALL rows.  x [B, L, K] bf16, idx int64 [n] positions along L
    (the invalid anchors of the MEH token memory, head.py:1210-1213: 1 580 of 33 600 at 640 px)."""

    @staticmethod
    def forward(ctx, x, weight, bias, idx):
        require_gpu(x, weight, bias)
        B, L, K = x.shape
        N = weight.shape[0]
        x2 = _c(x.reshape(-1, K))
        w16 = _c(bf16_of(weight))
        b32 = _c(bias.float())
        y = torch.empty(B, L, N, device=x.device, dtype=torch.bfloat16)
        rec = KERNEL_EVENTS.get('tamtr_linear_bf16')
        if rec is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        call('tamtr_linear_bf16', ptr(x2), ptr(w16), ptr(b32), ptr(y), B * L, N, K, stream_ptr())
        if rec is not None:
            e1.record()
            rec.append((e0, e1, 2.0 * B * L * N * K))
        if idx.numel():
            y[:, idx] = b32.to(torch.bfloat16)
        ctx.save_for_backward(x2, w16, idx)
        ctx.cfg = (x.shape, weight.dtype, bias.dtype)
        return y

    @staticmethod
    def backward(ctx, gy):
        x2, w16, idx = ctx.saved_tensors
        (B, L, K), w_dt, b_dt = ctx.cfg
        N = w16.shape[0]
        g3 = _c(gy.to(torch.bfloat16))
        g2 = g3.view(-1, N)
        gx = torch.empty(B, L, K, device=g2.device, dtype=torch.bfloat16)
        if K % 128 == 0 and N % 64 == 0:
            call('tamtr_linear_bf16', ptr(g2), ptr(_c(w16.t())), None, ptr(gx), B * L, K, N, stream_ptr())
        else:
            gx = (g2 @ w16).view(B, L, K)
        gw = dw_splitk(g2, x2)
        if idx.numel():
            gx.index_fill_(1, idx, 0)   # (not `gx[:, idx] = 0`: assigning a Python scalar through an index synchronises the host)
            gi, xi = g3[:, idx].reshape(-1, N), x2.view(B, L, K)[:, idx].reshape(-1, K)
            gw = gw - (gi.t() @ xi).float()
        gb = colsum(g2).to(b_dt)
        return gx, gw.to(w_dt), gb, None


def linear_bf16_zero_rows(x, weight, bias, idx):
    """linear_bf16 of x [B, L, K] with the rows idx (along L) treated as zero; see _LinearBF16ZeroRows."""
    return _LinearBF16ZeroRows.apply(x, weight, bias, idx)


class _Fanout(torch.autograd.Function):
    """n handles on one tensor for n consumers; the backward adds their n gradients in ONE pass (csrc/fold.hip tamtr_sum_n) instead of
    autograd's n - 1 pairwise accumulations (three 2-read-1-write passes over the 550 MB token memory gradient per step)."""

    @staticmethod
    def forward(ctx, x, n):
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        live = [g for g in gs if g is not None]
        if len(live) == 1:
            return live[0], None
        return _sum_handles(live, live[0].shape, live[0].dtype, live[0].device), None


def _sum_handles(live, shape, dtype, device):
    """Sum of the gradients of fan-out handles in one pass (tamtr_sum_n) into a buffer of our own; zeros when nobody sent one."""
    if not live:
        return torch.zeros(shape, dtype=dtype, device=device)
    if len(live) == 1:
        return live[0].clone()
    g0 = live[0]
    if (g0.is_cuda and g0.dtype in (torch.float32, torch.bfloat16) and g0.numel() % 4 == 0 and len(live) <= 8
            and all(g.dtype == g0.dtype and g.shape == g0.shape for g in live)):
        live = [_c(g) for g in live]
        out = torch.empty_like(live[0])
        src = (ctypes.c_void_p * len(live))(*[g.data_ptr() for g in live])
        call('tamtr_sum_n', ctypes.cast(src, ctypes.c_void_p), len(live), ptr(out), out.numel(), dtype_code(out), stream_ptr())
        return out
    acc = live[0].clone()
    for g in live[1:]:
        acc += g
    return acc


class _EncSelect(torch.autograd.Function):
    """The encoder side of the MEH query selection as ONE node (reference head.py:1205-1245, `_get_decoder_input`): token memory
    x [B, L, K] bf16 -> enc_output (Linear with the invalid-anchor rows treated as zero + LayerNorm) -> enc_score_head -> top-k over the
    tokens by best class score -> the picked rows of the normalised memory and of the scores.  Plus `n_dec` handles on x for the decoder
    layers' value projections (what ops.fanout gives).

    Why one node: everything downstream reads only the num_queries picked rows per image (4 800 of 537 600 at the bench shape), so the
    gradient of this whole branch with respect to the LayerNorm output, the Linear output and x is ZERO outside those rows.  Autograd
    on the separate ops runs the dense backward anyway - zero fill + sorted index_put + add for the two gathers, the score head's dX / dW
    over all rows, LayerNorm backward over all rows, the 512 x 512 dX GEMM and the split dW over all rows, a fourth 550 MB operand in the
    fan-out sum: ~1.9 ms per step - multiplying zeros.  Here the backward gathers the picked rows (Linear input, Linear output) in the
    forward, keeps NOTHING of the three [B, L, .] intermediates, does the whole chain on [B * num_queries, .] rows in fp32, and adds the
    rows' dX into the sum of the decoder handles' gradients (its own buffer).  Same function, same gradient (sums over rows lose only
    exact zeros); the forward is the same kernels as before."""

    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, eps, ws, bs, invalid, nq, fixed_top, n_dec):
        require_gpu(x, w, b)
        B, L, K = x.shape
        N = w.shape[0]
        if x.dtype != torch.bfloat16:
            raise _lib.TamtrHipError('enc_select needs a bf16 token memory')
        x2 = _c(x.reshape(-1, K))
        w16, b32 = _c(bf16_of(w)), _c(b.float())
        y = torch.empty(B, L, N, device=x.device, dtype=torch.bfloat16)
        rec = KERNEL_EVENTS.get('tamtr_linear_bf16')
        if rec is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        call('tamtr_linear_bf16', ptr(x2), ptr(w16), ptr(b32), ptr(y), B * L, N, K, stream_ptr())
        if rec is not None:
            e1.record()
            rec.append((e0, e1, 2.0 * B * L * N * K))
        if invalid.numel():
            y[:, invalid] = b32.to(torch.bfloat16)
        g32, be32 = _c(gamma.float()), _c(beta.float())
        mem = torch.empty_like(y)
        stats = torch.empty(B * L, 2, device=x.device, dtype=torch.float32)
        call('tamtr_layernorm_fwd', ptr(y), ptr(g32), ptr(be32), ptr(mem), ptr(stats), B * L, N, float(eps), dtype_code(y), stream_ptr())
        ws16 = bf16_of(ws)
        scores = torch.nn.functional.linear(mem, ws16, bs.to(torch.bfloat16))
        top = torch.topk(scores.max(-1).values, nq, dim=1).indices if fixed_top is None else fixed_top.to(x.device)
        bi = torch.arange(B, device=x.device).unsqueeze(-1)
        top_feat, enc_scores = mem[bi, top], scores[bi, top]
        valid = torch.ones(L, device=x.device, dtype=torch.bool)
        if invalid.numel():
            valid.index_fill_(0, invalid, False)   # (not `valid[invalid] = False`: assigning a Python scalar through an index synchronises the host)
        ctx.save_for_backward(top, x.view(B, L, K)[bi, top], y[bi, top], stats.view(B, L, 2)[bi, top], top_feat, valid[top], w16, ws16, g32)
        ctx.cfg = (x.shape, w.dtype, b.dtype, gamma.dtype, beta.dtype, ws.dtype, bs.dtype, fixed_top is None)
        ctx.mark_non_differentiable(top)
        return (top_feat, enc_scores, top) + tuple(x.view_as(x) for _ in range(int(n_dec)))

    @staticmethod
    def backward(ctx, g_feat, g_sc, _g_top, *g_dec):
        top, xr, yr, st, top_feat, vr, w16, ws16, g32 = ctx.saved_tensors
        (B, L, K), w_dt, b_dt, ga_dt, be_dt, ws_dt, bs_dt, distinct = ctx.cfg
        N, nc = w16.shape[0], ws16.shape[0]
        R = top.numel()
        gx = _sum_handles([g for g in g_dec if g is not None], (B, L, K), xr.dtype, xr.device)   # ours: the rows are added in place
        ge = g_sc.reshape(R, nc).float() if g_sc is not None else torch.zeros(R, nc, device=top.device)
        gm = ge @ ws16.float()                                                   # d/d(memory rows): score head ...
        if g_feat is not None:
            gm = gm + g_feat.reshape(R, N).float()                               # ... + the picked features' own gradient
        gws = ge.t() @ top_feat.reshape(R, N).float()
        gbs = ge.sum(0)
        st = st.reshape(R, 2)
        xh = (yr.reshape(R, N).float() - st[:, :1]) * st[:, 1:]                  # normalised rows from the forward's (mean, rstd)
        ggam, gbet = (gm * xh).sum(0), gm.sum(0)
        gxh = gm * g32
        gy = st[:, 1:] * (gxh - gxh.mean(-1, keepdim=True) - xh * (gxh * xh).mean(-1, keepdim=True))
        m = vr.reshape(R, 1).float()                                             # rows of invalid anchors entered the Linear as zeros
        xm = xr.reshape(R, K).float() * m
        gw, gb = gy.t() @ xm, gy.sum(0)
        gxr = ((gy @ w16.float()) * m).to(gx.dtype).view(B, -1, K)
        bi = torch.arange(B, device=top.device).unsqueeze(-1)
        if distinct:
            gx[bi, top] = gx[bi, top] + gxr                                      # top-k indices are distinct per image: plain gather / scatter
        else:
            gx.index_put_((bi, top), gxr, accumulate=True)                       # injected picks may repeat a row: the (sorted, ordered) accumulating form
        return (gx, gw.to(w_dt), gb.to(b_dt), ggam.to(ga_dt), gbet.to(be_dt), None, gws.to(ws_dt), gbs.to(bs_dt), None, None, None, None)


def enc_select(x, lin, norm, score_head, invalid, nq, fixed_top=None, n_dec=0):
    """(top_feat [B, nq, hd], enc_scores [B, nq, nc], top [B, nq], n_dec handles on x) - see _EncSelect."""
    out = _EncSelect.apply(x, lin.weight, lin.bias, norm.weight, norm.bias, norm.eps, score_head.weight, score_head.bias, invalid, int(nq),
                           fixed_top, int(n_dec))
    return out[0], out[1], out[2], list(out[3:])


def enc_select_ok(x, lin, norm, score_head):
    """The one-node query selection serves the bf16 HIP path in training (TAMTR_ENC_SELECT=dense: the separate ops, A/B)."""
    return (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 3 and torch.is_grad_enabled() and x.requires_grad
            and isinstance(lin, torch.nn.Linear) and isinstance(score_head, torch.nn.Linear) and lin.bias is not None and score_head.bias is not None
            and lin.in_features % 64 == 0 and lin.out_features % 128 == 0 and lin.out_features in (32, 64, 128, 256, 512, 1024)
            and _os.environ.get('TAMTR_ENC_SELECT') != 'dense')


class _EmbedRows(torch.autograd.Function):
    """weight[idx] for a SMALL table and MANY lookups (the denoising queries' class embeddings, reference models/utils/ops.py:215-216:
    3 072 lookups into 11 rows at the bench shape).  torch's backward of that index is its sorted `index_put` accumulate, which walks the
    duplicates of a row one after the other: 293 us for a [11, 256] gradient.  Here: dW = onehot(idx)^T @ g, one tiny GEMM, fixed order."""

    @staticmethod
    def forward(ctx, weight, idx):
        ctx.save_for_backward(idx)
        ctx.rows = weight.shape[0]
        return weight.index_select(0, idx)

    @staticmethod
    def backward(ctx, g):
        idx, = ctx.saved_tensors
        onehot = (idx.view(1, -1) == torch.arange(ctx.rows, device=idx.device).view(-1, 1)).to(torch.float32)   # [rows, n]
        return (onehot @ g.reshape(idx.numel(), -1).float()).to(g.dtype).view(ctx.rows, *g.shape[1:]), None


def embed_rows(weight, idx):
    """weight[idx] (idx 1-D int64); on the GPU with a gradient the one-hot product backward of _EmbedRows (TAMTR_EMBED_ROWS=torch: plain index)."""
    if (weight.is_cuda and idx.dim() == 1 and weight.dim() == 2 and weight.shape[0] <= 4096 and weight.requires_grad and torch.is_grad_enabled()
            and _os.environ.get('TAMTR_EMBED_ROWS') != 'torch'):
        return _EmbedRows.apply(weight, idx)
    return weight[idx]


def fanout(x, n):
    """x -> n tensors with x's values (views), whose gradients are summed in one kernel; for tensors that feed several heavy consumers."""
    if n <= 1 or not (x.requires_grad and torch.is_grad_enabled()):
        return (x,) * max(n, 1)
    return _Fanout.apply(x, int(n))


def linear_bf16(x, weight, bias=None):
    """x [..., K] bf16, weight [N, K], bias [N] -> [..., N] bf16 (transformer.py:273 value_proj)."""
    return _LinearBF16.apply(x, weight, bias)


# ------------------------------------------------------------------------------------------------ trunk: 1x1 convolutions' weight gradient
_CONV1X1_NO_MASTER = _os.environ.get('TAMTR_CONV1X1_MASTER') == '0'   # A/B switch: the weight gradient rounded to bf16 for the cast group's copy


class _Conv1x1CL(torch.autograd.Function):
    """y = conv2d(x, w) for a 1x1 / stride 1 / ungrouped convolution on a channels-last map, with the WEIGHT gradient taken off MIOpen.
    The forward stays on the library (tuned tables).  MIOpen's weight-gradient solvers for these shapes (and, for some, its input-gradient
    solvers) split the reduction across workgroups and add with atomics into a buffer they zero with hipMemsetAsync - a memset NODE in a recorded
    graph, the one node kind that does not replay in order under the HIP runtime's AQL packet capture (profiles/r04_packet_capture_bisect.txt:
    inf / NaN weight gradients on exactly the trunk's 1x1 convolutions).  A 1x1 convolution IS a per-pixel linear map, so its backward is
    two GEMMs over the [B*H*W, C] views of the channels-last maps: dX = dY W (library GEMM) and dW [C2, C1] = dY^T X as the row-sliced
    batched product + ordered slab sum of the tall linears (dw_splitk, slab_sum) - fp32 result, bitwise reproducible, no memset, no atomics."""

    @staticmethod
    def forward(ctx, x, w, master=None):
        # master: the fp32 parameter `w` is this step's bf16 copy of (model._CastGroup).  The weight gradient comes out of the slab sum in
        # fp32: it goes to the master as it is - not rounded to bf16 for the copy's sake and widened again by the cast group's backward
        # (one cast kernel per 1x1 convolution and step, ~100 on the TAM-TR-s trunk) - and the copy gets no gradient.
        xp = x if _is_cl(x) else _pack_cl(x)            # a channel slice of a wider map: packed once, kept for the backward
        ctx.save_for_backward(xp, w)
        ctx.to_master = master is not None
        return torch.nn.functional.conv2d(xp, w)

    @staticmethod
    def backward(ctx, gy):
        xp, w = ctx.saved_tensors
        B, C1, H, W = xp.shape
        C2 = w.shape[0]
        gy = gy.to(xp.dtype)
        gp = gy if _is_cl(gy) else (_pack_cl(gy) if _cl_pitch(gy) else gy.contiguous(memory_format=torch.channels_last))
        g2, x2 = gp.permute(0, 2, 3, 1).reshape(B * H * W, C2), xp.permute(0, 2, 3, 1).reshape(B * H * W, C1)   # views of packed NHWC maps
        gx = gw = None
        if ctx.needs_input_grad[0]:   # dX [M, C1] = dY [M, C2] W [C2, C1]: a library GEMM (MIOpen's input-gradient solvers for some of these shapes memset too)
            gx = torch.mm(g2, w.view(C2, C1)).view(B, H, W, C1).permute(0, 3, 1, 2)
        if ctx.to_master:
            return gx, None, (dw_splitk(g2, x2).view(C2, C1, 1, 1) if ctx.needs_input_grad[2] else None)
        if ctx.needs_input_grad[1]:
            gw = dw_splitk(g2, x2).view(C2, C1, 1, 1).to(w.dtype)
        return gx, gw, None


def _pack_cl(x):
    """Packed channels-last copy of a channel slice of a channels-last map (no autograd: for use inside Functions)."""
    B, C, H, W = x.shape
    out = torch.empty((B, C, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    call('tamtr_copy_rows', ptr(x), _cl_pitch(x), ptr(out), C, B * H * W, C, dtype_code(x), stream_ptr())
    return out


def conv1x1_cl_ok(x, conv):
    """A plain 1x1 / stride 1 / unpadded / ungrouped / unbiased nn.Conv2d on a channels-last (or channel-slice) fp32 / bf16 CUDA map whose
    weight needs a gradient: the case _Conv1x1CL serves."""
    return (x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16) and torch.is_grad_enabled()
            and isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None
            and (conv.weight.requires_grad or getattr(conv.weight, '_tamtr_master', None) is not None)
            and bool(_cl_pitch(x)) and not x.is_contiguous() and x.shape[1] % 8 == 0 and x.data_ptr() % 16 == 0 and _cl_pitch(x) % 8 == 0)


def conv2d_module(conv, x):
    """conv(x) for an nn.Conv2d of the trunk; 1x1 convolutions on channels-last maps take _Conv1x1CL (library forward / input gradient,
    own weight gradient), everything else the module itself."""
    if conv1x1_cl_ok(x, conv):
        w = conv.weight
        if torch.is_autocast_enabled('cuda') and w.dtype != x.dtype and x.dtype == torch.get_autocast_dtype('cuda'):
            w = w.to(x.dtype)    # (what autocast does inside conv2d; the trunk normally hands in the layer's bf16 copies already)
        if w.dtype == x.dtype:
            master = None if _CONV1X1_NO_MASTER else getattr(w, '_tamtr_master', None)   # set by model.token_memory on the cast group's copies
            if master is not None and master.requires_grad and master.dtype == torch.float32 and master.shape == w.shape:
                return _Conv1x1CL.apply(x, w, master)
            return _Conv1x1CL.apply(x, w)
    return conv(x)


# ------------------------------------------------------------------------------------------------ a-7 self-attention
_mask_cache = []   # [(mask tensor, its _version, packed words)]: the entry keeps the mask alive, so its address cannot be reused


def pack_mask(mask):
    """bool [Q,Q] (True = blocked) -> int32 bit words [Q, ceil(Q/32)]; cached per mask tensor (reused by all layers).
    The hit test is object identity + version: an address/shape key could match a NEW mask allocated in a freed one's block."""
    if _mask_cache and _mask_cache[0][0] is mask and _mask_cache[0][1] == mask._version:
        return _mask_cache[0][2]
    Q = mask.shape[1]
    W = (Q + 31) // 32
    m = torch.zeros(mask.shape[0], W * 32, dtype=torch.int64, device=mask.device)
    m[:, :Q] = mask.to(torch.int64)
    words = (m.view(mask.shape[0], W, 32) << torch.arange(32, device=mask.device)).sum(-1)
    words = torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32).contiguous()
    _mask_cache[:] = [(mask, mask._version, words)]
    return words


class _SelfAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, nh, mask_bits):
        require_gpu(q, k, v)
        B, Q, C = q.shape
        dh = C // nh
        for t in (q, k, v):
            if t.stride(-1) != 1 or t.stride(0) != Q * t.stride(1):
                raise _lib.TamtrHipError('self_attention operands must be row-strided views [B,Q,C] of a packed projection')
        if not (q.dtype == k.dtype == v.dtype):
            raise _lib.TamtrHipError('self_attention operands must share a dtype')
        o = torch.empty(B, Q, C, device=q.device, dtype=q.dtype)
        lse = torch.empty(B, nh, Q, device=q.device, dtype=torch.float32)
        call('tamtr_selfattn_fwd', ptr(q), ptr(k), ptr(v), ptr(mask_bits), ptr(o), ptr(lse), B, Q, nh, dh, q.stride(1), k.stride(1),
             v.stride(1), dtype_code(q), stream_ptr())
        ctx.save_for_backward(q, k, v, o, lse, mask_bits)
        ctx.nh = nh
        return o

    @staticmethod
    def backward(ctx, go):
        q, k, v, o, lse, mask_bits = ctx.saved_tensors
        nh = ctx.nh
        B, Q, C = q.shape
        go = _c(go.to(q.dtype))
        gq, gk, gv = torch.empty_like(o), torch.empty_like(o), torch.empty_like(o)
        ws = torch.empty(B, nh, Q, device=q.device, dtype=torch.float32)
        call('tamtr_selfattn_bwd', ptr(go), ptr(q), ptr(k), ptr(v), ptr(o), ptr(lse), ptr(mask_bits), ptr(gq), ptr(gk), ptr(gv),
             ptr(ws), B, Q, nh, C // nh, q.stride(1), k.stride(1), v.stride(1), dtype_code(q), stream_ptr())
        return gq, gk, gv, None, None


def self_attention(q, k, v, nh, attn_mask=None):
    """softmax(q k^T / sqrt(dh) + mask) v per head; q,k,v [B,Q,C] (may be column slices of a packed projection);
    attn_mask bool [Q,Q], True = blocked (transformer.py:546)."""
    bits = pack_mask(attn_mask) if attn_mask is not None else None
    return _SelfAttention.apply(q, k, v, int(nh), bits)


# ------------------------------------------------------------------------------------------------ a-9 selective scan
def _scan_row_sums(Bn, KD, device):
    """Workspace of the scan backward's per-(image, row) sums over time (include/tamtr_hip.h: grow)."""
    return torch.empty(Bn, KD, _lib.lib().tamtr_selective_scan_row_sums(), device=device, dtype=torch.float32)


def _split_row_sums(grow, R):
    """Add the images (a fixed-order reduction: no float atomics, bitwise reproducible) and cut the row into d(Wdt) [KD, R], dA [KD, 16],
    dD [KD], d(bias) [KD]."""
    rs = slab_sum(grow)
    return rs[:, 16:16 + R], rs[:, :16], rs[:, 48], rs[:, 49]


class _SelectiveScan(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, delta, A, Bm, Cm, D, dbias, xmode):
        require_gpu(u, delta, A, Bm, Cm, D, dbias)
        Bn, KD, L = delta.shape
        K, N = Bm.shape[1], Bm.shape[2]
        u, delta, A, Bm, Cm, D, dbias = (_c(t.float()) for t in (u, delta, A, Bm, Cm, D, dbias))
        chunk = _lib.lib().tamtr_selective_scan_chunk()
        nchunk = (L + chunk - 1) // chunk
        y = torch.empty_like(delta)
        hstate = torch.empty(Bn, KD, nchunk, N, device=u.device, dtype=torch.float32)
        call('tamtr_selective_scan_fwd', ptr(u), ptr(delta), ptr(A), ptr(Bm), ptr(Cm), ptr(D), ptr(dbias), ptr(y), ptr(hstate), Bn, K,
             KD // K, N, L, int(xmode), stream_ptr())
        ctx.save_for_backward(u, delta, A, Bm, Cm, D, dbias, hstate)
        ctx.xmode = int(xmode)
        return y

    @staticmethod
    def backward(ctx, gy):
        u, delta, A, Bm, Cm, D, dbias, hstate = ctx.saved_tensors
        Bn, KD, L = delta.shape
        K, N = Bm.shape[1], Bm.shape[2]
        gy = _c(gy.float())
        gu, gdelta = torch.empty_like(delta), torch.empty_like(delta)
        gB, gC = torch.empty_like(Bm), torch.empty_like(Cm)
        grow = _scan_row_sums(Bn, KD, u.device)
        nslab = _lib.lib().tamtr_selective_scan_bwd_slabs(KD // K)
        ws = torch.empty(2 * nslab * Bm.numel(), device=u.device, dtype=torch.float32)  # per-workgroup dB/dC slabs
        call('tamtr_selective_scan_bwd', ptr(gy), ptr(u), ptr(delta), ptr(A), ptr(Bm), ptr(Cm), ptr(D), ptr(dbias), ptr(hstate), ptr(gu),
             ptr(gdelta), ptr(grow), ptr(gB), ptr(gC), ptr(ws), Bn, K, KD // K, N, L, ctx.xmode, stream_ptr())
        _, gA, gD, gbias = _split_row_sums(grow, 0)
        if ctx.xmode:  # [B, 4*Dk, L] per direction (un-reversed) -> gradient of the two stored copies [B, 2, Dk, L]
            g4 = gu.view(Bn, 4, KD // 4, L)
            gu = g4[:, :2] + g4[:, 2:]
        return gu, gdelta, gA, gB, gC, gD, gbias, None


def selective_scan(u, delta, A, Bm, Cm, D, delta_bias):
    """S6 scan with softplus(delta + bias): u, delta [B,K*Dk,L]; A [K*Dk,16]; Bm, Cm [B,K,16,L]; D, delta_bias [K*Dk]."""
    return _SelectiveScan.apply(u, delta, A, Bm, Cm, D, delta_bias, 0)


def selective_scan_cross_delta(u2, delta, A, Bm, Cm, D, delta_bias):
    """Cross-scan layout with a materialised delta (kept for testing the layout on its own)."""
    return _SelectiveScan.apply(u2, delta, A, Bm, Cm, D, delta_bias, 1)


class _SelectiveScanDtProj(torch.autograd.Function):
    """Scan with the dt projection fused in: delta = Wdt . dtr is formed inside the kernels (never materialised)."""

    @staticmethod
    def forward(ctx, u, dtr, Wdt, A, Bm, Cm, D, dbias, xmode):
        require_gpu(u, dtr, Wdt, A, Bm, Cm, D, dbias)
        Bn, K, R, L = dtr.shape
        KD, N = A.shape
        u, dtr, Wdt, A, Bm, Cm, D, dbias = (_c(t.float()) for t in (u, dtr, Wdt, A, Bm, Cm, D, dbias))
        chunk = _lib.lib().tamtr_selective_scan_chunk()
        nchunk = (L + chunk - 1) // chunk
        y = torch.empty(Bn, KD, L, device=u.device, dtype=torch.float32)
        hstate = torch.empty(Bn, KD, nchunk, N, device=u.device, dtype=torch.float32)
        call('tamtr_selective_scan_dtproj_fwd', ptr(u), ptr(dtr), ptr(Wdt), ptr(A), ptr(Bm), ptr(Cm), ptr(D), ptr(dbias), ptr(y),
             ptr(hstate), Bn, K, KD // K, N, R, L, int(xmode), 0, stream_ptr())
        ctx.save_for_backward(u, dtr, Wdt, A, Bm, Cm, D, dbias, hstate)
        ctx.xmode = int(xmode)
        return y

    @staticmethod
    def backward(ctx, gy):
        u, dtr, Wdt, A, Bm, Cm, D, dbias, hstate = ctx.saved_tensors
        Bn, K, R, L = dtr.shape
        KD, N = A.shape
        gy = _c(gy.float())
        gu = torch.empty(Bn, KD, L, device=u.device, dtype=torch.float32)
        gdelta = torch.empty(Bn, KD, L, device=u.device, dtype=torch.float32)  # workspace between the two backward kernels
        gdtr = torch.empty_like(dtr)
        gB, gC = torch.empty_like(Bm), torch.empty_like(Cm)
        grow = _scan_row_sums(Bn, KD, dtr.device)
        nslab = _lib.lib().tamtr_selective_scan_bwd_slabs(KD // K)
        ws = torch.empty(2 * nslab * Bm.numel(), device=u.device, dtype=torch.float32)
        call('tamtr_selective_scan_dtproj_bwd', ptr(gy), ptr(u), ptr(dtr), ptr(Wdt), ptr(A), ptr(Bm), ptr(Cm), ptr(D), ptr(dbias),
             ptr(hstate), ptr(gu), ptr(gdelta), ptr(gdtr), ptr(grow), ptr(gB), ptr(gC), ptr(ws), Bn, K,
             KD // K, N, R, L, ctx.xmode, 0, stream_ptr())
        gW, gA, gD, gbias = _split_row_sums(grow, R)
        if ctx.xmode:
            g4 = gu.view(Bn, 4, KD // 4, L)
            gu = g4[:, :2] + g4[:, 2:]
        return gu, gdtr, gW, gA, gB, gC, gD, gbias, None


class _CrossScanInput(torch.autograd.Function):
    """u2 [B, 2, D, H*W] fp32 = SiLU(xc) in its row-major and column-major flattening (what CrossScan, csms6s.py:4-14, needs
    in the pair layout of the scan kernels) as one autograd node: three kernels forward, three backward, instead of the
    silu / float / stack / transpose chain and its slice-and-add autograd graph."""

    @staticmethod
    def forward(ctx, xc):
        B, D, H, W = xc.shape
        a = torch.nn.functional.silu(xc)  # in xc's dtype, as the reference's act (vmamba.py:951)
        u2 = torch.empty(B, 2, D, H * W, device=xc.device, dtype=torch.float32)
        u2[:, 0].view(B, D, H, W).copy_(a)
        u2[:, 1].view(B, D, W, H).copy_(a.transpose(2, 3))
        ctx.save_for_backward(xc)
        return u2

    @staticmethod
    def backward(ctx, g2):
        (xc,) = ctx.saved_tensors
        B, D, H, W = xc.shape
        ga = g2[:, 0].view(B, D, H, W) + g2[:, 1].view(B, D, W, H).transpose(2, 3)
        return torch.ops.aten.silu_backward(ga.to(xc.dtype), xc)


def cross_scan_input(xc):
    return _CrossScanInput.apply(xc)


class _DWConvSiluCross(torch.autograd.Function):
    """SS2D front end on the in_proj output as it lies: xz [B, H, W, 2*D] (xi = the first D channels of every pixel) ->
    u2 [B, 2, D, H*W] fp32 = SiLU(dwconv3x3(xi) + bias) in both flattenings (csrc/dwconv.hip)."""

    @staticmethod
    def forward(ctx, xz, weight, bias, D):
        require_gpu(xz, weight)
        xz = _c(xz)
        B, H, W, C2 = xz.shape
        w = _c(weight.float().reshape(D, 9))
        bvec = _c(bias.float()) if bias is not None else None
        u2 = torch.empty(B, 2, D, H * W, device=xz.device, dtype=torch.float32)
        call('tamtr_dwconv_silu_cross_fwd', ptr(xz), C2, ptr(w), ptr(bvec), ptr(u2), B, D, H, W, dtype_code(xz), 0, stream_ptr())
        ctx.save_for_backward(xz, w, bvec if bvec is not None else w.new_empty(0))
        ctx.cfg = (D, weight.shape, weight.dtype, None if bias is None else bias.dtype)
        return u2

    @staticmethod
    def backward(ctx, g2):
        xz, w, bvec = ctx.saved_tensors
        D, w_shape, w_dt, b_dt = ctx.cfg
        B, H, W, C2 = xz.shape
        tiles = _lib.lib().tamtr_dwconv_tiles(H, W)
        gxz = torch.zeros_like(xz)  # the second half (z) gets its gradient from the gate path; autograd adds the two
        ws = torch.empty(B, tiles, D, 10, device=xz.device, dtype=torch.float32)
        call('tamtr_dwconv_silu_cross_bwd', ptr(_c(g2.float())), ptr(xz), C2, ptr(w), ptr(bvec if b_dt is not None else None), ptr(gxz), C2,
             ptr(ws), B, D, H, W, dtype_code(xz), 0, stream_ptr())
        gwb = slab_sum(ws.view(-1, D, 10))
        gw = gwb[:, :9].reshape(w_shape).to(w_dt)
        gb = gwb[:, 9].to(b_dt) if b_dt is not None else None
        return gxz, gw, gb, None


def dwconv_silu_cross(xz, weight, bias, D):
    return _DWConvSiluCross.apply(xz, weight, bias, D)


def _split_len(L, min_len=1024, max_split=16):
    S = 1
    while S < max_split and L % (2 * S) == 0 and L // (2 * S) >= min_len:
        S *= 2
    return S


class _XProjCross(torch.autograd.Function):
    """x_proj of SS2D (vmamba.py:962-970) on the pair layout: wx [4, C, D] (C = R + 2N), u2 [B, 2, D, L] ->
    dtr [B,4,R,L], Bs [B,4,N,L], Cs [B,4,N,L] (fp32, contiguous, un-reversed).  Directions k and k+2 share a base copy, so it
    is two [2C, D] x [D, L] products; their weight gradient (a reduction over L per image) runs as a batched GEMM over L
    slices, and the output assembly has an explicit backward (one concatenation per copy instead of ~30 slice kernels)."""

    @staticmethod
    def forward(ctx, wx, u2, R, N):
        cdt = torch.get_autocast_dtype('cuda') if torch.is_autocast_enabled('cuda') else torch.float32
        C = R + 2 * N
        ub = u2.to(cdt)
        wa, wb = torch.cat([wx[0], wx[2]], 0).to(cdt), torch.cat([wx[1], wx[3]], 0).to(cdt)
        with torch.autocast('cuda', enabled=False):
            xa, xb = torch.matmul(wa, ub[:, 0]), torch.matmul(wb, ub[:, 1])  # [B, 2C, L]
        parts = []
        for lo, n in ((0, R), (R, N), (R + N, N)):
            parts.append(torch.stack([xa[:, lo:lo + n], xb[:, lo:lo + n], xa[:, C + lo:C + lo + n], xb[:, C + lo:C + lo + n]], 1).float())
        ctx.save_for_backward(ub, wa, wb)
        ctx.cfg = (R, N, wx.dtype, u2.dtype)
        return tuple(parts)

    @staticmethod
    def backward(ctx, gdtr, gBs, gCs):
        ub, wa, wb = ctx.saved_tensors
        R, N, w_dt, u_dt = ctx.cfg
        cdt = ub.dtype
        Bn, _, D, L = ub.shape
        C = R + 2 * N
        S = _split_len(L)
        gws = []
        gu2 = torch.empty(Bn, 2, D, L, device=ub.device, dtype=u_dt)  # each product is cast straight into its plane (no stack + cast)
        with torch.autocast('cuda', enabled=False):
            for i, w in ((0, wa), (1, wb)):
                gx = torch.cat([gdtr[:, i], gBs[:, i], gCs[:, i], gdtr[:, i + 2], gBs[:, i + 2], gCs[:, i + 2]], 1).to(cdt)  # [B, 2C, L]
                gu2[:, i].copy_(torch.matmul(w.t(), gx))
                ga = gx.view(Bn, 2 * C, S, L // S).transpose(1, 2).reshape(Bn * S, 2 * C, L // S)
                ua = ub[:, i].reshape(Bn, D, S, L // S).transpose(1, 2).reshape(Bn * S, D, L // S)
                gws.append(slab_sum(torch.bmm(ga, ua.transpose(1, 2))))  # [2C, D]
        gwx = torch.stack([gws[0][:C], gws[1][:C], gws[0][C:], gws[1][C:]], 0).to(w_dt)
        return gwx, gu2, None, None


def x_proj_cross(wx, u2, R, N):
    return _XProjCross.apply(wx, u2, R, N)


class _SelectiveScanCrossMerged(torch.autograd.Function):
    """Cross-scan + fused dt projection + CrossMerge (csms6s.py:4-46) as one autograd node: forward returns the merged map
    [B, Dk, H*W]; backward hands the scan kernel the merged gradient in its two flattenings instead of four planes (the
    slice/add autograd graph of the merge cost 5.4 ms per level-0 block: 1.7 GB zero fills, copies and adds)."""

    @staticmethod
    def forward(ctx, u2, dtr, Wdt, A, Bm, Cm, D, dbias, H, W, token_major=False):
        require_gpu(u2, dtr, Wdt, A, Bm, Cm, D, dbias)
        u2, dtr, Wdt, A, Bm, Cm, D, dbias = (_c(t.float()) for t in (u2, dtr, Wdt, A, Bm, Cm, D, dbias))
        Bn, _, Dk, L = u2.shape
        K, R, N = 4, dtr.shape[2], A.shape[1]
        chunk = _lib.lib().tamtr_selective_scan_chunk()
        nchunk = (L + chunk - 1) // chunk
        y = torch.empty(Bn, K, Dk, L, device=u2.device, dtype=torch.float32)
        hstate = torch.empty(Bn, K * Dk, nchunk, N, device=u2.device, dtype=torch.float32)
        call('tamtr_selective_scan_dtproj_fwd', ptr(u2), ptr(dtr), ptr(Wdt), ptr(A), ptr(Bm), ptr(Cm), ptr(D), ptr(dbias), ptr(y),
             ptr(hstate), Bn, K, Dk, N, R, L, 1, 0, stream_ptr())
        ctx.save_for_backward(u2, dtr, Wdt, A, Bm, Cm, D, dbias, hstate)
        ctx.hw = (H, W, token_major)
        if token_major:  # CrossMerge straight into [B, L, Dk] (what out_norm / out_proj consume): one tiled-transpose kernel
            ymT = torch.empty(Bn, L, Dk, device=u2.device, dtype=torch.float32)
            call('tamtr_cross_merge_fwd', ptr(y), ptr(ymT), Bn, Dk, H, W, 0, stream_ptr())
            return ymT
        ym = y[:, 0] + y[:, 2]
        ym += (y[:, 1] + y[:, 3]).view(Bn, Dk, W, H).transpose(2, 3).reshape(Bn, Dk, L)
        return ym

    @staticmethod
    def backward(ctx, gm):
        u2, dtr, Wdt, A, Bm, Cm, D, dbias, hstate = ctx.saved_tensors
        H, W, token_major = ctx.hw
        Bn, K, R, L = dtr.shape
        Dk, N = u2.shape[2], A.shape[1]
        KD = K * Dk
        g2 = torch.empty(Bn, 2, Dk, L, device=u2.device, dtype=torch.float32)
        if token_major:
            call('tamtr_cross_merge_bwd', ptr(_c(gm.float())), ptr(g2), Bn, Dk, H, W, 0, stream_ptr())
        else:
            g2[:, 0] = gm
            g2[:, 1].view(Bn, Dk, W, H).copy_(gm.view(Bn, Dk, H, W).transpose(2, 3))
        gu = torch.empty(Bn, KD, L, device=u2.device, dtype=torch.float32)
        gdelta = torch.empty(Bn, KD, L, device=u2.device, dtype=torch.float32)
        gdtr = torch.empty_like(dtr)
        gB, gC = torch.empty_like(Bm), torch.empty_like(Cm)
        grow = _scan_row_sums(Bn, KD, dtr.device)
        nslab = _lib.lib().tamtr_selective_scan_bwd_slabs(Dk)
        ws = torch.empty(2 * nslab * Bm.numel(), device=u2.device, dtype=torch.float32)
        call('tamtr_selective_scan_dtproj_bwd', ptr(g2), ptr(u2), ptr(dtr), ptr(Wdt), ptr(A), ptr(Bm), ptr(Cm), ptr(D), ptr(dbias),
             ptr(hstate), ptr(gu), ptr(gdelta), ptr(gdtr), ptr(grow), ptr(gB), ptr(gC), ptr(ws), Bn, K,
             Dk, N, R, L, 3, 0, stream_ptr())
        gW, gA, gD, gbias = _split_row_sums(grow, R)
        g4 = gu.view(Bn, 4, Dk, L)
        return g4[:, :2] + g4[:, 2:], gdtr, gW, gA, gB, gC, gD, gbias, None, None, None


def selective_scan_cross_merged(u2, dtr, Wdt, A, Bm, Cm, D, delta_bias, H, W, token_major=False):
    """selective_scan_cross followed by the cross-merge: returns [B, Dk, H*W] in row-major pixel order, or with
    token_major=True [B, H*W, Dk] (needs Dk % 32 == 0)."""
    return _SelectiveScanCrossMerged.apply(u2, dtr, Wdt, A, Bm, Cm, D, delta_bias, H, W, token_major)


class _LNGate(torch.autograd.Function):
    """out = LayerNorm(x; gamma, beta) * SiLU(z) with z = the second half of the channels-last in_proj output xz [B,H,W,2D]
    (vmamba.py:1005-1008,1029-1036) - csrc/ss2d_out.hip.  x fp32 [B, L, D] -> out [B, L, D] in xz's dtype."""

    @staticmethod
    def forward(ctx, x, xz, gamma, beta, eps):
        require_gpu(x, xz, gamma, beta)
        x, xz = _c(x.float()), _c(xz)
        D = x.shape[-1]
        ntok = x.numel() // D
        g32, b32 = _c(gamma.float()), _c(beta.float())
        out = torch.empty(x.shape, device=x.device, dtype=xz.dtype)
        stats = torch.empty(ntok, 2, device=x.device, dtype=torch.float32)
        call('tamtr_ln_gate_fwd', ptr(x), ptr(xz), xz.shape[-1], ptr(g32), ptr(b32), ptr(out), ptr(stats), ntok, D, float(eps),
             dtype_code(xz), stream_ptr())
        ctx.save_for_backward(x, xz, g32, b32, stats)
        ctx.cfg = (gamma.dtype, beta.dtype)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, xz, g32, b32, stats = ctx.saved_tensors
        D = x.shape[-1]
        ntok = x.numel() // D
        gout = _c(gout.to(xz.dtype))
        gx = torch.empty_like(x)
        gxz = torch.zeros_like(xz)  # the xi half gets its gradient from the conv path; autograd adds the two
        nblk = _lib.lib().tamtr_ln_gate_blocks(ntok)
        part = torch.empty(nblk, 2, D, device=x.device, dtype=torch.float32)
        call('tamtr_ln_gate_bwd', ptr(gout), ptr(x), ptr(xz), xz.shape[-1], ptr(g32), ptr(b32), ptr(stats), ptr(gx), ptr(gxz), ptr(part),
             ntok, D, dtype_code(xz), stream_ptr())
        gsum = slab_sum(part)
        return gx, gxz, gsum[0].to(ctx.cfg[0]), gsum[1].to(ctx.cfg[1]), None


def ln_gate(x, xz, gamma, beta, eps=1e-5):
    return _LNGate.apply(x, xz, gamma, beta, eps)


# ------------------------------------------------------------------------------------------------ a-10 loss terms (csrc/detrloss.hip)
class _DetrLayerLosses(torch.autograd.Function):
    """(class, bbox, giou) terms of all stacked decoder layers of one DETRLoss._layers call (loss.py:85-166,282-326 of the reference's
    models/utils: varifocal class loss on the matched IoU, 5 x L1, 2 x (1 - RIOU)) in three launches forward and two backward
    (csrc/detrloss.hip) instead of ~190 elementwise / gather / scatter / reduce launches each way.  fp32; sums in a fixed order."""

    @staticmethod
    def forward(ctx, pb, ps, gt_bboxes, gt_cls, li, bi, si, gi, n, gains):
        require_gpu(pb, ps, gt_bboxes, gt_cls, li, bi, si, gi)
        pb, ps, gt_bboxes = _c(pb.float()), _c(ps.float()), _c(gt_bboxes.float())
        gt_cls, li, bi, si, gi = (_c(t.long()) for t in (gt_cls, li, bi, si, gi))
        Lr, B, nq, nc = ps.shape
        dev = pb.device
        tgt = torch.empty(Lr, B, nq, device=dev, dtype=torch.int64).fill_(nc)      # (fill kernels, not memsets: tam-tr_amd/graphs.py)
        score = torch.empty(Lr, B, nq, device=dev, dtype=torch.float32).fill_(0.0)
        pair = torch.empty(2, Lr * n, device=dev, dtype=torch.float32)
        partial = torch.empty(Lr * _lib.lib().tamtr_detr_blocks(B * nq), device=dev, dtype=torch.float32)
        out = torch.empty(3, Lr, device=dev, dtype=torch.float32)
        gc, gb, gg = (float(v) for v in gains)
        call('tamtr_detr_layers_fwd', ptr(pb), ptr(ps), ptr(gt_bboxes), ptr(gt_cls), ptr(li), ptr(bi), ptr(si), ptr(gi), Lr, B, nq, nc, n, ptr(tgt),
             ptr(score), ptr(pair[0]), ptr(pair[1]), ptr(partial), gc, gb, gg, ptr(out), stream_ptr())
        ctx.save_for_backward(pb, ps, gt_bboxes, li, bi, si, gi, tgt, score)
        ctx.cfg = (n, gc, gb, gg)
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, g_cls, g_box, g_iou):
        pb, ps, gt_bboxes, li, bi, si, gi, tgt, score = ctx.saved_tensors
        n, gc, gb, gg = ctx.cfg
        Lr, B, nq, nc = ps.shape
        up = torch.stack([g_cls, g_box, g_iou]).float().contiguous()
        gpb = torch.empty_like(pb).fill_(0.0)
        gps = torch.empty_like(ps)
        call('tamtr_detr_layers_bwd', ptr(pb), ptr(ps), ptr(gt_bboxes), ptr(li), ptr(bi), ptr(si), ptr(gi), ptr(tgt), ptr(score), ptr(up), Lr, B, nq, nc, n,
             gc, gb, gg, ptr(gpb), ptr(gps), stream_ptr())
        return gpb, gps, None, None, None, None, None, None, None, None


def detr_layer_losses(pb, ps, gt_bboxes, gt_cls, li, bi, si, gi, n, gains):
    """pb [Lr,B,nq,4], ps [Lr,B,nq,nc], flat matched pairs (layer-major, n per layer) -> three [Lr] tensors (class, bbox, giou)."""
    return _DetrLayerLosses.apply(pb, ps, gt_bboxes, gt_cls, li, bi, si, gi, int(n), tuple(gains))


@torch.no_grad()
def detr_match_cost(ps, pb, gt_bboxes, gt_cls, gains, alpha, gamma):
    """The Hungarian matcher's cost (models/utils/ops.py:84-112) of all layers at once: ps [Lr,B,nq,nc] logits, pb [Lr,B,nq,4] ->
    C f32 [Lr, B, nq, G] over the flattened box list, non-finite entries zeroed."""
    require_gpu(ps, pb, gt_bboxes, gt_cls)
    Lr, B, nq, nc = ps.shape
    G = gt_bboxes.shape[0]
    ps, pb, gt_bboxes, gt_cls = _c(ps.float()), _c(pb.float()), _c(gt_bboxes.float()), _c(gt_cls.long())
    C = torch.empty(Lr, B, nq, G, device=ps.device, dtype=torch.float32)
    call('tamtr_detr_match_cost', ptr(ps), ptr(pb), ptr(gt_bboxes), ptr(gt_cls), Lr * B * nq, nc, G, float(gains[0]), float(gains[1]), float(gains[2]),
         float(alpha), float(gamma), ptr(C), stream_ptr())
    return C


# ------------------------------------------------------------------------------------------------ x_proj of SS2D (csrc/xproj.hip)
_SS2D_PLANES_F32 = _os.environ.get('TAMTR_SS2D_PLANES') == 'f32'   # A/B switch: the cross-scan planes in fp32 also in bf16 mode (rounds 1-3)


def ss2d_bf16_planes(dtype, D, L, R, N):
    """bf16 mode keeps SS2D's big time-indexed planes in bf16 (include/tamtr_hip.h "bf16 PLANES"): every kernel of the chain has that form
    only on its vector path and with the own x_proj kernels."""
    return dtype == torch.bfloat16 and not _SS2D_PLANES_F32 and L % 8 == 0 and xproj_ok(dtype, D, L, R, N)


_XPROJ_LIB = _os.environ.get('TAMTR_XPROJ') == 'torch'   # A/B switch: the torch-op form (cast, two batched library GEMMs, stack, ...)


def xproj_ok(cdt, D, L, R, N):
    """What tamtr_xproj_{fwd, bwd_dx, bwd_dw} take: the bf16 mode, d_state 16, d_inner a multiple of 256, L a multiple of 8."""
    return not _XPROJ_LIB and cdt == torch.bfloat16 and N == 16 and 1 <= R <= 32 and D % 256 == 0 and L % 8 == 0


def xproj_pack_weight(wx):
    """x_proj_weight [4, C, D] -> bf16 [2, MP, D]: rows [W_i ; W_(i+2)] per stored copy i, zero-padded to a multiple of 32 rows."""
    C = wx.shape[1]
    w = torch.stack([torch.cat([wx[0], wx[2]], 0), torch.cat([wx[1], wx[3]], 0)]).to(torch.bfloat16)
    pad = (-2 * C) % 32
    return _c(torch.nn.functional.pad(w, (0, 0, 0, pad)) if pad else w)


def xproj_pack_weight_t(wcat, C):
    """[2, MP, D] -> bf16 [2, D, KP]: the transposed blocks for the d/d(u2) product, zero-padded to a multiple of 16 columns."""
    w = wcat[:, :2 * C].transpose(1, 2)
    pad = (-2 * C) % 16
    return _c(torch.nn.functional.pad(w, (0, pad)) if pad else w)


class _SS2DCore(torch.autograd.Function):
    """SS2D between in_proj and out_proj as ONE autograd node (vmamba.py:949-1008): depthwise 3x3 + SiLU + cross-scan layout ->
    x_proj -> selective scan with the dt projection inside -> cross-merge -> out_norm x SiLU(z).  Same kernels and the same
    arithmetic as dwconv_silu_cross / x_proj_cross / selective_scan_cross_merged / ln_gate chained; what the single node buys is
    the backward's buffer plan, which autograd otherwise dictates:
      * d/d(u2) = fold of the scan's four planes + the two x_proj products in one pass (csrc/fold.hip) instead of a slice add, a cast
        and an accumulation (5 reads + 3 writes of an 839 MB plane pair at level 0 -> 3 + 1);
      * d/d(xz): the gate kernel writes the z half and the depthwise-conv kernel the xi half of ONE buffer (before: two zero-filled
        [B,H,W,2D] maps and their sum)."""

    @staticmethod
    def forward(ctx, xz, conv_w, conv_b, wx, Wdt, A, Ds, dbias, gamma, beta, eps, R, N):
        require_gpu(xz, conv_w, wx, Wdt, A, Ds, dbias, gamma, beta)
        xz = _c(xz)
        B, H, W, C2 = xz.shape
        D, L, K = C2 // 2, H * W, 4
        # front end
        cw = _c(conv_w.float().reshape(D, 9))
        cb = _c(conv_b.float()) if conv_b is not None else None
        # bf16 mode: the big time-indexed planes that only cross HBM between kernels (u2, y, and in the backward d(y), d(u), d(u2)) are bf16;
        # the recurrence, its states and every gradient sum stay fp32 (include/tamtr_hip.h "bf16 PLANES")
        p16 = ss2d_bf16_planes(xz.dtype, D, L, R, N)
        pdt, pc = (torch.bfloat16, 1) if p16 else (torch.float32, 0)
        u2 = torch.empty(B, 2, D, L, device=xz.device, dtype=pdt)
        call('tamtr_dwconv_silu_cross_fwd', ptr(xz), C2, ptr(cw), ptr(cb), ptr(u2), B, D, H, W, dtype_code(xz), pc, stream_ptr())
        # x_proj on the two copies (directions k and k + 2 share one)
        cdt = xz.dtype if xz.dtype == torch.bfloat16 else torch.float32
        C = R + 2 * N
        own_xp = xproj_ok(cdt, D, L, R, N)
        if own_xp:   # csrc/xproj.hip: u2 read once as f32, the three outputs written in the scan's layout
            wcat = xproj_pack_weight(wx)
            dtr, Bs, Cs = (torch.empty(B, K, n, L, device=xz.device, dtype=torch.float32) for n in (R, N, N))
            call('tamtr_xproj_fwd', ptr(u2), ptr(wcat), ptr(dtr), ptr(Bs), ptr(Cs), B, D, L, R, pc, stream_ptr())
            ub = wa = wb = None
        else:
            wcat = None
            ub = u2.to(cdt)
            wa, wb = torch.cat([wx[0], wx[2]], 0).to(cdt), torch.cat([wx[1], wx[3]], 0).to(cdt)
            with torch.autocast('cuda', enabled=False):
                xa, xb = torch.matmul(wa, ub[:, 0]), torch.matmul(wb, ub[:, 1])  # [B, 2C, L]
            dtr, Bs, Cs = (torch.stack([xa[:, lo:lo + n], xb[:, lo:lo + n], xa[:, C + lo:C + lo + n], xb[:, C + lo:C + lo + n]], 1).float()
                           for lo, n in ((0, R), (R, N), (R + N, N)))
        # scan + cross-merge, token-major
        Wdt32, A32, D32, db32 = (_c(t.float()) for t in (Wdt, A, Ds, dbias))
        chunk = _lib.lib().tamtr_selective_scan_chunk()
        y = torch.empty(B, K, D, L, device=xz.device, dtype=pdt)
        hstate = torch.empty(B, K * D, (L + chunk - 1) // chunk, N, device=xz.device, dtype=torch.float32)
        call('tamtr_selective_scan_dtproj_fwd', ptr(u2), ptr(dtr), ptr(Wdt32), ptr(A32), ptr(Bs), ptr(Cs), ptr(D32), ptr(db32), ptr(y),
             ptr(hstate), B, K, D, N, R, L, 1, pc, stream_ptr())
        ymT = torch.empty(B, L, D, device=xz.device, dtype=torch.float32)
        call('tamtr_cross_merge_fwd', ptr(y), ptr(ymT), B, D, H, W, pc, stream_ptr())
        del y
        # out_norm x SiLU(z)
        g32, b32 = _c(gamma.float()), _c(beta.float())
        out = torch.empty(B, L, D, device=xz.device, dtype=xz.dtype)
        stats = torch.empty(B * L, 2, device=xz.device, dtype=torch.float32)
        call('tamtr_ln_gate_fwd', ptr(ymT), ptr(xz), C2, ptr(g32), ptr(b32), ptr(out), ptr(stats), B * L, D, float(eps), dtype_code(xz),
             stream_ptr())
        none = cw.new_empty(0)
        ctx.save_for_backward(xz, cw, cb if cb is not None else none, u2, ub if ub is not None else none, wa if wa is not None else none,
                              wb if wb is not None else none, wcat if wcat is not None else none, dtr, Bs, Cs, Wdt32, A32, D32, db32, hstate,
                              ymT, g32, b32, stats)
        ctx.own_xp = own_xp
        ctx.cfg = (R, N, H, W, conv_w.shape, conv_w.dtype, None if conv_b is None else conv_b.dtype, wx.dtype, Wdt.dtype, A.dtype, Ds.dtype,
                   dbias.dtype, gamma.dtype, beta.dtype)
        return out

    @staticmethod
    def backward(ctx, gout):
        (xz, cw, cb, u2, ub, wa, wb, wcat, dtr, Bs, Cs, Wdt32, A32, D32, db32, hstate, ymT, g32, b32, stats) = ctx.saved_tensors
        R, N, H, W, cw_shape, cw_dt, cb_dt, wx_dt, wdt_dt, a_dt, d_dt, db_dt, ga_dt, be_dt = ctx.cfg
        B, _, _, C2 = xz.shape
        D, L, K, C = C2 // 2, H * W, 4, R + 2 * N
        dev = xz.device
        # gate: d/d(ymT), the z half of d/d(xz), d(gamma), d(beta)
        gout = _c(gout.to(xz.dtype))
        gy = torch.empty_like(ymT)
        gxz = torch.empty_like(xz)   # z half from the gate kernel here, xi half from the depthwise-conv kernel below
        nblk = _lib.lib().tamtr_ln_gate_blocks(B * L)
        part = torch.empty(nblk, 2, D, device=dev, dtype=torch.float32)
        call('tamtr_ln_gate_bwd', ptr(gout), ptr(ymT), ptr(xz), C2, ptr(g32), ptr(b32), ptr(stats), ptr(gy), ptr(gxz), ptr(part), B * L, D,
             dtype_code(xz), stream_ptr())
        gsum = slab_sum(part)
        # cross-merge and scan
        pdt = u2.dtype                       # the planes' element type (forward's choice)
        pc = 1 if pdt == torch.bfloat16 else 0
        g2 = torch.empty(B, 2, D, L, device=dev, dtype=pdt)
        call('tamtr_cross_merge_bwd', ptr(gy), ptr(g2), B, D, H, W, pc, stream_ptr())
        del gy
        gu = torch.empty(B, K * D, L, device=dev, dtype=pdt)
        # d(delta) workspace: only the operand of gdtr = Wdt^T d(delta); in bf16 mode (gdtr is rounded to bf16 below anyway) kept in bf16
        ws16 = pc or (xz.dtype == torch.bfloat16 and L % 4 == 0)
        gdelta = torch.empty(B, K * D, L, device=dev, dtype=torch.bfloat16 if ws16 else torch.float32)
        gdtr = torch.empty_like(dtr)
        gB, gC = torch.empty_like(Bs), torch.empty_like(Cs)
        grow = _scan_row_sums(B, K * D, dev)
        nslab = _lib.lib().tamtr_selective_scan_bwd_slabs(D)
        ws = torch.empty(2 * nslab * Bs.numel(), device=dev, dtype=torch.float32)
        call('tamtr_selective_scan_dtproj_bwd', ptr(g2), ptr(u2), ptr(dtr), ptr(Wdt32), ptr(A32), ptr(Bs), ptr(Cs), ptr(D32), ptr(db32),
             ptr(hstate), ptr(gu), ptr(gdelta), ptr(gdtr), ptr(grow), ptr(gB), ptr(gC), ptr(ws), B, K, D, N, R, L,
             3, int(bool(ws16)) | (2 * pc), stream_ptr())
        gW, gA, gD, gdb = _split_row_sums(grow, R)
        del gdelta, ws, g2
        gu2 = torch.empty(B, 2, D, L, device=dev, dtype=pdt)
        if ctx.own_xp:
            # csrc/xproj.hip: d/d(u2) = fold of the scan's four planes + Wcat^T G in one pass; dWcat as per-slice partial tiles + ordered sum
            wT = xproj_pack_weight_t(wcat, C)
            call('tamtr_xproj_bwd_dx', ptr(gu), ptr(gdtr), ptr(gB), ptr(gC), ptr(wT), ptr(gu2), B, D, L, R, pc, stream_ptr())
            del gu
            nsl = _lib.lib().tamtr_xproj_dw_slices(L)
            part = torch.empty(B * nsl, 2, 2 * C, D, device=dev, dtype=torch.float32)
            call('tamtr_xproj_bwd_dw', ptr(u2), ptr(gdtr), ptr(gB), ptr(gC), ptr(part), B, D, L, R, pc, stream_ptr())
            gws = slab_sum(part)
            del part
        else:
            # x_proj backward: per copy one [D, 2C] x [2C, L] product and the weight gradient as a batched GEMM over L slices
            cdt = ub.dtype
            S = _split_len(L)
            gws, ms = [], []
            with torch.autocast('cuda', enabled=False):
                for i, w in ((0, wa), (1, wb)):
                    gx = torch.cat([gdtr[:, i], gB[:, i], gC[:, i], gdtr[:, i + 2], gB[:, i + 2], gC[:, i + 2]], 1).to(cdt)  # [B, 2C, L]
                    ms.append(torch.matmul(w.t(), gx))                                                                         # [B, D, L]
                    ga = gx.view(B, 2 * C, S, L // S).transpose(1, 2).reshape(B * S, 2 * C, L // S)
                    ua = ub[:, i].reshape(B, D, S, L // S).transpose(1, 2).reshape(B * S, D, L // S)
                    gws.append(slab_sum(torch.bmm(ga, ua.transpose(1, 2))))  # [2C, D]
            call('tamtr_fold_add', ptr(gu), ptr(_c(ms[0])), ptr(_c(ms[1])), ptr(gu2), B, D * L, dtype_code(ms[0]), stream_ptr())
            del gu, ms
        gwx = torch.stack([gws[0][:C], gws[1][:C], gws[0][C:], gws[1][C:]], 0).to(wx_dt)
        # front end: the xi half of d/d(xz), d(conv weight), d(conv bias)
        tiles = _lib.lib().tamtr_dwconv_tiles(H, W)
        wsd = torch.empty(B, tiles, D, 10, device=dev, dtype=torch.float32)
        call('tamtr_dwconv_silu_cross_bwd', ptr(gu2), ptr(xz), C2, ptr(cw), ptr(cb if cb_dt is not None else None), ptr(gxz), C2, ptr(wsd), B,
             D, H, W, dtype_code(xz), pc, stream_ptr())
        gwb = slab_sum(wsd.view(-1, D, 10))
        gcw = gwb[:, :9].reshape(cw_shape).to(cw_dt)
        gcb = gwb[:, 9].to(cb_dt) if cb_dt is not None else None
        return (gxz, gcw, gcb, gwx, gW.to(wdt_dt), gA.to(a_dt), gD.to(d_dt), gdb.to(db_dt), gsum[0].to(ga_dt), gsum[1].to(be_dt), None, None,
                None)


def ss2d_core(xz, conv_w, conv_b, x_proj_weight, Wdt, A, Ds, dt_bias, out_norm_weight, out_norm_bias, eps, R, N):
    """xz [B,H,W,2D] (in_proj output) -> out_norm(cross-merged scan) * SiLU(z) as [B, H*W, D] in xz's dtype; see _SS2DCore."""
    return _SS2DCore.apply(xz, conv_w, conv_b, x_proj_weight, Wdt, A, Ds, dt_bias, out_norm_weight, out_norm_bias, float(eps), int(R), int(N))


def selective_scan_cross(u2, dtr, Wdt, A, Bm, Cm, D, delta_bias):
    """Cross-scan layout + fused dt projection: u2 [B,2,Dk,L] (row-major / column-major copies), dtr [B,4,R,L] low-rank dt
    factors, Wdt [4*Dk, R]; Bm, Cm [B,4,16,L]; everything stored UN-reversed (directions 2, 3 walk the buffers backwards).
    Returns y [B, 4*Dk, L] (un-reversed)."""
    return _SelectiveScanDtProj.apply(u2, dtr, Wdt, A, Bm, Cm, D, delta_bias, 1)


@torch.no_grad()
def lsap_assign(cost, gt_groups):
    """Device-side Hungarian assignment (models/utils/ops.py:98-119 without the host round trip).
    cost f32 [bs, nq, G] on the GPU, gt_groups: python list of boxes per image (sum == G).
    Returns int64 device tensors (batch_idx, query_idx, gt_idx) of length sum(min(nq, n_b)): scipy's pairs in scipy's
    order, gt_idx already offset into the flattened box list.  No synchronisation."""
    require_gpu(cost)
    bs, nq, G = cost.shape
    groups = [int(n) for n in gt_groups]
    if len(groups) != bs or sum(groups) != G:
        raise _lib.TamtrHipError(f'lsap_assign: group sizes {groups} do not tile cost {tuple(cost.shape)}')
    m = sum(min(nq, n) for n in groups)
    out = torch.empty(3, m, device=cost.device, dtype=torch.int64)
    if m:
        cost = _c(cost.float())
        sizes = (ctypes.c_int32 * bs)(*groups)
        call('tamtr_lsap_assign', ptr(cost), ctypes.cast(sizes, ctypes.c_void_p), bs, nq, G, ptr(out[0]), ptr(out[1]), ptr(out[2]),
             stream_ptr())
    return out[0], out[1], out[2]


def img_augment(src, inv_affine, luts, flags, out_hw, border=114):
    """The pixel half of the training transforms for a whole batch (affine warp -> HSV look-up -> flips -> CHW float / 255;
    ultralytics/data/augment.py:415-420,590-609,636-666,920-926).  src u8 [B, SH, SW, 3], inv_affine f64 [B, 6]
    (destination -> source), luts u8 [B, 3, 256], flags i32 [B] (bit 0 up-down, bit 1 left-right) -> f32 [B, 3, H, W].
    Bit-identical to the host kernels of libtamtr_host.so followed by `img.float() / 255` on the device."""
    require_gpu(src, inv_affine, luts, flags)
    B, SH, SW, ch = src.shape
    H, W = out_hw
    if (ch != 3 or src.dtype != torch.uint8 or inv_affine.dtype != torch.float64 or luts.dtype != torch.uint8 or flags.dtype != torch.int32
            or tuple(inv_affine.shape) != (B, 6) or tuple(luts.shape) != (B, 3, 256) or tuple(flags.shape) != (B,)):
        raise _lib.TamtrHipError('img_augment: expected src u8 [B,SH,SW,3], inv_affine f64 [B,6], luts u8 [B,3,256], flags i32 [B]')
    src, inv_affine, luts, flags = _c(src), _c(inv_affine), _c(luts), _c(flags)
    out = torch.empty(B, 3, H, W, device=src.device, dtype=torch.float32)
    call('tamtr_img_augment_u8', ptr(src), ptr(inv_affine), ptr(luts), ptr(flags), ptr(out), B, SH, SW, H, W, int(border), stream_ptr())
    return out


class _CPAM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        require_gpu(x)
        x = _c(x)
        B, C, H, W = x.shape
        Hp, Wp = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        p = torch.empty(B, C, Hp, Wp, dtype=x.dtype, device=x.device)   # ChannelAttentionModule.Maxpool 3/2/1 (block.py:274): csrc/pool.hip
        idx = torch.empty(B, C, Hp, Wp, dtype=torch.uint8, device=x.device)
        call('tamtr_maxpool_fwd', ptr(x), ptr(p), ptr(idx), B, C, H, W, 3, 2, 1, 0, dtype_code(x), stream_ptr())
        out = torch.empty_like(x)
        s2 = torch.empty(B, 8, H, W, device=x.device, dtype=torch.float32)
        arg = torch.empty(B, 8, H, W, device=x.device, dtype=torch.int32)
        call('tamtr_cpam_fwd', ptr(x), ptr(p), ptr(out), ptr(s2), ptr(arg), B, C, H, W, dtype_code(x), stream_ptr())
        ctx.save_for_backward(x, p, idx, s2, arg)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, p, idx, s2, arg = ctx.saved_tensors
        B, C, H, W = x.shape
        gout = _c(gout.to(x.dtype))
        dxd, du, dp = torch.empty_like(x), torch.empty_like(x), torch.empty_like(p)
        call('tamtr_cpam_bwd', ptr(gout), ptr(x), ptr(p), ptr(s2), ptr(arg), ptr(dxd), ptr(du), ptr(dp), B, C, H, W, dtype_code(x),
             stream_ptr())
        dx = torch.empty_like(x)
        call('tamtr_maxpool_bwd', ptr(dp), ptr(idx), ptr(dxd), ptr(dx), B, C, H, W, 3, 2, 1, 0, dtype_code(x), stream_ptr())  # + the direct term
        return dx


class _CPAMCL(torch.autograd.Function):
    """CPAM on a channels-last map where it lies (csrc/cpam.hip tamtr_cpam_cl_*, csrc/pool.hip with nhwc = 1): no transposing copies."""

    @staticmethod
    def forward(ctx, x):
        require_gpu(x)
        B, C, H, W = x.shape
        Hp, Wp = H // 2, W // 2
        cl = torch.channels_last
        p = torch.empty((B, C, Hp, Wp), dtype=x.dtype, device=x.device, memory_format=cl)
        idx = torch.empty((B, C, Hp, Wp), dtype=torch.uint8, device=x.device, memory_format=cl)
        out = torch.empty_like(x, memory_format=cl)
        s2 = torch.empty(B, H, W, 8, device=x.device, dtype=torch.float32)
        arg = torch.empty(B, H, W, 8, device=x.device, dtype=torch.int32)
        call('tamtr_cpam_cl_fwd', ptr(x), ptr(p), ptr(idx), ptr(out), ptr(s2), ptr(arg), B, C, H, W, dtype_code(x), stream_ptr())   # pool + gates
        ctx.save_for_backward(x, p, idx, s2, arg)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, p, idx, s2, arg = ctx.saved_tensors
        B, C, H, W = x.shape
        gout = gout.to(x.dtype)
        if not _is_cl(gout):
            gout = gout.contiguous(memory_format=torch.channels_last)
        cl = torch.channels_last
        dxd, du, dp = torch.empty_like(x, memory_format=cl), torch.empty_like(x, memory_format=cl), torch.empty_like(p, memory_format=cl)
        dx = torch.empty_like(x, memory_format=cl)
        call('tamtr_cpam_cl_bwd', ptr(gout), ptr(x), ptr(p), ptr(idx), ptr(s2), ptr(arg), ptr(dxd), ptr(du), ptr(dp), ptr(dx), B, C, H, W, dtype_code(x),
             stream_ptr())
        return dx


def cpam_cl_ok(x):
    """What tamtr_cpam_cl_* take: a packed channels-last map, even H and W, a chunk (C / 8 channels) = a power-of-two number of 16-byte vectors."""
    if not (x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16) and _is_cl(x) and x.data_ptr() % 16 == 0):
        return False
    B, C, H, W = x.shape
    V = 8 if x.dtype == torch.bfloat16 else 4
    lpc = C // (8 * V)
    return (H % 2 == 0 and W % 2 == 0 and C % (8 * V) == 0 and lpc >= 1 and lpc & (lpc - 1) == 0 and lpc <= 64 and C // V <= 256 and 256 % (C // V) == 0
            and _os.environ.get('TAMTR_CPAM') != 'nchw')


def cpam(x):
    """CPAM (extra_modules/block.py:271-308): x [B,C,H,W] fp32/bf16 -> channel gate sigmoid(up2(maxpool3s2(x))) * x followed by
    the per-chunk (8 chunks) spatial gate sigmoid(max over the chunk's channels).  One fused kernel after the pool.
    Channels-last maps (the trunk's layout) take the channels-last kernels; the NCHW kernels remain for NCHW maps (deterministic mode)
    and for channel counts the lane mapping does not cover (there a channels-last map is repacked on the way in and out)."""
    if cpam_cl_ok(x):
        return _CPAMCL.apply(x)
    if _is_cl(x):
        return to_channels_last(_CPAM.apply(to_nchw(x)))
    return _CPAM.apply(x)


class _LayerNorm(torch.autograd.Function):
    """LayerNorm over the last axis in the activation's own dtype (VSSBlock.norm / norm2) - csrc/ss2d_out.hip."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        require_gpu(x, gamma, beta)
        x = _c(x)
        D = x.shape[-1]
        ntok = x.numel() // D
        g32, b32 = _c(gamma.float()), _c(beta.float())
        out = torch.empty_like(x)
        stats = torch.empty(ntok, 2, device=x.device, dtype=torch.float32)
        call('tamtr_layernorm_fwd', ptr(x), ptr(g32), ptr(b32), ptr(out), ptr(stats), ntok, D, float(eps), dtype_code(x), stream_ptr())
        ctx.save_for_backward(x, g32, stats)
        ctx.cfg = (gamma.dtype, beta.dtype)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, g32, stats = ctx.saved_tensors
        D = x.shape[-1]
        ntok = x.numel() // D
        gout = _c(gout.to(x.dtype))
        gx = torch.empty_like(x)
        part = torch.empty(_lib.lib().tamtr_ln_gate_blocks(ntok), 2, D, device=x.device, dtype=torch.float32)
        call('tamtr_layernorm_bwd', ptr(gout), ptr(x), ptr(g32), ptr(stats), ptr(gx), ptr(part), ntok, D, dtype_code(x), stream_ptr())
        gsum = slab_sum(part)
        return gx, gsum[0].to(ctx.cfg[0]), gsum[1].to(ctx.cfg[1]), None


def layer_norm(x, gamma, beta, eps=1e-5):
    return _LayerNorm.apply(x, gamma, beta, eps)


def layer_norm_module(norm, x):
    """norm(x) for an nn.LayerNorm over the last axis: on the GPU the wave-per-token kernel (one launch forward; one + an ordered partial sum
    backward, where torch launches three), in x's dtype - what torch's autocast gives for fp32 inputs; else the module."""
    D = x.shape[-1]
    if (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and len(norm.normalized_shape) == 1 and norm.elementwise_affine and norm.bias is not None
            and D in (32, 64, 128, 256, 512, 1024) and _os.environ.get('TAMTR_DECODER_LN') != 'torch'):
        return _LayerNorm.apply(x, norm.weight, norm.bias, norm.eps)
    return norm(x)


class _BNAct(torch.autograd.Function):
    """Training-mode BatchNorm2d (+ SiLU) on an NCHW map - csrc/bn.hip.  running_mean / running_var are updated in place."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, silu):
        require_gpu(x, gamma, beta)
        x = _c(x)
        B, C = x.shape[:2]
        HW = x.numel() // (B * C)
        g32, b32 = _c(gamma.float()), _c(beta.float())
        y = torch.empty_like(x)
        mr = torch.empty(C, 2, device=x.device, dtype=torch.float32)
        part = torch.empty(C * _lib.lib().tamtr_bn_slices(B, HW) * 3, device=x.device, dtype=torch.float32)
        call('tamtr_bn_act_fwd', ptr(x), ptr(g32), ptr(b32), ptr(running_mean), ptr(running_var), ptr(y), ptr(mr), ptr(part), B, C, HW,
             float(eps), float(momentum), int(bool(silu)), dtype_code(x), stream_ptr())
        ctx.save_for_backward(x, g32, b32, mr)
        ctx.cfg = (int(bool(silu)), gamma.dtype, beta.dtype)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, g32, b32, mr = ctx.saved_tensors
        act, g_dt, b_dt = ctx.cfg
        B, C = x.shape[:2]
        HW = x.numel() // (B * C)
        gy = _c(gy.to(x.dtype))
        gx = torch.empty_like(x)
        gg, gb = torch.empty(C, device=x.device, dtype=torch.float32), torch.empty(C, device=x.device, dtype=torch.float32)
        part = torch.empty(C * _lib.lib().tamtr_bn_slices(B, HW) * 2, device=x.device, dtype=torch.float32)
        call('tamtr_bn_act_bwd', ptr(gy), ptr(x), ptr(g32), ptr(b32), ptr(mr), ptr(gx), ptr(gg), ptr(gb), ptr(part), B, C, HW, act,
             dtype_code(x), stream_ptr())
        return gx, gg.to(g_dt), gb.to(b_dt), None, None, None, None, None


class _BNActCL(torch.autograd.Function):
    """Same on a channels-last [N, C] map (token-major)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, silu, residual=None):
        require_gpu(x, gamma, beta)
        x = _c(x)
        N, C = x.shape
        g32, b32 = _c(gamma.float()), _c(beta.float())
        y = torch.empty_like(x)
        mr = torch.empty(C, 2, device=x.device, dtype=torch.float32)
        part = torch.empty(C * _lib.lib().tamtr_bncl_blocks(N, C, dtype_code(x)) * 3, device=x.device, dtype=torch.float32)
        res = None if residual is None else _c(residual.to(x.dtype))
        call('tamtr_bncl_act_fwd', ptr(x), ptr(g32), ptr(b32), ptr(running_mean), ptr(running_var), ptr(res), ptr(y), ptr(mr), ptr(part), N, C,
             float(eps), float(momentum), int(bool(silu)), dtype_code(x), stream_ptr())
        ctx.save_for_backward(x, g32, b32, mr)
        ctx.cfg = (int(bool(silu)), gamma.dtype, beta.dtype, None if residual is None else residual.dtype)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, g32, b32, mr = ctx.saved_tensors
        act, g_dt, b_dt, res_dt = ctx.cfg
        N, C = x.shape
        g_res = None if res_dt is None else gy.to(res_dt)   # y = act(bn(x)) + residual: the shortcut's gradient is gy itself
        gy = _gy_rows(gy, x)   # (a channel slice of a concatenation's gradient is read in place through its row pitch)
        gx = torch.empty_like(x)
        gg, gb = torch.empty(C, device=x.device, dtype=torch.float32), torch.empty(C, device=x.device, dtype=torch.float32)
        part = torch.empty(C * _lib.lib().tamtr_bncl_blocks(N, C, dtype_code(x)) * 2 + 2 * C, device=x.device, dtype=torch.float32)
        call('tamtr_bncl_act_bwd', ptr(gy), gy.stride(0), ptr(x), ptr(g32), ptr(b32), ptr(mr), ptr(gx), ptr(gg), ptr(gb), ptr(part), N, C, act,
             dtype_code(x), stream_ptr())
        return gx, gg.to(g_dt), gb.to(b_dt), None, None, None, None, None, g_res


def _gy_rows(gy, x):
    """gy as the kernels take it: x's dtype, unit column stride, 16-byte aligned rows (a channel slice of a wider map is fine)."""
    C = x.shape[1]
    gy = gy.to(x.dtype)
    v = 8 if (x.dtype == torch.bfloat16 and C % 8 == 0) else 4
    if not (gy.stride(1) == 1 and gy.stride(0) >= C and gy.stride(0) % v == 0 and gy.data_ptr() % 16 == 0 and (C & (C - 1)) == 0):
        gy = gy.contiguous()
    return gy


class _BN2ActCL(torch.autograd.Function):
    """y = act(bn1(x1) + bn2(x2)) on channels-last [N, C] maps, both BatchNorms in training mode: RepConvN's training form
    (extra_modules/block.py:66-69) in one apply pass and one backward pair (csrc/bn.hip tamtr_bncl2_act_*)."""

    @staticmethod
    def forward(ctx, x1, ga1, be1, rm1, rv1, x2, ga2, be2, rm2, rv2, eps, momentum, silu):
        require_gpu(x1, x2, ga1, ga2)
        x1, x2 = _c(x1), _c(x2)
        N, C = x1.shape
        p1 = [_c(t.float()) for t in (ga1, be1, ga2, be2)]
        y = torch.empty_like(x1)
        mr = torch.empty(2, C, 2, device=x1.device, dtype=torch.float32)
        part = torch.empty(2 * C * _lib.lib().tamtr_bncl_blocks(N, C, dtype_code(x1)) * 3, device=x1.device, dtype=torch.float32)
        call('tamtr_bncl2_act_fwd', ptr(x1), ptr(p1[0]), ptr(p1[1]), ptr(rm1), ptr(rv1), ptr(x2), ptr(p1[2]), ptr(p1[3]), ptr(rm2), ptr(rv2),
             ptr(y), ptr(mr), ptr(part), N, C, float(eps), float(momentum), int(bool(silu)), dtype_code(x1), stream_ptr())
        ctx.save_for_backward(x1, x2, *p1, mr)
        ctx.cfg = (int(bool(silu)), ga1.dtype, be1.dtype, ga2.dtype, be2.dtype)
        return y

    @staticmethod
    def backward(ctx, gy):
        x1, x2, g1, b1, g2, b2, mr = ctx.saved_tensors
        act, dt_g1, dt_b1, dt_g2, dt_b2 = ctx.cfg
        N, C = x1.shape
        gy = _gy_rows(gy, x1)
        gx1, gx2 = torch.empty_like(x1), torch.empty_like(x2)
        gg = torch.empty(4, C, device=x1.device, dtype=torch.float32)
        part = torch.empty(C * _lib.lib().tamtr_bncl_blocks(N, C, dtype_code(x1)) * 3 + 3 * C, device=x1.device, dtype=torch.float32)
        call('tamtr_bncl2_act_bwd', ptr(gy), gy.stride(0), ptr(x1), ptr(x2), ptr(g1), ptr(b1), ptr(g2), ptr(b2), ptr(mr), ptr(gx1), ptr(gx2),
             gg[0].data_ptr(), gg[1].data_ptr(), gg[2].data_ptr(), gg[3].data_ptr(), ptr(part), N, C, act, dtype_code(x1), stream_ptr())
        return (gx1, gg[0].to(dt_g1), gg[1].to(dt_b1), None, None, gx2, gg[2].to(dt_g2), gg[3].to(dt_b2), None, None, None, None, None)


class _BNCatCL(torch.autograd.Function):
    """feats [B, sum L_i, C] = torch.cat([BatchNorm_i(y_i).view(B, L_i, C) for i], 1) with every level's BatchNorm (training mode, no
    activation) written straight into its segment of the result, and the backward reading its segment of d(feats) where it lies
    (csrc/bn.hip tamtr_bncl_act_seg_*): the MEH token memory (head.py:1087,1202-1219) without the 550 MB concatenation copy and the three
    slice copies of its gradient.  y_i [B * L_i, C]; running statistics are updated in place like ops.bn_act."""

    @staticmethod
    def forward(ctx, B, eps, moms, n, *args):
        ys, gammas, betas, rms, rvs = (args[k * n:(k + 1) * n] for k in range(5))
        require_gpu(*ys)
        ys = [_c(y) for y in ys]
        C, dt = ys[0].shape[1], ys[0].dtype
        Ls = [y.shape[0] // B for y in ys]
        Lt = sum(Ls)
        feats = torch.empty(B, Lt, C, device=ys[0].device, dtype=dt)
        saved, off = [], 0
        for y, ga, be, rm, rv, mom, L in zip(ys, gammas, betas, rms, rvs, moms, Ls):
            N = B * L
            g32, b32 = _c(ga.float()), _c(be.float())
            mr = torch.empty(C, 2, device=y.device, dtype=torch.float32)
            part = torch.empty(C * _lib.lib().tamtr_bncl_blocks(N, C, dtype_code(y)) * 3, device=y.device, dtype=torch.float32)
            call('tamtr_bncl_act_seg_fwd', ptr(y), ptr(g32), ptr(b32), ptr(rm), ptr(rv), feats.data_ptr() + off * C * feats.element_size(), L, Lt * C,
                 ptr(mr), ptr(part), N, C, float(eps), float(mom), 0, dtype_code(y), stream_ptr())
            saved += [y, g32, b32, mr]
            off += L
        ctx.save_for_backward(*saved)
        ctx.cfg = (B, n, Ls, [g.dtype for g in gammas], [b.dtype for b in betas])
        return feats

    @staticmethod
    def backward(ctx, gf):
        B, n, Ls, gdt, bdt = ctx.cfg
        saved = ctx.saved_tensors
        Lt = sum(Ls)
        C = saved[0].shape[1]
        gf = _c(gf.to(saved[0].dtype))
        gxs, ggs, gbs, off = [], [], [], 0
        for i, L in enumerate(Ls):
            y, g32, b32, mr = saved[4 * i:4 * i + 4]
            N = B * L
            gx = torch.empty_like(y)
            gg, gb = torch.empty(C, device=y.device, dtype=torch.float32), torch.empty(C, device=y.device, dtype=torch.float32)
            part = torch.empty(C * _lib.lib().tamtr_bncl_blocks(N, C, dtype_code(y)) * 2 + 2 * C, device=y.device, dtype=torch.float32)
            call('tamtr_bncl_act_seg_bwd', gf.data_ptr() + off * C * gf.element_size(), L, Lt * C, ptr(y), ptr(g32), ptr(b32), ptr(mr), ptr(gx), ptr(gg),
                 ptr(gb), ptr(part), N, C, 0, dtype_code(y), stream_ptr())
            gxs.append(gx); ggs.append(gg.to(gdt[i])); gbs.append(gb.to(bdt[i]))
            off += L
        return (None, None, None, None, *gxs, *ggs, *gbs, *([None] * (2 * n)))


def bn_cat_cl(ys, bns, B):
    """The token memory: per level y_i [B * L_i, C] -> BatchNorm_i (training mode, batch statistics, no activation) -> [B, sum L_i, C],
    each level written into its segment directly.  All BatchNorms share eps; C a power of two."""
    moms = [_bn_tick(bn) for bn in bns]
    n = len(ys)
    return _BNCatCL.apply(int(B), float(bns[0].eps), tuple(float(m) for m in moms), n, *ys, *[bn.weight for bn in bns], *[bn.bias for bn in bns],
                          *[bn.running_mean for bn in bns], *[bn.running_var for bn in bns])


def bn_cat_cl_ok(ys, bns):
    C = ys[0].shape[1]
    return (all(y.is_cuda and y.dim() == 2 and y.shape[1] == C and y.dtype == ys[0].dtype and y.dtype in (torch.float32, torch.bfloat16) for y in ys)
            and C & (C - 1) == 0 and bn_cl_ok(C, ys[0].dtype) and all(bn.training and bn.track_running_stats and bn.affine and bn.eps == bns[0].eps for bn in bns)
            and _os.environ.get('TAMTR_BN_CAT') != 'torch')


_BN_COUNTERS = None


def begin_bn_counter_batch():
    """Defer the `num_batches_tracked += 1` of every bn_act call (170 one-element kernels per step) to end_bn_counter_batch()."""
    global _BN_COUNTERS
    _BN_COUNTERS = []
    return _BN_COUNTERS


def end_bn_counter_batch():
    global _BN_COUNTERS
    pending, _BN_COUNTERS = _BN_COUNTERS, None
    if pending:
        torch._foreach_add_(pending, 1)


def bn_cl_ok(C, dtype):
    """Channel counts the channels-last BatchNorm kernels take (include/tamtr_hip.h: tamtr_bncl_act_fwd)."""
    v = 8 if dtype == torch.bfloat16 and C % 8 == 0 else 4
    return C % 4 == 0 and C <= 1024 and C // v <= 256 and 256 % (C // v) == 0


def _bn_tick(bn):
    """num_batches_tracked += 1 (deferred to one multi-tensor kernel inside a counter batch) and the momentum of this call."""
    if bn.track_running_stats:
        if _BN_COUNTERS is not None and bn.momentum is not None:
            _BN_COUNTERS.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked += 1
    return bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)


def bn_act(x, bn, silu, residual=None):
    """act(bn(x)) [+ residual] for an nn.BatchNorm2d in training mode (batch statistics; running stats and num_batches_tracked
    updated).  residual: only with the channels-last [N, C] form."""
    mom = _bn_tick(bn)
    rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
    if x.dim() == 2:   # [N, C] token-major / channels-last map
        return _BNActCL.apply(x, bn.weight, bn.bias, rm, rv, bn.eps, mom, silu, residual)
    y = _BNAct.apply(x, bn.weight, bn.bias, rm, rv, bn.eps, mom, silu)
    return y if residual is None else y + residual


def bn2_act(x1, bn1, x2, bn2, silu):
    """act(bn1(x1) + bn2(x2)) on channels-last [N, C] maps, both BatchNorm2d in training mode with running statistics."""
    if bn1.eps != bn2.eps or bn1.momentum != bn2.momentum or not (bn1.track_running_stats and bn2.track_running_stats):
        raise _lib.TamtrHipError('bn2_act: the two BatchNorms must share eps / momentum and track running statistics')
    mom = _bn_tick(bn1)
    _bn_tick(bn2)
    return _BN2ActCL.apply(x1, bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var, x2, bn2.weight, bn2.bias, bn2.running_mean,
                           bn2.running_var, bn1.eps, mom, silu)
