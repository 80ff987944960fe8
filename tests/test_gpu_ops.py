"""GPU parity: HIP kernels (through the C ABI) vs the CPU oracle and the reference-generated golden vectors.

fp32 kernels are held to 1e-3 relative or better (north_star tolerance; most are ~1e-6); the bf16 variants are held to
bf16 rounding (documented per test).  Run with `pytest -m gpu` on an MI355X.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import T, assert_close, check_param_grads, check_summary
from oracle import tamtr_oracle as O
from weights import rnd, urnd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import tamtr_amd.ops as ops
    return ops


def dev(t, dtype=None):
    t = t.detach().cuda()
    return t.to(dtype) if dtype is not None else t


def _gate_ref(x, gk, bias, v, nh):
    """oracle gate with the pieces the kernel takes as inputs (gk = gl(guide), v = proj_conv(x))."""
    B, C, H, W = x.shape
    hc = C // nh
    e = x.view(B, nh, hc, H * W)
    aw = torch.einsum('bmcp,bnmc->bmpn', e, gk.view(B, -1, nh, hc)).max(-1).values / hc ** 0.5 + bias[None, :, None]
    return (v.view(B, nh, hc, H * W) * torch.sigmoid(aw).unsqueeze(2)).view(B, C, H, W)


@pytest.mark.parametrize('B,C,nh,H,W,Tn', [(2, 64, 2, 12, 12, 10), (2, 256, 8, 9, 7, 7), (3, 128, 4, 5, 20, 1),
                                           (1, 64, 1, 8, 8, 37), (2, 96, 2, 6, 6, 80)])
def test_gate_kernel_fp32(ops, B, C, nh, H, W, Tn):
    x, gk, v = rnd((B, C, H, W), 1), rnd((B, Tn, C), 2, 0.3), rnd((B, C, H, W), 3)
    bias, cot = rnd((nh,), 4, 0.2), rnd((B, C, H, W), 5)
    xr, gr, vr, br = (t.clone().requires_grad_() for t in (x, gk, v, bias))
    ref = _gate_ref(xr, gr, br, vr, nh)
    (ref * cot).sum().backward()
    xd, gd, vd, bd = (dev(t).requires_grad_() for t in (x, gk, v, bias))
    out = ops.maxsigmoid_gate(xd, gd, bd, vd, nh)
    (out * dev(cot)).sum().backward()
    assert_close(out, ref, 1e-4, 1e-5, 'gate out')
    for n, a, b in (('dx', xd, xr), ('dgk', gd, gr), ('dv', vd, vr), ('dbias', bd, br)):
        assert_close(a.grad, b.grad, 1e-3, 1e-4, n)


def test_gate_kernel_bf16(ops):
    """bf16 I/O, fp32 accumulate: output within bf16 rounding (2^-8 relative) of the fp32 oracle on bf16-rounded inputs."""
    B, C, nh, H, W, Tn = 2, 128, 4, 16, 16, 10
    x, gk, v = rnd((B, C, H, W), 1).bfloat16(), rnd((B, Tn, C), 2, 0.3), rnd((B, C, H, W), 3).bfloat16()
    bias = rnd((nh,), 4, 0.2)
    ref = _gate_ref(x.float(), gk, bias, v.float(), nh)
    out = ops.maxsigmoid_gate(dev(x), dev(gk), dev(bias), dev(v), nh)
    assert out.dtype == torch.bfloat16
    assert_close(out.float(), ref, 1e-2, 1e-2, 'gate bf16')


@pytest.mark.parametrize('tag', ['A', 'B', 'C'])
def test_msdeform_core_golden(ops, golden, tag):
    """Against the reference's own output (F.grid_sample path), fixtures msdeform_core.npz."""
    fx = golden('msdeform_core')
    v, loc, aw = (dev(T(fx[f'{tag}.{k}'])).requires_grad_() for k in ('value', 'loc', 'aw'))
    out = ops.ms_deform_attn_core(v, fx[f'{tag}.shapes'].tolist(), loc, aw)
    assert_close(out, fx[f'{tag}.out'], 1e-4, 1e-5, 'core out')
    (out * dev(T(fx[f'{tag}.cot']))).sum().backward()
    assert_close(v.grad, fx[f'{tag}.g_value'], 1e-3, 1e-5, 'g_value')
    assert_close(loc.grad, fx[f'{tag}.g_loc'], 1e-3, 5e-4, 'g_loc')
    assert_close(aw.grad, fx[f'{tag}.g_aw'], 1e-3, 1e-5, 'g_aw')


def test_msdeform_core_edges(ops):
    """all samples outside the maps -> exactly zero output and gradients; integer-aligned samples; D=64 bf16."""
    B, Q, M, D = 1, 5, 8, 64
    shapes = [(4, 6), (2, 3)]
    L = sum(h * w for h, w in shapes)
    v = rnd((B, L, M, D), 1)
    loc = torch.full((B, Q, M, 2, 4, 2), 3.0)
    aw = torch.full((B, Q, M, 2, 4), 1 / 8)
    vd = dev(v).requires_grad_()
    out = ops.ms_deform_attn_core(vd, shapes, dev(loc), dev(aw))
    assert float(out.abs().max()) == 0.0
    out.sum().backward()
    assert float(vd.grad.abs().max()) == 0.0
    loc = urnd((B, Q, M, 2, 4, 2), 2, -0.1, 1.1)
    ref = O.ms_deform_attn_core(v.bfloat16().float(), shapes, loc, aw)
    out = ops.ms_deform_attn_core(dev(v).bfloat16(), shapes, dev(loc), dev(aw))
    assert out.dtype == torch.bfloat16
    assert_close(out.float(), ref, 1e-2, 1e-2, 'bf16 core')


def test_msdeform_core_full_size_properties(ops):
    """BASELINE size (B=16, L=33600, Q=292, 8 heads x 64): size-independent properties instead of an oracle run:
    (1) linearity in value; (2) constant value field + weights summing to 1 inside the map -> the constant."""
    B, Q, M, D = 16, 292, 8, 64
    shapes = [(160, 160), (80, 80), (40, 40)]
    L = sum(h * w for h, w in shapes)
    g = torch.Generator(device='cuda').manual_seed(0)
    v1 = torch.randn(B, L, M, D, device='cuda', generator=g)
    v2 = torch.randn(B, L, M, D, device='cuda', generator=g)
    loc = 0.1 + 0.8 * torch.rand(B, Q, M, 3, 4, 2, device='cuda', generator=g)
    aw = torch.softmax(torch.randn(B, Q, M, 12, device='cuda', generator=g), -1).view(B, Q, M, 3, 4)
    o1, o2 = ops.ms_deform_attn_core(v1, shapes, loc, aw), ops.ms_deform_attn_core(v2, shapes, loc, aw)
    o12 = ops.ms_deform_attn_core(v1 + 2 * v2, shapes, loc, aw)
    assert_close(o12, o1 + 2 * o2, 1e-4, 1e-4, 'linearity')
    const = torch.full_like(v1, 0.75)
    assert_close(ops.ms_deform_attn_core(const, shapes, loc, aw), torch.full_like(o1, 0.75), 1e-5, 1e-5, 'partition of unity')


def test_contrastive_golden(ops, golden):
    fx = golden('contrastive')
    x, w = dev(T(fx['x'])).requires_grad_(), dev(T(fx['w'])).requires_grad_()
    ls, bi = dev(T(fx['logit_scale'])).requires_grad_(), dev(T(fx['bias'])).requires_grad_()
    out = ops.contrastive_logits(x, w, ls, bi)
    assert_close(out, fx['out'], 1e-4, 1e-4)
    (out * dev(T(fx['cot']))).sum().backward()
    check_summary(fx, 'gin.x', x.grad, 1e-3, 1e-5)
    check_summary(fx, 'gin.w', w.grad, 1e-3, 1e-5)
    check_param_grads(fx, '', {'bias': bi.grad, 'logit_scale': ls.grad}, 1e-3, 1e-3)


@pytest.mark.parametrize('B,Q,K,C', [(1, 1, 1, 64), (3, 17, 80, 128), (2, 300, 10, 512), (1, 5, 3, 2048), (2, 100, 16, 256), (2, 181, 17, 256)])
def test_contrastive_shapes(ops, B, Q, K, C):
    x, w = rnd((B, Q, C), 1, 2.0), rnd((B, K, C), 2)
    P = O.View({'logit_scale': torch.tensor(2.3), 'bias': torch.tensor([-9.5])})
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    ref = O.contrastive_head(xr, wr, P)
    cot = rnd(ref.shape, 3)
    (ref * cot).sum().backward()
    xd, wd = dev(x).requires_grad_(), dev(w).requires_grad_()
    out = ops.contrastive_logits(xd, wd, dev(P['logit_scale']), dev(P['bias']))
    (out * dev(cot)).sum().backward()
    assert_close(out, ref, 1e-4, 1e-4)
    assert_close(xd.grad, xr.grad, 1e-3, 1e-5)
    assert_close(wd.grad, wr.grad, 1e-3, 1e-5)
    xz = torch.zeros(1, 2, C)
    outz = ops.contrastive_logits(dev(xz), dev(w[:1]), dev(P['logit_scale']), dev(P['bias']))
    assert_close(outz, O.contrastive_head(xz, w[:1], P), 1e-6, 1e-6, 'zero row (eps clamp)')


def test_cpu_tensor_is_refused():
    import tamtr_amd.ops as ops
    from tamtr_amd import TamtrHipError
    with pytest.raises(TamtrHipError):
        ops.contrastive_logits(torch.zeros(1, 1, 64), torch.zeros(1, 1, 64), torch.zeros(()), torch.zeros(1))


def _scipy_pairs(cost, groups):
    """What HungarianMatcher does in the reference (models/utils/ops.py:98-119): scipy per image on its own column block."""
    from scipy.optimize import linear_sum_assignment
    bi, si, gi, off = [], [], [], 0
    for b, n in enumerate(groups):
        r, c = linear_sum_assignment(cost[b, :, off:off + n].numpy())
        bi += [b] * len(r); si += r.tolist(); gi += (c + off).tolist()
        off += n
    return bi, si, gi


@pytest.mark.parametrize('nq,groups,kind', [
    (100, [8] * 16, 'float'),                       # the bench configuration
    (100, [0, 3, 17, 1, 0, 42], 'float'),           # ragged, empty images
    (100, [150, 100, 99, 101], 'float'),            # more boxes than queries: scipy does not transpose
    (300, [500, 7], 'float'),                       # reference default nq with a crowded VisDrone-like image (unstaged costs)
    (100, [8] * 4, 'ties'),                         # small-integer costs: many exact ties, tie rule must equal scipy's
    (37, [5, 40, 37], 'const'),                     # constant matrix (non-finite costs are zeroed by the matcher)
    (64, [64, 1], 'ties'),
])
def test_lsap_assign_equals_scipy(ops, nq, groups, kind):
    g = torch.Generator().manual_seed(nq + len(groups))
    G = sum(groups)
    if kind == 'float':
        cost = torch.randn(len(groups), nq, G, generator=g) * 3
    elif kind == 'ties':
        cost = torch.randint(0, 4, (len(groups), nq, G), generator=g).float()
    else:
        cost = torch.zeros(len(groups), nq, G)
    bi, si, gi = ops.lsap_assign(cost.cuda(), groups)
    rb, rs, rg = _scipy_pairs(cost, groups)
    assert bi.tolist() == rb
    if kind == 'float':
        assert si.tolist() == rs and gi.tolist() == rg
    else:  # same optimum, and (the solver restates scipy's scan order and tie rule) the very same pairs
        tot = cost[bi.cpu(), si.cpu(), gi.cpu()].double().sum()
        assert float(tot) == float(cost[rb, rs, rg].double().sum())
        assert si.tolist() == rs and gi.tolist() == rg


def test_lsap_assign_argument_checks(ops):
    import tamtr_amd
    with pytest.raises(tamtr_amd.TamtrHipError):
        ops.lsap_assign(torch.zeros(2, 10, 5).cuda(), [2, 2])          # sizes do not tile the columns
    with pytest.raises(tamtr_amd.TamtrHipError):
        ops.lsap_assign(torch.zeros(2, 10, 5), [2, 3])                 # CPU tensor: no fallback
    bi, si, gi = ops.lsap_assign(torch.zeros(3, 10, 0).cuda(), [0, 0, 0])
    assert bi.numel() == si.numel() == gi.numel() == 0


@pytest.mark.parametrize('B,C,H,W', [(2, 64, 12, 16), (1, 128, 6, 6), (2, 16, 2, 2), (1, 256, 10, 4), (1, 8, 14, 30)])
def test_cpam_vs_oracle(ops, B, C, H, W):
    """Fused CPAM gates vs the oracle's restatement of extra_modules/block.py:271-308 (forward and dx), fp32."""
    x = rnd((B, C, H, W), 5) * 2
    cot = rnd((B, C, H, W), 6)
    xr = x.clone().requires_grad_()
    ref = O.cpam(xr)
    (ref * cot).sum().backward()
    xd = dev(x).requires_grad_()
    out = ops.cpam(xd)
    (out * dev(cot)).sum().backward()
    assert_close(out, ref, 1e-5, 1e-6, 'cpam out')
    assert_close(xd.grad, xr.grad, 1e-4, 1e-5, 'cpam dx')


def test_cpam_bf16_and_argument_checks(ops):
    import tamtr_amd
    x = rnd((2, 128, 16, 16), 7)
    cot = rnd((2, 128, 16, 16), 8)
    xr = x.bfloat16().float().requires_grad_()
    ref = O.cpam(xr)
    (ref * cot.bfloat16().float()).sum().backward()
    xd = dev(x, torch.bfloat16).requires_grad_()
    out = ops.cpam(xd)
    (out * dev(cot, torch.bfloat16)).sum().backward()
    assert out.dtype == torch.bfloat16
    # bf16 storage of out / p / du (8 mantissa bits), fp32 arithmetic inside
    assert_close(out.float(), ref, 1e-2, 1e-2, 'cpam bf16 out')
    assert_close(xd.grad.float(), xr.grad, 3e-2, 3e-2, 'cpam bf16 dx')
    with pytest.raises(tamtr_amd.TamtrHipError):
        ops.cpam(torch.zeros(1, 16, 5, 4).cuda())      # odd height: the reference fails on it too
    with pytest.raises(tamtr_amd.TamtrHipError):
        ops.cpam(torch.zeros(1, 12, 4, 4).cuda())      # channels not divisible into 8 chunks
    with pytest.raises(tamtr_amd.TamtrHipError):
        ops.cpam(torch.zeros(1, 16, 4, 4))             # CPU tensor


@pytest.mark.parametrize('B,C,H,W,dt', [(2, 512, 10, 12, torch.bfloat16), (2, 256, 20, 16, torch.bfloat16), (1, 128, 36, 40, torch.bfloat16),
                                          (2, 64, 12, 16, torch.bfloat16), (2, 128, 10, 6, torch.float32), (1, 32, 8, 8, torch.float32), (1, 512, 4, 2, torch.float32)])
def test_cpam_channels_last_kernels(ops, B, C, H, W, dt):
    """CPAM on channels-last maps (tamtr_cpam_cl_* + the NHWC max-pool; extra_modules/block.py:271-308): against the oracle's restatement
    (fp32 reference of the same rounded input) and against the NCHW kernels on the repacked map - same formulas per element, so the two
    kernel families agree to rounding of the chunk sums - at the trunk's channel counts (chunks of 2, 4, 8 and 16 lanes); the result is
    channels-last, no transposing copy is launched."""
    x = (rnd((B, C, H, W), 5) * 2).to(dt)
    cot = rnd((B, C, H, W), 6).to(dt)
    xr = x.float().clone().requires_grad_()
    ref = O.cpam(xr)
    (ref * cot.float()).sum().backward()
    xc = x.cuda().contiguous(memory_format=torch.channels_last)
    assert ops.cpam_cl_ok(xc)
    xd = xc.clone(memory_format=torch.preserve_format).requires_grad_()
    out = ops.cpam(xd)
    assert out.dtype == dt and ops.is_cl(out)
    (out.float() * cot.cuda().float()).sum().backward()
    assert ops.is_cl(xd.grad)
    t_o, t_g = (1e-5, 1e-4) if dt == torch.float32 else (1e-2, 3e-2)
    assert_close(out.float(), ref, t_o, t_o if dt == torch.bfloat16 else 1e-6, 'cpam (channels-last) out')
    assert_close(xd.grad.float(), xr.grad, t_g, t_g if dt == torch.bfloat16 else 1e-5, 'cpam (channels-last) dx')
    # the NCHW kernels on the same values
    xn = x.cuda().contiguous().requires_grad_()
    on = ops._CPAM.apply(xn)
    (on.float() * cot.cuda().float()).sum().backward()
    if dt == torch.float32:
        assert_close(out.contiguous(), on, 1e-6, 1e-6, 'out, channels-last vs NCHW kernels')   # same formulas; the compiler contracts the taps' products differently
        assert_close(xd.grad.contiguous(), xn.grad, 1e-5, 1e-6, 'dx, channels-last vs NCHW kernels')
    else:
        assert_close(out.float(), on.float(), 1e-2, 1e-2, 'out, channels-last vs NCHW kernels')
        assert_close(xd.grad.float(), xn.grad.float(), 3e-2, 3e-2, 'dx, channels-last vs NCHW kernels')
    # channel counts outside the lane mapping fall back to the NCHW kernels behind a repack
    odd = torch.zeros(1, 24, 4, 4, device='cuda').contiguous(memory_format=torch.channels_last)
    assert not ops.cpam_cl_ok(odd) and ops.cpam(odd).shape == odd.shape


def test_cpam_full_size_properties(ops):
    """BASELINE-size site (bs 16, 128 x 160 x 160, bf16): output bounded by |x| (two sigmoid gates), equals the fp32 kernel."""
    g = torch.Generator(device='cuda').manual_seed(0)
    x = torch.randn(16, 128, 160, 160, device='cuda', generator=g)
    o32 = ops.cpam(x)
    o16 = ops.cpam(x.bfloat16())
    assert bool((o32.abs() <= x.abs() + 1e-6).all())
    assert float((o16.float() - o32).abs().max()) < 0.05


@pytest.mark.parametrize('B,D,H,W,dt', [(2, 32, 16, 16, torch.float32), (1, 64, 13, 21, torch.float32), (2, 32, 40, 40, torch.float32),
                                        (1, 96, 8, 5, torch.bfloat16)])
def test_dwconv_silu_cross(ops, B, D, H, W, dt):
    """SS2D front end (vmamba.py:949-952 + CrossScan csms6s.py:4-14): fused kernel vs conv2d + SiLU + the two flattenings in
    torch on the CPU, forward and all three gradients; xi is the first half of the channels-last in_proj output."""
    import torch.nn.functional as F
    xz = rnd((B, H, W, 2 * D), 1).to(dt).float()
    w, bias = rnd((D, 1, 3, 3), 2, 0.4), rnd((D,), 3, 0.2)
    cot = rnd((B, 2, D, H * W), 4)
    xr, wr, br = xz.clone().requires_grad_(), w.clone().requires_grad_(), bias.clone().requires_grad_()
    a = F.silu(F.conv2d(xr[..., :D].permute(0, 3, 1, 2), wr, br, padding=1, groups=D))
    ref = torch.stack([a.flatten(2), a.transpose(2, 3).flatten(2)], 1)
    (ref * cot).sum().backward()
    xd, wd, bd = dev(xz, dt).requires_grad_(), dev(w).requires_grad_(), dev(bias).requires_grad_()
    out = ops.dwconv_silu_cross(xd, wd, bd, D)
    (out * dev(cot)).sum().backward()
    tol = 1e-5 if dt == torch.float32 else 1e-2
    assert_close(out, ref, tol, tol, 'dwconv out')
    assert_close(xd.grad.float(), xr.grad, 10 * tol, 10 * tol, 'dwconv dx')   # second half of the channels stays zero
    assert_close(wd.grad, wr.grad, 10 * tol, 10 * tol * (B * H * W) ** 0.5, 'dwconv dw')
    assert_close(bd.grad, br.grad, 10 * tol, 10 * tol * (B * H * W) ** 0.5, 'dwconv db')


def test_dwconv_argument_checks(ops):
    import tamtr_amd
    with pytest.raises(tamtr_amd.TamtrHipError):
        ops.dwconv_silu_cross(torch.zeros(1, 4, 4, 48).cuda(), torch.zeros(24, 1, 3, 3).cuda(), None, 24)   # D % 32 != 0
    with pytest.raises(tamtr_amd.TamtrHipError):
        ops.dwconv_silu_cross(torch.zeros(1, 4, 4, 64), torch.zeros(32, 1, 3, 3), None, 32)                # CPU tensors


@pytest.mark.parametrize('D,dt', [(64, torch.float32), (256, torch.float32), (1024, torch.float32), (128, torch.bfloat16), (512, torch.bfloat16)])
def test_ln_gate(ops, D, dt):
    """out_norm + SiLU(z) gate (vmamba.py:1005-1008,1029-1036) in one kernel vs LayerNorm * silu in torch on the CPU: forward and the
    gradients of x, z (inside xz), gamma, beta."""
    import torch.nn.functional as F
    B, H, W = 2, 5, 7
    x = rnd((B, H * W, D), 1) * 1.5 + 0.3
    xz = rnd((B, H, W, 2 * D), 2).to(dt).float()
    gamma, beta = 1 + 0.2 * rnd((D,), 3), 0.1 * rnd((D,), 4)
    cot = rnd((B, H * W, D), 5).to(dt).float()
    xr, zr, gr, br = x.clone().requires_grad_(), xz.clone().requires_grad_(), gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    ref = F.layer_norm(xr, (D,), gr, br, 1e-5) * F.silu(zr[..., D:]).reshape(B, H * W, D)
    (ref * cot).sum().backward()
    xd, zd, gd, bd = dev(x).requires_grad_(), dev(xz, dt).requires_grad_(), dev(gamma).requires_grad_(), dev(beta).requires_grad_()
    out = ops.ln_gate(xd, zd, gd, bd, 1e-5)
    assert out.dtype == dt and out.shape == (B, H * W, D)
    (out.float() * dev(cot)).sum().backward()
    tol = 2e-5 if dt == torch.float32 else 2e-2
    assert_close(out.float(), ref, tol, tol, 'ln_gate out')
    assert_close(xd.grad, xr.grad, 5 * tol, 5 * tol, 'ln_gate dx')
    assert_close(zd.grad.float(), zr.grad, 5 * tol, 5 * tol, 'ln_gate dxz')   # first half (xi) stays zero
    assert_close(gd.grad, gr.grad, 5 * tol, 5 * tol * (B * H * W) ** 0.5, 'ln_gate dgamma')
    assert_close(bd.grad, br.grad, 5 * tol, 5 * tol * (B * H * W) ** 0.5, 'ln_gate dbeta')


@pytest.mark.parametrize('B,D,H,W', [(2, 32, 16, 16), (1, 64, 13, 21), (1, 32, 40, 8)])
def test_cross_merge_kernels(ops, B, D, H, W):
    """CrossMerge (csms6s.py:26-34) into token-major layout and its transpose: pure data movement + adds, exact."""
    from tamtr_amd._lib import call, ptr, stream_ptr
    L = H * W
    y4 = dev(rnd((B, 4, D, L), 1))
    ymT = torch.empty(B, L, D, device='cuda')
    call('tamtr_cross_merge_fwd', ptr(y4), ptr(ymT), B, D, H, W, 0, stream_ptr())
    ref = y4[:, 0] + y4[:, 2] + (y4[:, 1] + y4[:, 3]).view(B, D, W, H).transpose(2, 3).reshape(B, D, L)
    assert_close(ymT, ref.transpose(1, 2), 1e-6, 1e-6, 'cross merge fwd')
    g = dev(rnd((B, L, D), 2))
    g2 = torch.empty(B, 2, D, L, device='cuda')
    call('tamtr_cross_merge_bwd', ptr(g), ptr(g2), B, D, H, W, 0, stream_ptr())
    gm = g.transpose(1, 2)
    assert torch.equal(g2[:, 0], gm.contiguous())
    assert torch.equal(g2[:, 1], gm.reshape(B, D, H, W).transpose(2, 3).reshape(B, D, L))


@pytest.mark.parametrize('D,dt', [(32, torch.float32), (128, torch.float32), (512, torch.float32), (256, torch.bfloat16), (128, torch.bfloat16),
                                  (64, torch.bfloat16), (1024, torch.bfloat16)])
def test_layer_norm_kernel(ops, D, dt):
    """VSSBlock.norm / norm2 (vmamba.py:1190,1222): wave-per-token LayerNorm in the activation dtype vs F.layer_norm on the CPU (bf16 rows of
    64 / 128 elements take the several-tokens-per-wave kernels: 211 tokens leave the last wave's token slots partly empty)."""
    import torch.nn.functional as F
    n = 211
    x = (rnd((n, D), 1) * 2 + 0.5).to(dt).float()
    gamma, beta = 1 + 0.2 * rnd((D,), 2), 0.1 * rnd((D,), 3)
    cot = rnd((n, D), 4).to(dt).float()
    xr, gr, br = x.clone().requires_grad_(), gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    ref = F.layer_norm(xr, (D,), gr, br, 1e-5)
    (ref * cot).sum().backward()
    xd, gd, bd = dev(x, dt).requires_grad_(), dev(gamma).requires_grad_(), dev(beta).requires_grad_()
    out = ops.layer_norm(xd, gd, bd, 1e-5)
    assert out.dtype == dt
    (out.float() * dev(cot)).sum().backward()
    tol = 2e-5 if dt == torch.float32 else 2e-2
    assert_close(out.float(), ref, tol, tol, 'ln out')
    assert_close(xd.grad.float(), xr.grad, 5 * tol, 5 * tol, 'ln dx')
    assert_close(gd.grad, gr.grad, 5 * tol, 5 * tol * n ** 0.5, 'ln dgamma')
    assert_close(bd.grad, br.grad, 5 * tol, 5 * tol * n ** 0.5, 'ln dbeta')


@pytest.mark.parametrize('B,C,H,W,silu,dt', [(4, 8, 20, 20, True, torch.float32), (2, 3, 7, 9, False, torch.float32), (2, 16, 96, 96, True, torch.float32),
                                             (3, 5, 3, 3, True, torch.float32), (2, 32, 40, 40, True, torch.bfloat16)])
def test_bn_act_vs_torch(ops, B, C, H, W, silu, dt):
    """Training-mode BatchNorm2d (+SiLU) kernels vs nn.BatchNorm2d + F.silu on the CPU (conv.py:36-40 with the reference's eps 1e-3 /
    momentum 0.03): output, running statistics, and the gradients of x, gamma, beta."""
    import torch.nn as nn
    import torch.nn.functional as F
    x = (rnd((B, C, H, W), 1) * 1.7 + 0.4).to(dt).float()
    cot = rnd((B, C, H, W), 2).to(dt).float()
    ref_bn = nn.BatchNorm2d(C, eps=1e-3, momentum=0.03)
    with torch.no_grad():
        ref_bn.weight.copy_(1 + 0.3 * rnd((C,), 3)); ref_bn.bias.copy_(0.2 * rnd((C,), 4))
        ref_bn.running_mean.copy_(0.1 * rnd((C,), 5)); ref_bn.running_var.copy_(1 + 0.1 * rnd((C,), 6).abs())
    import copy
    dev_bn = copy.deepcopy(ref_bn).cuda()
    xr = x.clone().requires_grad_()
    z = ref_bn(xr)
    ref = F.silu(z) if silu else z
    (ref * cot).sum().backward()
    xd = dev(x, dt).requires_grad_()
    out = ops.bn_act(xd, dev_bn, silu)
    (out.float() * dev(cot)).sum().backward()
    tol = 2e-5 if dt == torch.float32 else 2e-2
    assert out.dtype == dt
    assert_close(out.float(), ref, tol, tol, 'bn out')
    assert_close(dev_bn.running_mean, ref_bn.running_mean, 1e-5, 1e-6, 'running_mean')
    assert_close(dev_bn.running_var, ref_bn.running_var, 1e-5, 1e-6, 'running_var')
    assert int(dev_bn.num_batches_tracked) == int(ref_bn.num_batches_tracked) == 1
    n = B * H * W
    assert_close(xd.grad.float(), xr.grad, 10 * tol, 10 * tol, 'bn dx')
    assert_close(dev_bn.weight.grad, ref_bn.weight.grad, 10 * tol, 10 * tol * n ** 0.5, 'bn dgamma')
    assert_close(dev_bn.bias.grad, ref_bn.bias.grad, 10 * tol, 10 * tol * n ** 0.5, 'bn dbeta')


@pytest.mark.parametrize('N,C,dt', [(1000, 512, torch.float32), (333, 64, torch.float32), (4096, 256, torch.bfloat16), (130, 1024, torch.float32)])
def test_bn_channels_last_vs_torch(ops, N, C, dt):
    """Token-major BatchNorm (the MEH input projection's BatchNorm over [B*L, hd], head.py:1087) vs nn.BatchNorm1d on the CPU."""
    import copy
    import torch.nn as nn
    x = (rnd((N, C), 1) * 1.3 - 0.7).to(dt).float()
    cot = rnd((N, C), 2).to(dt).float()
    ref_bn = nn.BatchNorm1d(C, eps=1e-3, momentum=0.03)
    with torch.no_grad():
        ref_bn.weight.copy_(1 + 0.3 * rnd((C,), 3)); ref_bn.bias.copy_(0.2 * rnd((C,), 4))
    dev_bn = copy.deepcopy(ref_bn).cuda()
    xr = x.clone().requires_grad_()
    ref = ref_bn(xr)
    (ref * cot).sum().backward()
    xd = dev(x, dt).requires_grad_()
    out = ops.bn_act(xd, dev_bn, False)
    (out.float() * dev(cot)).sum().backward()
    tol = 2e-5 if dt == torch.float32 else 2e-2
    assert_close(out.float(), ref, tol, tol, 'bncl out')
    assert_close(dev_bn.running_mean, ref_bn.running_mean, 1e-5, 1e-6, 'running_mean')
    assert_close(dev_bn.running_var, ref_bn.running_var, 1e-5, 1e-6, 'running_var')
    assert_close(xd.grad.float(), xr.grad, 10 * tol, 10 * tol, 'bncl dx')
    assert_close(dev_bn.weight.grad, ref_bn.weight.grad, 10 * tol, 10 * tol * N ** 0.5, 'bncl dgamma')
    assert_close(dev_bn.bias.grad, ref_bn.bias.grad, 10 * tol, 10 * tol * N ** 0.5, 'bncl dbeta')


# ------------------------------------------------------------------------------------------------ BASELINE-size properties
def test_bn_act_full_size_statistics(ops):
    """[16, 64, 320, 320] bf16 (the first trunk layer at 640 px, bs 16): after BatchNorm every channel has mean beta and standard
    deviation gamma (size-independent property); running statistics moved by momentum * (batch - running)."""
    import torch.nn as nn
    g = torch.Generator(device='cuda').manual_seed(0)
    x = (torch.randn(16, 64, 320, 320, device='cuda', generator=g) * 2.5 + 1.0).bfloat16()
    bn = nn.BatchNorm2d(64, eps=1e-3, momentum=0.03).cuda()
    with torch.no_grad():
        bn.weight.copy_(torch.linspace(0.5, 2.0, 64)); bn.bias.copy_(torch.linspace(-1, 1, 64))
    y = ops.bn_act(x, bn, False).detach().float()
    assert float((y.mean((0, 2, 3)) - bn.bias.detach()).abs().max()) < 2e-2
    assert float((y.std((0, 2, 3)) / bn.weight.detach() - 1).abs().max()) < 2e-2
    xm = x.float().mean((0, 2, 3))
    assert_close(bn.running_mean, 0.03 * xm, 1e-3, 1e-4, 'running mean after one step')
    assert int(bn.num_batches_tracked) == 1


def test_dwconv_full_size_identity_kernel(ops):
    """Level-0 shape (bs 16, 160 x 160, d_inner 256): with the centre-tap kernel the op is SiLU(xi) laid out in both flattenings."""
    g = torch.Generator(device='cuda').manual_seed(1)
    B, H, W, D = 16, 160, 160, 256
    xz = torch.randn(B, H, W, 2 * D, device='cuda', generator=g).bfloat16()
    w = torch.zeros(D, 1, 3, 3, device='cuda'); w[:, 0, 1, 1] = 1.0
    u2 = ops.dwconv_silu_cross(xz, w, None, D)
    a = torch.nn.functional.silu(xz[..., :D].float()).permute(0, 3, 1, 2)   # [B, D, H, W]
    assert_close(u2[:, 0].view(B, D, H, W)[3:5], a[3:5], 1e-6, 1e-6, 'row-major plane')
    assert_close(u2[:, 1].view(B, D, W, H)[11], a[11].transpose(1, 2), 1e-6, 1e-6, 'column-major plane')
    assert float((u2[:, 0].sum() - u2[:, 1].sum()).abs()) < 1e-3 * float(u2[:, 0].abs().sum())


def test_ln_gate_full_size_row_statistics(ops):
    """409 600 tokens x 256 channels: with gamma = 1, beta = 0 the un-gated rows have zero mean and unit variance."""
    g = torch.Generator(device='cuda').manual_seed(2)
    ntok, D = 16 * 160 * 160, 256
    x = torch.randn(1, ntok, D, device='cuda', generator=g) * 3 + 2
    xz = torch.zeros(1, ntok, 1, 2 * D, device='cuda', dtype=torch.bfloat16)
    xz[..., D:] = 20.0   # SiLU(20) = 20 (to 2e-9): the gate becomes a known constant
    out = ops.ln_gate(x, xz, torch.ones(D, device='cuda'), torch.zeros(D, device='cuda'), 1e-5).float() / 20.0
    assert float(out.mean(-1).abs().max()) < 2e-2 and float((out.var(-1, unbiased=False) - 1).abs().max()) < 3e-2


def test_img_augment_matches_host_kernels_bit_for_bit(ops, tmp_path):
    """tamtr_img_augment_u8 (affine warp -> HSV tables -> flips -> CHW float / 255 on the device) against the host kernels of
    libtamtr_host.so / their numpy twin followed by torch's own `.float() / 255` on the device: identical bits, for crops,
    zooms, rotations, every flip combination, the no-HSV flag, and through the dataset with device_augment on and off."""
    import random
    from PIL import Image
    from oracle import imgproc_np as NP
    from tamtr_amd import data as D
    g = np.random.default_rng(12)
    B, SH, SW, H, W = 8, 70, 90, 48, 64
    src = g.integers(0, 256, (B, SH, SW, 3), dtype=np.uint8)
    mats = [np.array([[1, 0, 0], [0, 1, 0]], np.float32), np.array([[1, 0, -13], [0, 1, -11]], np.float32),
            np.array([[0.61, 0.07, 4.3], [-0.05, 0.66, 2.9]], np.float32), np.array([[1.9, 0, -50.2], [0, 1.9, -30.7]], np.float32),
            np.array([[0.8, 0.6, 10], [-0.6, 0.8, 30]], np.float32), np.array([[0.11, 0, 20], [0, 0.11, 20]], np.float32),
            np.array([[1, 0, 0.5], [0, 1, 0.25]], np.float32), np.array([[-1, 0, 63], [0, -1, 47]], np.float32)]
    gains = [np.array([1 + 0.015 * g.uniform(-1, 1), 1 + 0.7 * g.uniform(-1, 1), 1 + 0.4 * g.uniform(-1, 1)]) for _ in range(B)]
    luts = np.stack([D.hsv_luts(gn) for gn in gains])
    flags = np.arange(B, dtype=np.int32)              # 0..7: all flip combinations, 4..7 with the HSV step off
    inv = np.stack([D.invert_affine(m) for m in mats])
    got = ops.img_augment(torch.from_numpy(src).cuda(), torch.from_numpy(inv).cuda(), torch.from_numpy(luts).cuda(),
                          torch.from_numpy(flags).cuda(), (H, W))
    want = []
    for b in range(B):
        pic = D.warp_affine_u8(src[b], mats[b], W, H, 114)
        assert np.array_equal(pic, NP.warp_affine_u8(src[b], mats[b], W, H, 114))
        if not flags[b] & 4:
            pic = NP.hsv_lut_u8(pic, *luts[b])
        if flags[b] & 1:
            pic = pic[::-1]
        if flags[b] & 2:
            pic = pic[:, ::-1]
        want.append(np.ascontiguousarray(pic.transpose(2, 0, 1)))
    want = torch.from_numpy(np.stack(want)).cuda()
    u8 = (got * 255).round().to(torch.uint8)
    for b in range(B):
        assert torch.equal(u8[b], want[b]), f'image {b}: {(u8[b] != want[b]).float().mean().item():.4f} of the pixels differ'
    assert torch.equal(got, want.float() / 255)
    # through the dataset: same seeds, pixel work on the host vs on the device
    (tmp_path / 'images').mkdir(), (tmp_path / 'labels').mkdir()
    for i in range(4):
        Image.fromarray(g.integers(0, 255, (60 + 7 * i, 100 - 9 * i, 3), dtype=np.uint8)).save(tmp_path / 'images' / f'{i}.png')
        (tmp_path / 'labels' / f'{i}.txt').write_text(f'{i} 0.5 0.5 0.4 0.4\n')
    names = ['a', 'b', 'c', 'd']
    tf = D.TextFeatures.synthetic(names + [''], dim=8)
    batches = []
    for dev_aug in (False, True):
        ds = D.PromptDetDataset(str(tmp_path / 'images'), names, imgsz=64, augment=True, hyp={'degrees': 8.0, 'flipud': 0.5},
                                batch_size=4, device_augment=dev_aug)
        random.seed(3), np.random.seed(3)
        batches.append(D.preprocess_batch(D.collate([ds[i] for i in range(4)]), tf, 'cuda'))
    a, b = batches
    assert b['img'].shape == (4, 3, 64, 64) and torch.equal(a['img'], b['img']) and 'src' not in b
    assert torch.equal(a['bboxes'], b['bboxes']) and torch.equal(a['txt_feats'], b['txt_feats'])
    with pytest.raises(Exception):
        ops.img_augment(torch.zeros(1, 4, 4, 3), torch.zeros(1, 6, dtype=torch.float64), torch.zeros(1, 3, 256, dtype=torch.uint8),
                        torch.zeros(1, dtype=torch.int32), (4, 4))


# ------------------------------------------------------------------------------------------------ layout edges (csrc/layout.hip)
@pytest.mark.parametrize('B,C,H,W,dt', [(2, 64, 16, 16, torch.bfloat16), (3, 50, 7, 11, torch.float32), (2, 128, 40, 40, torch.bfloat16),
                                        (1, 6, 5, 13, torch.bfloat16), (2, 256, 20, 24, torch.float32)])
def test_relayout_is_a_pure_permutation(ops, B, C, H, W, dt):
    """NHWC <-> NCHW repacking against torch's own .contiguous(): bit-identical both ways, also from a channel slice of a wider
    channels-last map (`cv1(x).chunk(2, 1)`), and as an autograd edge (the gradient comes back in the input's layout)."""
    x = rnd((B, C, H, W), 5).to(dt).cuda()
    xcl = x.contiguous(memory_format=torch.channels_last)
    a = ops.to_nchw(xcl)
    assert a.is_contiguous() and torch.equal(a, x)
    b = ops.to_channels_last(x)
    assert b.is_contiguous(memory_format=torch.channels_last) and torch.equal(b, x) and b.stride() == xcl.stride()
    wide = rnd((B, 2 * C, H, W), 6).to(dt).cuda().contiguous(memory_format=torch.channels_last)
    for half in wide.chunk(2, 1):
        assert ops._cl_pitch(half) == 2 * C
        c = ops.to_nchw(half)
        assert c.is_contiguous() and torch.equal(c, half.contiguous())
    xg = xcl.clone().requires_grad_()
    cot = rnd((B, C, H, W), 7).to(dt).cuda()
    (ops.to_nchw(xg) * cot).sum().backward()
    assert torch.equal(xg.grad, cot) and xg.grad.is_contiguous(memory_format=torch.channels_last)


# ------------------------------------------------------------------------------------------------ max pooling (csrc/pool.hip)
@pytest.mark.parametrize('B,C,H,W,k,s,p,dt,cl', [(2, 16, 20, 20, 5, 1, 2, torch.bfloat16, True), (2, 8, 17, 23, 3, 2, 1, torch.float32, False),
                                                 (1, 32, 40, 40, 3, 2, 1, torch.bfloat16, False), (2, 6, 9, 9, 5, 1, 2, torch.float32, True),
                                                 (1, 4, 8, 10, 2, 2, 0, torch.float32, False), (2, 8, 12, 16, 5, 1, 2, torch.float32, False),
                                                 (2, 6, 10, 8, 2, 2, 0, torch.bfloat16, False)])
def test_max_pool_vs_torch(ops, B, C, H, W, k, s, p, dt, cl):
    """SPPELAN's 5/1/2 pools (block.py:255-268) and CPAM's 3/2/1 pool (block.py:274) against F.max_pool2d on the CPU: values
    bit-identical, gradients routed to the same winners (bf16 inputs tie often: the first maximum in window order must win)."""
    x = rnd((B, C, H, W), 11).to(dt)
    xr = x.float().clone().requires_grad_()
    ref = F.max_pool2d(xr, k, s, p)
    cot = rnd(tuple(ref.shape), 12).to(dt).float()
    (ref * cot).sum().backward()
    xd = x.cuda()
    if cl:
        xd = xd.contiguous(memory_format=torch.channels_last)
    xd.requires_grad_()
    out = ops.max_pool2d(xd, k, s, p)
    assert out.is_contiguous(memory_format=torch.channels_last if cl else torch.contiguous_format)
    assert torch.equal(out.float().cpu(), ref.detach())
    (out.float() * cot.cuda()).sum().backward()
    assert_close(xd.grad.float().cpu(), xr.grad, 1e-2 if dt == torch.bfloat16 else 1e-6, 1e-6, 'maxpool dx')


# ------------------------------------------------------------------------------------------------ channel concatenation in NHWC
def test_cat_and_pack_channels_vs_torch(ops):
    """ops.cat_channels / pack_channels (csrc/layout.hip tamtr_copy_rows) against torch.cat / .contiguous() on channels-last maps,
    including the two halves of a chunk(2, 1); gradients come back as channel slices with the right values."""
    B, H, W = 2, 12, 20
    a = rnd((B, 32, H, W), 1).bfloat16().cuda().contiguous(memory_format=torch.channels_last).requires_grad_()
    b = rnd((B, 16, H, W), 2).bfloat16().cuda().contiguous(memory_format=torch.channels_last).requires_grad_()
    h0, h1 = a.chunk(2, 1)
    out = ops.cat_channels([h0, h1, b, ops.pack_channels(h1)])
    ref = torch.cat([h0, h1, b, h1], 1)
    assert out.is_contiguous(memory_format=torch.channels_last) and torch.equal(out, ref)
    cot = rnd(tuple(ref.shape), 3).bfloat16().cuda()
    (out.float() * cot.float()).sum().backward()
    ga, gb = a.grad.clone(), b.grad.clone()
    a.grad = b.grad = None
    (ref.float() * cot.float()).sum().backward()
    assert torch.equal(ga, a.grad) and torch.equal(gb, b.grad)
    odd = rnd((B, 6, H, W), 4).cuda().contiguous(memory_format=torch.channels_last)          # fp32, width not a multiple of 4: scalar path
    assert torch.equal(ops.cat_channels([odd, odd[:, 1:4]]), torch.cat([odd, odd[:, 1:4]], 1))


def test_bn_channels_last_backward_reads_a_channel_slice(ops):
    """The gradient of `torch.cat([bn_a(x), bn_b(y)], 1)` reaches each BatchNorm as a channel slice (row pitch 96 instead of 64 / 32):
    tamtr_bncl_act_bwd reads it in place; same result as with packed gradients."""
    import copy
    import torch.nn as nn
    N = 3000
    xa, xb = rnd((N, 64), 1).bfloat16().cuda(), rnd((N, 32), 2).bfloat16().cuda()
    bna, bnb = nn.BatchNorm1d(64, eps=1e-3, momentum=0.03).cuda(), nn.BatchNorm1d(32, eps=1e-3, momentum=0.03).cuda()
    with torch.no_grad():
        bna.weight.copy_(1 + 0.3 * rnd((64,), 3).cuda()); bnb.weight.copy_(1 + 0.3 * rnd((32,), 4).cuda())
    cot = rnd((N, 96), 5).bfloat16().cuda()
    res = []
    for packed in (False, True):
        a, b = xa.clone().requires_grad_(), xb.clone().requires_grad_()
        m1, m2 = copy.deepcopy(bna), copy.deepcopy(bnb)
        ya, yb = ops.bn_act(a, m1, True), ops.bn_act(b, m2, False)
        if packed:
            (ya.float() * cot[:, :64].float()).sum().backward()
            (yb.float() * cot[:, 64:].float()).sum().backward()
        else:
            (torch.cat([ya, yb], 1).float() * cot.float()).sum().backward()
        res.append((a.grad, b.grad, m1.weight.grad, m2.bias.grad))
    for u, v in zip(*res):
        assert_close(u.float(), v.float(), 1e-6, 1e-6, 'bncl backward, strided vs packed gy')


@pytest.mark.parametrize('B,Ls,C,dt', [(3, (400, 100, 25), 256, torch.bfloat16), (2, (77, 30, 9), 64, torch.float32), (1, (640, 160, 40), 256, torch.bfloat16)])
def test_bn_cat_writes_the_token_memory_in_place(ops, B, Ls, C, dt):
    """ops.bn_cat_cl = torch.cat([BatchNorm_i(y_i).view(B, L_i, C)], 1) (head.py:1202-1219) with every level written straight into its
    segment and the backward reading its segment of the gradient in place: the same kernels on other addresses, so every output,
    running statistic and gradient is BIT-identical to the concatenation of ops.bn_act results."""
    import copy
    import torch.nn as nn
    ys = [(rnd((B * L, C), 10 + i) * (1 + i) - 0.3 * i).to(dt).cuda() for i, L in enumerate(Ls)]
    bns = []
    for i in range(len(Ls)):
        bn = nn.BatchNorm2d(C, eps=1e-5, momentum=0.1).cuda()
        with torch.no_grad():
            bn.weight.copy_(1 + 0.3 * rnd((C,), 20 + i).cuda()); bn.bias.copy_(0.2 * rnd((C,), 30 + i).cuda())
        bns.append(bn)
    cot = rnd((B, sum(Ls), C), 40).to(dt).cuda()
    res = []
    for seg in (False, True):
        xs = [y.clone().requires_grad_() for y in ys]
        ms = [copy.deepcopy(bn) for bn in bns]
        if seg:
            assert ops.bn_cat_cl_ok(xs, ms)
            f = ops.bn_cat_cl(xs, ms, B)
        else:
            f = torch.cat([ops.bn_act(x, m, False).view(B, L, C) for x, m, L in zip(xs, ms, Ls)], 1)
        assert f.shape == (B, sum(Ls), C) and f.is_contiguous()
        (f.float() * cot.float()).sum().backward()
        res.append([f.detach()] + [x.grad for x in xs] + [m.weight.grad for m in ms] + [m.bias.grad for m in ms]
                   + [m.running_mean for m in ms] + [m.running_var for m in ms] + [m.num_batches_tracked for m in ms])
    for k, (u, v) in enumerate(zip(*res)):
        assert torch.equal(u, v), f'item {k}: segmented and concatenated forms differ by {(u.float() - v.float()).abs().max().item():.3e}'
    # eval-mode BatchNorms (running statistics) are not this kernel's case
    for m in bns:
        m.eval()
    assert not ops.bn_cat_cl_ok(ys, bns)


@pytest.mark.parametrize('scale,dt', [(2.0, torch.bfloat16), (0.5, torch.bfloat16), (2.0, torch.float32), (0.5, torch.float32)])
def test_nearest_resampling_channels_last(ops, scale, dt):
    """nn.Upsample(scale_factor=2.0 | 0.5, mode='nearest') (TAMTR.yaml layers 11/14/19/22/27/30) on a channels-last map: values
    identical to torch's, gradient = sum of the four copies / scatter to the even pixels."""
    import torch.nn as nn
    from tamtr_amd.backbone import Upsample
    B, C, H, W = 2, 16, 6, 10
    x = rnd((B, C, H, W), 21).to(dt)
    xr = x.float().clone().requires_grad_()
    ref = nn.Upsample(scale_factor=scale, mode='nearest')(xr)
    cot = rnd(tuple(ref.shape), 22).to(dt).float()
    (ref * cot).sum().backward()
    xd = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_()
    out = Upsample(scale_factor=scale, mode='nearest')(xd)
    assert out.is_contiguous(memory_format=torch.channels_last) and torch.equal(out.float().cpu(), ref.detach())
    (out.float() * cot.cuda()).sum().backward()
    assert_close(xd.grad.float().cpu(), xr.grad, 1e-2 if dt == torch.bfloat16 else 1e-6, 1e-6, 'resample dx')


# ------------------------------------------------------------------------------------------------ RepConvN / bottleneck forms of BatchNorm
@pytest.mark.parametrize('N,C,dt,silu', [(3000, 64, torch.bfloat16, True), (777, 32, torch.float32, True), (4100, 128, torch.float32, False)])
def test_bn_pair_and_shortcut_vs_torch(ops, N, C, dt, silu):
    """tamtr_bncl2_act_*: y = act(bn1(x1) + bn2(x2)) (RepConvN, extra_modules/block.py:66-69) and tamtr_bncl_act_fwd's residual input
    (x + cv2(cv1(x)), block.py:100-102) against nn.BatchNorm1d / SiLU / add on the CPU: outputs, running statistics, all gradients."""
    import copy
    import torch.nn as nn
    x1, x2, r = ((rnd((N, C), s_) * 1.3 + 0.2 * s_).to(dt).float() for s_ in (1, 2, 3))
    cot = rnd((N, C), 4).to(dt).float()
    ref1, ref2 = nn.BatchNorm1d(C, eps=1e-3, momentum=0.03), nn.BatchNorm1d(C, eps=1e-3, momentum=0.03)
    with torch.no_grad():
        ref1.weight.copy_(1 + 0.3 * rnd((C,), 5)); ref1.bias.copy_(0.2 * rnd((C,), 6))
        ref2.weight.copy_(1 - 0.2 * rnd((C,), 7)); ref2.bias.copy_(0.1 * rnd((C,), 8))
    d1, d2 = copy.deepcopy(ref1).cuda(), copy.deepcopy(ref2).cuda()
    act = torch.nn.functional.silu if silu else (lambda t: t)
    tol = 2e-5 if dt == torch.float32 else 2e-2
    # pair
    a, b = x1.clone().requires_grad_(), x2.clone().requires_grad_()
    ref = act(ref1(a) + ref2(b))
    (ref * cot).sum().backward()
    ad, bd = dev(x1, dt).requires_grad_(), dev(x2, dt).requires_grad_()
    out = ops.bn2_act(ad, d1, bd, d2, silu)
    (out.float() * dev(cot)).sum().backward()
    assert_close(out.float(), ref, tol, tol, 'pair out')
    for m_d, m_r in ((d1, ref1), (d2, ref2)):
        assert_close(m_d.running_mean, m_r.running_mean, 1e-5, 1e-6, 'running_mean')
        assert_close(m_d.running_var, m_r.running_var, 1e-5, 1e-6, 'running_var')
        assert int(m_d.num_batches_tracked) == 1
        assert_close(m_d.weight.grad, m_r.weight.grad, 10 * tol, 10 * tol * N ** 0.5, 'pair dgamma')
        assert_close(m_d.bias.grad, m_r.bias.grad, 10 * tol, 10 * tol * N ** 0.5, 'pair dbeta')
    assert_close(ad.grad.float(), a.grad, 10 * tol, 10 * tol, 'pair dx1')
    assert_close(bd.grad.float(), b.grad, 10 * tol, 10 * tol, 'pair dx2')
    # shortcut
    for m in (ref1, d1):
        m.zero_grad(set_to_none=True)
    a, rr = x1.clone().requires_grad_(), r.clone().requires_grad_()
    ref = act(ref1(a)) + rr
    (ref * cot).sum().backward()
    ad, rd = dev(x1, dt).requires_grad_(), dev(r, dt).requires_grad_()
    out = ops.bn_act(ad, d1, silu, rd)
    (out.float() * dev(cot)).sum().backward()
    assert_close(out.float(), ref, tol, tol, 'shortcut out')
    assert_close(ad.grad.float(), a.grad, 10 * tol, 10 * tol, 'shortcut dx')
    assert_close(rd.grad.float(), rr.grad, tol, tol, 'shortcut dres')
    assert_close(d1.weight.grad, ref1.weight.grad, 10 * tol, 10 * tol * N ** 0.5, 'shortcut dgamma')


def test_chunk_and_multi_input_cat_kernels(ops):
    """ops.chunk2_channels (kernel-backed backward) and the one-launch concatenation of four inputs (tamtr_cat_rows) - the pattern of
    RepNCSPELAN4.forward (extra_modules/block.py:147-152) - against torch.chunk / torch.cat, forward and backward."""
    B, C, H, W = 2, 64, 10, 12
    base = rnd((B, C, H, W), 1).bfloat16().cuda().contiguous(memory_format=torch.channels_last)
    extra = [rnd((B, c, H, W), 2 + i).bfloat16().cuda().contiguous(memory_format=torch.channels_last) for i, c in enumerate((32, 16))]
    cot = rnd((B, C + 48, H, W), 9).bfloat16().cuda()
    grads = []
    for mine in (True, False):
        x = base.clone().requires_grad_()
        es = [e.clone().requires_grad_() for e in extra]
        y0, y1 = ops.chunk2_channels(x) if mine else x.chunk(2, 1)
        out = (ops.cat_channels if mine else (lambda t: torch.cat(t, 1)))([y0, y1 * 2, es[0], es[1]])
        (out.float() * cot.float()).sum().backward()
        grads.append((out.detach(), x.grad, es[0].grad, es[1].grad))
    for a, b in zip(*grads):
        assert torch.equal(a, b)


@pytest.mark.parametrize('dt,n,shape', [(torch.bfloat16, 4, (5, 37, 64)), (torch.float32, 3, (5, 37, 64)), (torch.bfloat16, 2, (5, 37, 64)),
                                        (torch.bfloat16, 8, (3, 37, 12)), (torch.float32, 5, (1, 1, 4)), (torch.bfloat16, 4, (16, 2100, 256))])
def test_fanout_sums_consumer_gradients_in_one_pass(ops, dt, n, shape):
    """ops.fanout: n handles on one tensor whose n gradients are added by tamtr_sum_n (the MEH token memory feeds enc_output and every
    decoder layer's value_proj) - same total gradient as letting autograd accumulate them."""
    x = rnd(shape, 1).to(dt).cuda()          # (odd and even numbers of 4-element groups, a one-group tensor, a token-memory-sized one)
    ws = [rnd(shape, 2 + i).to(dt).cuda() for i in range(n)]
    a = x.clone().requires_grad_()
    hs = ops.fanout(a, n)
    assert len(hs) == n and all(torch.equal(h, a) for h in hs)
    sum((h * w).float().sum() for h, w in zip(hs, ws)).backward()
    b = x.clone().requires_grad_()
    sum((b * w).float().sum() for w in ws).backward()
    tol = 1e-6 if dt == torch.float32 else 2e-2
    assert_close(a.grad.float(), b.grad.float(), tol, tol * float(b.grad.abs().max()), 'fanout grad')


@pytest.mark.parametrize('M,N', [(16 * 33600, 512), (4672, 512), (100003, 1024), (5000, 64), (40000, 2048)])
def test_colsum_bias_gradient_kernel(M, N):
    """ops.colsum (tamtr_colsum_bf16: the bias gradient of the token-wise linears, db = column sums of dY over B*L tokens) against a
    float64 sum of the same bf16 values; same bits on a second call."""
    import tamtr_amd.ops as ops
    g = torch.Generator(device='cuda').manual_seed(M % 1000)
    x = (torch.randn(M, N, device='cuda', generator=g) + 0.1).bfloat16()
    a, b = ops.colsum(x), ops.colsum(x)
    assert a.dtype == torch.float32 and a.shape == (N,) and torch.equal(a, b)
    ref = x.double().sum(0)
    err = float((a.double() - ref).abs().max())
    assert err <= 1e-5 * float(x.double().abs().sum(0).max()), err       # fp32 accumulation over <= 264 rows per partial, then <= 2048 partials


@pytest.mark.parametrize('shape,dt', [((1600, 256, 10), torch.float32), ((3200, 2, 512), torch.float32), ((16, 1024, 64), torch.float32),
                                      ((64, 512, 2048), torch.float32), ((256, 80, 256), torch.bfloat16), ((33, 8), torch.float32),
                                      ((2048, 512), torch.float32), ((1, 4, 4), torch.float32), ((400, 12), torch.bfloat16)])
def test_slab_sum_is_the_ordered_row_sum(shape, dt):
    """ops.slab_sum (tamtr_slab_sum_rows: the last stage of the package's two-stage reductions - LayerNorm / depthwise-conv partial rows,
    the scan's per-image rows, split-K slices) against a float64 sum of the same values, at the shapes the step produces; the same bits
    on a second call (fixed order, no atomics) and on operands at another address."""
    import tamtr_amd.ops as ops
    g = torch.Generator(device='cuda').manual_seed(sum(shape))
    x = (torch.randn(*shape, device='cuda', generator=g) + 0.05).to(dt)
    a, b = ops.slab_sum(x), ops.slab_sum(x.clone())
    assert a.dtype == torch.float32 and a.shape == x.shape[1:] and torch.equal(a, b)
    ref = x.double().sum(0)
    err = float((a.double() - ref).abs().max())
    assert err <= 2e-6 * float(x.double().abs().sum(0).max()) + 1e-30, err


@pytest.mark.parametrize('B,C1,C2,H,W,dt,sliced', [(16, 256, 128, 40, 40, torch.bfloat16, False), (4, 64, 64, 80, 80, torch.bfloat16, True),
                                                   (2, 32, 32, 13, 21, torch.float32, False), (2, 1536, 512, 20, 20, torch.bfloat16, False),
                                                   (3, 128, 256, 20, 20, torch.float32, True)])
def test_conv1x1_weight_gradient_off_the_library(ops, B, C1, C2, H, W, dt, sliced):
    """ops.conv2d_module on the trunk's 1x1 convolutions (nn/modules/conv.py:23-40 with k = 1): forward and d/d(input) are the library's,
    the weight gradient is the row-sliced product + ordered slab sum (no memset node, no atomics).  Against plain nn.Conv2d autograd on
    the same operands (fp32 reference of the same bf16 values for the weight gradient); packed input and a channel slice of a wider map;
    the same bits on a second backward."""
    import torch.nn as nn
    torch.manual_seed(C1 + H)
    conv = nn.Conv2d(C1, C2, 1, bias=False).cuda()
    wide = (rnd((B, (2 if sliced else 1) * C1, H, W), 1)).to(dt).cuda().contiguous(memory_format=torch.channels_last)
    x0 = wide.chunk(2, 1)[1] if sliced else wide
    cot = rnd((B, C2, H, W), 2).to(dt).cuda().contiguous(memory_format=torch.channels_last)
    w = conv.weight.detach().to(dt)

    def run(fn):
        x, wl = x0.detach().clone(memory_format=torch.preserve_format).requires_grad_(), w.clone().requires_grad_()
        if sliced:   # keep the slice geometry: a view of a leaf
            leaf = wide.detach().clone(memory_format=torch.preserve_format).requires_grad_()
            x = leaf.chunk(2, 1)[1]
        y = fn(x, wl)
        gx, gw = torch.autograd.grad(y, [leaf if sliced else x, wl], cot)
        return y.detach(), gx, gw

    class _M:   # conv2d_module reads the module's attributes and weight
        pass
    def own(x, wl):
        m = nn.Conv2d(C1, C2, 1, bias=False).cuda()
        m.weight = nn.Parameter(wl.detach())
        assert ops.conv1x1_cl_ok(x, m)
        return ops._Conv1x1CL.apply(x, wl)
    y, gx, gw = run(own)
    y2, gx2, gw2 = run(own)
    yr, gxr, gwr = run(lambda x, wl: torch.nn.functional.conv2d(x, wl))
    assert torch.equal(gw, gw2) and torch.equal(gx, gx2)      # the backward is two GEMMs in a fixed order
    assert_close(y.float(), yr.float(), 1e-2 if dt == torch.bfloat16 else 1e-5, (1e-2 if dt == torch.bfloat16 else 1e-5) * float(yr.float().abs().max()), 'y')   # (the library's forward is not bitwise reproducible for every shape)
    assert_close(gx.float(), gxr.float(), 2e-2 if dt == torch.bfloat16 else 1e-4, (2e-2 if dt == torch.bfloat16 else 1e-4) * float(gxr.float().abs().max()), 'dX')
    ref = torch.einsum('bohw,bihw->oi', cot.double(), x0.double()).view(C2, C1, 1, 1)      # exact dW of the same (rounded) operands
    tol = 1e-2 if dt == torch.bfloat16 else 2e-4                                            # own: fp32 sums rounded once to the weight's dtype
    assert_close(gw.double(), ref, tol, tol * float(ref.abs().max()), 'dW')
    err_own, err_lib = float((gw.double() - ref).norm() / ref.norm()), float((gwr.double() - ref).norm() / ref.norm())
    assert err_own <= 1.5 * err_lib + 1e-6, (err_own, err_lib)                              # no worse than the library's own weight gradient
    if dt == torch.bfloat16:
        # the weight as a bf16 COPY of an fp32 master (model._CastGroup): the fp32 slab sum goes to the master unrounded, the copy gets nothing
        master = conv.weight.detach().clone().requires_grad_()
        x, copy16 = x0.detach().clone(memory_format=torch.preserve_format).requires_grad_(), master.detach().to(dt).requires_grad_()
        if sliced:
            x = wide.detach().clone(memory_format=torch.preserve_format).requires_grad_().chunk(2, 1)[1]
        copy16._tamtr_master = master
        m = nn.Conv2d(C1, C2, 1, bias=False).cuda()
        del m.weight
        m.weight = copy16           # (what torch.func.functional_call does for the duration of the layer's forward)
        ym = ops.conv2d_module(m, x)
        gm, gc = torch.autograd.grad(ym, [master, copy16], cot, allow_unused=True)
        assert gc is None and gm.dtype == torch.float32 and ym.dtype == y.dtype and ym.shape == y.shape
        assert torch.equal(gm.to(dt), gw), 'the master receives the same sum, before its rounding to bf16'
        assert float((gm.double() - ref).norm() / ref.norm()) <= err_own + 1e-9


@pytest.mark.parametrize('shadows', [False, True])
def test_fused_optim_step_equals_clip_adamw_ema(shadows):
    """engine.FusedOptimStep (csrc/optim.hip: clip_grad_norm_ + AdamW + EMA in four launches) against torch.nn.utils.clip_grad_norm_,
    torch.optim.AdamW and engine.ModelEMA on a twin model, over six steps: three parameter groups with their own lr / weight decay
    (engine.build_optimizer's layout), a channels-last convolution weight, BatchNorm statistics (EMA-only entries), a parameter that gets
    no gradient on some steps (its Adam step count must lag, like torch's), gradients far above and below the clipping norm, a changing
    learning rate (warm-up).  Parameters, moments, step counts, EMA weights and the clipped gradients agree at fp32 rounding.
    shadows=True: the kernel also keeps a bf16 copy of every parameter (`p._tamtr_bf16`, ops.bf16_of): after every step each copy is the
    master rounded to bf16, bit for bit; a copy whose master was changed behind the stepper's back is not handed out (tensor version) and is
    re-derived by the next step(); model.load_state_dict refreshes them."""
    import copy
    import torch.nn as nn
    import tamtr_amd.ops as ops
    from tamtr_amd.engine import FusedOptimStep, ModelEMA, build_optimizer

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = nn.Conv2d(8, 16, 3, padding=1, bias=False)   # (a bias in front of BatchNorm has a zero gradient: rounding noise that Adam
            self.bn = nn.BatchNorm2d(16)                              # would blow up to +-lr, differently in two runs of the library convolution)
            self.fc = nn.Linear(16, 5003)                 # > one 8 192-element chunk, odd length: the scalar tail
            self.sometimes = nn.Linear(16, 7)
            self.never = nn.Linear(3, 3)

        def forward(self, x, use):
            h = torch.relu(self.bn(self.conv(x))).mean((2, 3))
            y = self.fc(h).pow(2).mean()
            return y + self.sometimes(h).sum() if use else y
    torch.manual_seed(0)
    a = Net().cuda().train()
    a.conv.weight.data = a.conv.weight.data.contiguous(memory_format=torch.channels_last)
    b = copy.deepcopy(a)
    oa, ob = build_optimizer(a, 'AdamW', lr=1e-2, decay=1e-2), build_optimizer(b, 'AdamW', lr=1e-2, decay=1e-2)
    ea, eb = ModelEMA(a, tau=3), ModelEMA(b, tau=3)
    st = FusedOptimStep.create(b, ob, eb, max_norm=0.1, shadows=shadows)
    assert st is not None
    if shadows:
        assert len(st.shadows) == len(list(b.parameters()))
        for p_ in b.parameters():   # there before the first forward, in the master's layout
            sh = ops.bf16_shadow(p_)
            assert sh is not None and sh.dtype == torch.bfloat16 and sh.stride() == p_.stride() and torch.equal(sh, p_.detach().bfloat16())
            assert ops.bf16_of(p_) is sh
    else:
        assert all(ops.bf16_shadow(p_) is None for p_ in b.parameters())
    g = torch.Generator(device='cuda').manual_seed(1)
    for step in range(6):
        x = torch.randn(4, 8, 6, 6, device='cuda', generator=g).contiguous(memory_format=torch.channels_last) * (10.0 if step % 2 else 1e-3)
        for grp_a, grp_b in zip(oa.param_groups, ob.param_groups):
            grp_a['lr'] = grp_b['lr'] = 1e-2 * (step + 1) / 6
        for m_, o_ in ((a, oa), (b, ob)):
            o_.zero_grad(set_to_none=True)
            m_(x, step not in (1, 4)).backward()
        want_norm = torch.nn.utils.clip_grad_norm_([p for p in a.parameters() if p.grad is not None], max_norm=0.1)
        oa.step()
        ea.update(a)
        got = st.step()
        assert abs(float(got[0]) - float(want_norm)) <= 1e-5 * float(want_norm), (float(got[0]), float(want_norm))
        for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
            assert_close(pb, pa, 1e-5, 2e-6, f'step {step} param {n}')   # (atol: an element whose gradient is rounding noise moves by ~lr * noise ratio)
            assert (pa.grad is None) == (pb.grad is None)
            if pa.grad is not None:
                # (the two norms differ in the order of their fp32 sums, asserted to 1e-5 above: the clip coefficient carries that into the
                # clipped gradient and the moments; Adam's m / sqrt(v) cancels it in the parameter)
                assert_close(pb.grad, pa.grad, 1e-4, 1e-9, f'step {step} clipped grad {n}')
                sa, sb = oa.state[pa], ob.state[pb]
                assert float(sa['step']) == float(sb['step']), n
                assert_close(sb['exp_avg'], sa['exp_avg'], 1e-4, 1e-10, f'exp_avg {n}')
                assert_close(sb['exp_avg_sq'], sa['exp_avg_sq'], 2e-4, 1e-14, f'exp_avg_sq {n}')
        for (k, va), vb in zip(ea.ema.state_dict().items(), eb.ema.state_dict().values()):
            if va.dtype.is_floating_point:
                assert_close(vb, va, 1e-5, 2e-6, f'step {step} ema {k}')
        assert ea.updates == eb.updates == step + 1
        if shadows:
            for n, p_ in b.named_parameters():
                sh = ops.bf16_shadow(p_)
                assert sh is not None and torch.equal(sh, p_.detach().bfloat16()), f'step {step}: bf16 copy of {n}'
            if step == 2:   # a master changed by something else: its copy is withheld until the next step() has re-derived it
                with torch.no_grad():
                    b.never.weight.mul_(2.0); a.never.weight.mul_(2.0)          # (never gets a gradient: only the stale-copy path can fix it)
                    b.fc.bias.add_(0.25); a.fc.bias.add_(0.25)                  # (gets one: the kernel rewrites it)
                assert ops.bf16_shadow(b.never.weight) is None and ops.bf16_shadow(b.fc.bias) is None
                assert torch.equal(ops.bf16_of(b.never.weight), b.never.weight.detach().bfloat16())
    assert float(ob.state[b.sometimes.weight]['step']) == 4.0 and float(ob.state[b.never.weight]['step']) == 0.0 and b.never.weight.grad is None
    # the optimizer's state_dict round-trips (the step counts are ordinary tensors to it) and the stepper notices replaced state
    sd = copy.deepcopy(ob.state_dict())
    ob.load_state_dict(sd)
    b(x, True).backward()
    st.step()
    assert float(ob.state[b.fc.weight]['step']) == 7.0
    if shadows:
        addr = {n: ops.bf16_shadow(p_).data_ptr() for n, p_ in b.named_parameters()}
        assert {id(p_): s_.data_ptr() for p_, s_ in st.shadows} == {id(p_): ops.bf16_shadow(p_).data_ptr() for p_ in b.parameters()}
        sd_m = {k: v.clone() * 0.5 if v.dtype.is_floating_point else v.clone() for k, v in b.state_dict().items()}
        b.load_state_dict(sd_m)
        for n, p_ in b.named_parameters():
            sh = ops.bf16_shadow(p_)
            assert sh is not None and sh.data_ptr() == addr[n] and torch.equal(sh, p_.detach().bfloat16()), n
        st.drop_shadows()
        assert all(ops.bf16_shadow(p_) is None for p_ in b.parameters())


@pytest.mark.parametrize('ddt', [torch.float32, torch.bfloat16])
def test_box_refine_vs_torch_formula(ops, ddt):
    """ops.box_refine = sigmoid(delta + inverse_sigmoid(ref)) (nn/modules/transformer.py:881-887, nn/modules/utils.py:46-52) against the
    torch expression, values and both gradients, including references on and beyond the clamps (0, 1, below eps, above 1 - eps, outside
    [0, 1]) where torch's clamp gradients decide."""
    g = torch.Generator().manual_seed(5)
    ref = torch.rand(16, 292, 4, generator=g)
    edge = torch.tensor([0.0, 1.0, 5e-6, 1 - 5e-6, 3e-5, 1 - 3e-5, -0.2, 1.3, 0.5, 2e-5])   # (not eps itself: float(1e-5) < 1e-5, a tie the fp64 reference breaks the other way)
    ref.view(-1)[:edge.numel()] = edge
    delta = (torch.randn(16, 292, 4, generator=g) * 2).to(ddt)
    cot = torch.randn(16, 292, 4, generator=g)

    def formula(d, r):
        x = r.clamp(min=0, max=1)
        return torch.sigmoid(d + torch.log(x.clamp(min=1e-5) / (1 - x).clamp(min=1e-5)))
    dr, rr = delta.double().requires_grad_(), ref.double().requires_grad_()
    want = formula(dr, rr)
    (want * cot.double()).sum().backward()
    dd, rd = delta.cuda().requires_grad_(), ref.cuda().requires_grad_()
    got = ops.box_refine(dd, rd)
    assert got.dtype == torch.float32 and got.shape == ref.shape
    (got * cot.cuda()).sum().backward()
    assert_close(got, want, 1e-5, 1e-6, 'refined boxes')
    assert dd.grad.dtype == ddt
    tol = 1e-5 if ddt == torch.float32 else 1e-2
    assert_close(dd.grad.float(), dr.grad, tol, tol * float(dr.grad.abs().max()) * 1e-1, 'd/d(delta)')
    assert_close(rd.grad, rr.grad, 1e-4, 1e-5 * float(rr.grad.abs().max()), 'd/d(ref)')
    # a detached reference (what the decoder passes between layers) gets no gradient buffer at all
    d2 = delta.cuda().requires_grad_()
    ops.box_refine(d2, ref.cuda()).sum().backward()
    assert d2.grad is not None


@pytest.mark.parametrize('M,K,N,xdt,bias', [(4672, 512, 512, torch.float32, True), (4672, 512, 1024, torch.bfloat16, True), (1600, 1024, 512, torch.float32, True),
                                           (292, 512, 96, torch.float32, True), (640, 64, 8, torch.bfloat16, False), (4672, 512, 4, torch.float32, True),
                                           (4672, 4, 1024, torch.float32, True)])
def test_linear_master_vs_autocast_linear(ops, M, K, N, xdt, bias):
    """ops.linear on the decoder side's nn.Linear layers (transformer.py:539-558,869-889) against the module under bf16 autocast: the same
    forward (same library GEMM on the same bf16 operands), the same input gradient, and weight / bias gradients that are the fp32 sums
    autocast's path rounds to bf16 - compared with the exact sums of the same bf16 operands, they must be at least as close."""
    import torch.nn as nn
    torch.manual_seed(M + N)
    lin = nn.Linear(K, N, bias=bias).cuda()
    x0 = (rnd((M, K), 1)).to(xdt).cuda()
    cot = rnd((M, N), 2).bfloat16().cuda()
    res = []
    for own in (False, True):
        lin.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            assert ops.linear_master_ok(x, lin)
            y = ops.linear(x, lin) if own else lin(x)
        assert y.dtype == torch.bfloat16
        y.backward(cot)
        res.append((y.detach(), x.grad, lin.weight.grad.clone(), None if not bias else lin.bias.grad.clone()))
    (y0, gx0, gw0, gb0), (y1, gx1, gw1, gb1) = res
    assert torch.equal(y1, y0) and gx1.dtype == xdt
    assert_close(gx1.float(), gx0.float(), 1e-2, 1e-2 * float(gx0.float().abs().max()), 'dX')
    x16, w16 = x0.bfloat16().double(), lin.weight.detach().bfloat16().double()
    gw_ref, gb_ref = cot.double().t() @ x16, cot.double().sum(0)
    assert gw1.dtype == torch.float32
    e_own, e_auto = float((gw1.double() - gw_ref).norm() / gw_ref.norm()), float((gw0.double() - gw_ref).norm() / gw_ref.norm())
    assert e_own <= (e_auto + 1e-7 if N % 8 == 0 and K % 8 == 0 else 3 * e_auto + 1e-3) and e_own <= 5e-3, (e_own, e_auto)   # (skinny layers: bf16 partial products)
    if bias:
        e_own, e_auto = float((gb1.double() - gb_ref).norm() / gb_ref.norm()), float((gb0.double() - gb_ref).norm() / gb_ref.norm())
        assert e_own <= e_auto + 1e-7 and e_own <= 1e-5, (e_own, e_auto)
    # a second backward gives the same bits (no atomics, fixed order)
    lin.zero_grad(set_to_none=True)
    x = x0.clone().requires_grad_()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        ops.linear(x, lin).backward(cot)
    assert torch.equal(lin.weight.grad, gw1) and (not bias or torch.equal(lin.bias.grad, gb1))
    # packed projection rows (the self-attention's in_proj): row blocks of one parameter
    wp, bp = nn.Parameter(rnd((3 * N, K), 3).cuda() * K ** -0.5), nn.Parameter(rnd((3 * N,), 4).cuda() * 0.1)
    x = x0.clone().requires_grad_()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        ya = ops.linear_rows(x, wp, bp, 0, 2 * N)
        yr = torch.nn.functional.linear(x, wp[:2 * N], bp[:2 * N])
    assert torch.equal(ya, yr)
    ya.float().pow(2).sum().backward()
    assert wp.grad.shape == wp.shape and float(wp.grad[2 * N:].abs().max()) == 0 and float(wp.grad[:2 * N].abs().max()) > 0


@pytest.mark.parametrize('B,D,H,W,R', [(2, 256, 24, 40, 8), (1, 512, 16, 24, 16), (1, 1024, 8, 16, 32), (1, 256, 36, 28, 8)])
def test_ss2d_bf16_planes_are_the_f32_kernels_rounded_once(ops, B, D, H, W, R):
    """bf16 mode keeps SS2D's big time-indexed planes (u2, y, d(y), d(u), d(u2)) in bf16 between the kernels (include/tamtr_hip.h "bf16
    PLANES").  Every kernel of the chain in that form against its fp32-plane form: fed the SAME (bf16-representable) values it must
    produce the same numbers - bit for bit where the output is fp32, and exactly the fp32 form's output rounded to bf16 where it is a plane."""
    import tamtr_amd._lib as L_
    from tamtr_amd._lib import call, ptr, stream_ptr
    L, N, K, C = H * W, 16, 4, R + 32
    g = torch.Generator(device='cuda').manual_seed(D + L)
    rn = lambda *sh: torch.randn(*sh, device='cuda', generator=g)   # noqa: E731
    sp = stream_ptr()
    # ---- front end: depthwise conv + SiLU -> u2
    xz = rn(B, H, W, 2 * D).bfloat16()
    cw, cb = rn(D, 9) * 0.3, rn(D) * 0.1
    u32, u16 = torch.empty(B, 2, D, L, device='cuda'), torch.empty(B, 2, D, L, device='cuda', dtype=torch.bfloat16)
    call('tamtr_dwconv_silu_cross_fwd', ptr(xz), 2 * D, ptr(cw), ptr(cb), ptr(u32), B, D, H, W, 1, 0, sp)
    call('tamtr_dwconv_silu_cross_fwd', ptr(xz), 2 * D, ptr(cw), ptr(cb), ptr(u16), B, D, H, W, 1, 1, sp)
    assert torch.equal(u16, u32.bfloat16()), 'dwconv forward plane'
    uf = u16.float()
    # ---- x_proj forward
    wx = rn(4, C, D) * D ** -0.5
    wcat = ops.xproj_pack_weight(wx)
    o32 = [torch.empty(B, 4, n, L, device='cuda') for n in (R, N, N)]
    o16 = [torch.empty(B, 4, n, L, device='cuda') for n in (R, N, N)]
    call('tamtr_xproj_fwd', ptr(uf), ptr(wcat), ptr(o32[0]), ptr(o32[1]), ptr(o32[2]), B, D, L, R, 0, sp)
    call('tamtr_xproj_fwd', ptr(u16), ptr(wcat), ptr(o16[0]), ptr(o16[1]), ptr(o16[2]), B, D, L, R, 1, sp)
    for a, b, nm in zip(o16, o32, ('dtr', 'Bs', 'Cs')):
        assert torch.equal(a, b), 'x_proj forward ' + nm
    dtr, Bs, Cs = o32
    # ---- scan forward
    Wdt, A, Dv, db = rn(K * D, R) * R ** -0.5, -torch.exp(rn(K * D, N) * 0.5), rn(K * D), rn(K * D) * 0.5 - 1.0
    chunk = L_.lib().tamtr_selective_scan_chunk()
    nck = (L + chunk - 1) // chunk
    y32, y16 = torch.empty(B, K, D, L, device='cuda'), torch.empty(B, K, D, L, device='cuda', dtype=torch.bfloat16)
    h32, h16 = torch.empty(B, K * D, nck, N, device='cuda'), torch.empty(B, K * D, nck, N, device='cuda')
    call('tamtr_selective_scan_dtproj_fwd', ptr(uf), ptr(dtr), ptr(Wdt), ptr(A), ptr(Bs), ptr(Cs), ptr(Dv), ptr(db), ptr(y32), ptr(h32), B, K, D, N, R, L, 1, 0, sp)
    call('tamtr_selective_scan_dtproj_fwd', ptr(u16), ptr(dtr), ptr(Wdt), ptr(A), ptr(Bs), ptr(Cs), ptr(Dv), ptr(db), ptr(y16), ptr(h16), B, K, D, N, R, L, 1, 1, sp)
    assert torch.isfinite(y32).all()
    assert torch.equal(h16, h32), 'scan checkpoints'
    assert torch.equal(y16, y32.bfloat16()), 'scan forward plane'
    # ---- cross merge forward / backward
    m32, m16 = torch.empty(B, L, D, device='cuda'), torch.empty(B, L, D, device='cuda')
    call('tamtr_cross_merge_fwd', ptr(y16.float()), ptr(m32), B, D, H, W, 0, sp)
    call('tamtr_cross_merge_fwd', ptr(y16), ptr(m16), B, D, H, W, 1, sp)
    assert torch.equal(m16, m32), 'cross merge forward'
    gm = rn(B, L, D)
    g32, g16 = torch.empty(B, 2, D, L, device='cuda'), torch.empty(B, 2, D, L, device='cuda', dtype=torch.bfloat16)
    call('tamtr_cross_merge_bwd', ptr(gm), ptr(g32), B, D, H, W, 0, sp)
    call('tamtr_cross_merge_bwd', ptr(gm), ptr(g16), B, D, H, W, 1, sp)
    assert torch.equal(g16, g32.bfloat16()), 'cross merge backward plane'
    # ---- scan backward
    nslab = L_.lib().tamtr_selective_scan_bwd_slabs(D)
    outs = []
    for p16 in (0, 1):
        gu = torch.empty(B, K * D, L, device='cuda', dtype=torch.bfloat16 if p16 else torch.float32)
        gdelta = torch.empty(B, K * D, L, device='cuda', dtype=torch.bfloat16)
        gdtr, gB, gC = torch.empty_like(dtr), torch.empty_like(Bs), torch.empty_like(Cs)
        grow = torch.empty(B, K * D, L_.lib().tamtr_selective_scan_row_sums(), device='cuda')
        ws = torch.empty(2 * nslab * Bs.numel(), device='cuda')
        call('tamtr_selective_scan_dtproj_bwd', ptr(g16 if p16 else g16.float()), ptr(u16 if p16 else uf), ptr(dtr), ptr(Wdt), ptr(A), ptr(Bs), ptr(Cs), ptr(Dv),
             ptr(db), ptr(h32), ptr(gu), ptr(gdelta), ptr(gdtr), ptr(grow), ptr(gB), ptr(gC), ptr(ws), B, K, D, N, R, L, 3, 1 | (2 * p16), sp)
        outs.append((gu, gdtr, gB, gC, grow))
    assert torch.equal(outs[1][0], outs[0][0].bfloat16()), 'scan backward d(u) plane'
    for a, b, nm in zip(outs[1][1:], outs[0][1:], ('gdtr', 'gB', 'gC', 'row sums')):
        assert torch.equal(a, b), 'scan backward ' + nm
    gu16, gdtr, gB, gC, _ = outs[1]
    # ---- x_proj backward
    wT = ops.xproj_pack_weight_t(wcat, C)
    d32, d16 = torch.empty(B, 2, D, L, device='cuda'), torch.empty(B, 2, D, L, device='cuda', dtype=torch.bfloat16)
    call('tamtr_xproj_bwd_dx', ptr(gu16.float()), ptr(gdtr), ptr(gB), ptr(gC), ptr(wT), ptr(d32), B, D, L, R, 0, sp)
    call('tamtr_xproj_bwd_dx', ptr(gu16), ptr(gdtr), ptr(gB), ptr(gC), ptr(wT), ptr(d16), B, D, L, R, 1, sp)
    assert torch.equal(d16, d32.bfloat16()), 'x_proj backward d(u2) plane'
    nsl = L_.lib().tamtr_xproj_dw_slices(L)
    p32, p16_ = (torch.full((B * nsl, 2, 2 * C, D), float('nan'), device='cuda') for _ in range(2))
    call('tamtr_xproj_bwd_dw', ptr(uf), ptr(gdtr), ptr(gB), ptr(gC), ptr(p32), B, D, L, R, 0, sp)
    call('tamtr_xproj_bwd_dw', ptr(u16), ptr(gdtr), ptr(gB), ptr(gC), ptr(p16_), B, D, L, R, 1, sp)
    assert torch.isfinite(p32).all() and torch.equal(p16_, p32), 'x_proj weight gradient'
    # ---- front end backward
    tiles = L_.lib().tamtr_dwconv_tiles(H, W)
    res = []
    for pc, gpl in ((0, d16.float()), (1, d16)):
        gxz, wsd = torch.zeros_like(xz), torch.empty(B, tiles, D, 10, device='cuda')
        call('tamtr_dwconv_silu_cross_bwd', ptr(gpl), ptr(xz), 2 * D, ptr(cw), ptr(cb), ptr(gxz), 2 * D, ptr(wsd), B, D, H, W, 1, pc, sp)
        res.append((gxz, wsd))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), 'dwconv backward'
    # argument checks: bf16 planes only beside bf16 activations, only on the vector path
    with pytest.raises(L_.TamtrHipError):
        call('tamtr_dwconv_silu_cross_fwd', ptr(xz.float()), 2 * D, ptr(cw), ptr(cb), ptr(u16), B, D, H, W, 0, 1, sp)
    with pytest.raises(L_.TamtrHipError):
        call('tamtr_selective_scan_dtproj_fwd', ptr(u16), ptr(dtr), ptr(Wdt), ptr(A), ptr(Bs), ptr(Cs), ptr(Dv), ptr(db), ptr(y16), ptr(h16), B, K, D, N, R, L - 1, 1, 1, sp)


@pytest.mark.parametrize('B,D,L,R', [(2, 256, 1048, 8), (1, 512, 400, 16), (1, 1024, 136, 32), (3, 256, 64, 8)])
def test_xproj_kernels_vs_fp32_products_of_the_same_bf16_operands(ops, B, D, L, R):
    """csrc/xproj.hip (x_proj of SS2D on the pair layout, vmamba.py:962-975): the forward's three outputs, d/d(u2) with the fold of the
    scan's four planes, and the weight gradient's partial tiles + ordered sum, against fp64 products of the same operands rounded to
    bf16 (what the bf16 library GEMMs they replace computed); ragged pixel counts (not a multiple of 32 / of the 1 024-pixel slice);
    the same bits on a second call."""
    import tamtr_amd._lib as L_
    N, C = 16, R + 32
    g = torch.Generator(device='cuda').manual_seed(D + L)
    rn = lambda *sh: torch.randn(*sh, device='cuda', generator=g)   # noqa: E731
    u2, wx = rn(B, 2, D, L), rn(4, C, D) * D ** -0.5
    r16 = lambda t: t.bfloat16().double()                             # noqa: E731
    wcat = ops.xproj_pack_weight(wx)
    assert wcat.shape == (2, -(-2 * C // 32) * 32, D) and wcat.dtype == torch.bfloat16
    outs = [[torch.empty(B, 4, n, L, device='cuda') for n in (R, N, N)] for _ in range(2)]
    for o in outs:
        ops.call('tamtr_xproj_fwd', ops.ptr(u2), ops.ptr(wcat), ops.ptr(o[0]), ops.ptr(o[1]), ops.ptr(o[2]), B, D, L, R, 0, ops.stream_ptr())
    assert all(torch.equal(a, b) for a, b in zip(*outs))
    dtr, Bs, Cs = outs[0]
    for i in range(2):
        W = torch.cat([r16(wx[i]), r16(wx[i + 2])], 0)                                   # [2C, D]
        ref = torch.einsum('md,bdl->bml', W, r16(u2[:, i]))                               # [B, 2C, L]
        got = torch.cat([dtr[:, i], Bs[:, i], Cs[:, i], dtr[:, i + 2], Bs[:, i + 2], Cs[:, i + 2]], 1).double()
        assert_close(got, ref, 8e-3, 8e-3 * float(ref.abs().max()), f'x_proj forward copy {i}')   # one bf16 rounding of the product
    # backward
    gdtr, gB, gC, gu = rn(B, 4, R, L), rn(B, 4, N, L), rn(B, 4, N, L), rn(B, 4, D, L)
    wT = ops.xproj_pack_weight_t(wcat, C)
    assert wT.shape == (2, D, -(-2 * C // 16) * 16)
    gu2 = [torch.empty(B, 2, D, L, device='cuda') for _ in range(2)]
    for o in gu2:
        ops.call('tamtr_xproj_bwd_dx', ops.ptr(gu), ops.ptr(gdtr), ops.ptr(gB), ops.ptr(gC), ops.ptr(wT), ops.ptr(o), B, D, L, R, 0, ops.stream_ptr())
    assert torch.equal(*gu2)
    nsl = L_.lib().tamtr_xproj_dw_slices(L)
    parts = [torch.full((B * nsl, 2, 2 * C, D), float('nan'), device='cuda') for _ in range(2)]
    for o in parts:
        ops.call('tamtr_xproj_bwd_dw', ops.ptr(u2), ops.ptr(gdtr), ops.ptr(gB), ops.ptr(gC), ops.ptr(o), B, D, L, R, 0, ops.stream_ptr())
    assert torch.equal(*parts) and torch.isfinite(parts[0]).all()
    gws = ops.slab_sum(parts[0])
    for i in range(2):
        W = torch.cat([r16(wx[i]), r16(wx[i + 2])], 0)
        G = torch.cat([gdtr[:, i], gB[:, i], gC[:, i], gdtr[:, i + 2], gB[:, i + 2], gC[:, i + 2]], 1)    # [B, 2C, L]
        prod = torch.einsum('md,bml->bdl', W, r16(G))
        ref = gu[:, i].double() + gu[:, i + 2].double() + prod
        tol = 8e-3 * float(prod.abs().max())                                             # the product is rounded to bf16 before the fold
        assert_close(gu2[0][:, i].double(), ref, 1e-6, tol, f'd/d(u2) copy {i}')
        refw = torch.einsum('bml,bdl->md', r16(G), r16(u2[:, i]))
        assert_close(gws[i].double(), refw, 1e-4, 1e-4 * float(refw.abs().max()), f'dWcat copy {i}')


@pytest.mark.parametrize('Lr,B,nq,nc,counts', [(4, 16, 100, 10, [8] * 16), (3, 4, 192, 10, [8, 3, 0, 5]), (2, 2, 37, 80, [1, 40]), (1, 3, 50, 7, [2, 2, 2])])
def test_fused_loss_terms_and_matcher_cost_equal_the_torch_form(Lr, B, nq, nc, counts):
    """csrc/detrloss.hip against the elementwise torch form of tam-tr_amd/loss.py (which the CPU suite pins on the reference's fixtures:
    loss / matcher / riou): the matcher's cost matrices of all layers, the assignment that follows from them, the three terms per layer, and
    their gradients with respect to boxes and logits - ragged box counts incl. an image without boxes, the dn branch's fixed pairs, 80 classes."""
    import tamtr_amd.loss as LS
    g = torch.Generator().manual_seed(Lr * 100 + nq)
    G = sum(counts)
    pb0 = torch.cat([0.2 + 0.6 * torch.rand(Lr, B, nq, 2, generator=g), 0.03 + 0.3 * torch.rand(Lr, B, nq, 2, generator=g)], -1).cuda()
    ps0 = (torch.randn(Lr, B, nq, nc, generator=g) * 2 - 3).cuda()
    gtb = torch.cat([0.2 + 0.6 * torch.rand(G, 2, generator=g), 0.03 + 0.3 * torch.rand(G, 2, generator=g)], -1).cuda()
    gtc = torch.randint(0, nc, (G,), generator=g).cuda()
    crit = LS.DETRLoss(nc=nc, use_vfl=True).cuda()
    out, mats = {}, {}
    LS._TORCH_LOSS = True
    try:
        ms_torch = crit.matcher(pb0, ps0, gtb, gtc, counts)
    finally:
        LS._TORCH_LOSS = False
    for mode in ('torch', 'fused'):
        LS._TORCH_LOSS = mode == 'torch'
        try:
            ms = crit.matcher(pb0, ps0, gtb, gtc, counts)
            mats[mode] = [m.flat for m in ms]
            for tag, match in (('hungarian', None), ('fixed', ms_torch[0])):      # `match` given = the dn branch (one pair list for every layer)
                pb, ps = pb0.clone().requires_grad_(), ps0.clone().requires_grad_()
                crit.fixed_matches = ms_torch if match is None else None      # the same pairs on both paths
                terms = crit._layers(pb, ps, gtb, gtc, counts, match)
                w = torch.arange(1, 3 * Lr + 1, device='cuda', dtype=torch.float32).view(3, Lr) / Lr          # distinct upstream gradients
                gpb, gps = torch.autograd.grad((torch.stack(terms) * w).sum(), [pb, ps])
                out[mode, tag] = ([t.detach() for t in terms], gpb, gps)
        finally:
            LS._TORCH_LOSS = False
            crit.fixed_matches = None
    for a, b in zip(mats['torch'], mats['fused']):
        assert all(torch.equal(x, y) for x, y in zip(a, b)), 'assignments differ'
    for tag in ('hungarian', 'fixed'):
        (tt, bt, st), (tf, bf, sf) = out['torch', tag], out['fused', tag]
        for name, x, y in zip(('class', 'bbox', 'giou'), tf, tt):
            assert_close(x, y, 2e-5, 1e-6, f'{tag} {name} terms')
        assert_close(bf, bt, 2e-4, 2e-6 * float(bt.abs().max()), f'{tag} d/d(boxes)')
        assert_close(sf, st, 2e-4, 2e-6 * float(st.abs().max()), f'{tag} d/d(logits)')
        assert int((bf != 0).any(-1).sum()) == int((bt != 0).any(-1).sum())      # exactly the matched rows carry a box gradient


# ------------------------------------------------------------------------------------------------ next-3: proj_conv on MFMA
@pytest.mark.parametrize('B,C,H,W,sliced', [(2, 64, 160, 160, True), (2, 128, 80, 80, True), (2, 256, 40, 40, True),
                                            (1, 64, 13, 21, False), (3, 128, 8, 16, False), (1, 64, 320, 320, True)])
def test_conv3x3_cl_stats_vs_fp32_conv(ops, B, C, H, W, sliced):
    """tamtr_conv3x3_cl_stats_fwd (the gate's value branch, block.py:205,223: Conv2d 3x3 s1 p1 no bias + training BatchNorm
    statistics) at the five TIAGELAN sites' shapes (640 px), the 1280 px level-0 shape and two ragged maps, against torch's CPU fp32
    convolution of the same bf16 values.  Output: one bf16 rounding of an fp32 sum (2^-8 relative + accumulation-order noise);
    statistics: exactly those of the stored values (double-precision check), the running update as nn.BatchNorm2d does it."""
    torch.manual_seed(C + H)
    wide = (rnd((B, 2 * C if sliced else C, H, W), 1) * 1.5).bfloat16()
    conv = torch.nn.Conv2d(C, C, 3, 1, 1, bias=False)
    bn = torch.nn.BatchNorm2d(C, eps=1e-3, momentum=0.03)
    with torch.no_grad():
        conv.weight.copy_(rnd((C, C, 3, 3), 2, (9 * C) ** -0.5 * 2.0))
        bn.running_mean.copy_(rnd((C,), 3, 0.1)); bn.running_var.copy_(1 + 0.1 * rnd((C,), 4).abs())
    x_cpu = (wide.chunk(2, 1)[1] if sliced else wide).float()
    ref = F.conv2d(x_cpu, conv.weight.detach().bfloat16().float(), padding=1)          # fp32, same bf16 operands
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
    conv, bn = conv.cuda(), bn.cuda().train()
    wd = wide.cuda().contiguous(memory_format=torch.channels_last)
    x = wd.chunk(2, 1)[1] if sliced else wd
    assert ops.conv3x3_cl_ok(x, conv)
    y, mr = ops.conv3x3_cl_stats(x, conv, bn)
    assert y.dtype == torch.bfloat16 and y.shape == (B, C, H, W) and y.is_contiguous(memory_format=torch.channels_last)
    scale = float(ref.abs().max())
    assert_close(y.float(), ref, 2 ** -7, 2e-3 * scale, 'conv output')
    yd = y.float().cpu().double()
    mean, var = yd.mean((0, 2, 3)), yd.var((0, 2, 3), unbiased=False)
    assert_close(mr[:, 0], mean, 1e-4, 1e-5 * scale, 'batch mean')
    assert_close(mr[:, 1], (var + 1e-3).rsqrt(), 1e-4, 0, 'batch rstd')
    n = B * H * W
    assert_close(bn.running_mean, 0.97 * rm0.double() + 0.03 * mean, 1e-5, 1e-6, 'running_mean')
    assert_close(bn.running_var, 0.97 * rv0.double() + 0.03 * var * n / (n - 1), 1e-4, 1e-6, 'running_var')
    assert int(bn.num_batches_tracked) == 1


@pytest.mark.parametrize('B,L,hd,nc,nq,n_dec,dup', [(2, 340, 128, 10, 16, 3, False), (3, 1000, 512, 10, 50, 1, False), (1, 84, 256, 4, 8, 0, False),
                                                    (2, 340, 128, 10, 16, 1, True)])
def test_enc_select_row_sparse_backward_matches_the_dense_graph(ops, B, L, hd, nc, nq, n_dec, dup):
    """ops.enc_select (the MEH query selection as one autograd node whose backward runs on the picked rows only, reference head.py:1205-1245)
    against the same chain as separate ops (linear_bf16_zero_rows -> LayerNorm kernel -> score head -> top-k -> two gathers, ops.fanout for the
    decoder handles) and against a float64 autograd graph of the same bf16 operands: outputs bit-equal to the separate ops (same kernels), every
    gradient within bf16 rounding of the dense one and at least as close to the float64 graph."""
    import torch.nn as nn
    torch.manual_seed(B * L)
    lin, norm, head = nn.Linear(hd, hd).cuda(), nn.LayerNorm(hd).cuda(), nn.Linear(hd, nc).cuda()
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5)
        norm.bias.uniform_(-0.2, 0.2)
    invalid = torch.tensor(sorted({0, 3, L // 2, L - 1}), device='cuda')
    x0 = torch.randn(B, L, hd, device='cuda').bfloat16()
    cot_f, cot_s = torch.randn(B, nq, hd, device='cuda'), torch.randn(B, nq, nc, device='cuda')
    cot_d = [torch.randn(B, L, hd, device='cuda').bfloat16() for _ in range(n_dec)]
    params = [lin.weight, lin.bias, norm.weight, norm.bias, head.weight, head.bias]
    bi = torch.arange(B, device='cuda').unsqueeze(-1)

    def grads(loss, x):
        for p in params:
            p.grad = None
        loss.backward()
        return [x.grad.float()] + [p.grad.clone() for p in params]

    # (a) separate ops
    xa = x0.clone().requires_grad_()
    f_enc, *hs = ops.fanout(xa, 1 + n_dec)
    mem = ops.layer_norm(ops.linear_bf16_zero_rows(f_enc, lin.weight, lin.bias, invalid), norm.weight, norm.bias, norm.eps)
    sc = F.linear(mem, head.weight.bfloat16(), head.bias.bfloat16())
    top = torch.topk(sc.max(-1).values, nq, dim=1).indices
    if dup:                      # injected picks (fixed_topk) that name a row twice: its gradient is the sum of both uses
        top = top.clone()
        top[:, 1] = top[:, 0]
    fa, sa = mem[bi, top], sc[bi, top]
    ga = grads((fa.float() * cot_f).sum() + (sa.float() * cot_s).sum() + sum((h.float() * c.float()).sum() for h, c in zip(hs, cot_d)), xa)
    # (b) one node
    xb = x0.clone().requires_grad_()
    assert ops.enc_select_ok(xb, lin, norm, head)
    fb, sb, tb, hb = ops.enc_select(xb, lin, norm, head, invalid, nq, top if dup else None, n_dec)
    assert torch.equal(tb, top) and torch.equal(fb, fa) and torch.equal(sb, sa) and len(hb) == n_dec and all(torch.equal(h, x0) for h in hb)
    assert not tb.requires_grad
    gb = grads((fb.float() * cot_f).sum() + (sb.float() * cot_s).sum() + sum((h.float() * c.float()).sum() for h, c in zip(hb, cot_d)), xb)
    # (c) float64 graph of the same operands (weights as the forward rounds them), the same picks
    xc = x0.double().requires_grad_()
    pc = [p.detach().clone().double().requires_grad_() for p in params]
    w64, ws64 = (p.detach().bfloat16().double() + (p - p.detach()) for p in (pc[0], pc[4]))   # rounded value, straight-through gradient
    valid = torch.ones(L, 1, device='cuda', dtype=torch.float64)
    valid[invalid] = 0
    y = F.linear(xc * valid, w64, pc[1])
    m64 = F.layer_norm(y, (hd,), pc[2], pc[3], norm.eps)
    s64 = F.linear(m64, ws64, pc[5])
    ((m64[bi, top] * cot_f.double()).sum() + (s64[bi, top] * cot_s.double()).sum() + sum((xc * c.double()).sum() for c in cot_d)).backward()
    gc = [xc.grad.float()] + [p.grad.float() for p in pc]
    names = ['x', 'lin.weight', 'lin.bias', 'norm.weight', 'norm.bias', 'score.weight', 'score.bias']
    for n, a, b, c in zip(names, ga, gb, gc):
        scale = float(c.abs().max()) + 1e-12
        ea, eb = float((a.float() - c).abs().max()) / scale, float((b.float() - c).abs().max()) / scale
        assert eb <= max(1.25 * ea, 1e-2), (n, ea, eb)       # the row-wise fp32 backward is no further from the float64 graph than the dense bf16 one
        assert eb <= 3e-2, (n, ea, eb)
    # rows nobody picked get exactly the decoder handles' gradient; the invalid anchors' rows never get any from this branch
    want = sum(c.float() for c in cot_d) if cot_d else torch.zeros_like(x0, dtype=torch.float32)
    picked = torch.zeros(B, L, dtype=torch.bool, device='cuda')
    picked.index_put_((bi, top), torch.ones((), dtype=torch.bool, device='cuda'))
    assert torch.equal(gb[0][~picked], want.bfloat16().float()[~picked]) or n_dec > 1
    if n_dec <= 1:
        inv_picked = picked[:, invalid]
        assert torch.equal(gb[0][:, invalid][inv_picked], want.bfloat16().float()[:, invalid][inv_picked])


@pytest.mark.parametrize('rows,D,n', [(11, 256, 3072), (81, 512, 500), (2, 8, 1)])
def test_embed_rows_backward_is_the_index_backward(ops, rows, D, n):
    """ops.embed_rows (the denoising queries' class-embedding lookup, reference models/utils/ops.py:215-216): weight[idx] with the weight gradient as
    onehot^T @ g instead of torch's sorted index_put over thousands of duplicates - same values as the plain index, same gradient, same bits twice."""
    torch.manual_seed(rows + n)
    w0 = torch.randn(rows, D, device='cuda')
    idx = torch.randint(0, rows, (n,), device='cuda')
    cot = torch.randn(n, D, device='cuda')
    wa = w0.clone().requires_grad_()
    (wa[idx] * cot).sum().backward()
    grads = []
    for _ in range(2):
        wb = w0.clone().requires_grad_()
        out = ops.embed_rows(wb, idx)
        assert torch.equal(out, w0[idx])
        (out * cot).sum().backward()
        grads.append(wb.grad.clone())
    assert torch.equal(grads[0], grads[1])
    assert_close(grads[0], wa.grad, 1e-5, 1e-5 * float(wa.grad.abs().max()) + 1e-6, 'embed_rows grad')
    assert ops.embed_rows(w0, idx).grad_fn is None and torch.equal(ops.embed_rows(w0, idx), w0[idx])      # no gradient wanted: the plain index


@pytest.mark.parametrize('B,Q,M,D,shapes,P', [(2, 37, 8, 64, [(12, 10), (6, 5), (3, 3)], 4), (1, 292, 8, 64, [(20, 20), (10, 10), (5, 5)], 4)])
def test_value_proj_msda_pair_gives_the_bias_gradient_from_query_sized_operands(ops, B, Q, M, D, shapes, P):
    """ops.value_proj_msda (MSDeformAttn.value_proj + the sampling core as one node, reference transformer.py:273-311): the same kernels as the two
    separate nodes - output, d/dx, d/dW, d/dloc, d/daw bit-equal - and the bias gradient as sum_{b,q} gout * colw (ABI 34: the on-map weight of every
    (image, query, head), emitted by tamtr_msdeform_attn_bwd_sorted) instead of a pass over d(value); colw against its definition computed in torch,
    sampling locations reaching over the map's edges."""
    import torch.nn as nn
    torch.manual_seed(Q)
    L, N, nl = sum(h * w for h, w in shapes), M * D, len(shapes)
    lin = nn.Linear(N, N).cuda()
    x0 = torch.randn(B, L, N, device='cuda').bfloat16()
    loc0 = torch.rand(B, Q, M, nl, P, 2, device='cuda') * 1.3 - 0.15
    aw0 = torch.softmax(torch.randn(B, Q, M, nl * P, device='cuda'), -1).view(B, Q, M, nl, P)
    cot = torch.randn(B, Q, N, device='cuda').bfloat16()
    assert ops.value_proj_msda_ok(x0, lin, M, Q, P)

    def run(paired):
        x, loc, aw = x0.clone().requires_grad_(), loc0.clone().requires_grad_(), aw0.clone().requires_grad_()
        lin.weight.grad = lin.bias.grad = None
        if paired:
            out = ops.value_proj_msda(x, lin, M, shapes, loc, aw)
        else:
            out = ops.ms_deform_attn_core(ops.linear_bf16(x, lin.weight, lin.bias).view(B, L, M, D), shapes, loc, aw)
        (out.float() * cot.float()).sum().backward()
        return out.detach(), x.grad, lin.weight.grad.clone(), lin.bias.grad.clone(), loc.grad, aw.grad

    a, b = run(False), run(True)
    for name, u, v in zip(('out', 'dx', 'dW', 'dloc', 'daw'), (a[0], a[1], a[2], a[4], a[5]), (b[0], b[1], b[2], b[4], b[5])):
        assert torch.equal(u, v), name
    # colw by its definition, and the bias gradient from it
    colw = torch.zeros(B, Q, M, device='cuda')
    for l, (H, W) in enumerate(shapes):
        x, y = loc0[..., l, :, 0] * W - 0.5, loc0[..., l, :, 1] * H - 0.5
        xf, yf = x.floor(), y.floor()
        fx, fy = x - xf, y - yf
        sx = ((xf >= 0) & (xf < W)).float() * (1 - fx) + ((xf + 1 >= 0) & (xf + 1 < W)).float() * fx
        sy = ((yf >= 0) & (yf < H)).float() * (1 - fy) + ((yf + 1 >= 0) & (yf + 1 < H)).float() * fy
        colw += (aw0[..., l, :] * sx * sy).sum(-1)
    assert float(colw.min()) < 0.999 and float(colw.max()) <= 1.0 + 1e-5          # some weight does fall off the map in this draw
    want = (cot.view(B * Q, M, D).double() * colw.view(B * Q, M, 1).double()).sum(0).view(N)
    scale = float(want.abs().max())
    assert float((b[3].double() - want).abs().max()) <= 1e-5 * scale + 1e-6, 'db from colw'
    assert float((a[3].double() - want).abs().max()) <= 2e-2 * scale, 'the column sums of the bf16 d(value) agree with it to bf16 rounding'
