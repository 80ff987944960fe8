#!/usr/bin/env python3
"""Micro-benchmarks of the individual HIP kernels at the BASELINE shapes (640^2, bs 16).  Prints one line per kernel:
avg ms over interleaved rounds, algorithmic GB/s or TFLOP/s.   python tools/bench_kernels.py [scan|gemm|gate|msda|attn|all]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tamtr_amd.ops as ops  # noqa: E402


def timeit(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return sum(ms) / len(ms), ms[0]


def bench_scan():
    big = os.environ.get('TAMTR_BENCH_PX') == '1280'   # configs[4]: 1280 px, 8 images
    for lvl, (d_inner, L, R) in enumerate([(256, 102400, 8), (512, 25600, 16), (1024, 6400, 32)] if big else [(256, 25600, 8), (512, 6400, 16), (1024, 1600, 32)]):
        B, K, N = (8 if big else 16), 4, 16
        g = torch.Generator(device='cuda').manual_seed(0)
        side = int(L ** 0.5)
        u2 = torch.randn(B, 2, d_inner, L, device='cuda', generator=g)
        dtr = torch.randn(B, K, R, L, device='cuda', generator=g)
        Wdt = torch.randn(K * d_inner, R, device='cuda', generator=g) * R ** -0.5
        A = -torch.exp(torch.randn(K * d_inner, N, device='cuda', generator=g) * 0.3)
        Bm = torch.randn(B, K, N, L, device='cuda', generator=g)
        Cm = torch.randn(B, K, N, L, device='cuda', generator=g)
        D = torch.randn(K * d_inner, device='cuda', generator=g)
        bias = torch.randn(K * d_inner, device='cuda', generator=g) - 3
        ins = [t.requires_grad_() for t in (u2, dtr, Wdt, A, Bm, Cm, D, bias)]
        y = ops.selective_scan_cross(*ins)
        gy = torch.randn_like(y)
        f_avg, f_min = timeit(lambda: ops.selective_scan_cross(*ins))
        def fb():
            yy = ops.selective_scan_cross(*ins)
            torch.autograd.grad(yy, ins, gy)
        fb_avg, fb_min = timeit(fb)
        byt = y.numel() * 4
        print(f'scan (cross layout, fused dt proj) level {lvl} d_inner={d_inner} L={L} R={R}: fwd {f_avg:.2f} ms '
              f'({1.5 * byt / f_avg / 1e6:.0f} GB/s algorithmic: u2 + y), fwd+bwd {fb_avg:.2f} ms -> bwd {fb_avg - f_avg:.2f} ms')


def bench_gemm():
    M, N, K = 16 * 33600, 512, 512
    x = torch.randn(M, K, device='cuda').bfloat16()
    w = (torch.randn(N, K, device='cuda') * K ** -0.5)
    b = torch.randn(N, device='cuda')
    w16 = w.bfloat16()
    a, mn = timeit(lambda: ops.linear_bf16(x, w, b), n=10)
    ops.KERNEL_EVENTS['tamtr_linear_bf16'] = []
    for _ in range(20):
        ops.linear_bf16(x, w, b)
    torch.cuda.synchronize()
    ev = ops.KERNEL_EVENTS.pop('tamtr_linear_bf16')
    ks = sorted(e0.elapsed_time(e1) for e0, e1, _ in ev)
    ka = sum(ks) / len(ks)
    print(f'linear_bf16 M={M}: op {a * 1e3:.0f} us avg; kernel alone {ka * 1e3:.0f} us avg / {ks[0] * 1e3:.0f} us min = '
          f'{2 * M * N * K / ka / 1e9:.0f} TFLOP/s, {(M * K + M * N) * 2 / ka / 1e6:.0f} GB/s algorithmic')
    a, mn = timeit(lambda: torch.nn.functional.linear(x, w16, b.bfloat16()), n=10)
    print(f'torch/hipBLASLt same shape: {a * 1e3:.0f} us avg, {2 * M * N * K / a / 1e9:.0f} TFLOP/s')
    # what the same bytes cost as a pure stream (X read once, Y written once - the GEMM's algorithmic traffic): a device copy of X
    y = torch.empty_like(x)
    a, mn = timeit(lambda: y.copy_(x), n=20, warm=3)
    print(f'copy of X into a buffer of its size (the same 1.10 GB: {M * K * 2 / 1e6:.0f} MB read + {M * N * 2 / 1e6:.0f} MB written): '
          f'{a * 1e3:.0f} us avg / {mn * 1e3:.0f} min = {(M * K + M * N) * 2 / a / 1e6:.0f} GB/s')


def bench_gate():
    for (C, nh, H) in [(256, 8, 40), (128, 4, 80), (64, 2, 160)]:
        for dt in (torch.float32, torch.bfloat16):
            x = torch.randn(16, C, H, H, device='cuda').to(dt)
            v = torch.randn(16, C, H, H, device='cuda').to(dt)
            gk = torch.randn(16, 10, C, device='cuda')
            bias = torch.zeros(nh, device='cuda')
            a, mn = timeit(lambda: ops.maxsigmoid_gate(x, gk, bias, v, nh), n=10)
            byt = 3 * x.numel() * x.element_size()
            print(f'gate C={C} {H}x{H} {str(dt)[6:]}: {a * 1e3:.0f} us, {byt / a / 1e6:.0f} GB/s (x + v + out)')


def bench_gatecl():
    """The form bench.py runs at the five TIAGELAN sites (three shapes): channels-last forward-only gate, e = a channel slice of the cv1
    output, v = the RAW proj_conv output with its BatchNorm applied in the load (ops.maxsigmoid_gate_cl); next to it the NCHW kernel on
    packed planes of the same sizes.  Algorithmic bytes: e + v + out."""
    import torch.nn as nn
    big = os.environ.get('TAMTR_BENCH_PX') == '1280'
    for (C, nh, H) in ([(256, 8, 80), (128, 4, 160), (64, 2, 320)] if big else [(256, 8, 40), (128, 4, 80), (64, 2, 160)]):
        B = 8 if big else 16
        wide = torch.randn(B, 2 * C, H, H, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
        e = wide.chunk(2, 1)[1]
        v = torch.randn(B, C, H, H, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
        gk = torch.randn(B, 10, C, device='cuda')
        bias = torch.zeros(nh, device='cuda')
        bn = nn.BatchNorm2d(C, eps=1e-3, momentum=0.03).cuda().train()
        st = torch.stack([torch.randn(C, device='cuda') * 0.1, torch.rand(C, device='cuda') + 0.5], 1).contiguous()
        a, mn = timeit(lambda: ops.maxsigmoid_gate_cl(e, gk, bias, v, st, bn, nh), n=20, warm=3)
        xn, vn = e.contiguous(), v.contiguous()
        a2, mn2 = timeit(lambda: ops.maxsigmoid_gate(xn, gk, bias, vn, nh), n=20, warm=3)
        byt = 3 * v.numel() * 2
        print(f'gate_cl C={C} {H}x{H} x{B} bf16: {a * 1e3:.1f} us avg / {mn * 1e3:.1f} min, {byt / mn / 1e6:.0f} GB/s (e + v + out = {byt / 1e6:.0f} MB); '
              f'NCHW gate_fwd on packed planes {a2 * 1e3:.1f} us avg / {mn2 * 1e3:.1f} min')


def bench_msda():
    B, Q, M, D = 16, 292, 8, 64
    shapes = [(160, 160), (80, 80), (40, 40)]
    L = sum(h * w for h, w in shapes)
    for dt in (torch.float32, torch.bfloat16):
        v = torch.randn(B, L, M, D, device='cuda').to(dt).requires_grad_()
        loc = (0.5 + 0.15 * torch.randn(B, Q, M, 3, 4, 2, device='cuda')).requires_grad_()
        aw = torch.softmax(torch.randn(B, Q, M, 12, device='cuda'), -1).view(B, Q, M, 3, 4).requires_grad_()
        a, _ = timeit(lambda: ops.ms_deform_attn_core(v, shapes, loc, aw), n=10)
        out = ops.ms_deform_attn_core(v, shapes, loc, aw)
        go = torch.randn_like(out)
        def fb():
            o = ops.ms_deform_attn_core(v, shapes, loc, aw)
            torch.autograd.grad(o, [v, loc, aw], go)
        ab, _ = timeit(fb, n=10)
        gathered = B * Q * M * 12 * 4 * D * v.element_size()
        print(f'msda {str(dt)[6:]}: fwd {a * 1e3:.0f} us ({gathered / a / 1e6:.0f} GB/s gathered), fwd+bwd {ab * 1e3:.0f} us')


def bench_attn():
    B, Q, nh, dh = 16, 292, 8, 64
    for dt in (torch.float32, torch.bfloat16):
        p = torch.randn(B, Q, 3 * nh * dh, device='cuda').to(dt).requires_grad_()
        C = nh * dh
        mask = torch.zeros(Q, Q, dtype=torch.bool, device='cuda')
        mask[192:, :192] = True
        f = lambda: ops.self_attention(p[..., :C], p[..., C:2 * C], p[..., 2 * C:], nh, mask)
        a, _ = timeit(f, n=10)
        go = torch.randn(B, Q, C, device='cuda').to(dt)
        def fb():
            torch.autograd.grad(f(), [p], go)
        ab, _ = timeit(fb, n=10)
        print(f'self-attn {str(dt)[6:]}: fwd {a * 1e3:.0f} us, fwd+bwd {ab * 1e3:.0f} us ({4 * B * nh * Q * Q * dh / a / 1e9:.1f} TFLOP/s fwd)')


def bench_cpam():
    for (C, H) in [(512, 40), (256, 80), (128, 160)]:
        for dt in (torch.float32, torch.bfloat16):
            x = torch.randn(16, C, H, H, device='cuda').to(dt).requires_grad_()
            p, idx = torch.nn.functional.max_pool2d(x.detach(), 3, 2, 1, return_indices=True)
            out, s2 = torch.empty_like(x), torch.empty(16, 8, H, H, device='cuda')
            arg = torch.empty(16, 8, H, H, device='cuda', dtype=torch.int32)
            code = 0 if dt == torch.float32 else 1
            from tamtr_amd._lib import call, ptr, stream_ptr
            k, _ = timeit(lambda: call('tamtr_cpam_fwd', ptr(x.detach()), ptr(p), ptr(out), ptr(s2), ptr(arg), 16, C, H, H, code, stream_ptr()), n=10)
            a, _ = timeit(lambda: ops.cpam(x), n=10)
            go = torch.randn_like(x)
            ab, _ = timeit(lambda: torch.autograd.grad(ops.cpam(x), [x], go), n=10)
            byt = (2 * x.numel() + p.numel()) * x.element_size()
            print(f'cpam C={C} {H}x{H} {str(dt)[6:]}: gate kernel {k * 1e3:.0f} us = {byt / k / 1e6:.0f} GB/s (x + p + out); '
                  f'op with max-pool {a * 1e3:.0f} us, fwd+bwd {ab * 1e3:.0f} us')


def bench_dwconv():
    for (D, H) in [(256, 160), (512, 80), (1024, 40)]:
        xz = torch.randn(16, H, H, 2 * D, device='cuda').bfloat16().requires_grad_()
        w = (torch.randn(D, 1, 3, 3, device='cuda') * 0.3).requires_grad_()
        b = torch.zeros(D, device='cuda').requires_grad_()
        a, _ = timeit(lambda: ops.dwconv_silu_cross(xz, w, b, D), n=10)
        u2 = ops.dwconv_silu_cross(xz, w, b, D)
        go = torch.randn_like(u2)
        ab, _ = timeit(lambda: torch.autograd.grad(ops.dwconv_silu_cross(xz, w, b, D), [xz, w, b], go), n=10)
        byt = 16 * H * H * D * (2 + 8)
        print(f'dwconv+silu+cross d_inner={D} {H}x{H} bf16 in: fwd {a * 1e3:.0f} us = {byt / a / 1e6:.0f} GB/s (xi + two fp32 planes), '
              f'fwd+bwd {ab * 1e3:.0f} us')


def bench_lsap():
    g = torch.Generator(device='cuda').manual_seed(0)
    for nq, groups in [(100, [8] * 16), (300, [120] * 16)]:
        cost = torch.randn(len(groups), nq, sum(groups), device='cuda', generator=g)
        a, mn = timeit(lambda: ops.lsap_assign(cost, groups), n=20)
        print(f'lsap nq={nq} boxes/img={groups[0]} x {len(groups)} images: {a * 1e3:.0f} us avg, {mn * 1e3:.0f} us min (one launch, no host round trip)')


def bench_planes():
    """Every kernel of the SS2D chain with its big planes in fp32 and in bf16 (include/tamtr_hip.h "bf16 PLANES") at the three level shapes:
    time and algorithmic bytes per second of the planes + the other operands it reads / writes."""
    import tamtr_amd._lib as L_
    from tamtr_amd._lib import call, ptr, stream_ptr
    B, K, N = 16, 4, 16
    sp = stream_ptr()
    for D, H, R in [(256, 160, 8), (512, 80, 16), (1024, 40, 32)]:
        W, L, C = H, H * H, R + 32
        rn = lambda *sh: torch.randn(*sh, device='cuda')   # noqa: E731
        xz = rn(B, H, W, 2 * D).bfloat16()
        cw, cb = rn(D, 9) * 0.3, rn(D) * 0.1
        wx = rn(4, C, D) * D ** -0.5
        wcat = ops.xproj_pack_weight(wx)
        wT = ops.xproj_pack_weight_t(wcat, C)
        Wdt, A, Dv, db = rn(K * D, R) * R ** -0.5, -torch.exp(rn(K * D, N) * 0.5), rn(K * D), rn(K * D) * 0.5 - 1.0
        chunk = L_.lib().tamtr_selective_scan_chunk()
        hst = torch.empty(B, K * D, (L + chunk - 1) // chunk, N, device='cuda')
        dtr, Bs, Cs = (torch.empty(B, 4, n, L, device='cuda') for n in (R, N, N))
        ymT, gm = torch.empty(B, L, D, device='cuda'), rn(B, L, D)
        gdelta = torch.empty(B, K * D, L, device='cuda', dtype=torch.bfloat16)
        gdtr, gB, gC = torch.empty_like(dtr), torch.empty_like(Bs), torch.empty_like(Cs)
        grow = torch.empty(B, K * D, L_.lib().tamtr_selective_scan_row_sums(), device='cuda')
        ws = torch.empty(2 * L_.lib().tamtr_selective_scan_bwd_slabs(D) * Bs.numel(), device='cuda')
        nsl = L_.lib().tamtr_xproj_dw_slices(L)
        part = torch.empty(B * nsl, 2, 2 * C, D, device='cuda')
        gxz = torch.zeros_like(xz)
        wsd = torch.empty(B, L_.lib().tamtr_dwconv_tiles(H, W), D, 10, device='cuda')
        pl = B * D * L                       # elements of one [B, D, L] plane
        small = (R + 2 * N) * 4 * B * L * 4  # dtr + Bs + Cs (or their gradients), f32
        rows = []
        for pc, pdt in ((0, torch.float32), (1, torch.bfloat16)):
            e = 2 if pc else 4
            u2, y = torch.empty(B, 2, D, L, device='cuda', dtype=pdt), torch.empty(B, K, D, L, device='cuda', dtype=pdt)
            g2, gu, gu2 = torch.empty_like(u2), torch.empty(B, K * D, L, device='cuda', dtype=pdt), torch.empty_like(u2)
            steps = [
                ('dwconv_cross_fwd', lambda: call('tamtr_dwconv_silu_cross_fwd', ptr(xz), 2 * D, ptr(cw), ptr(cb), ptr(u2), B, D, H, W, 1, pc, sp), pl * 2 + 2 * pl * e),
                ('xproj_fwd', lambda: call('tamtr_xproj_fwd', ptr(u2), ptr(wcat), ptr(dtr), ptr(Bs), ptr(Cs), B, D, L, R, pc, sp), 2 * pl * e + small),
                ('selscan_fwd', lambda: call('tamtr_selective_scan_dtproj_fwd', ptr(u2), ptr(dtr), ptr(Wdt), ptr(A), ptr(Bs), ptr(Cs), ptr(Dv), ptr(db), ptr(y), ptr(hst), B, K, D, N, R, L, 1, pc, sp), 8 * pl * e + small),
                ('cross_merge_fwd', lambda: call('tamtr_cross_merge_fwd', ptr(y), ptr(ymT), B, D, H, W, pc, sp), 4 * pl * e + pl * 4),
                ('cross_merge_bwd', lambda: call('tamtr_cross_merge_bwd', ptr(gm), ptr(g2), B, D, H, W, pc, sp), pl * 4 + 2 * pl * e),
                ('selscan_bwd', lambda: call('tamtr_selective_scan_dtproj_bwd', ptr(g2), ptr(u2), ptr(dtr), ptr(Wdt), ptr(A), ptr(Bs), ptr(Cs), ptr(Dv), ptr(db), ptr(hst), ptr(gu), ptr(gdelta),
                                             ptr(gdtr), ptr(grow), ptr(gB), ptr(gC), ptr(ws), B, K, D, N, R, L, 3, 1 | (2 * pc), sp), 12 * pl * e + 8 * pl * 2 + 2 * small),
                ('xproj_bwd_dx', lambda: call('tamtr_xproj_bwd_dx', ptr(gu), ptr(gdtr), ptr(gB), ptr(gC), ptr(wT), ptr(gu2), B, D, L, R, pc, sp), 6 * pl * e + small),
                ('xproj_bwd_dw', lambda: call('tamtr_xproj_bwd_dw', ptr(u2), ptr(gdtr), ptr(gB), ptr(gC), ptr(part), B, D, L, R, pc, sp), 2 * pl * e + small),
                ('dwconv_cross_bwd', lambda: call('tamtr_dwconv_silu_cross_bwd', ptr(gu2), ptr(xz), 2 * D, ptr(cw), ptr(cb), ptr(gxz), 2 * D, ptr(wsd), B, D, H, W, 1, pc, sp), 2 * pl * e + 2 * pl * 2),
            ]
            # realistic values in the planes the later kernels read
            u2.copy_(rn(B, 2, D, L)); g2.copy_(rn(B, 2, D, L)); gu.copy_(rn(B, K * D, L)); gu2.copy_(rn(B, 2, D, L))
            for name, fn, byt in steps:
                a, mn = timeit(fn, n=10, warm=2)
                rows.append((name, pc, mn, byt))
            del u2, y, g2, gu, gu2
        tot = [0.0, 0.0]
        for i in range(len(rows) // 2):
            (name, _, t32, b32), (_, _, t16, b16) = rows[i], rows[i + len(rows) // 2]
            tot[0] += t32; tot[1] += t16
            print(f'planes d_inner={D} L={L}: {name:18s} f32 {t32 * 1e3:7.0f} us ({b32 / t32 / 1e9:5.2f} TB/s)   bf16 {t16 * 1e3:7.0f} us ({b16 / t16 / 1e9:5.2f} TB/s)   {t16 / t32:.2f}x')
        print(f'planes d_inner={D} L={L}: chain total      f32 {tot[0] * 1e3:7.0f} us   bf16 {tot[1] * 1e3:7.0f} us   saves {(tot[0] - tot[1]) * 1e3:.0f} us')


def bench_projconv():
    """The gate's value branch at the five TIAGELAN sites (three shapes, 16 images, bf16 channels-last slice of the cv1 output): the MFMA
    kernel with the statistics in its epilogue against the library convolution + tamtr_bncl_stats."""
    import torch.nn as nn
    for C, S in [(64, 160), (128, 80), (256, 40)]:
        wide = torch.randn(16, 2 * C, S, S, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
        x = wide.chunk(2, 1)[1]
        conv = nn.Conv2d(C, C, 3, 1, 1, bias=False).cuda()
        bn = nn.BatchNorm2d(C, eps=1e-3, momentum=0.03).cuda().train()
        flop = 2 * 9 * C * C * 16 * S * S
        own, own_min = timeit(lambda: ops.conv3x3_cl_stats(x, conv, bn), n=20, warm=3)
        w16 = conv.weight.detach().bfloat16().contiguous(memory_format=torch.channels_last)
        def lib():
            v = torch.nn.functional.conv2d(x, w16, padding=1)
            return ops.bn_stats_cl(v.permute(0, 2, 3, 1).reshape(-1, C), bn)
        torch.backends.cudnn.benchmark = True
        libt, lib_min = timeit(lib, n=20, warm=3)
        print(f'proj_conv {C} ch x {S}^2 x 16: own conv + stats {own * 1e3:.0f} us avg / {own_min * 1e3:.0f} min '
              f'({flop / own_min / 1e9:.0f} TFLOP/s), library conv + stats kernel {libt * 1e3:.0f} us avg / {lib_min * 1e3:.0f} min')


if __name__ == '__main__':
    which = sys.argv[1:] or ['all']
    for name in ('scan', 'gemm', 'gate', 'gatecl', 'msda', 'attn', 'cpam', 'dwconv', 'lsap', 'projconv', 'planes'):
        if name in which or 'all' in which:
            globals()['bench_' + name]()
