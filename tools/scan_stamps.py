#!/usr/bin/env python3
"""Phase timing inside selscan_bwd_kernel (variant built with -DSCAN_STAMP): TAMTR_HIP_LIB=.../variants/libtamtr_stamp.so python3 tools/scan_stamps.py [level]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tamtr_amd.ops as ops
from tamtr_amd._lib import call, ptr, stream_ptr, lib
lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 0
Dk, L, R = [(256, 25600, 8), (512, 6400, 16), (1024, 1600, 32)][lvl]
B, K, N = 16, 4, 16
g = torch.Generator(device='cuda').manual_seed(0)
rn = lambda *s: torch.randn(*s, device='cuda', generator=g)
u2, dtr, Wdt = rn(B, 2, Dk, L), rn(B, K, R, L), rn(K * Dk, R) * R ** -0.5
A, Bm, Cm, D, bias = -torch.exp(rn(K * Dk, N) * 0.3), rn(B, K, N, L), rn(B, K, N, L), rn(K * Dk), rn(K * Dk) - 3
chunk = lib().tamtr_selective_scan_chunk(); nchunk = (L + chunk - 1) // chunk
y = torch.empty(B, K, Dk, L, device='cuda'); hstate = torch.empty(B, K * Dk, nchunk, N, device='cuda')
call('tamtr_selective_scan_dtproj_fwd', ptr(u2), ptr(dtr), ptr(Wdt), ptr(A), ptr(Bm), ptr(Cm), ptr(D), ptr(bias), ptr(y), ptr(hstate), B, K, Dk, N, R, L, 1, 0, stream_ptr())
gy = rn(B, K * Dk, L)
gu, gdelta, gdtr = torch.empty(B, K * Dk, L, device='cuda'), torch.empty(B, K * Dk, L, device='cuda'), torch.empty_like(dtr)
gW, gA, gB, gC, gD, gb = torch.zeros_like(Wdt), torch.zeros_like(A), torch.empty_like(Bm), torch.empty_like(Cm), torch.zeros_like(D), torch.zeros_like(bias)
ws = torch.empty(2 * lib().tamtr_selective_scan_bwd_slabs(Dk) * Bm.numel(), device='cuda')
for _ in range(2):
    call('tamtr_selective_scan_dtproj_bwd', ptr(gy), ptr(u2), ptr(dtr), ptr(Wdt), ptr(A), ptr(Bm), ptr(Cm), ptr(D), ptr(bias), ptr(hstate), ptr(gu), ptr(gdelta),
         ptr(gdtr), ptr(gW), ptr(gA), ptr(gB), ptr(gC), ptr(gD), ptr(gb), ptr(ws), B, K, Dk, N, R, L, 1, stream_ptr())
torch.cuda.synchronize()
st = gu.flatten()[:8].double().cpu() * 64
names = ['chunk-top barrier', 'tile staging + barrier', 'row prologue (next-row requests, dt projection, softplus)', 'the 16 states', 'row epilogue (stores, row sums)',
         'fold + barriers', 'slab stores', '-']
tot = float(st[:8].sum())
print(f'level {lvl}: wave 0 of workgroup (0,0): {tot:.0f} ticks total over {nchunk} chunks')
for n, v in zip(names, st[:8]):
    print(f'  {n:<58s} {100 * float(v) / tot:5.1f} %   {float(v) / nchunk:9.0f} ticks per chunk')

nwg = (Dk // 32) * B * K
w = gu.flatten()[16:16 + 4 * nwg].view(nwg, 4).double().cpu()
t0, t1, xcc = w[:, 0], w[:, 1], w[:, 2]
base = t0.min()
dur = ((t1 - t0) % 16777216.0) / 100.0   # us (100 MHz, 24-bit window)
start = (t0 - base) / 100.0
q = torch.quantile(dur, torch.tensor([0.0, 0.25, 0.5, 0.75, 0.9, 0.99, 1.0], dtype=torch.float64))
print(f'workgroups {nwg}: all started within {start.max():.0f} us; duration quantiles 0/25/50/75/90/99/100 %: ' + ' / '.join(f'{v:.0f}' for v in q) + ' us')
print('per XCC: count, median, max duration (us):', [(int((xcc == i).sum()), int(dur[xcc == i].median()), int(dur[xcc == i].max())) for i in range(8)])
nx = Dk // 32
idx = torch.arange(nwg)
kdir, bx = (idx // nx) % K, idx % nx
print('per direction k median/max:', [(int(dur[kdir == i].median()), int(dur[kdir == i].max())) for i in range(K)])
print('per row block (blockIdx.x) median/max:', [(int(dur[bx == i].median()), int(dur[bx == i].max())) for i in range(min(nx, 8))])
slow = torch.nonzero(dur > q[3] * 1.15).flatten()
print(f'{len(slow)} workgroups 15 % over the 75 % quantile; their XCCs: {sorted(set(int(xcc[i]) for i in slow))}')
