#!/usr/bin/env python3
"""Phase timing inside linear_bf16_wstat_kernel (variant built with -DWS_STAMP): TAMTR_HIP_LIB=.../variants/libtamtr_wsstamp.so python3 tools/gemm_stamps_ws.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tamtr_amd._lib import call, ptr, stream_ptr
M, N, K = 16 * 33600, 512, 512
x = torch.randn(M, K, device='cuda').bfloat16(); w = (torch.randn(N, K, device='cuda') * K ** -0.5).bfloat16(); b = torch.randn(N, device='cuda')
y = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
for _ in range(3):
    call('tamtr_linear_bf16', ptr(x), ptr(w), ptr(b), ptr(y), M, N, K, stream_ptr())
torch.cuda.synchronize()
d = y.view(-1)[:64].view(torch.int64).cpu()
names = ['vmcnt wait', 'barrier', 'waves 4-7: DMA issue', 'MFMA chain (33 MFMA + 32 ds_read)', 'convert / permlane / stores', 'waves 0-3: DMA issue']
for wv, off in (('wave 0 (early)', 0), ('wave 4 (late)', 8)):
    v = d[off:off + 7].tolist(); nb = v[6]; tot = sum(v[:6])
    print(f'{wv}: {nb} blocks, {tot / nb:.0f} ticks per block')
    for n, t in zip(names, v[:6]):
        print(f'   {n:<72s} {t / nb:8.0f} ticks per block  {100 * t / tot:5.1f} %')
