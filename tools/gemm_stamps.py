import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tamtr_amd.ops as ops
M, N, K = 16 * 33600, 512, 512
x = torch.randn(M, K, device='cuda').bfloat16(); w = (torch.randn(N, K, device='cuda') * K ** -0.5)
for _ in range(3):
    y = ops.linear_bf16(x, w, None)
torch.cuda.synchronize()
d = y.view(torch.int64).flatten()[:256 * 8 * 8].view(256, 8, 8).double().cpu()
tot = d[..., 4]
print('per-wave kernel cycles (s_memtime ticks): mean %.0f  min %.0f max %.0f; steps %d' % (tot.mean(), tot.min(), tot.max(), int(d[0, 0, 5])))
for i, n in enumerate(['wait vmcnt+barrier', 'issue glds', 'frag reads + MFMA', 'epilogue']):
    print(f'  {n:<22s} {100 * (d[..., i] / tot).mean():5.1f}%   per step {d[..., i].mean() / d[0, 0, 5]:.0f} ticks')
print('  by wave (wait%):', [round(float(100 * (d[:, w, 0] / tot[:, w]).mean()), 1) for w in range(8)])
