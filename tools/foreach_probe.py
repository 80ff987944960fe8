import torch, time
ws=[torch.randn(64,64,3,3,device='cuda') for _ in range(170)]
outs=[torch.empty_like(w,dtype=torch.bfloat16) for w in ws]
outs32=[torch.empty_like(w) for w in ws]
def t(f,n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); a=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-a)/n*1e3
print('foreach fp32->bf16 ms', t(lambda: torch._foreach_copy_(outs, ws)))
print('foreach fp32->fp32 ms', t(lambda: torch._foreach_copy_(outs32, ws)))
print('loop fp32->bf16 ms', t(lambda: [o.copy_(w) for o,w in zip(outs,ws)]))
