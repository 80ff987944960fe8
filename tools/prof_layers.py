#!/usr/bin/env python3
"""Per-layer forward time (events around each top-level layer of the TAMTR graph, and around the parts of the MEH head)
plus total fwd / bwd / optimizer split at the BASELINE shape.  python tools/prof_layers.py [--dtype bf16|fp32]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch  # noqa: E402
from tamtr_amd.model import RTDETRDetectionWorldModel  # noqa: E402

dtype = 'bf16' if '--dtype' not in sys.argv else sys.argv[sys.argv.index('--dtype') + 1]
torch.manual_seed(0)
model = RTDETRDetectionWorldModel(nc=10).cuda().train()
model.autocast_dtype = torch.bfloat16 if dtype == 'bf16' else None
batch = synth_batch(16, 640, 1, 'cuda')
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True)
recs = {}


def wrap(mod, name):
    orig = mod.forward

    def fwd(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig(*a, **k)
        e1.record()
        recs.setdefault(name, []).append((e0, e1))
        return out
    mod.forward = fwd


for i, m in enumerate(model.model):
    wrap(m, f'{i:02d} {m.type}')
head = model.model[-1]
for i, b in enumerate(head.VSSBlocks):
    wrap(b, f'41.VSS{i}')
    wrap(b.op, f'41.VSS{i}.ss2d')
for i, l in enumerate(head.decoder.layers):
    wrap(l, f'41.dec{i}')
    wrap(l.cross_attn, f'41.dec{i}.cross')
    wrap(l.self_attn, f'41.dec{i}.self')
wrap(head.enc_output, '41.enc_output')


def step():
    opt.zero_grad(set_to_none=True)
    t = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    t[0].record()
    loss, _ = model(batch)
    t[1].record()
    loss.backward()
    t[2].record()
    torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.grad is not None], 0.1)
    opt.step()
    t[3].record()
    return t


for _ in range(2):
    step()
torch.cuda.synchronize()
recs.clear()
t0 = time.perf_counter()
ts = [step() for _ in range(3)]
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 3 * 1e3
f = sum(t[0].elapsed_time(t[1]) for t in ts) / 3
b = sum(t[1].elapsed_time(t[2]) for t in ts) / 3
o = sum(t[2].elapsed_time(t[3]) for t in ts) / 3
print(f'{dtype}: wall {wall:.1f} ms/step = fwd+loss {f:.1f} + bwd {b:.1f} + clip/AdamW {o:.1f}')
tot = 0
for k in sorted(recs):
    ms = sum(a.elapsed_time(bb) for a, bb in recs[k]) / 3
    if '.' not in k.split()[0]:
        tot += ms
    print(f'  {k:<28s} {ms:8.2f} ms')
print(f'  sum of top-level layers {tot:.1f} ms (rest of fwd = loss + matcher)')
