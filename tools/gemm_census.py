#!/usr/bin/env python3
"""One eager training step under torch.profiler: every library GEMM call (aten::mm / addmm / bmm / baddbmm) with its operand shapes, device
time, algorithmic bytes and the time those bytes take at 5 TB/s - which library GEMMs are far from their byte time."""
import os, sys, collections

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch
from tamtr_amd.model import RTDETRDetectionWorldModel
from tamtr_amd import tuning
from torch.profiler import profile, ProfilerActivity

tuning.use_tuned_convolutions('shipped')
torch.manual_seed(0)
model = RTDETRDetectionWorldModel(nc=10).cuda().train()
model.autocast_dtype = torch.bfloat16
batch = synth_batch(16, 640, 1, 'cuda')
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True)


def step():
    opt.zero_grad(set_to_none=True)
    loss, _ = model(batch)
    loss.backward()
    opt.step()
    torch.cuda.synchronize()


step(); step()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
agg = collections.defaultdict(lambda: [0.0, 0])
for e in prof.events():
    if e.name in ('aten::mm', 'aten::addmm', 'aten::bmm', 'aten::baddbmm') and e.device_time > 0:
        k = (e.name, str(e.input_shapes))
        agg[k][0] += e.device_time
        agg[k][1] += 1
tot = 0
for (name, shp), (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    tot += t
    print(f'{t / 1e3:8.3f} ms  n={n:3d}  avg {t / n:8.1f} us  {name:14s} {shp[:150]}')
print(f'total {tot / 1e3:.2f} ms in {sum(v[1] for v in agg.values())} calls')
