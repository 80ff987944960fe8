#!/usr/bin/env python3
"""Which torch ops (with which shapes) are behind the step's generic elementwise / copy kernels: one eager training step under
torch.profiler with shapes, grouped by (op, input shapes), sorted by device time."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch
from tamtr_amd.model import RTDETRDetectionWorldModel
from torch.profiler import profile, ProfilerActivity

torch.manual_seed(0)
model = RTDETRDetectionWorldModel(nc=10).cuda().train()
model.autocast_dtype = torch.bfloat16
batch = synth_batch(16, 640, 1, 'cuda')
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True)
def step():
    opt.zero_grad(set_to_none=True)
    loss, _ = model(batch); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
want = sys.argv[1:] or ['aten::copy_', 'aten::cat', 'aten::add', 'aten::add_', 'aten::mul', 'aten::sum', 'aten::fill_', 'aten::zero_', 'aten::silu', 'aten::_to_copy']
rows = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.name in want and e.device_time > 0:
        k = (e.name, str(e.input_shapes)[:150])
        rows[k][0] += 1; rows[k][1] += e.device_time
tot = collections.defaultdict(float)
for (n, _), (c, t) in rows.items(): tot[n] += t
for n, t in sorted(tot.items(), key=lambda x: -x[1]): print(f'{n:16s} {t/1e3:8.2f} ms')
for (n, s), (c, t) in sorted(rows.items(), key=lambda x: -x[1][1])[:70]:
    print(f'{t/1e3:7.3f} ms  n={c:3d}  {n:14s} {s}')
