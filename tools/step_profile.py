#!/usr/bin/env python3
"""One eager training step under torch.profiler: the heaviest ops by device time and by host time (what makes a mode slow).
TAMTR_DETERMINISTIC=1 python3 tools/step_profile.py   ->  the deterministic mode"""
import os, sys, time

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch
from tamtr_amd.model import RTDETRDetectionWorldModel
from tamtr_amd import tuning
from torch.profiler import profile, ProfilerActivity

if os.environ.get('TAMTR_DETERMINISTIC') == '1':
    print(tuning.use_deterministic_convolutions(search=False), flush=True)
B = int(os.environ.get('BS', 16))
torch.manual_seed(0)
model = RTDETRDetectionWorldModel(nc=10).cuda().train()
model.autocast_dtype = torch.bfloat16
batch = synth_batch(B, 640, 1, 'cuda')
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True)


def step(tag=None):
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    loss, _ = model(batch)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    loss.backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    opt.step()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    if tag:
        print(f'{tag}: forward+loss {t1 - t0:.3f} s, backward {t2 - t1:.3f} s, optimizer {t3 - t2:.3f} s', flush=True)


step('step 0 (kernel selection)')
step('step 1')
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step('profiled')
ev = [e for e in prof.events()]
by_dev = sorted(ev, key=lambda e: -e.device_time)[:14]
print('--- by device time')
for e in by_dev:
    print(f'{e.device_time / 1e3:10.2f} ms  cpu {e.cpu_time / 1e3:9.2f} ms  {e.name[:60]:60s} {str(e.input_shapes)[:110]}')
by_cpu = sorted([e for e in ev if e.name.startswith(('aten::', 'hip', 'Tamtr', '_'))], key=lambda e: -e.self_cpu_time_total)[:14]
print('--- by self host time')
for e in by_cpu:
    print(f'{e.self_cpu_time_total / 1e3:10.2f} ms  {e.name[:60]:60s} {str(e.input_shapes)[:110]}')
