#!/usr/bin/env python3
"""Is the step host-bound?  Issue time (Python returns, GPU still running) vs synchronised time, forward and backward."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch
from tamtr_amd.model import RTDETRDetectionWorldModel
from tamtr_amd.tuning import use_tuned_convolutions
use_tuned_convolutions('shipped')
torch.manual_seed(0)
model = RTDETRDetectionWorldModel(nc=10).cuda().train()
model.autocast_dtype = torch.bfloat16
batch = synth_batch(16, 640, 1, 'cuda')
if os.environ.get('STATIC_PART', 'graph') == 'graph':
    model.capture_static_part(batch['img'], batch['txt_feats'])
for _ in range(3):
    model.zero_grad(set_to_none=True)
    l, _ = model(batch); l.backward()
torch.cuda.synchronize()
for it in range(3):
    model.zero_grad(set_to_none=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    l, _ = model(batch)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    l.backward()
    t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
    print(f'fwd: issue {1e3*(t1-t0):.1f} ms, +sync {1e3*(t2-t1):.1f} ms | bwd: issue {1e3*(t3-t2):.1f} ms, +sync {1e3*(t4-t3):.1f} ms | total {1e3*(t4-t0):.1f}')
import torch.utils.cpp_extension  # noqa
print('threads', torch.get_num_threads(), 'cpus', len(os.sched_getaffinity(0)))
