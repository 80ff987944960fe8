#!/usr/bin/env python3
"""Which operations of one training step synchronise the host with the GPU?  torch.cuda.set_sync_debug_mode('warn') around one step of the
bench's model (static part kernel by kernel), warnings printed with the Python frame that issued them.
    python3 tools/micro/sync_probe.py            (TAMTR_ENC_SELECT=dense for the A/B)"""
import os, sys, traceback, warnings
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import synth_batch
from tamtr_amd.model import RTDETRDetectionWorldModel
from tamtr_amd.tuning import use_tuned_convolutions
from tamtr_amd.engine import FusedOptimStep, ModelEMA

use_tuned_convolutions('shipped')
torch.manual_seed(0)
dev = torch.device('cuda', 0)
model = RTDETRDetectionWorldModel(nc=10).to(dev).train()
model.autocast_dtype = torch.bfloat16
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-4, fused=True)
ema = ModelEMA(model)
stepper = FusedOptimStep.create(model, opt, ema, max_norm=0.1, shadows=True)
batch = synth_batch(16, 640, 1, dev)


def step():
    opt.zero_grad(set_to_none=True)
    loss, _ = model(batch)
    loss.backward()
    stepper.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
seen = []


def show(message, category, filename, lineno, file=None, line=None):
    frames = [f for f in traceback.extract_stack() if '/tam-tr_amd/' in f.filename or f.filename.endswith('bench.py')]
    where = ' <- '.join(f'{os.path.basename(f.filename)}:{f.lineno} {f.name}' for f in frames[-3:][::-1]) or f'{filename}:{lineno}'
    seen.append((str(message)[:90], where))


warnings.showwarning = show
warnings.simplefilter('always')
torch.cuda.set_sync_debug_mode('warn')
step()
torch.cuda.set_sync_debug_mode('default')
torch.cuda.synchronize()
print(f'{len(seen)} synchronising calls in one step (TAMTR_ENC_SELECT={os.environ.get("TAMTR_ENC_SELECT", "rows")}):')
for m, w in seen:
    print('  ', m, '|', w)
