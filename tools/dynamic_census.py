#!/usr/bin/env python3
"""What would stand in the way of recording the label-dependent part of the step (denoising groups, query selection, decoder, heads,
12-term loss: forward and backward) as a third pair of HIP graphs under AQL packet capture?  One eager training step under torch.profiler
with stacks; for every memset and every torch multi-workgroup-capable reduction issued between the token memory and the parameter
gradients: the aten op, its input shapes and the innermost frame of this package.  Also counts the kernels of that part."""
import collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch
from tamtr_amd.model import RTDETRDetectionWorldModel
from tamtr_amd.tuning import use_tuned_convolutions
from torch.profiler import profile, ProfilerActivity, record_function

use_tuned_convolutions('shipped')
torch.manual_seed(0)
model = RTDETRDetectionWorldModel(nc=10).cuda().train()
model.autocast_dtype = torch.bfloat16
batch = synth_batch(16, 640, 1, 'cuda')
head = model.model[-1]
_decode = head.decode
feats_box = {}


def decode(feats, shapes, text, b=None):
    feats_box['f'] = feats
    with record_function('DYN:forward'):
        return _decode(feats, shapes, text, b)
head.decode = decode


def run():
    model.zero_grad(set_to_none=True)
    loss, _ = model(batch)
    f = feats_box['f']
    dyn = [p for n, p in head.named_parameters() if not n.startswith(('VSSBlocks', 'input_proj'))]
    with record_function('DYN:backward'):
        return torch.autograd.grad(loss, [f] + dyn, allow_unused=True)


for _ in range(3):
    run()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    run()
    torch.cuda.synchronize()
ev = prof.events()
wins = [(e.time_range.start, e.time_range.end, e.name) for e in ev if e.name.startswith('DYN:')]
# the loss is computed after decode() returns: extend the forward window to the start of the backward window
fw = [w for w in wins if w[2] == 'DYN:forward'][0]
bw = [w for w in wins if w[2] == 'DYN:backward'][0]
lo, hi = fw[0], bw[1]
rows = collections.defaultdict(lambda: [0, 0.0])
nk = collections.Counter()
for e in ev:
    ks = getattr(e, 'kernels', None) or []
    if e.device_type != torch.autograd.DeviceType.CPU or not ks or not (lo <= e.time_range.start <= hi):
        continue
    for k in ks:
        nm = k.name
        kind = 'memset' if ('emset' in nm or 'fillBuffer' in nm) else ('memcpy' if 'emcpy' in nm or 'copyBuffer' in nm else
                                                                       ('reduce' if ('reduce_kernel' in nm and 'at::native' in nm) else 'kernel'))
        nk[kind] += 1
        if kind in ('memset', 'memcpy', 'reduce'):
            frame = next((f for f in (e.stack or []) if 'tam-tr_amd' in f or 'tamtr_amd' in f), (e.stack or ['?'])[0] if e.stack else '?')
            rows[(kind, e.name, str(e.input_shapes)[:90], frame.strip()[-90:])][0] += 1
            rows[(kind, e.name, str(e.input_shapes)[:90], frame.strip()[-90:])][1] += k.duration
print('# label-dependent part of one eager step (decode + loss forward, their backward down to the token memory): kernels by kind:', dict(nk))
for (kind, op, shp, fr), (n, us) in sorted(rows.items(), key=lambda x: (x[0][0], -x[1][0])):
    print(f'{kind:7s} n={n:3d} {us:7.1f} us  {op:30s} {shp:90s} {fr}')
