#!/usr/bin/env python3
"""Which dtype-conversion / copy kernels does one eager training step (640 px, 16 images, bf16) still launch, and who asks for them?
One step under torch.profiler with shapes; every kernel whose name says copy / cast (torch's *_copy_kernel, direct_copy, StoreWithCast,
multi_tensor copies) is attributed to its aten op, input shapes and dtype pair, split into the static part (trunk + VSS + input projection:
what the recorded graphs replay) and the label-dependent part (decode + loss).  Prints a table sorted by launches."""
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


def decode(feats, shapes, text, b=None):
    with record_function('DYN:forward'):
        return _decode(feats, shapes, text, b)
head.decode = decode
feats_hook = {}
_tm = model.token_memory


def token_memory(*a, **k):
    f, s = _tm(*a, **k)
    if f.requires_grad:
        f.register_hook(lambda g: feats_hook.__setitem__('t', torch.cuda.Event()) or None)
    return f, s
model.token_memory = token_memory


def run():
    model.zero_grad(set_to_none=True)
    loss, _ = model(batch)
    loss.backward()


for _ in range(3):
    run()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    run()
    torch.cuda.synchronize()
ev = prof.events()
dyn = [(e.time_range.start, e.time_range.end) for e in ev if e.name == 'DYN:forward']
rows = collections.defaultdict(lambda: [0, 0.0])
total = collections.Counter()
for e in ev:
    ks = getattr(e, 'kernels', None) or []
    if e.device_type != torch.autograd.DeviceType.CPU or not ks:
        continue
    for k in ks:
        nm = k.name
        total['kernels'] += 1
        if not any(t in nm for t in ('copy_kernel', 'direct_copy', 'StoreWithCast', 'LoadWithCast', 'CatArray', 'copyBuffer')):
            continue
        kind = ('to_bf16' if 'bfloat16_copy' in nm else 'bf16_to_f32' if 'bfloat16tofloat32' in nm else 'cast_other' if 'WithCast' in nm
                else 'cat' if 'CatArray' in nm else 'copy')
        total[kind] += 1
        rows[(kind, e.name, str(e.input_shapes)[:100])][0] += 1
        rows[(kind, e.name, str(e.input_shapes)[:100])][1] += k.duration
print('# one eager step: kernels', total['kernels'], {k: v for k, v in total.items() if k != 'kernels'})
for (kind, op, shp), (n, us) in sorted(rows.items(), key=lambda x: (x[0][0], -x[1][0])):
    print(f'{kind:12s} n={n:3d} {us:8.1f} us  {op:34s} {shp}')
