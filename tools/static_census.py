#!/usr/bin/env python3
"""Which ops of the RECORDED part of the step (trunk + VSS blocks + input projection, forward and backward) put memset nodes or torch's
generic reduction kernels into the HIP graphs?  (Those are the nodes that do not replay correctly under AQL packet capture:
profiles/r04_packet_capture_bisect.txt.)  One eager forward + backward of model.token_memory under torch.profiler with stacks; for every
memset and every at::native reduce kernel: the aten op that issued it, its input shapes and the innermost frame of this package.

    python3 tools/static_census.py [--all-generic]      (--all-generic: list every at::native kernel, not only reductions / memsets)"""
import collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch
from tamtr_amd.model import RTDETRDetectionWorldModel
from tamtr_amd.tuning import use_tuned_convolutions
from torch.profiler import profile, ProfilerActivity

use_tuned_convolutions('shipped')
torch.manual_seed(0)
model = RTDETRDetectionWorldModel(nc=10).cuda().train()
model.autocast_dtype = torch.bfloat16
b = synth_batch(16, 640, 1, 'cuda')
img, txt = b['img'], b['txt_feats'].float()
dp = torch.ones(3, 2, 16, device='cuda')
params = [p for n, p in model.named_parameters() if not n.startswith('model.41.') or '.VSSBlocks.' in n or '.input_proj.' in n]


def run():
    feats, _ = model.token_memory(img, txt, autocast_cache=False, drop_scales=dp)
    return torch.autograd.grad(feats, params, torch.ones_like(feats), allow_unused=True)


for _ in range(3):
    run()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    run()
    torch.cuda.synchronize()
allg = '--all-generic' in sys.argv
rows = collections.defaultdict(lambda: [0, 0.0])
nk = collections.Counter()
for e in prof.events():
    ks = getattr(e, 'kernels', None) or []
    if e.device_type != torch.autograd.DeviceType.CPU or not ks:
        continue
    for k in ks:
        nm = k.name
        kind = 'memset' if ('emset' in nm or 'fillBuffer' in nm) else ('reduce' if ('reduce_kernel' in nm and 'at::native' in nm) else
                                                                       ('generic' if 'at::native' in nm or 'elementwise' in nm else None))
        nk[kind] += 1
        if kind in ('memset', 'reduce') or (allg and kind == 'generic'):
            frame = next((f for f in (e.stack or []) if 'tam-tr_amd' in f or 'tamtr_amd' in f), (e.stack or ['?'])[0] if e.stack else '?')
            key = (kind, e.name, str(e.input_shapes)[:110], frame.strip()[-110:])
            rows[key][0] += 1
            rows[key][1] += k.duration
print('# kernels of one eager pass over the recorded part by kind:', dict(nk))
for (kind, op, shp, fr), (n, us) in sorted(rows.items(), key=lambda x: (x[0][0], -x[1][0])):
    print(f'{kind:8s} n={n:3d} {us:8.1f} us  {op:28s} {shp:110s} {fr}')
