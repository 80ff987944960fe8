#!/usr/bin/env python3
"""Where do the memset nodes come from that the recorded backward contains when bench.py runs with more than one rank (8 of them in the 2-rank
rehearsal, none with one rank)?  Run under torchrun: every rank sets itself up exactly as bench.py does (process group, ranked table seeding,
model, fused optimizer step with weight copies, gradient reducer), then profiles one eager pass over the recorded part and lists the memset
kernels with the aten op that issued them; rank 0 prints.

    python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/micro/ddp_memset_probe.py"""
import collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch.distributed as dist
from bench import synth_batch
from tamtr_amd import dist as tdist
from tamtr_amd.engine import FusedOptimStep, ModelEMA
from tamtr_amd.model import RTDETRDetectionWorldModel
from tamtr_amd.tuning import use_tuned_convolutions_ranked
from torch.profiler import profile, ProfilerActivity

rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
if world > 1:
    dist.init_process_group('gloo', rank=rank, world_size=world)
torch.cuda.set_device(0)
if os.environ.get('PROBE_SERIAL') == '1' and world > 1:   # one rank after the other: is it the concurrency on the shared GPU, or the set-up?
    for r in range(rank):
        dist.barrier()
print(f'[rank {rank}] OMP_NUM_THREADS={os.environ.get("OMP_NUM_THREADS")} torch threads {torch.get_num_threads()}', flush=True)
use_tuned_convolutions_ranked('shipped', None, rank=rank, world=world)
torch.manual_seed(0)
model = RTDETRDetectionWorldModel(nc=10).cuda().train()
model.autocast_dtype = torch.bfloat16
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-4, fused=True)
stepper = FusedOptimStep.create(model, opt, ModelEMA(model), max_norm=0.1, shadows=True)
reducer = tdist.GradReducer(model.named_parameters(), skip=lambda n: '.attn.' in n, late=lambda n: 'denoising_class_embed' in n) if world > 1 else None
b = synth_batch(16, 640, 1 + rank, 'cuda')
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
rows = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    ks = getattr(e, 'kernels', None) or []
    if e.device_type != torch.autograd.DeviceType.CPU or not ks:
        continue
    for k in ks:
        if 'emset' in k.name or 'fillBuffer' in k.name:
            frame = next((f for f in (e.stack or []) if 'tam-tr_amd' in f or 'tamtr_amd' in f), (e.stack or ['?'])[0] if e.stack else '?')
            rows[(e.name, str(e.input_shapes)[:120], frame.strip()[-100:])][0] += 1
if os.environ.get('PROBE_SERIAL') == '1' and world > 1:
    for r in range(rank, world - 1):
        dist.barrier()
elif world > 1:
    dist.barrier()
print(f'[rank {rank}] memset kernels: {sum(v[0] for v in rows.values())}', flush=True)
if rank == 0:
    print(f'# world {world}: memset kernels of one eager pass over the recorded part: {sum(v[0] for v in rows.values())}')
    for (op, shp, fr), (n, _) in sorted(rows.items(), key=lambda x: -x[1][0]):
        print(f'  n={n:3d} {op:30s} {shp:120s} {fr}')
if world > 1:
    dist.destroy_process_group()
