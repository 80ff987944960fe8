#!/usr/bin/env python3
"""Per-phase GPU time and launch count of one eager training step (640 px, 16 images, bf16), forward and backward separately.

The step is run kernel by kernel (no graph replay) with a device synchronisation at every phase edge, under torch.profiler; every kernel
is attributed to the phase whose host-side window contains its start.  Forward edges are function boundaries (trunk -> VSS + input
projection -> query selection -> decoder layers -> loss); backward edges are tensor hooks on the tensors that cross them (the token
memory, the three trunk maps), which fire when the gradient of that tensor is complete.

    python3 tools/step_phases.py [--top 6] [--json out.json]

Prints a table (phase, GPU ms, launches, of which torch-generic elementwise / reduce / copy / fill, library GEMM) and the top kernels of
every phase; the synchronisations make the SUM larger than a free-running step - the figures are shares, not a step time."""
import argparse, collections, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch
from tamtr_amd.model import RTDETRDetectionWorldModel
from tamtr_amd.tuning import use_tuned_convolutions
from torch.profiler import profile, ProfilerActivity, record_function

ap = argparse.ArgumentParser()
ap.add_argument('--top', type=int, default=6)
ap.add_argument('--json', default=None)
ap.add_argument('--batch', type=int, default=16)
ap.add_argument('--imgsz', type=int, default=640)
args = ap.parse_args()

use_tuned_convolutions('shipped')
torch.manual_seed(0)
model = RTDETRDetectionWorldModel(nc=10).cuda().train()
model.autocast_dtype = torch.bfloat16
head = model.model[-1]
batch = synth_batch(args.batch, args.imgsz, 1, 'cuda')
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-4, fused=True)
marks = []          # (label, profiler-clock ns) in issue order; a mark is taken AFTER a device synchronisation
ON = [False]


def mark(label):
    if ON[0]:
        torch.cuda.synchronize()
        with record_function('MARK:' + label):
            pass


def edge_fn(obj, name, before, after):
    """Wrap obj.name so that mark(before) runs on entry and mark(after) on return."""
    fn = getattr(obj, name)

    def wrapped(*a, **k):
        if before:
            mark(before)
        out = fn(*a, **k)
        if after:
            mark(after)
        return out
    setattr(obj, name, wrapped)


# forward edges: [trunk] encode [vss+proj] decode: _get_decoder_input [query selection] decoder [decoder] criterion [loss]
_encode = head.encode


def encode(x, drop_scales=None):
    mark('fwd trunk')
    if ON[0] and torch.is_grad_enabled():
        for i, t in enumerate(x):
            if t.requires_grad:
                t.register_hook(lambda g, i=i: mark(f'bwd vss+proj (until trunk map {i})'))
    out = _encode(x, drop_scales)
    mark('fwd vss + input_proj')
    if ON[0] and out[0].requires_grad:
        out[0].register_hook(lambda g: mark('bwd loss + decoder + query selection'))
    return out
head.encode = encode
edge_fn(head, '_get_decoder_input', None, 'fwd query selection (cdn + enc_output + top-k)')
edge_fn(head.decoder, 'forward', None, 'fwd decoder (3 layers + heads)')
model.criterion = model.init_criterion()
edge_fn(model.criterion, 'forward', None, 'fwd loss (4 + 4 layers, device assignment)')


def step():
    opt.zero_grad(set_to_none=True)
    mark('optimizer.zero_grad')
    loss, _ = model(batch)
    mark('fwd tail (stack of the 12 terms)')
    loss.backward()
    mark('bwd trunk')
    torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.grad is not None], max_norm=0.1)
    opt.step()
    mark('clip + AdamW')


for _ in range(3):
    step()
torch.cuda.synchronize()
ON[0] = True
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
ON[0] = False

ev = prof.events()
mk = sorted(((e.time_range.start, e.name[5:]) for e in ev if e.name.startswith('MARK:')), key=lambda x: x[0])
kern = sorted(((e.time_range.start, e.time_range.end - e.time_range.start, e.name) for e in ev
               if e.device_type == torch.autograd.DeviceType.CUDA and not e.name.startswith('MARK:')), key=lambda x: x[0])
GENERIC = ('at::native::', 'elementwise_kernel', 'reduce_kernel', 'fillBuffer', 'copyBuffer', 'CatArray', 'multi_tensor_apply', 'index', 'gather', 'scatter', 'sort', 'topk')
GEMM = ('Cijk_', 'ck::', '_ZN2ck', 'igemm_', 'gemm', 'Gemm')
phases = collections.OrderedDict()
mi = 0
for t0, dur, name in kern:
    while mi < len(mk) and mk[mi][0] <= t0:
        mi += 1
    label = mk[mi][1] if mi < len(mk) else 'after the last mark'
    p = phases.setdefault(label, {'us': 0.0, 'n': 0, 'generic_us': 0.0, 'generic_n': 0, 'lib_us': 0.0, 'lib_n': 0, 'reduce_n': 0, 'memset_n': 0, 'k': collections.Counter(), 'kn': collections.Counter()})
    p['us'] += dur
    p['n'] += 1
    own = '(anonymous namespace)' in name and 'at::native' not in name
    if not own and any(s in name for s in GENERIC):
        p['generic_us'] += dur
        p['generic_n'] += 1
    elif not own and any(s in name for s in GEMM) or name.startswith(('SubTensorOp', 'naive_conv', 'batched_transpose')):
        p['lib_us'] += dur
        p['lib_n'] += 1
    p['reduce_n'] += 'reduce_kernel' in name
    p['memset_n'] += 'fillBuffer' in name
    short = name.split('(')[0][-70:] if 'anonymous' not in name else name.split('::')[-1].split('(')[0][:70]
    p['k'][short] += dur
    p['kn'][short] += 1
order = [m[1] for m in mk]
seen = []
for l in order:
    if l in phases and l not in seen:
        seen.append(l)
tot_us, tot_n = sum(p['us'] for p in phases.values()), sum(p['n'] for p in phases.values())
print(f'# one eager step, {args.imgsz} px, {args.batch} images, bf16; synchronised at every phase edge.  total {tot_us / 1e3:.1f} ms of kernels in {tot_n} launches')
print(f'{"phase":58s} {"GPU ms":>8s} {"launches":>9s} {"torch-generic ms / n":>22s} {"library (GEMM, conv) ms / n":>28s} {"reduce_kernel":>14s} {"memset":>7s}')
for l in seen + [x for x in phases if x not in seen]:
    p = phases[l]
    print(f'{l:58s} {p["us"] / 1e3:8.2f} {p["n"]:9d} {p["generic_us"] / 1e3:14.2f} / {p["generic_n"]:5d} {p["lib_us"] / 1e3:20.2f} / {p["lib_n"]:5d} {p["reduce_n"]:14d} {p["memset_n"]:7d}')
print()
for l in seen:
    p = phases[l]
    print(f'== {l}: {p["us"] / 1e3:.2f} ms, {p["n"]} launches')
    for k, us in p['k'].most_common(args.top):
        print(f'     {us / 1e3:7.3f} ms  n={p["kn"][k]:4d}  {k}')
if args.json:
    json.dump({l: {k: (v if not isinstance(v, collections.Counter) else dict(v.most_common(12))) for k, v in p.items()} for l, p in phases.items()}, open(args.json, 'w'), indent=1)
