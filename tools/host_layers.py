#!/usr/bin/env python3
"""Host issue time per section of a training step (perf_counter around each top-level layer / head part / the loss, NO
synchronisation inside the step) - where the Python + dispatch + launch time goes when the step is host-bound."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch
from tamtr_amd.model import RTDETRDetectionWorldModel
torch.set_num_threads(8)
torch.manual_seed(0)
model = RTDETRDetectionWorldModel(nc=10).cuda().train()
model.autocast_dtype = torch.bfloat16
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True)
batch = synth_batch(16, 640, 1, 'cuda')
recs = {}


def wrap(obj, attr, name):
    orig = getattr(obj, attr)

    def f(*a, **k):
        t0 = time.perf_counter()
        out = orig(*a, **k)
        recs[name] = recs.get(name, 0.0) + time.perf_counter() - t0
        return out
    setattr(obj, attr, f)


for i, m in enumerate(model.model):
    wrap(m, 'forward', f'{i:02d} {m.type}')
head = model.model[-1]
for i, b in enumerate(head.VSSBlocks):
    wrap(b, 'forward', f'  41.VSS{i}')
for i, l in enumerate(head.decoder.layers):
    wrap(l, 'forward', f'  41.dec{i}')
wrap(head, '_get_encoder_input', '  41.enc_input')
wrap(head, '_get_decoder_input', '  41.dec_input')
import tamtr_amd.head as H
wrap(H, 'get_cdn_group', '  41.cdn')
model.criterion = model.init_criterion()
wrap(model.criterion, 'forward', 'loss (criterion)')
wrap(model.criterion.matcher, 'forward', '  matcher')


def step():
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    loss, _ = model(batch)
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.grad is not None], 0.1)
    opt.step()
    t3 = time.perf_counter()
    return t1 - t0, t2 - t1, t3 - t2


for _ in range(3):
    step()
torch.cuda.synchronize()
recs.clear()
N = 5
tot = [0.0, 0.0, 0.0]
t0 = time.perf_counter()
for _ in range(N):
    for i, v in enumerate(step()):
        tot[i] += v
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f'wall {1e3 * wall / N:.1f} ms/step; host issue {1e3 * t_issue / N:.1f} = fwd+loss {1e3 * tot[0] / N:.1f} + bwd {1e3 * tot[1] / N:.1f} + clip/opt {1e3 * tot[2] / N:.1f}')
for k, v in recs.items():
    print(f'  {k:32s} {1e3 * v / N:7.2f} ms host')
