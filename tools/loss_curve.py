#!/usr/bin/env python3
"""Does the bf16 mode TRAIN like fp32?  ("a 5 % loss / 2-logit shift is not obviously mAP-neutral".)  The dataset is not in
the image, so the closest experiment available: the same model, the same initial weights, the same fixed synthetic batch (16 images, 8 boxes
each: something the model can fit), the same optimizer (AdamW 1e-4, clip 0.1, EMA) and the same per-step seeds for the denoising noise -
once in bf16 mode (what bench.py measures) and once in fp32 - for --steps optimisation steps each; the loss of every step is recorded.
Printed: the two curves at a few steps, their smoothed relative difference, and the relative difference of the mean loss over the last
quarter of the run.

    python3 tools/loss_curve.py --steps 200 --out profiles/r04_loss_curve_bf16_vs_fp32.json"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch
from tamtr_amd.engine import FusedOptimStep, ModelEMA
from tamtr_amd.model import RTDETRDetectionWorldModel
from tamtr_amd.tuning import use_tuned_convolutions

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=200)
ap.add_argument('--batch', type=int, default=16)
ap.add_argument('--imgsz', type=int, default=640)
ap.add_argument('--lr', type=float, default=1e-4)
ap.add_argument('--out', default=None)
args = ap.parse_args()
use_tuned_convolutions('shipped')
torch.manual_seed(0)
base = RTDETRDetectionWorldModel(nc=10).cuda().train()
state = {k: v.clone() for k, v in base.state_dict().items()}
batch = synth_batch(args.batch, args.imgsz, 1, 'cuda')
del base
curves, secs = {}, {}
for mode in ('bf16', 'fp32'):
    torch.manual_seed(0)
    model = RTDETRDetectionWorldModel(nc=10).cuda().train()
    model.load_state_dict(state)
    model.autocast_dtype = torch.bfloat16 if mode == 'bf16' else None
    opt = torch.optim.AdamW(model.parameters(), lr=args.lr, weight_decay=1e-4, betas=(0.9, 0.999), fused=True)
    st = FusedOptimStep.create(model, opt, ModelEMA(model), max_norm=0.1, shadows=mode == 'bf16')
    losses = []
    t0 = time.time()
    for i in range(args.steps):
        opt.zero_grad(set_to_none=True)
        torch.manual_seed(1000 + i)          # the step's denoising noise and DropPath draws: the same in both modes
        loss, _ = model(batch)
        loss.backward()
        st.step()
        losses.append(loss.detach())
    torch.cuda.synchronize()
    secs[mode] = time.time() - t0
    curves[mode] = [float(v) for v in torch.stack(losses).float().cpu()]
    st.drop_shadows()
    del model, opt, st
    print(f'[{mode}] {args.steps} steps in {secs[mode]:.1f} s; loss {curves[mode][0]:.3f} -> {curves[mode][-1]:.3f}', file=sys.stderr, flush=True)


def smooth(c, k=10):
    return [sum(c[max(0, i - k + 1):i + 1]) / len(c[max(0, i - k + 1):i + 1]) for i in range(len(c))]


a, b = smooth(curves['bf16']), smooth(curves['fp32'])
rel = [abs(x - y) / y for x, y in zip(a, b)]
q = max(1, args.steps // 4)
tail = (sum(curves['bf16'][-q:]) / q, sum(curves['fp32'][-q:]) / q)
marks = sorted({0, 1, 2, 5, 10, 20, 50, 100, 150, args.steps - 1} & set(range(args.steps)))
out = {'what': 'loss per optimisation step on one fixed synthetic batch, bf16 mode against fp32, same initial weights / optimizer / per-step seeds',
       'steps': args.steps, 'batch': args.batch, 'imgsz': args.imgsz, 'lr': args.lr, 'seconds': secs,
       'loss_at': {str(i): {'bf16': curves['bf16'][i], 'fp32': curves['fp32'][i]} for i in marks},
       'smoothed_rel_diff': {'mean': sum(rel) / len(rel), 'max': max(rel), 'last_quarter_mean': sum(rel[-q:]) / q},
       'mean_loss_last_quarter': {'bf16': tail[0], 'fp32': tail[1], 'rel_diff': abs(tail[0] - tail[1]) / tail[1]},
       'loss_drop': {'bf16': curves['bf16'][0] - tail[0], 'fp32': curves['fp32'][0] - tail[1]},
       'curves': curves}
print(json.dumps({k: v for k, v in out.items() if k != 'curves'}, indent=1))
if args.out:
    json.dump(out, open(args.out, 'w'))
