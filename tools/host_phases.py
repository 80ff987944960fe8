#!/usr/bin/env python3
"""Host issue time of the benchmarked step by phase: wall-clock time the Python thread spends in each phase of bench.py's step while the
GPU runs behind it (no synchronisation inside the loop; one at the end).  With the static part replayed from HIP graphs.

    [PACKET_CAPTURE=1] python3 tools/host_phases.py [--steps 20] [--static-part graph|eager]

Prints ms per step per phase (mean over the steps), the synchronised step time and the process CPU time per step."""
import argparse, os, sys, time
if 'PACKET_CAPTURE' in os.environ:      # default: the runtime's default (on)
    os.environ['DEBUG_CLR_GRAPH_PACKET_CAPTURE'] = os.environ['PACKET_CAPTURE']
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch
from tamtr_amd.engine import FusedOptimStep, ModelEMA
from tamtr_amd.model import RTDETRDetectionWorldModel
from tamtr_amd.tuning import use_tuned_convolutions

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--static-part', default='graph')
ap.add_argument('--optim-step', default='fused', choices=['fused', 'torch'])
args = ap.parse_args()
use_tuned_convolutions('shipped')
torch.set_num_threads(8)
torch.manual_seed(0)
model = RTDETRDetectionWorldModel(nc=10).cuda().train()
model.autocast_dtype = torch.bfloat16
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-4, fused=True)
ema = ModelEMA(model)
stepper = FusedOptimStep.create(model, opt, ema, 0.1) if args.optim_step == 'fused' else None
batch = synth_batch(16, 640, 1, 'cuda')
if args.static_part == 'graph':
    model.capture_static_part(batch['img'], batch['txt_feats'], verify=False)
head = model.model[-1]
T = {}
last = [0.0]


def lap(name):
    t = time.perf_counter()
    T[name] = T.get(name, 0.0) + t - last[0]
    last[0] = t


def wrap(obj, name, label_before, label_after):
    fn = getattr(obj, name)

    def w(*a, **k):
        lap(label_before)
        out = fn(*a, **k)
        lap(label_after)
        return out
    setattr(obj, name, w)


wrap(head, 'decode', 'fwd: labels upload + static part (graph launch)', 'fwd: decode (cdn, query selection, decoder)')
model.criterion = model.init_criterion()
wrap(model.criterion, 'forward', 'fwd: prediction split / stack', 'fwd: loss')


def step():
    last[0] = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    lap('zero_grad')
    loss, _ = model(batch)
    lap('fwd: tail')
    loss.backward()
    lap('backward (eager decoder + loss nodes, then the graph launch)')
    if stepper is not None:
        stepper.step()
        lap('clip + AdamW + EMA (engine.FusedOptimStep)')
        return
    torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.grad is not None], max_norm=0.1)
    lap('clip_grad_norm_')
    opt.step()
    lap('AdamW (fused)')
    ema.update(model)
    lap('EMA update')


def thread_cpu():
    """{tid: (name, cpu seconds)} of this process' threads (/proc/self/task/*/stat: utime + stime)."""
    out, tck = {}, os.sysconf('SC_CLK_TCK')
    for tid in os.listdir('/proc/self/task'):
        try:
            f = open(f'/proc/self/task/{tid}/stat').read()
            name = f[f.index('(') + 1:f.rindex(')')]
            rest = f[f.rindex(')') + 2:].split()
            out[int(tid)] = (name, (int(rest[11]) + int(rest[12])) / tck)
        except (OSError, ValueError):
            pass
    return out


for _ in range(5):
    step()
torch.cuda.synchronize()
T.clear()
th0 = thread_cpu()
c0, t0 = time.process_time(), time.perf_counter()
for _ in range(args.steps):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'# host issue time per step by phase ({args.steps} steps, static part: {args.static_part}, DEBUG_CLR_GRAPH_PACKET_CAPTURE={os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "unset = on")}, '
      f'{torch.get_num_threads()} torch threads)')
for k, v in T.items():
    print(f'  {v / args.steps * 1e3:8.2f} ms  {k}')
print(f'  {sum(T.values()) / args.steps * 1e3:8.2f} ms  sum = issue time of a step;  loop wall {(t1 - t0) / args.steps * 1e3:.2f} ms/step, with the final drain {(t2 - t0) / args.steps * 1e3:.2f} ms/step, '
      f'process CPU {(time.process_time() - c0) / args.steps * 1e3:.1f} ms/step')
th1 = thread_cpu()
rows = sorted(((th1[t][1] - th0.get(t, (None, 0.0))[1], th1[t][0], t) for t in th1), reverse=True)
print('  CPU time per step by thread (ms): ' + ', '.join(f'{n}[{t}{" = main" if t == os.getpid() else ""}] {d / args.steps * 1e3:.1f}' for d, n, t in rows[:8] if d > 0))
