#!/usr/bin/env python3
"""Replay check of the recorded static part under a given HIP-runtime configuration (one configuration per process: the runtime reads
its switches when it starts).  Used to look for the cause of the garbage replays seen with AQL packet capture of graph nodes
(profiles/r02_graph_capture_findings.txt):

    PACKET_CAPTURE=1 [HIP_FORCE_DEV_KERNARG=0 | DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1 | AMD_SERIALIZE_KERNEL=3 | ...] python3 tools/graph_probe.py [tag]

Prints one JSON line: GraphedPart.verify over 6 replays (token memory + 552 gradients against eager) and the host / wall time of a
graphed training step."""
import json, os, sys, time
os.environ['DEBUG_CLR_GRAPH_PACKET_CAPTURE'] = os.environ.get('PACKET_CAPTURE', '0')
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch
from tamtr_amd.model import RTDETRDetectionWorldModel

from tamtr_amd.tuning import use_tuned_convolutions
use_tuned_convolutions(os.environ.get('CONV_TUNING', 'shipped'))   # as bench.py: with MIOpen's heuristic the backward records memset-based solvers
tag = sys.argv[1] if len(sys.argv) > 1 else ''
B, S = int(os.environ.get('BS', 16)), int(os.environ.get('IMG', 640))
torch.manual_seed(0)
model = RTDETRDetectionWorldModel(nc=10).cuda().train()
model.autocast_dtype = torch.bfloat16
batch = synth_batch(B, S, 1, 'cuda')
model.capture_static_part(batch['img'], batch['txt_feats'], verify=False)
gp = model._static[0]
gp.static_in[2].copy_(model.model[-1].draw_drop_scales(B, 'cuda'))
chk = gp.verify(replays=6)
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True)
rec = []
for i in range(6):
    torch.cuda.synchronize()
    c0, t0 = time.process_time(), time.perf_counter()
    opt.zero_grad(set_to_none=True)
    loss, _ = model(batch)
    loss.backward()
    opt.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    rec.append((round((t1 - t0) * 1e3, 1), round((time.perf_counter() - t0) * 1e3, 1), round((time.process_time() - c0) * 1e3, 1)))
flags = {k: v for k, v in os.environ.items() if k.startswith(('DEBUG_', 'HIP_FORCE', 'AMD_SERIALIZE', 'GPU_', 'ROC_'))}
print(json.dumps({'tag': tag, 'flags': flags, 'ok': chk['ok'], 'out_rel_max': chk['out_rel_max'], 'grad_l2_rel_max': chk['grad_l2_rel_max'],
                  'grad_rel_max': chk['grad_rel_max'], 'eager_noise_out': chk['eager_noise_out'], 'eager_noise_grad_l2': chk['eager_noise_grad_l2'],
                  'informative': chk['informative_grads'], 'nonfinite': [r['nonfinite_grads'] for r in chk['replays']],
                  'worst': [r['worst_informative_grad'] for r in chk['replays']], 'step_ms(issue, wall, cpu)': rec[2:], 'loss': float(loss)}), flush=True)
