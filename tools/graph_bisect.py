#!/usr/bin/env python3
"""Which nodes of the recorded static part break under the HIP runtime's AQL packet capture of graph nodes?

One sub-part per process (the runtime reads its switches when it starts):

    PACKET_CAPTURE=1 PART=vss0|vss1|vss2|proj|trunk|all [BLAS=default|cublas|cublaslt] [BS=16] [IMG=640] python3 tools/graph_bisect.py [tag]

vss<i>: one VSS block of the head alone on a random map of its level (no MIOpen call in it: eager execution is bitwise reproducible,
so a replay is held to eager exactly); proj: the three input projections; trunk: model.model[:-1] (MIOpen; eager noise applies);
all: the whole static part as bench.py records it.  BLAS switches torch's preferred GEMM backend (rocBLAS / hipBLASLt) for the
process.  Prints one JSON line: per-replay counts and the parameters whose gradient differs from eager, with their relative error.
"""
import json, os, sys
os.environ['DEBUG_CLR_GRAPH_PACKET_CAPTURE'] = os.environ.get('PACKET_CAPTURE', '0')
import torch
import torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch
from tamtr_amd.graphs import GraphedPart
from tamtr_amd.model import RTDETRDetectionWorldModel

tag = sys.argv[1] if len(sys.argv) > 1 else ''
part = os.environ.get('PART', 'vss0')
blas = os.environ.get('BLAS', 'default')
B, S = int(os.environ.get('BS', 16)), int(os.environ.get('IMG', 640))
REPLAYS = int(os.environ.get('REPLAYS', 4))
if blas != 'default':
    torch.backends.cuda.preferred_blas_library(blas)
if os.environ.get('TUNED', '1') == '1':      # the shipped convolution tables: eager run-to-run noise 3e-3 instead of ~1 (MIOpen heuristic)
    from tamtr_amd.tuning import use_tuned_convolutions
    use_tuned_convolutions('shipped')
torch.manual_seed(0)
model = RTDETRDetectionWorldModel(nc=10).cuda().train()
model.autocast_dtype = torch.bfloat16
head = model.model[-1]
dims = [(128, S // 4), (256, S // 8), (512, S // 16)]


class _AC(nn.Module):
    """forward under the step's bf16 autocast, cache off (as model.token_memory runs the recorded function)."""

    def forward(self, *a):
        with torch.autocast('cuda', dtype=torch.bfloat16, cache_enabled=False):
            return self.run(*a)


class VssPart(_AC):
    def __init__(self, i):
        super().__init__()
        self.blk = head.VSSBlocks[i]

    def run(self, x, scales):
        return self.blk(x, scales)


class ProjPart(_AC):
    def __init__(self):
        super().__init__()
        self.proj = head.input_proj

    def run(self, a, b, c):
        return torch.cat([head._project_level(i, t)[0] for i, t in enumerate((a, b, c))], 1)


class TrunkPart(_AC):
    def __init__(self):
        super().__init__()
        self.trunk = nn.ModuleList(model.model[:-1])

    def run(self, img, txt):
        from tamtr_amd import ops
        from tamtr_amd.modules import TIAGELAN
        x = img.contiguous(memory_format=torch.channels_last)
        counters = ops.begin_bn_counter_batch()
        y = []
        for m in model.model[:-1]:
            if m.f != -1:
                x = y[m.f] if isinstance(m.f, int) else [x if j == -1 else y[j] for j in m.f]
            x = m(x, txt) if isinstance(m, TIAGELAN) else m(x)
            y.append(x if m.i in model.save else None)
        ops.end_bn_counter_batch()
        return torch.cat([y[j].float().flatten(1) for j in head.f], 1)


g = torch.Generator(device='cuda').manual_seed(3)
if part.startswith('vss'):
    i = int(part[3:])
    c, hw = dims[i]
    mod = VssPart(i)
    args = (torch.randn(B, hw, hw, c, device='cuda', generator=g).bfloat16(), torch.ones(2, B, device='cuda'))
elif part == 'proj':
    mod = ProjPart()
    args = tuple(torch.randn(B, hw, hw, c, device='cuda', generator=g).bfloat16() for c, hw in dims)
elif part == 'trunk':
    mod = TrunkPart()
    b = synth_batch(B, S, 1, 'cuda')
    args = (b['img'], b['txt_feats'])
else:
    b = synth_batch(B, S, 1, 'cuda')
    model.capture_static_part(b['img'], b['txt_feats'], verify=False)
    gp = model._static[0]
    mod = None
if mod is not None:
    mod.train()
    gp = GraphedPart(mod, args, warmup=3)
chk = gp.verify(replays=REPLAYS)
# per-tensor table of what went wrong: redo one replay and list the gradients that are off (verify() keeps only the worst)
saved = gp._snapshot_buffers()
live = [i for i, gr in enumerate(gp.static_grads) if gr is not None]
cot = torch.randn(gp.static_out.shape, device='cuda', generator=g).to(gp.static_out.dtype)
with torch.enable_grad():
    out = gp.module(*gp.static_in)
    ref = torch.autograd.grad(out, [gp.params[i] for i in live], cot, allow_unused=True)
ref = [None if r is None else r.detach().float().clone() for r in ref]
out = out.detach().float().clone()
bad = {}
for rep in range(REPLAYS):
    gp._restore_buffers(saved)
    gp.static_gout.copy_(cot)
    gp.fwd.replay()
    junk = torch.full((1 << 22,), float('nan'), device='cuda')
    del junk
    gp.bwd.replay()
    torch.cuda.synchronize()
    eo = float((gp.static_out.float() - out).norm() / out.norm())
    if not eo < 1e-3:
        bad.setdefault('<output>', []).append(eo)
    for i, r in zip(live, ref):
        if r is None:
            continue
        d = float((gp.static_grads[i].float() - r).norm() / r.norm().clamp_min(1e-30))
        if not d < float(os.environ.get('OFF_TOL', '1e-2')):
            bad.setdefault(gp.names[i], []).append(d)
gp._restore_buffers(saved)
flags = {k: v for k, v in os.environ.items() if k.startswith(('DEBUG_', 'HIP_FORCE', 'AMD_SERIALIZE', 'GPU_', 'ROC_', 'TAMTR_'))}
print(json.dumps({'tag': tag, 'part': part, 'blas': blas, 'flags': flags, 'ok': chk['ok'], 'conclusive': chk['conclusive'],
                  'eager_noise_grad_l2': chk['eager_noise_grad_l2'], 'grad_l2_rel_max': chk['grad_l2_rel_max'], 'out_rel_max': chk['out_rel_max'],
                  'census': getattr(gp, 'census', None), 'grads': len(live), 'nonfinite': [r['nonfinite_grads'] for r in chk['replays']],
                  'off_tensors': {k: [f'{v:.2e}' for v in vs] for k, vs in sorted(bad.items())}}), flush=True)
