#!/usr/bin/env python3
"""Staged HIP-graph capture experiments on the step (one stage per process: `python3 tools/try_graph.py <stage>`).
  fwd     torch.cuda.graph around the trunk forward (no grad)
  layer0  make_graphed_callables on the first Conv block alone (MIOpen conv + csrc/bn.hip), forward + backward
  trunk   make_graphed_callables on graph layers 0..8 (backbone)
  static  model.capture_static_part(): trunk + VSS + input projection; 10 eager vs 10 graphed training steps, same seeds"""
import faulthandler, os, sys, time
faulthandler.enable()
os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', os.environ.get('PACKET_CAPTURE', '0'))  # PACKET_CAPTURE=1 reproduces the garbage replays
import torch
import torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch
from tamtr_amd.model import RTDETRDetectionWorldModel

stage = sys.argv[1]
B, S = int(os.environ.get('BS', 16)), int(os.environ.get('IMG', 640))
torch.manual_seed(0)
model = RTDETRDetectionWorldModel(nc=10).cuda().train()
model.autocast_dtype = torch.bfloat16
batch = synth_batch(B, S, 1, 'cuda')
img, txt = batch['img'], batch['txt_feats'].float()
print(f'stage {stage}: bs {B} img {S}', flush=True)

if stage == 'memset':
    # does a replayed graph keep memset nodes (torch.zeros / zero_ of contiguous tensors, reduce semaphores) ordered with the kernels
    # around them?  z = zeros; z += x; s = z.sum() chained 200 times on one capture stream, replayed 50 times.
    x = torch.ones(1 << 22, device='cuda')
    tall = torch.ones(1 << 18, 64, device='cuda', dtype=torch.bfloat16)
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    def body():
        outs = []
        for i in range(200):
            z = torch.zeros(1 << 22, device='cuda')
            z.add_(x)
            outs.append(z.sum())                     # multi-block reduction: semaphores zeroed by a memset
            outs.append(tall.sum(0, dtype=torch.float32)[i % 64])
        return torch.stack(outs)
    with torch.cuda.stream(st):
        ref = body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        out = body()
    bad = 0
    for rep in range(50):
        g.replay()
        junk = torch.full((1 << 24,), float('nan'), device='cuda'); del junk
        bad += int((out != ref).sum())
    torch.cuda.synchronize()
    print('memset-in-graph mismatches over 50 replays x 400 values:', bad, flush=True)

elif stage == 'fwd':
    with torch.no_grad():
        for _ in range(3):
            model.token_memory(img, txt)
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                ref, _ = model.token_memory(img, txt, autocast_cache=False)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        print('warm', flush=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out, _ = model.token_memory(img, txt, autocast_cache=False)
        print('captured', flush=True)
        g.replay(); torch.cuda.synchronize()
        print('replayed; finite', bool(torch.isfinite(out.float()).all()), flush=True)
        t0 = time.perf_counter()
        for _ in range(10): g.replay()
        torch.cuda.synchronize()
        print(f'graph forward {(time.perf_counter() - t0) * 100:.2f} ms', flush=True)
        t0 = time.perf_counter()
        for _ in range(10): model.token_memory(img, txt)
        torch.cuda.synchronize()
        print(f'eager forward {(time.perf_counter() - t0) * 100:.2f} ms', flush=True)

elif stage in ('layer0', 'trunk'):
    n = 1 if stage == 'layer0' else 9

    class Part(nn.Module):
        def __init__(self, layers):
            super().__init__()
            self.layers = nn.ModuleList(layers)

        def forward(self, x):
            with torch.autocast('cuda', dtype=torch.bfloat16, cache_enabled=False):
                x = x.contiguous(memory_format=torch.channels_last)
                for l in self.layers:
                    x = l(x)
            return x
    part = Part(model.model[:n]).train()
    x = img.clone().requires_grad_()
    for _ in range(2):
        part(x).float().square().mean().backward()
    torch.cuda.synchronize()
    print('eager ok', flush=True)
    g = torch.cuda.make_graphed_callables(part, (x,), num_warmup_iters=3, allow_unused_input=True)
    print('captured', flush=True)
    for _ in range(3):
        part.zero_grad(set_to_none=True)
        l = g(x).float().square().mean()
        l.backward()
    torch.cuda.synchronize()
    print('graphed steps ok', float(l), flush=True)

elif stage == 'static':
    import copy
    steps = int(os.environ.get('STEPS', 10))

    def run(m, n):
        opt = torch.optim.AdamW(m.parameters(), lr=1e-4, fused=True)
        losses = []
        for i in range(n + 3):
            if i == 3:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            torch.manual_seed(100 + i)
            opt.zero_grad(set_to_none=True)
            loss, _ = m(batch)
            loss.backward()
            opt.step()
            losses.append(loss.detach())
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, [float(l) for l in losses]
    twin = copy.deepcopy(model)
    ms_e, le = run(twin, steps)
    print(f'eager  : {ms_e:.1f} ms/step  losses {le[:4]} .. {le[-1]:.4f}', flush=True)
    model.capture_static_part(img, txt)
    print('captured', flush=True)
    ms_g, lg = run(model, steps)
    print(f'graphed: {ms_g:.1f} ms/step  losses {lg[:4]} .. {lg[-1]:.4f}', flush=True)

elif stage == 'check':
    # same weights, DropPath off: gradients of the graphed static part vs eager, twice (replay consistency), then host phase times
    for m in model.modules():
        if hasattr(m, 'drop_prob'):
            m.drop_prob = 0.0

    def grads(m):
        m.zero_grad(set_to_none=True)
        torch.manual_seed(7)
        loss, _ = m(batch)
        loss.backward()
        return float(loss), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
    le, ge = grads(model)
    le2, ge2 = grads(model)
    model.capture_static_part(img, txt)
    for rep in range(3):
        lg, gg = grads(model)
        worst = []
        for k in ge:
            d = (gg[k].float() - ge[k].float()).abs().max() / (ge[k].float().abs().max() + 1e-12) if k in gg else float('inf')
            n = (ge2[k].float() - ge[k].float()).abs().max() / (ge[k].float().abs().max() + 1e-12)
            worst.append((float(d), float(n), k))
        worst.sort(reverse=True)
        print(f'rep {rep}: loss eager {le:.5f} (again {le2:.5f}) graphed {lg:.5f}; params with grad {len(ge)} vs {len(gg)}; '
              f'nonfinite {sum(1 for v in gg.values() if not torch.isfinite(v).all())}', flush=True)
        for d, n, k in worst[:6]:
            print(f'   rel diff {d:.3e} (eager run-to-run {n:.3e})  {k}', flush=True)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True)
    for i in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        loss, _ = model(batch); t1 = time.perf_counter()
        loss.backward(); t2 = time.perf_counter()
        opt.step(); t3 = time.perf_counter()
        torch.cuda.synchronize(); t4 = time.perf_counter()
        print(f'step {i}: fwd issue {1e3*(t1-t0):.1f}  bwd issue {1e3*(t2-t1):.1f}  opt issue {1e3*(t3-t2):.1f}  drain {1e3*(t4-t3):.1f}  total {1e3*(t4-t0):.1f} ms  loss {float(loss):.4f}', flush=True)

elif stage == 'bisect':
    # GraphedPart over graph layers [0, n): replay vs eager, values and gradients, for growing n; then the VSS blocks alone
    from tamtr_amd.graphs import GraphedPart
    from tamtr_amd.modules import TIAGELAN
    for m in model.modules():
        if hasattr(m, 'drop_prob'):
            m.drop_prob = 0.0

    class Upto(nn.Module):
        def __init__(self, model, n):
            super().__init__()
            self.layers, self.n = nn.ModuleList(model.model[:n]), n
            object.__setattr__(self, 'save', model.save)

        def forward(self, x, t):
            with torch.autocast('cuda', dtype=torch.bfloat16, cache_enabled=False):
                x = x.contiguous(memory_format=torch.channels_last)
                y = []
                for m in self.layers:
                    if m.f != -1:
                        x = y[m.f] if isinstance(m.f, int) else [x if j == -1 else y[j] for j in m.f]
                    x = m(x, t) if isinstance(m, TIAGELAN) else m(x)
                    y.append(x if m.i in self.save else None)
            return x.float()

    class Vss(nn.Module):
        def __init__(self, blk, c, hw):
            super().__init__()
            self.blk = blk
            self.shape = (B, hw, hw, c)

        def forward(self, x, t):
            with torch.autocast('cuda', dtype=torch.bfloat16, cache_enabled=False):
                return self.blk(x).float()

    def rel(a, b):
        return float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-20))

    def probe(name, part, args):
        cot = torch.randn_like(part(*args)).detach()
        def eager():
            part.zero_grad(set_to_none=True)
            o = part(*args); o.backward(cot)
            return o.detach().clone(), {k: p.grad.clone() for k, p in part.named_parameters() if p.grad is not None}
        o1, g1 = eager(); o2, g2 = eager()
        noise = max([rel(o2, o1)] + [rel(g2[k], g1[k]) for k in g1])
        gp = GraphedPart(part, args)
        for rep in range(REPS):
            part.zero_grad(set_to_none=True)
            o = gp(*args)
            if CHURN:   # eager allocations and kernels between the two replays, as the decoder + loss do
                junk = [torch.full((1 << 22,), float('nan'), device='cuda') for _ in range(64)]
                del junk
            o.backward(cot)
            g = {k: p.grad for k, p in part.named_parameters() if p.grad is not None}
            bad = sorted(((rel(g[k], g1[k]) if k in g else float('inf'), k) for k in g1), reverse=True)
            nonfin = sum(1 for v in g.values() if not torch.isfinite(v).all())
            print(f'{name} rep {rep}: out rel {rel(o, o1):.2e}  worst grad rel {bad[0][0]:.2e} ({bad[0][1]})  nonfinite grads {nonfin}  '
                  f'[eager run-to-run {noise:.2e}]  grads {len(g)}/{len(g1)}', flush=True)
        del gp
    which = os.environ.get('PARTS', '1,3,9,17,41,vss').split(',')
    REPS = int(os.environ.get('REPS', 3))
    CHURN = os.environ.get('CHURN') == '1'
    for w in which:
        if w == 'static':
            from tamtr_amd.model import _StaticPart
            probe('static', _StaticPart(model).train(), (img, txt))
        elif w in ('vp', 'tv'):
            head = model.model[-1]

            class VP(nn.Module):   # VSS blocks + input projection on given maps (deterministic in eager mode)
                def __init__(self):
                    super().__init__()
                    self.vss, self.proj = head.VSSBlocks, head.input_proj

                def forward(self, a, b, c):
                    with torch.autocast('cuda', dtype=torch.bfloat16, cache_enabled=False):
                        return head.encode([a, b, c])[0].float()

            class TV(nn.Module):   # trunk + VSS blocks, no projection
                def __init__(self):
                    super().__init__()
                    self.trunk, self.vss = nn.ModuleList(model.model[:-1]), head.VSSBlocks
                    object.__setattr__(self, 'up', Upto(model, 41))

                def forward(self, x, t):
                    with torch.autocast('cuda', dtype=torch.bfloat16, cache_enabled=False):
                        x = x.contiguous(memory_format=torch.channels_last)
                        y = []
                        for m in self.trunk:
                            if m.f != -1:
                                x = y[m.f] if isinstance(m.f, int) else [x if j == -1 else y[j] for j in m.f]
                            x = m(x, t) if isinstance(m, TIAGELAN) else m(x)
                            y.append(x if m.i in model.save else None)
                        toks = [blk(f.permute(0, 2, 3, 1)) for blk, f in zip(self.vss, [y[j] for j in head.f])]
                        return torch.cat([t_.flatten(1) for t_ in toks], 1).float()
            if w == 'vp':
                maps = tuple(torch.randn(B, c, S // d, S // d, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
                             for c, d in ((128, 4), (256, 8), (512, 16)))
                probe('vss+proj', VP().train(), maps)
            else:
                probe('trunk+vss', TV().train(), (img, txt))
        elif w == 'vss':
            head = model.model[-1]
            for i, (c, hw) in enumerate(((128, S // 4), (256, S // 8), (512, S // 16))):
                x = torch.randn(B, hw, hw, c, device='cuda').to(torch.bfloat16)
                probe(f'vss{i}', Vss(head.VSSBlocks[i], c, hw).train(), (x, txt))
        else:
            probe(f'layers[0:{w})', Upto(model, int(w)).train(), (img, txt))
