#!/usr/bin/env python3
"""bench.py - images/sec of one TAM-TR training step (fwd + 12-term RIOU loss + bwd + AdamW) at 640x640, bs 16 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype bf16|fp32] [--no-cpu-baseline]
    N > 1: either under a launcher (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py
    --gpus N ...: RANK / WORLD_SIZE in the environment) or plainly as `python bench.py --gpus N`: with no WORLD_SIZE in the environment
    the script starts its own N ranks as a CHILD `torch.distributed.run` (before anything touches the GPU), relays their output and
    exits with their code - what the reference's trainer does for device='0,1,..' (engine/trainer.py:161-189, utils/dist.py:50-62).

Workload = BASELINE.json configs[1]/[2] ("TAM-TR-s" := the reference's only graph, TAMTR.yaml, 42.1 M params - SURVEY D3):
synthetic images rand(B,3,640,640), unit-norm 10x512 text features, 8 GT boxes per image (=> 192 denoising + 100 queries).
The step is the reference's training step (engine/trainer.py:328-357,471-479): forward through BaseModel.forward(dict),
loss, backward, gradient clip 0.1, AdamW(lr 1e-4, wd 1e-4), EMA update of the weights (utils/torch_utils.py:392-419).  Every rank runs bs 16 (weak scaling); gradients are summed
over ranks with bucketed RCCL all-reduces overlapped with the backward (tam-tr_amd/dist.py).

One JSON line on rank 0.  `roofline`: the MEH value-projection GEMM (tamtr_linear_bf16, the dominant dense contraction of
the head: M = 16*33600, N = K = 512; value_proj x 3 layers + enc_output forward, and the value projections' dX products), timed live with
events on the launch stream inside the timed steps, priced against the dense bf16 MFMA peak; `traffic` = HBM bytes per launch
from the PMC passes committed under profiles/ (same kernel, same shape).  `cpu_baseline`: the CPU oracle (oracle/, a port - the
reference's own end-to-end path cannot run on CPU) on this box's host cores, bounded sample: configs[0]'s 8 images, one
warm-up step on 2 of them + 1 timed fwd+bwd step on all 8.  `config.bf16_vs_fp32`: relative difference of the bf16-mode loss from the fp32-mode loss of the SAME
model on the SAME batch, measured before the warm-up (the GPU parity suite holds the fp32 mode to the CPU oracle at 1e-3).
`config.graph_vs_eager`: after the timed steps, ONE more forward + backward through the replayed graphs and one executed kernel by
kernel on the same weights, batch and seeds: relative difference of the loss and of the whole gradient (L2), next to the same
figure between two eager passes (`eager_noise_*`: MIOpen's split-K / atomic solvers are not bitwise reproducible).
`config.static_part_check`: the capture-time check (GraphedPart.verify: token memory and all 552 parameter gradients of a replay
against eager execution).  `host_cpu_ms_per_step`: process CPU time (user + system, all threads of the rank) per timed step, max and
per rank - the share of a step the host spends issuing work, which is what N ranks on one host compete for.
`config.conv_tuning`: how MIOpen picked the trunk's convolution kernels (tam-tr_amd/tuning.py: its own timed search, looked up in immediate mode from
the tables shipped in the repo).  `config.static_part`: "hip-graph" when the shape-static part of the step (trunk, VSS blocks, input projection: ~3/4 of the
launches) is replayed as two HIP graphs (model.capture_static_part), "eager" otherwise (--static-part eager).
For N > 1 the process group must be RCCL (`nccl`): the line records backend and world size, anything else is refused.
"""
import argparse
import json
import os
import sys
import time


import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense, /opt/skills/guides/MI355X_MICROARCH.md "Peak BF16/FP16 MFMA ~2.5 PF dense"
HBM_PEAK_GBS = 8000.0           # HBM3E, same guide ("~8 TB/s"); a device copy of the GEMM's 1.10 GB reaches 5.45 TB/s (profiles/r03_gemm_vs_copy.txt)
MFMA_F32_PEAK_TFLOPS = 157.3


def synth_batch(B, S, seed, device):
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(B, 3, S, S, generator=g)
    txt = torch.nn.functional.normalize(torch.randn(B, 10, 512, generator=g), dim=-1)
    n = 8
    cls = torch.randint(0, 10, (B * n, 1), generator=g).float()
    xy = 0.2 + 0.6 * torch.rand(B * n, 2, generator=g)
    wh = 0.02 + 0.2 * torch.rand(B * n, 2, generator=g)
    bidx = torch.arange(B).repeat_interleave(n).float()
    b = {'img': img, 'txt_feats': txt, 'cls': cls, 'bboxes': torch.cat([xy, wh], 1), 'batch_idx': bidx}
    # image + text embeddings resident in HBM; the (tiny) label tensors stay on the host, where the reference's trainer leaves them
    # (RTDETRTrainer.preprocess_batch), and are uploaded inside the step
    return {k: v.to(device) if k in ('img', 'txt_feats') else v for k, v in b.items()}


class KernelTimer:
    """Events on the launch stream (torch's current stream) tightly around the tamtr_linear_bf16 C call (ops.KERNEL_EVENTS)."""

    def __init__(self):
        self.enabled = False

    def install(self):
        import tamtr_amd.ops as ops
        self.ops = ops

    def __setattr__(self, k, v):
        object.__setattr__(self, k, v)
        if k == 'enabled' and hasattr(self, 'ops'):
            if v:
                self.ops.KERNEL_EVENTS['tamtr_linear_bf16'] = []
            else:
                self.events = self.ops.KERNEL_EVENTS.pop('tamtr_linear_bf16', getattr(self, 'events', []))

    def summary(self):
        """Launches of the roofline shape only (the largest GEMM = the value projection / enc_output shape); the kernel also
        runs the three smaller input-projection GEMMs, which are not mixed into this figure."""
        ev = getattr(self, 'events', [])
        if not ev:
            return None
        top = max(f for _, _, f in ev)
        ms = sorted(a.elapsed_time(b) for a, b, f in ev if f == top)
        avg = sum(ms) / len(ms)
        return {'launches': len(ms), 'avg_ms': avg, 'min_ms': ms[0], 'tflops': top / (avg * 1e-3) / 1e12}


def event_bracket_overhead(dev):
    """What a pair of events on the launch stream adds around ONE kernel on this runtime: the same bracket around a kernel that itself
    runs ~2 us (a 1 024-element ordered sum), median of 50, between other work as in the step.  The roofline's `avg_ms` is the raw bracket
    (conservative); rocprofv3's kernel durations (profiles/) are shorter by about this much."""
    import tamtr_amd.ops as ops
    t = torch.ones(4, 1024, device=dev)
    big = torch.empty(1 << 24, device=dev)
    ms = []
    for _ in range(50):
        big.zero_()                                  # something in front, so the start marker waits like in the step
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.slab_sum(t); e1.record()
        ms.append((e0, e1))
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) for a, b in ms)
    return v[len(v) // 2]


def _hbm_view(args, ks):
    if args.dtype != 'bf16':
        return None
    M = args.batch * (args.imgsz // 4) ** 2 * 21 // 16
    gbs = (M * 512 + M * 512) * 2 / (ks['avg_ms'] * 1e-3) / 1e9
    return {'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS}


def gemm_traffic(args):
    """HBM bytes per launch of the roofline kernel from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE and
    WRITE_SIZE in separate runs of tools/gemm_only.py, corrected as the MI355X guide prescribes).  Counters cannot be collected
    from inside this process; null when the committed measurement is not for this shape."""
    for name in ('r04_gemm_pmc.json', 'r03_gemm_pmc.json'):   # (the kernel has not changed since round 3; the newest measurement wins)
        try:
            d = json.load(open(os.path.join(ROOT, 'profiles', name)))
            if d.get('M') == args.batch * (args.imgsz // 4) ** 2 * 21 // 16 and args.dtype == 'bf16':
                return d['hbm_bytes_per_launch']
        except Exception:
            pass
    return None


def host_cores():
    """CPUs this process may actually use: affinity mask, capped by the cgroup CPU quota (the GPU boxes give a 1-GPU job a
    share of a large host) and by 32."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 32))


def cpu_baseline(n_img=8, S=640, repeats=1):
    """The CPU oracle (port) on this box's host cores: fwd+bwd of the same graph on n_img images (BASELINE configs[0]: 8).  Bounded
    sample: one untimed warm-up step on 2 images (thread pools, allocator, the C scan twin's library), then `repeats` timed steps
    on the n_img-image batch (about 40 s each on 16 cores)."""
    cores = host_cores()
    os.environ['OMP_NUM_THREADS'] = str(cores)  # the C scan twin's OpenMP runtime (loaded lazily below)
    torch.set_num_threads(cores)
    from oracle import selscan_c, specs, tamtr_oracle as O
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    from weights import fill_state
    print(f'[bench] cpu_baseline: CPU oracle on {cores} host threads, {n_img} image(s) {S}x{S}', file=sys.stderr, flush=True)
    st = fill_state(specs.tamtr_model(10, vss=True), 7)
    for k, v in st.items():
        if v.dtype.is_floating_point and not k.endswith(('running_mean', 'running_var')):
            v.requires_grad_()
    full = synth_batch(max(n_img, 2), S, 1, 'cpu')  # BatchNorm in train mode needs > 1 value per channel at the deepest maps

    def cut(n):
        keep = full['batch_idx'] < n
        return {'img': full['img'][:n], 'txt_feats': full['txt_feats'][:n], 'cls': full['cls'][keep], 'bboxes': full['bboxes'][keep],
                'batch_idx': full['batch_idx'][keep]}
    times = []
    for it in range(1 + repeats):
        for v in st.values():
            v.grad = None
        b = cut(min(2, n_img) if it == 0 else n_img)
        t0 = time.time()
        torch.manual_seed(0)
        O.tamtr_loss(st, b, True, scan_fn=selscan_c.scan)[0].backward()
        dt = time.time() - t0
        print(f'[bench] cpu_baseline: {"warm-up (2 images)" if it == 0 else "timed"} step {dt:.1f} s', file=sys.stderr, flush=True)
        if it:
            times.append(dt)
    mean = sum(times) / len(times)
    return {'value': n_img / mean, 'unit': 'images/sec', 'cores': cores, 'kind': 'port',
            'sample': f'{n_img} image(s) {S}x{S} (BASELINE configs[0]), fp32 CPU oracle (torch CPU ops + C scan twin): warm-up step on 2 images, '
                      f'then {repeats} timed fwd+bwd step(s) on {n_img}, mean {mean:.1f} s (min {min(times):.1f}, max {max(times):.1f}); a single-step figure on a '
                      'shared host: 43 - 93 s across boxes of this pool (+-2x) - context, not a target'}


def graph_vs_eager(model, batch, seed=4321):
    """One forward + backward of the training objective through the replayed graphs and one kernel by kernel: same weights, batch,
    denoising draw and DropPath draw.  BatchNorm statistics are put back; parameters' .grad is not touched."""
    saved = {k: v.detach().clone() for k, v in model.state_dict().items()}
    params = [p for p in model.parameters() if p.requires_grad]
    held = model._static

    def run(use_graph):
        model._static = held if use_graph else None
        try:
            torch.manual_seed(seed)
            loss, _ = model(batch)
            gr = torch.autograd.grad(loss, params, allow_unused=True)
            flat = torch.cat([g.detach().float().reshape(-1) for g in gr if g is not None])
        finally:
            model._static = held
            model.load_state_dict(saved)
        return float(loss.detach()), flat
    lg, gg = run(True)
    le, ge = run(False)
    le2, ge2 = run(False)
    n = float(ge.norm())
    return {'loss_graph': lg, 'loss_eager': le, 'loss_rel': abs(lg - le) / abs(le), 'grad_l2_rel': float((gg - ge).norm()) / n,
            'eager_noise_loss_rel': abs(le2 - le) / abs(le), 'eager_noise_grad_l2_rel': float((ge2 - ge).norm()) / n,
            'finite': bool(torch.isfinite(gg).all())}


def other_kernels(args, dev):
    """The other kernels VERDICT r3 names, timed live (standalone, events on the launch stream, after the timed loop): the channels-last
    text gate at the largest TIAGELAN site (HBM-bound: e + v + out) and the selective scan at MEH level 0 (the step's largest kernel:
    VALU-issue-bound; priced against HBM on its algorithmic bytes, which is why its fraction is low - DESIGN 4)."""
    import torch.nn as nn
    import tamtr_amd.ops as ops
    out = []

    def t_ms(fn, n=10, warm=3):
        for _ in range(warm):
            fn()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a, b in ev:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in ev)
        return sum(ms) / len(ms)
    B, H = args.batch, args.imgsz // 4
    with torch.no_grad():
        C, nh = 64, 2
        wide = torch.randn(B, 2 * C, H, H, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
        e, v = wide.chunk(2, 1)[1], torch.randn(B, C, H, H, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
        gk, bias = torch.randn(B, 10, C, device=dev), torch.zeros(nh, device=dev)
        bn = nn.BatchNorm2d(C, eps=1e-3, momentum=0.03).to(dev).train()
        st = torch.stack([torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5], 1).contiguous()
        ms = t_ms(lambda: ops.maxsigmoid_gate_cl(e, gk, bias, v, st, bn, nh), n=20)
        byt = 3 * v.numel() * 2
        out.append({'kernel': f'gate_cl_fwd_kernel<bf16> (BTA-PAN text gate, {C} ch x {H}^2 x {B}: TIAGELAN site 32)', 'bound': 'hbm', 'achieved': byt / ms / 1e6,
                    'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': byt / ms / 1e6 / HBM_PEAK_GBS, 'avg_ms': ms, 'algorithmic_bytes': byt})
        del wide, e, v
    # the scan in the form the step runs it (tamtr_selective_scan_dtproj_* on the cross-scan pair layout; planes in bf16 in bf16 mode)
    import tamtr_amd._lib as L_
    from tamtr_amd._lib import call, ptr, stream_ptr
    D, L, R, K, N = 256, H * H, 8, 4, 16
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *sh: torch.randn(*sh, device=dev, generator=g)   # noqa: E731
    pc = 1 if args.dtype == 'bf16' else 0
    pdt, e = (torch.bfloat16, 2) if pc else (torch.float32, 4)
    u2, g2 = rn(B, 2, D, L).to(pdt), rn(B, 2, D, L).to(pdt)
    dtr, Wdt, A, Bs, Cs, Dv, db = rn(B, K, R, L), rn(K * D, R) * R ** -0.5, -torch.exp(rn(K * D, N) * 0.3), rn(B, K, N, L), rn(B, K, N, L), rn(K * D), rn(K * D) - 3
    y, gu = torch.empty(B, K, D, L, device=dev, dtype=pdt), torch.empty(B, K * D, L, device=dev, dtype=pdt)
    chunk = L_.lib().tamtr_selective_scan_chunk()
    hst = torch.empty(B, K * D, (L + chunk - 1) // chunk, N, device=dev)
    gdelta = torch.empty(B, K * D, L, device=dev, dtype=torch.bfloat16 if pc else torch.float32)
    gdtr, gB, gC = torch.empty_like(dtr), torch.empty_like(Bs), torch.empty_like(Cs)
    grow = torch.empty(B, K * D, L_.lib().tamtr_selective_scan_row_sums(), device=dev)
    ws = torch.empty(2 * L_.lib().tamtr_selective_scan_bwd_slabs(D) * Bs.numel(), device=dev)
    sp = stream_ptr()
    f_ms = t_ms(lambda: call('tamtr_selective_scan_dtproj_fwd', ptr(u2), ptr(dtr), ptr(Wdt), ptr(A), ptr(Bs), ptr(Cs), ptr(Dv), ptr(db), ptr(y), ptr(hst), B, K, D, N,
                             R, L, 1, pc, sp), n=5, warm=2)
    b_ms = t_ms(lambda: call('tamtr_selective_scan_dtproj_bwd', ptr(g2), ptr(u2), ptr(dtr), ptr(Wdt), ptr(A), ptr(Bs), ptr(Cs), ptr(Dv), ptr(db), ptr(hst), ptr(gu),
                             ptr(gdelta), ptr(gdtr), ptr(grow), ptr(gB), ptr(gC), ptr(ws), B, K, D, N, R, L, 3, 3 if pc else 0, sp), n=5, warm=2)
    state_steps = B * K * D * L * N
    small = B * K * (R + 2 * N) * L * 4                       # dtr + B + C (or their gradients), f32
    pl = B * D * L                                            # elements of one [B, D, L] plane
    f_bytes = (2 + 4) * pl * e + small                        # u2 read, y written
    b_bytes = (2 + 2 + 4) * pl * e + 2 * 4 * pl * (2 if pc else 4) + 2 * small   # d(y), u2 read, d(u) written; d(delta) workspace written + read
    for name, ms, byt in (('selscan_fwd_kernel (MEH level 0: d_inner 256, L %d, 4 directions, 16 states; planes %s)' % (L, 'bf16' if pc else 'f32'), f_ms, f_bytes),
                          ('selscan_bwd (kernel + dtproj_gdtr + slab_sum, same shape)', b_ms, b_bytes)):
        out.append({'kernel': name, 'bound': 'hbm', 'limiter': 'VALU issue (fp32 recurrence: exp + 3 FMA per step and state, DPP scans)', 'achieved': byt / ms / 1e6,
                    'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': byt / ms / 1e6 / HBM_PEAK_GBS, 'avg_ms': ms, 'algorithmic_bytes': byt,
                    'state_steps_per_s': state_steps / (ms * 1e-3)})
    return out


def _brief(chk):
    if not chk:
        return None
    return {k: chk[k] for k in ('ok', 'conclusive', 'grads', 'informative_grads', 'out_rel_max', 'grad_l2_rel_max', 'eager_noise_out', 'eager_noise_grad_l2',
                                'bound_out', 'bound_grad_l2')}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=16, help='images per GPU')
    ap.add_argument('--imgsz', type=int, default=640)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--rccl-solo', action='store_true',
                    help='--gpus 1 only: run the N > 1 form of the step (gradient buckets, all-reduces from the hooks, barriers, the all-rank agreement) '
                         'on a ONE-rank RCCL process group - every collective goes through the real backend, nothing is exchanged.  A rehearsal of the '
                         'data-parallel code path on a one-GPU box (two ranks cannot share a device under RCCL), not the N = 1 measurement')
    ap.add_argument('--static-part', default='graph', choices=['graph', 'eager'],
                    help='trunk + VSS blocks + input projection replayed as two HIP graphs (forward, backward) or launched kernel by kernel')
    ap.add_argument('--conv-tuning', default='shipped', choices=['shipped', 'search', 'off'],
                    help="MIOpen solver choice for the trunk's convolutions: immediate mode on the tables of its own timed search shipped under "
                         'tam-tr_amd/tuned/miopen (default), a fresh search (minutes), or its heuristic')
    ap.add_argument('--conv-db', default=None, help='directory the search writes its tables to (--conv-tuning search)')
    ap.add_argument('--grad-dtype', default='fp32', choices=['fp32', 'bf16'], help='dtype of the gradient buckets on the wire (N > 1)')
    ap.add_argument('--weight-shadows', default='on', choices=['on', 'off'],
                    help='bf16 mode with --optim-step fused: the optimizer kernel maintains the bf16 copies of the weights (engine.FusedOptimStep(shadows=True))')
    ap.add_argument('--optim-step', default='fused', choices=['fused', 'torch'],
                    help='clip + AdamW + EMA as the table-driven kernels of csrc/optim.hip (engine.FusedOptimStep) or as the three torch calls')
    ap.add_argument('--no-graph-check', action='store_true', help='skip the graph-vs-eager step and the standalone roofline_other launches after the timed loop (profiling runs: keeps the trace to the timed steps)')
    ap.add_argument('--cpu-baseline-images', type=int, default=8, help='images of the CPU-oracle sample (BASELINE configs[0]: 8)')
    args = ap.parse_args()

    import tamtr_amd  # noqa: F401  (raises if the HIP library is missing; loading it does not touch the GPU)
    from tamtr_amd import dist as tdist
    n_dev = torch.cuda.device_count()    # (counting devices does not initialise the GPU)
    if args.gpus > max(n_dev, 1) and os.environ.get('TAMTR_BENCH_ALLOW_GLOO') != '1':
        # one rank per GPU: with fewer devices than ranks two ranks would share a card and the line would not be an N-GPU measurement
        raise SystemExit(f'bench.py --gpus {args.gpus}: this node shows {n_dev} GPU(s) (torch.cuda.device_count()); one rank per GPU is needed. '
                         'Nothing was launched.  (TAMTR_BENCH_ALLOW_GLOO=1 rehearses the code path with ranks sharing a device over gloo.)')
    plan = tdist.launch_plan(args.gpus, os.environ, sys.argv[1:], __file__)
    if plan is not None:     # `python bench.py --gpus N`: become the launcher of N ranks; nothing below runs in this process
        raise SystemExit(tdist.self_launch(plan))
    from tamtr_amd.model import RTDETRDetectionWorldModel
    # stdout carries ONE line, the JSON.  Native libraries write there too - RCCL prints a five-line version banner to stdout when rank 0
    # creates its communicator (seen on the one-rank rehearsal, profiles/r04_rccl_one_rank_rehearsal.txt) - so from here on file descriptor 1
    # is stderr for everybody, and the line goes to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    solo = bool(args.rccl_solo)
    if solo and args.gpus != 1:
        raise SystemExit('--rccl-solo is the one-rank rehearsal: --gpus 1')
    rank, local, world = tdist.init_from_env(backend='nccl' if solo else None, solo=solo)
    grouped = world > 1 or solo     # a process group exists: the step runs its data-parallel form
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}')
    backend = torch.distributed.get_backend() if grouped else None
    if world > 1 and backend != 'nccl' and os.environ.get('TAMTR_BENCH_ALLOW_GLOO') != '1':
        # (a gloo run on one shared GPU is a rehearsal of the code path, not a measurement: set TAMTR_BENCH_ALLOW_GLOO=1 to run it)
        raise SystemExit(f'--gpus {world} needs the RCCL process group (backend "nccl"), got "{backend}"')
    local = local % max(torch.cuda.device_count(), 1)  # (rehearsals with more ranks than GPUs share a device)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    from tamtr_amd.tuning import use_tuned_convolutions_ranked
    # before the first convolution; rank 0 seeds the shared per-user table directory while the others wait (tuning.py)
    conv_tuning = use_tuned_convolutions_ranked(args.conv_tuning, args.conv_db, rank=rank, world=world)
    torch.set_num_threads(min(8, host_cores()))  # host side = launch issue + a few tiny CPU ops (dn RNG, scipy LSA): no 128-thread pools
    torch.manual_seed(0)
    model = RTDETRDetectionWorldModel(nc=10).to(dev).train()
    model.autocast_dtype = torch.bfloat16 if args.dtype == 'bf16' else None
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-4, betas=(0.9, 0.999), fused=True)
    from tamtr_amd.engine import FusedOptimStep, ModelEMA
    ema = ModelEMA(model)   # the reference's optimizer_step ends with ema.update(model) on every rank (trainer.py:259,478-479)
    # clip_grad_norm_(0.1) + AdamW.step() + ema.update() as four launches over a device table (csrc/optim.hip); --optim-step torch: the three torch calls
    # --weight-shadows on: the update kernel also keeps the bf16 compute copies of the weights (what autocast casts per use)
    shadows = args.weight_shadows == 'on' and args.dtype == 'bf16'
    stepper = FusedOptimStep.create(model, opt, ema, max_norm=0.1, shadows=shadows) if args.optim_step == 'fused' else None
    reducer = None
    if grouped:
        reducer = tdist.GradReducer(model.named_parameters(), skip=lambda n: '.attn.' in n, late=lambda n: 'denoising_class_embed' in n,
                                    grad_dtype=torch.bfloat16 if args.grad_dtype == 'bf16' else None, always_collective=solo)
    batch = synth_batch(args.batch, args.imgsz, 1 + rank, dev)
    timer = KernelTimer()
    timer.install()

    def step():
        if reducer is not None:
            reducer.prepare()
        else:
            opt.zero_grad(set_to_none=True)
        loss, items = model(batch)
        loss.backward()
        if reducer is not None:
            reducer.finish()
        if stepper is not None:
            stepper.step()
        else:
            torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.grad is not None], max_norm=0.1)
            opt.step()
            ema.update(model)
        return loss

    def fence():
        if grouped:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def note(msg):
        if rank == 0:
            print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)

    # bf16 mode against fp32 mode on the same weights and batch (forward only, same denoising seed), before anything is trained
    mode_err = None
    t_probe = time.perf_counter()
    if args.dtype == 'bf16':
        saved = {k: v.clone() for k, v in model.state_dict().items()}   # BatchNorm statistics move with a training forward
        with torch.no_grad():
            losses = {}
            for name, dt in (('fp32', None), ('bf16', torch.bfloat16)):
                model.autocast_dtype = dt
                torch.manual_seed(1234)
                losses[name] = float(model(batch)[0])
                model.load_state_dict(saved)
        model.autocast_dtype = torch.bfloat16
        mode_err = {'loss_fp32': losses['fp32'], 'loss_bf16': losses['bf16'], 'rel': abs(losses['bf16'] - losses['fp32']) / abs(losses['fp32'])}
        del saved
        torch.manual_seed(0)
        torch.cuda.synchronize()
        note(f'bf16 against fp32 forward on the same batch: {time.perf_counter() - t_probe:.1f} s (first convolutions: MIOpen handle, tables, code objects)')
    static_part = 'eager'
    if args.static_part == 'graph':
        try:
            t1 = time.perf_counter()
            model.capture_static_part(batch['img'], batch['txt_feats'], log=note)   # raises when a replay does not reproduce eager
            static_part = 'hip-graph'
            note(f'static part captured in {time.perf_counter() - t1:.1f} s (includes MIOpen kernel selection)')
        except Exception as e:  # noqa: BLE001 - report and run eagerly; the JSON line says which it was
            model.release_static_part()
            static_part = f'eager (capture failed: {type(e).__name__}: {e})'[:200]
            note(static_part)
        if grouped:   # every rank runs the same form of the step (and the same checks with collectives in them afterwards): graphs only if ALL captured
            ok = torch.tensor([1 if static_part == 'hip-graph' else 0], device=dev, dtype=torch.int32)
            torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN)
            if int(ok) == 0 and static_part == 'hip-graph':
                model.release_static_part()
                static_part = 'eager (another rank could not capture its static part)'
                note(static_part)
    note(f'model built on {world} GPU(s), dtype {args.dtype}; warm-up ({args.warmup} steps; the first one includes MIOpen kernel selection)')
    for i in range(args.warmup):
        t1 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        note(f'warm-up step {i}: {time.perf_counter() - t1:.2f} s')
    fence()
    timer.enabled = True
    c0 = time.process_time()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    host_ms = (time.process_time() - c0) / args.steps * 1e3   # CPU time of this rank's process (all threads) per step
    timer.enabled = False
    host_all = [host_ms]
    if grouped:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
        h = torch.zeros(world, device=dev, dtype=torch.float64)
        h[rank] = host_ms
        torch.distributed.all_reduce(h)
        host_all = h.tolist()
    note(f'timed {args.steps} steps: {dt / args.steps * 1e3:.1f} ms/step (host CPU {max(host_all):.1f} ms/step)')
    gve = None
    if static_part == 'hip-graph' and not args.no_graph_check:   # the replayed path against kernel-by-kernel execution, at the end of the run (every rank: same work)
        if reducer is None:
            opt.zero_grad(set_to_none=True)
        try:
            gve = graph_vs_eager(model, batch)
        except Exception as e:  # noqa: BLE001 - the measurement stands; the line says the check did not run
            gve = {'error': f'{type(e).__name__}: {e}'[:200]}
        note(f'graph vs eager: {gve}')
        fence()
    if rank == 0:
        ks = timer.summary()
        peak = MFMA_BF16_PEAK_TFLOPS if args.dtype == 'bf16' else MFMA_F32_PEAK_TFLOPS
        out = {
            'metric': f'images/sec fwd+bwd @{args.imgsz}x{args.imgsz} bs={args.batch}/GPU', 'value': world * args.batch * args.steps / dt, 'unit': 'images/sec',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'host_cpu_ms_per_step': {'max': max(host_all), 'per_rank': [round(v, 2) for v in host_all], 'host_cores': host_cores()},
            'config': {'workload': f'TAM-TR (TAMTR.yaml graph, 42.1M params) train step fwd+loss+bwd+clip+AdamW+EMA, {args.imgsz}x{args.imgsz}, '
                                   f'bs {args.batch}/GPU, 10 text prompts, 8 GT/img, full BTA-PAN+MEH HIP path',
                       'global_batch': world * args.batch, 'parallelism': f'dp{world}',
                       'reduced_precision': 'bf16 (BASELINE configs[4] names fp16: this build serves every reduced-precision configuration as bf16 - same MFMA rate on gfx950, fp32 exponent range, no loss scaler; DESIGN 7)', 'final_loss': float(loss.detach()),
                       'bf16_vs_fp32': mode_err, 'dist_backend': backend, 'dist_world_size': world,
                       **({'rccl_solo_rehearsal': 'one-rank RCCL group: the data-parallel form of the step with every collective issued (%d gradient buckets per step), '
                                                  'nothing exchanged - not the N = 1 measurement' % len(reducer.buckets)} if solo else {}),
                       'grad_bucket_dtype': (args.grad_dtype if grouped else None), 'static_part': static_part,
                       'optim_step': 'fused (csrc/optim.hip)' if stepper is not None else 'torch', 'weight_shadows': bool(stepper is not None and stepper.use_shadows), 'static_part_check': _brief(getattr(model, 'static_part_check', None)), 'graph_vs_eager': gve, 'conv_tuning': conv_tuning,
                       'hbm_GiB_peak': round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)},
            'roofline': None if ks is None else {
                'bound': 'mfma', 'kernel': 'linear_bf16_wstat_kernel<512> (MEH value_proj x3 + enc_output, and the dX of the value projections, M=%d N=K=512)'
                                           % (args.batch * (args.imgsz // 4) ** 2 * 21 // 16),
                'achieved': ks['tflops'], 'peak': peak, 'unit': 'TFLOP/s', 'frac': ks['tflops'] / peak, 'traffic': gemm_traffic(args),
                'avg_ms': ks['avg_ms'], 'launches': ks['launches'],
                # (an event pair around a ~2 us kernel, measured the same way right here: what the bracket itself adds on this runtime)
                'event_bracket_ms_around_a_2us_kernel': event_bracket_overhead(dev),
                # the same launches priced against the OTHER roof (the shape sits at the chip's balance point: 256 flop/B against 312):
                # algorithmic bytes (X read once + Y written once) per second over the 8 TB/s HBM peak
                'hbm_view': _hbm_view(args, ks)},
        }
        if args.dtype == 'bf16' and not args.no_graph_check:   # (profiling runs keep the trace to the timed steps)
            try:
                out['roofline_other'] = other_kernels(args, dev)
            except Exception as e:  # noqa: BLE001 - an extra; the line stands without it
                out['roofline_other'] = {'error': f'{type(e).__name__}: {e}'[:200]}
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(args.cpu_baseline_images, args.imgsz)
        else:
            out['cpu_baseline'] = None
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    if grouped:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
