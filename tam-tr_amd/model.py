"""TAMTR graph + the BaseModel plugin boundary.  Reference: ultralytics/nn/tasks.py:28-43 (BaseModel.forward),
:518-672 (RTDETRDetectionWorldModel.predict / loss), :841-972 (parse_model) and cfg/models/TAMTR/TAMTR.yaml.

`model` is an nn.Sequential whose children carry `.i .f .type .np` exactly as parse_model attaches them, so state_dict
keys are `model.{i}.…` and reference checkpoints' state_dicts load with strict=True.
"""
import torch
import torch.nn as nn

from .backbone import CPAM, Concat, Conv, RepNCSPELAN4, SPPELAN, Upsample
from .head import ManbaWorldDecoder
from .hostio import stager
from .loss import RTDETRDetectionLoss
from .modules import TIAGELAN

# [from, module, args] - same rows as cfg/models/TAMTR/TAMTR.yaml:9-67 (there is one scale only)
TAMTR_SPEC = [
    [-1, 'Conv', [64, 3, 2]], [-1, 'Conv', [128, 3, 2]], [-1, 'RepNCSPELAN4', [256, 128, 64, 1]],
    [-1, 'Conv', [256, 3, 2]], [-1, 'RepNCSPELAN4', [512, 256, 128, 1]], [-1, 'Conv', [512, 3, 2]],
    [-1, 'RepNCSPELAN4', [512, 512, 256, 1]], [-1, 'Conv', [512, 3, 2]], [-1, 'RepNCSPELAN4', [512, 512, 256, 1]],
    [-1, 'SPPELAN', [512, 256]],
    [-1, 'Conv', [512, 1, 1]], [-1, 'Upsample', [2.0]], [6, 'Conv', [512, 1, 1]], [4, 'Conv', [512, 1, 1]],
    [-1, 'Upsample', [0.5]], [[-1, -3, -4], 'Concat', [1]], [-1, 'TIAGELAN', [512, 512, 256, 1, 8]], [-1, 'CPAM', []],
    [-1, 'Conv', [256, 1, 1]], [-1, 'Upsample', [2.0]], [4, 'Conv', [256, 1, 1]], [2, 'Conv', [256, 1, 1]],
    [-1, 'Upsample', [0.5]], [[-1, -3, -4], 'Concat', [1]], [-1, 'TIAGELAN', [256, 256, 128, 1, 4]], [-1, 'CPAM', []],
    [-1, 'Conv', [128, 1, 1]], [-1, 'Upsample', [2.0]], [2, 'Conv', [128, 1, 1]], [0, 'Conv', [128, 1, 1]],
    [-1, 'Upsample', [0.5]], [[-1, -3, -4], 'Concat', [1]], [-1, 'TIAGELAN', [128, 128, 64, 1, 2]], [-1, 'CPAM', []],
    [-1, 'Conv', [128, 3, 2]], [[-1, 24], 'Concat', [1]], [-1, 'TIAGELAN', [256, 256, 128, 1, 4]], [-1, 'CPAM', []],
    [-1, 'Conv', [256, 3, 2]], [[-1, 16], 'Concat', [1]], [-1, 'TIAGELAN', [512, 512, 256, 1, 8]],
    [[32, 36, 40], 'ManbaWorldDecoder', ['nc', 512, 100, 4, 8, 3]],
]


def build_graph(spec, ch=3, nc=10):
    """Instantiate the rows; returns (nn.Sequential, save list).  Channel bookkeeping as parse_model (tasks.py:886-955)."""
    chans, layers, save = [], [], []
    for i, (f, name, args) in enumerate(spec):
        cin = ch if i == 0 else (chans[f] if isinstance(f, int) else None)
        if name == 'Conv':
            m, c2 = Conv(cin, *args), args[0]
        elif name == 'RepNCSPELAN4':
            m, c2 = RepNCSPELAN4(cin, *args), args[0]
        elif name == 'TIAGELAN':
            m, c2 = TIAGELAN(cin, *args), args[0]
        elif name == 'SPPELAN':
            m, c2 = SPPELAN(cin, *args), args[0]
        elif name == 'Upsample':
            m, c2 = Upsample(None, args[0], 'nearest'), cin
        elif name == 'Concat':
            m, c2 = Concat(*args), sum(chans[j] for j in f)
        elif name == 'CPAM':
            m, c2 = CPAM(cin), cin
        elif name == 'ManbaWorldDecoder':
            a = [nc if v == 'nc' else v for v in args]
            m, c2 = ManbaWorldDecoder(a[0], [chans[j] for j in f], *a[1:]), None
        elif name == 'Detect':      # the yolo head rows `[[..], 1, Detect, [nc]]` (tasks.py:923-924): args = [nc], ch from `f`
            from .detect import Detect
            m, c2 = Detect(nc if args[0] == 'nc' else args[0], [chans[j] for j in f]), None
        else:
            raise ValueError(name)
        m.i, m.f, m.type = i, f, name
        m.np = sum(p.numel() for p in m.parameters())
        save.extend(x % i for x in ([f] if isinstance(f, int) else f) if x != -1)
        layers.append(m)
        chans.append(c2)
    return nn.Sequential(*layers), sorted(set(save))


import os
# NHWC trunk on the GPU: MIOpen's bf16 convolutions are NHWC kernels and wrap every NCHW operand in a transpose (483 launches,
# 7.75 ms of the 640 px / bs 16 step).  Channels-last activations + channels-last k x k weights take them as they are (needs
# PYTORCH_MIOPEN_SUGGEST_NHWC=1, set by the package __init__).  TAMTR_CHANNELS_LAST=0 restores NCHW, and so does the deterministic mode
# (TAMTR_DETERMINISTIC=1) unless told otherwise: MIOpen 3.5 has no deterministic solver for NHWC bf16 convolutions other than its naive
# kernels (17 s per 16-image step), for NCHW it has (0.25 s).  Per model: set_channels_last().
_CHANNELS_LAST = os.environ.get('TAMTR_CHANNELS_LAST', '0' if os.environ.get('TAMTR_DETERMINISTIC') == '1' else '1') != '0'


class _CastGroup(torch.autograd.Function):
    """fp32 master conv weights of one graph layer -> bf16 compute copies in ONE multi-tensor kernel, and their gradients back
    to fp32 in one.  autocast casts every weight (and every weight gradient) with its own ~13 us kernel: 400 + 340 launches
    and ~10 ms of GPU time per step on this graph.  One group per top-level layer, so weight gradients still become
    available layer by layer during the backward (the DP reducer overlaps its all-reduce with what is left)."""

    @staticmethod
    def forward(ctx, *ws):
        ctx.set_materialize_grads(False)  # a copy nobody differentiates through (the discarded gate's proj_conv) keeps grad None
        outs = [torch.empty_like(w, dtype=torch.bfloat16) for w in ws]
        torch._foreach_copy_(outs, list(ws))
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        live = [i for i, g in enumerate(gs) if g is not None]  # e.g. the gate's proj_conv: evaluated, never used
        outs = [torch.empty_like(gs[i], dtype=torch.float32) for i in live]
        if live:
            torch._foreach_copy_(outs, [gs[i] for i in live])
        res = [None] * len(gs)
        for i, o in zip(live, outs):
            res[i] = o
        return tuple(res)


class _ShadowGroup(torch.autograd.Function):
    """_CastGroup when the optimizer keeps the bf16 copies up to date itself (engine.FusedOptimStep(shadows=True)): the forward hands out
    aliases of those copies - no kernel - and the backward is _CastGroup's (one multi-tensor cast of the bf16 gradients to fp32)."""

    @staticmethod
    def forward(ctx, n, *ts):
        ctx.set_materialize_grads(False)
        return tuple(s.view_as(s) for s in ts[n:])

    @staticmethod
    def backward(ctx, *gs):
        return (None,) + _CastGroup.backward(ctx, *gs) + (None,) * len(gs)


def _cast_group(masters):
    from . import ops
    sh = [ops.bf16_shadow(p) for p in masters]
    if all(s is not None for s in sh):
        return _ShadowGroup.apply(len(masters), *masters, *sh)
    return _CastGroup.apply(*masters)


def _conv_weight_names(m):
    names = getattr(m, '_tamtr_conv_names', None)
    if names is None:
        names = [f'{n}.weight' if n else 'weight' for n, sub in m.named_modules() if isinstance(sub, nn.Conv2d)]
        m._tamtr_conv_names = names
    return names


class _StaticPart(nn.Module):
    """token_memory() as a module that owns exactly the parameters it uses (what make_graphed_callables differentiates)."""

    def __init__(self, model):
        super().__init__()
        head = model.model[-1]
        self.trunk, self.vss, self.proj = nn.ModuleList(model.model[:-1]), head.VSSBlocks, head.input_proj
        object.__setattr__(self, '_owner', model)

    def forward(self, img, txt, drop_scales):
        return self._owner.token_memory(img, txt, autocast_cache=False, drop_scales=drop_scales)[0]


class RTDETRDetectionWorldModel(nn.Module):
    """forward(dict) -> (loss, loss_items)   [training batch: img, txt_feats, cls, bboxes, batch_idx]
       forward(tensor) -> predictions        [uses self.txt_feats set in advance]"""

    def __init__(self, cfg=None, ch=3, nc=10, verbose=False):
        super().__init__()
        self.yaml = {'nc': nc, 'ch': ch, 'spec': 'TAMTR'}
        self.nc = nc
        self.names = {i: f'{i}' for i in range(nc)}
        self.txt_feats = torch.randn(1, nc, 512)
        self.model, self.save = build_graph(cfg or TAMTR_SPEC, ch, nc)
        self.channels_last = False
        self.set_channels_last(_CHANNELS_LAST)
        self.stride = torch.Tensor([32])
        self.autocast_dtype = None  # torch.bfloat16 => bf16 activations through the trunk and the head GEMMs

    def set_channels_last(self, flag=True):
        """Trunk layout of THIS model: NHWC activations + channels-last k x k convolution weights (the default on the GPU), or NCHW
        (values, state_dict keys and shapes are the same either way; the deterministic mode runs NCHW, see _CHANNELS_LAST).  Recorded
        graphs belong to the layout they were recorded in: they are dropped."""
        self.release_static_part()
        fmt = torch.channels_last if flag else torch.contiguous_format
        for m in self.model[:-1].modules():
            if isinstance(m, nn.Conv2d) and m.kernel_size != (1, 1):
                m.weight.data = m.weight.data.contiguous(memory_format=fmt)
        self.channels_last = bool(flag)
        return self

    def forward(self, x, *args, **kwargs):
        if isinstance(x, dict):
            return self.loss(x, *args, **kwargs)
        return self.predict(x, *args, **kwargs)

    def init_criterion(self):
        return RTDETRDetectionLoss(nc=self.nc, use_vfl=True)

    def set_text_features(self, txt_feats):
        """Offline vocabulary (stands in for set_classes(), tasks.py:552-571, whose CLIP encoder is out of scope)."""
        self.txt_feats = txt_feats.reshape(-1, txt_feats.shape[-2], txt_feats.shape[-1])
        self.model[-1].nc = self.txt_feats.shape[1]

    def fuse(self, verbose=False):
        """Evaluation graph (nn/tasks.py:121-152, what valTAMTR.py / AutoBackend(fuse=True) run): every RepConvN collapses to one
        3x3 conv, every Conv's BatchNorm folds into its conv (the gates' proj_conv included); the head's input_proj
        (Sequential(Conv2d, BatchNorm2d), not a Conv) stays as it is.  Weights are frozen; not for training."""
        from .backbone import Conv, RepConvN
        if not self.is_fused():
            for m in list(self.modules()):
                if isinstance(m, RepConvN):
                    m.switch_to_deploy()
            for m in list(self.modules()):
                if isinstance(m, Conv):
                    m.fuse()
                if hasattr(m, '_tamtr_conv_names'):
                    del m._tamtr_conv_names
        return self

    def is_fused(self, thresh=10):
        return sum(isinstance(v, nn.BatchNorm2d) for v in self.modules()) < thresh

    def token_memory(self, x, txt, autocast_cache=True, drop_scales=None):
        """Trunk -> VSS blocks -> input projection: the part of the step whose shapes follow from the image size alone (what
        capture_static_part() records as HIP graphs).  Returns the token memory [B, L, hd] and the level shapes.
        drop_scales [3, 2, B]: this step's DropPath factors (head.draw_drop_scales); None = the blocks draw their own."""
        head = self.model[-1]
        with torch.autocast('cuda', dtype=self.autocast_dtype or torch.bfloat16, enabled=self.autocast_dtype is not None,
                            cache_enabled=autocast_cache):
            if self.channels_last and x.is_cuda:
                x = x.contiguous(memory_format=torch.channels_last)
            from . import ops
            counters = ops.begin_bn_counter_batch() if self.training else None
            y = []
            grouped_cast = self.autocast_dtype == torch.bfloat16 and x.is_cuda and torch.is_grad_enabled()
            for m in self.model[:-1]:
                if m.f != -1:
                    x = y[m.f] if isinstance(m.f, int) else [x if j == -1 else y[j] for j in m.f]
                args = (x, txt) if isinstance(m, TIAGELAN) else (x,)
                names = _conv_weight_names(m) if grouped_cast else ()
                if names:  # this layer's conv weights as bf16 copies from one kernel (see _CastGroup); same values autocast would use
                    masters = [m.get_parameter(n) for n in names]
                    w16 = _cast_group(masters)
                    for c, p in zip(w16, masters):   # ops.conv2d_module sends a 1x1 convolution's fp32 weight gradient straight to the master
                        c._tamtr_master = p
                    x = torch.func.functional_call(m, dict(zip(names, w16)), args)
                else:
                    x = m(*args)
                y.append(x if m.i in self.save else None)
            feats, shapes = head.encode([y[j] for j in head.f], drop_scales)
            if counters is not None:
                ops.end_bn_counter_batch()  # num_batches_tracked += 1 of every BatchNorm that ran, in one multi-tensor kernel
        return feats, shapes

    def capture_static_part(self, img, txt_feats, warmup=3, verify=True, log=None):
        """Record token_memory() - forward and backward - as two HIP graphs for this image shape, dtype mode and training state
        (graphs.GraphedPart): ~3/4 of the step's kernel launches become two graph launches.  Everything after
        the token memory (query selection, denoising groups, decoder, loss) has shapes that follow the labels and stays eager.
        `img` / `txt_feats`: tensors of the shapes the training loop will pass.  Undo with release_static_part().

        The model is left exactly as it was found: BatchNorm running statistics / counters moved by the warm-up passes are put back and
        no random numbers are drawn (the DropPath factors are an INPUT of the recorded function; the step draws them eagerly).
        verify=True: the recorded graphs are replayed once and held to an eager forward + backward on the same inputs
        (GraphedPart.verify: token memory and every parameter gradient); on a mismatch - or when eager execution itself is too noisy for the
        comparison to decide anything (`conclusive: False`; verify='loose' accepts that case) - the graphs are dropped, the reason is logged
        and RuntimeError is raised - callers (engine.fit, bench.py, tools/train.py) then run eagerly.  The result is kept in
        `self.static_part_check`."""
        import torch.version
        from .graphs import GraphedPart, VALIDATED_HIP
        self.release_static_part()
        part = _StaticPart(self)
        part.train(self.training)
        txt = txt_feats.to(device=img.device, dtype=torch.float32)
        if len(txt) != len(img):
            txt = txt.repeat(len(img), 1, 1)
        saved = [b.detach().clone() for b in part.buffers()]
        head = self.model[-1]
        dp = torch.ones(head.num_Blocks, 2, len(img), device=img.device)
        with torch.no_grad():
            _, shapes = self.token_memory(img, txt, drop_scales=dp)  # the level shapes (DropPath factors given: nothing is drawn)
        graphed = GraphedPart(part, (img.detach(), txt.detach(), dp), warmup=warmup, log=log)
        try:
            import time
            t0 = time.perf_counter()
            self.static_part_check = graphed.verify() if verify else None
            if log is not None and verify:
                log(f'graph capture: replay check against eager execution {time.perf_counter() - t0:.1f} s')
        finally:
            with torch.no_grad():
                for b, v in zip(part.buffers(), saved):
                    b.copy_(v)
        if verify and not self.static_part_check['ok']:
            chk = self.static_part_check
            msg = (f"HIP-graph replay of the static part does not reproduce eager execution (token memory rel {chk['out_rel_max']:.2e} / bound "
                   f"{chk['bound_out']:.1e}, all gradients rel {chk['grad_l2_rel_max']:.2e} / bound {chk['bound_grad_l2']:.1e}, worst informative tensor "
                   f"{chk['replays'][-1]['worst_informative_grad']} at {chk['replays'][-1]['worst_informative_over_bound']:.1f} x its bound; "
                   f"HIP {torch.version.hip}): running eagerly")
            (log or print)(msg)
            raise RuntimeError(msg)
        if verify and not self.static_part_check['conclusive']:
            # eager execution itself is not reproducible here (MIOpen's heuristic solvers, tables of another build): the comparison can only
            # tell "finite and within an order of magnitude", which a replay that is wrong by a few hundred per cent passes.  That is not
            # accepted silently: the graphs are dropped unless the caller asked for the loose check (verify='loose').
            chk = self.static_part_check
            msg = (f"HIP-graph replay check inconclusive: eager execution itself is not reproducible here (whole-gradient difference between two "
                   f"eager runs {chk['eager_noise_grad_l2']:.2e}, replay against eager {chk['grad_l2_rel_max']:.2e}); only finiteness and order of "
                   "magnitude could be checked - TAMTR_DETERMINISTIC=1 or the shipped convolution tables give a conclusive comparison")
            if verify != 'loose':
                (log or print)(msg + ": running eagerly (capture_static_part(verify='loose') accepts such a capture)")
                raise RuntimeError(msg)
            if log is not None:
                log(msg + " - accepted (verify='loose')")
        if log is not None and not any(str(torch.version.hip).startswith(v) for v in VALIDATED_HIP):
            log(f'HIP-graph replay: runtime {torch.version.hip} is not one this package was validated on {VALIDATED_HIP}; the replay check passed')
        self._static = (graphed, tuple(img.shape), img.dtype, self.autocast_dtype, self.training, shapes, tuple(txt.shape))
        return self

    def release_static_part(self):
        self._static = None

    def predict(self, x, profile=False, visualize=False, batch=None, augment=False, txt_feats=None):
        txt = (self.txt_feats if txt_feats is None else txt_feats).to(device=x.device, dtype=torch.float32)
        if len(txt) != len(x):
            txt = txt.repeat(len(x), 1, 1)
        head = self.model[-1]
        st = getattr(self, '_static', None)
        # this step's DropPath factors of the VSS blocks: one draw on the ordinary generator, the same for the replayed and the eager path
        dp = head.draw_drop_scales(len(x), x.device) if self.training and x.is_cuda else None
        if st is not None and st[1:5] == (tuple(x.shape), x.dtype, self.autocast_dtype, self.training) and st[6] == tuple(txt.shape) \
                and torch.is_grad_enabled():
            feats, shapes = st[0](x, txt, dp), st[5]
        else:  # other shapes (a tail batch, another prompt count), evaluation, no_grad: kernel by kernel
            feats, shapes = self.token_memory(x, txt, drop_scales=dp)
        with torch.autocast('cuda', dtype=self.autocast_dtype or torch.bfloat16, enabled=self.autocast_dtype is not None):
            return head.decode(feats, shapes, txt.clone(), batch)

    def loss(self, batch, preds=None):
        if not hasattr(self, 'criterion'):
            self.criterion = self.init_criterion()
        img = batch['img']
        dev = img.device
        # Labels arrive on the HOST in the reference's trainer (RTDETRTrainer.preprocess_batch regroups them on the CPU); keep
        # the host copies for the ragged group sizes and the denoising RNG and upload through the pinned ring, so that nothing in
        # the step waits for the GPU.  Device-resident labels are accepted too, at the cost of one synchronising read-back.
        host = {'cls': batch['cls'].detach().cpu().long().view(-1), 'bboxes': batch['bboxes'].detach().cpu().float(),
                'batch_idx': batch['batch_idx'].detach().cpu().long().view(-1)}
        counts = torch.bincount(host['batch_idx'], minlength=len(img)).tolist()
        st = stager()
        st.next_step()
        targets = {k: (batch[k].to(dev, host[k].dtype).view(host[k].shape) if batch[k].is_cuda else st.h2d(host[k], dev)) for k in host}
        targets.update(gt_groups=counts, host=host)
        preds = self.predict(img, batch=targets, txt_feats=batch['txt_feats']) if preds is None else preds
        dec_bboxes, dec_scores, enc_bboxes, enc_scores, dn_meta = preds if self.training else preds[1]
        dn_bboxes = dn_scores = None
        if dn_meta is not None:
            dn_bboxes, dec_bboxes = torch.split(dec_bboxes, dn_meta['dn_num_split'], dim=2)
            dn_scores, dec_scores = torch.split(dec_scores, dn_meta['dn_num_split'], dim=2)
        dec_bboxes = torch.cat([enc_bboxes.unsqueeze(0).to(dec_bboxes.dtype), dec_bboxes])
        dec_scores = torch.cat([enc_scores.unsqueeze(0).to(dec_scores.dtype), dec_scores])
        terms = self.criterion((dec_bboxes, dec_scores), targets, dn_bboxes=dn_bboxes, dn_scores=dn_scores, dn_meta=dn_meta)
        self.last_loss_terms = {k: v.detach() for k, v in terms.items()}  # detached: must not keep the step's autograd graph alive
        items = torch.stack([terms[k].detach() for k in ('loss_giou', 'loss_class', 'loss_bbox')])
        return torch.stack(list(terms.values())).sum(), items  # 12 terms (loss.py:384-416): one stack + one sum
