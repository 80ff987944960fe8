"""Train / validate harness around the hot path: the pieces of the reference's engine that decide numbers.

  build_optimizer      ultralytics/engine/trainer.py:624-681   (three parameter groups: biases, norm weights, decayed weights)
  lr schedule, warm-up ultralytics/engine/trainer.py:276-279,330-340
  train_step           ultralytics/engine/trainer.py:343-357,471-479   (autocast forward, backward, clip 0.1, step, EMA)
  ModelEMA             ultralytics/utils/torch_utils.py:392-419
  postprocess          ultralytics/models/rtdetrworld/val.py:102-128  (conf filter, class-offset NMS; torchvision is not in the
                       image: the greedy NMS is restated here)
  match_predictions    ultralytics/engine/validator.py:208-247, models/yolo/detect/val.py:169-183
  ap_per_class, compute_ap, smooth, box_iou   ultralytics/utils/metrics.py:49-68,941-946,999-1029,1032-1128
  Validator            models/rtdetrworld/val.py:130-173 + models/yolo/detect/val.py get_stats

  fit                  ultralytics/engine/trainer.py:262-281,285-420 (the epoch loop: warm-up, step, schedule, EMA validation,
                       last / best checkpoints) over data.py's loaders

Host-side PyTorch / numpy by design (the reference's is too).  Batches arrive as tensors (img, txt_feats, cls, bboxes, batch_idx
[, ori_shape]) - from data.py's loaders or synthetic; the CLIP text encoder stays out of scope (precomputed table, data.py).
"""
import math
import os
import time
from copy import deepcopy

import numpy as np
import torch
import torch.nn as nn


# ------------------------------------------------------------------------------------------------ training side
def build_optimizer(model, name='AdamW', lr=0.001, momentum=0.9, decay=1e-5, iterations=1e5):
    """Parameter groups exactly as the reference: [0] every '*bias*' (no decay), then weights with decay, then weights of
    normalisation layers (no decay).  name='auto' picks SGD(0.01) above 10 000 iterations, else AdamW(lr = 0.002*5/(4+nc))."""
    g = [], [], []
    norms = tuple(v for k, v in nn.__dict__.items() if 'Norm' in k)
    if name == 'auto':
        nc = getattr(model, 'nc', 10)
        lr_fit = round(0.002 * 5 / (4 + nc), 6)
        name, lr, momentum = ('SGD', 0.01, 0.9) if iterations > 10000 else ('AdamW', lr_fit, 0.9)
    for module_name, module in model.named_modules():
        for param_name, param in module.named_parameters(recurse=False):
            fullname = f'{module_name}.{param_name}' if module_name else param_name
            if 'bias' in fullname:
                g[2].append(param)
            elif isinstance(module, norms):
                g[1].append(param)
            else:
                g[0].append(param)
    if name in ('Adam', 'Adamax', 'AdamW', 'NAdam', 'RAdam'):
        opt = getattr(torch.optim, name)(g[2], lr=lr, betas=(momentum, 0.999), weight_decay=0.0)
    elif name == 'RMSProp':
        opt = torch.optim.RMSprop(g[2], lr=lr, momentum=momentum)
    elif name == 'SGD':
        opt = torch.optim.SGD(g[2], lr=lr, momentum=momentum, nesterov=True)
    else:
        raise NotImplementedError(f"Optimizer '{name}' not found in [Adam, AdamW, NAdam, RAdam, RMSProp, SGD, auto]")
    opt.add_param_group({'params': g[0], 'weight_decay': decay})
    opt.add_param_group({'params': g[1], 'weight_decay': 0.0})
    return opt


def linear_lr(epochs, lrf):
    """lf(epoch): 1 -> lrf linearly (trainer.py:278)."""
    return lambda x: (1 - x / epochs) * (1.0 - lrf) + lrf


def warmup(optimizer, ni, nw, lf_epoch, warmup_bias_lr=0.0, warmup_momentum=0.8, momentum=0.937):
    """Linear warm-up of lr (bias group from warmup_bias_lr, others from 0) and momentum over the first nw iterations
    (trainer.py:330-340; group 0 is the bias group)."""
    if ni > nw:
        return
    for j, x in enumerate(optimizer.param_groups):
        x['lr'] = float(np.interp(ni, [0, nw], [warmup_bias_lr if j == 0 else 0.0, x['initial_lr'] * lf_epoch]))
        if 'momentum' in x:
            x['momentum'] = float(np.interp(ni, [0, nw], [warmup_momentum, momentum]))


class ModelEMA:
    """EMA of every floating-point state_dict entry, decay ramp d(n) = decay * (1 - exp(-n / tau))."""

    def __init__(self, model, decay=0.9999, tau=2000, updates=0):
        self.ema = deepcopy(model).eval()
        self.updates = updates
        self.decay = lambda x: decay * (1 - math.exp(-x / tau))
        for p in self.ema.parameters():
            p.requires_grad_(False)
        self.enabled = True
        self._pairs = None      # (key, ema tensors, model tensors): walking two 1 200-entry state_dicts costs ~10 ms of host time per step

    def _tensors(self, model):
        first = next(model.parameters())
        key = (id(model), first.data_ptr(), len(model._modules))
        if self._pairs is None or self._pairs[0] != key:
            msd = model.state_dict()
            dst, src = [], []
            for k, v in self.ema.state_dict().items():
                if v.dtype.is_floating_point:
                    dst.append(v)
                    src.append(msd[k].detach())
            self._pairs = (key, dst, src)
        return self._pairs[1], self._pairs[2]

    @torch.no_grad()
    def update(self, model):
        if not self.enabled:
            return
        self.updates += 1
        d = self.decay(self.updates)
        dst, src = self._tensors(model)        # (parameters and buffers are updated in place, so the aliases stay valid)
        torch._foreach_mul_(dst, d)            # v = d * v + (1 - d) * m, a few multi-tensor kernels instead of 2 per tensor
        torch._foreach_add_(dst, src, alpha=1 - d)


class FusedOptimStep:
    """The reference's optimizer_step (ultralytics/engine/trainer.py:471-479): clip_grad_norm_(max_norm) -> optimizer.step() ->
    ema.update(model), for torch.optim.AdamW on the GPU, as four kernel launches over a device table that is built once
    (csrc/optim.hip tamtr_optim_step).  torch spends 9.4 ms of host time per step on the same work (grouping ~750 tensors into lists for
    its multi-tensor kernels, three times over: profiles/r04_host_phases.txt), and the host is this step's critical path.

    The optimizer object stays the owner of its state: `optimizer.state[p]` holds exp_avg / exp_avg_sq / step tensors as torch.optim.AdamW
    would have created them (state_dict() / load_state_dict() work; the step counts are 0-d views of one flat tensor), `param_groups`
    is read every step (warm-up and schedules change lr), the EMA object keeps its `updates` counter and decay ramp.
    Same arithmetic as torch's fused AdamW kernel and as ModelEMA.update (fp32); the gradient norm stays on the device.

        stepper = FusedOptimStep.create(model, optimizer, ema, max_norm=0.1)     # None when the combination is not served
        ...backward...;  stepper.step()        # instead of clip_grad_norm_ + optimizer.step() + ema.update(model)
    """

    @staticmethod
    def create(model, optimizer, ema=None, max_norm=0.1, shadows=False):
        """shadows=True: the update kernel also stores every new parameter value rounded to bf16 (`p._tamtr_bf16`, ops.bf16_of): the compute
        copies that bf16 autocast would cast out of the fp32 masters at every use in the next step.  They follow the masters as long as the
        masters are changed by THIS object's step() (or by load_state_dict on the model: a hook refreshes them); ops.bf16_of re-checks the
        tensor version on every use and falls back to a cast, so a stale copy is never read on the eager path - recorded graphs read the
        shadows' memory directly, so re-capture (model.capture_static_part) after switching shadows on or off."""
        ps = [p for g in optimizer.param_groups for p in g['params']]
        ok = (type(optimizer) is torch.optim.AdamW and len(optimizer.param_groups) <= 4 and ps
              and all(p.is_cuda and p.dtype == torch.float32 and (p.is_contiguous() or p.is_contiguous(memory_format=torch.channels_last)) for p in ps)
              and not any(g.get('amsgrad') or g.get('maximize') or g.get('capturable') or g.get('differentiable') for g in optimizer.param_groups))
        if not ok:
            return None
        st = FusedOptimStep(model, optimizer, ema, max_norm, shadows)
        if shadows:
            st._build()          # the copies exist before the first forward (and before a graph capture records their addresses)
        return st

    def __init__(self, model, optimizer, ema=None, max_norm=0.1, shadows=False):
        self.model, self.opt, self.ema, self.max_norm = model, optimizer, ema, float(max_norm)
        self.use_shadows = bool(shadows)
        self.shadows = []        # (parameter, bf16 copy) pairs
        self._key = None
        self._hook = None

    def refresh_shadows(self):
        """Re-derive every bf16 copy from its master (after the masters were changed by anything but step())."""
        with torch.no_grad():
            if self.shadows:
                torch._foreach_copy_([s for _, s in self.shadows], [p for p, _ in self.shadows])
            for p, s in self.shadows:
                s._tamtr_version = p._version

    def drop_shadows(self):
        for p, s in self.shadows:
            s._tamtr_version = -1     # (a recorded graph that still reads this copy re-derives it before every replay: graphs.GraphedPart.__call__)
            if hasattr(p, '_tamtr_bf16'):
                del p._tamtr_bf16
        self.shadows = []
        if self._hook is not None:
            self._hook.remove()
            self._hook = None

    def _state_key(self):
        first = self.opt.param_groups[0]['params'][0]
        st = self.opt.state.get(first, {})
        e = None if self.ema is None else next(self.ema.ema.parameters()).data_ptr()
        return (first.data_ptr(), st['exp_avg'].data_ptr() if 'exp_avg' in st else 0, e, sum(len(g['params']) for g in self.opt.param_groups))

    @torch.no_grad()
    def _build(self):
        from . import _lib
        dev = self.opt.param_groups[0]['params'][0].device
        entries = []                                   # (tensor, group, has_adam)
        for gi, g in enumerate(self.opt.param_groups):
            entries += [(p, gi, True) for p in g['params']]
        opt_ptrs = {p.data_ptr() for p, _, _ in entries}
        ema_of = {}
        if self.ema is not None:
            msd, esd = self.model.state_dict(), self.ema.ema.state_dict()
            by_ptr = {v.data_ptr(): k for k, v in msd.items() if v.dtype.is_floating_point}
            for p, _, _ in entries:
                k = by_ptr.get(p.data_ptr())
                if k is not None:
                    ema_of[id(p)] = esd[k]
            for k, v in msd.items():   # floating-point buffers (BatchNorm statistics) and parameters outside the optimizer: EMA only
                if v.dtype.is_floating_point and v.is_cuda and v.data_ptr() not in opt_ptrs:
                    entries.append((v, 0, False))
                    ema_of[id(v)] = esd[k]
        n = len(entries)
        steps = torch.zeros(n, device=dev, dtype=torch.float32)
        P, M, V, E, SH, numel, group = [], [], [], [], [], [], []
        old_sh = {id(p): s for p, s in self.shadows}
        self.shadows = []
        for i, (t, gi, adam) in enumerate(entries):
            sh = None
            if adam and self.use_shadows:
                sh = old_sh.get(id(t))
                if sh is None or sh.shape != t.shape or sh.stride() != t.stride() or sh.device != t.device:
                    sh = torch.empty_like(t, dtype=torch.bfloat16, memory_format=torch.preserve_format)
                t._tamtr_bf16 = sh
                self.shadows.append((t, sh))
            SH.append(sh.data_ptr() if sh is not None else 0)
            m = v = None
            if adam:
                st = self.opt.state[t]
                if 'exp_avg' not in st:               # what torch.optim.AdamW._init_group creates on its first step
                    st['exp_avg'] = torch.zeros_like(t, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(t, memory_format=torch.preserve_format)
                elif 'step' in st:
                    steps[i] = float(st['step'])
                st['step'] = steps[i]                 # 0-d view: state_dict() / load_state_dict() see an ordinary step tensor
                m, v = st['exp_avg'], st['exp_avg_sq']
                if m.stride() != t.stride() or v.stride() != t.stride():
                    raise ValueError('FusedOptimStep: optimizer state with other strides than its parameter')
            e = ema_of.get(id(t))
            if e is not None and (e.stride() != t.stride() or e.dtype != torch.float32):
                raise ValueError('FusedOptimStep: EMA copy with another layout than the model tensor')
            P.append(t.data_ptr()); M.append(m.data_ptr() if m is not None else 0); V.append(v.data_ptr() if v is not None else 0)
            E.append(e.data_ptr() if e is not None else 0); numel.append(t.numel()); group.append(gi)
        chunk = _lib.lib().tamtr_optim_chunk()
        ct, co = [], []
        for i, nel in enumerate(numel):
            for off in range(0, nel, chunk):
                ct.append(i); co.append(off)
        I64 = lambda x: torch.tensor(x, dtype=torch.int64, device=dev)   # noqa: E731
        self.refresh_shadows()
        if self.use_shadows and self._hook is None and hasattr(self.model, 'register_load_state_dict_post_hook'):
            self._hook = self.model.register_load_state_dict_post_hook(lambda *_: self.refresh_shadows())
        self.tab = {'p': I64(P), 'm': I64(M), 'v': I64(V), 'e': I64(E), 'sh': I64(SH) if self.use_shadows else None, 'numel': I64(numel), 'group': torch.tensor(group, dtype=torch.uint8, device=dev),
                    'ct': torch.tensor(ct, dtype=torch.int32, device=dev), 'co': I64(co), 'step': steps,
                    'partial': torch.empty(len(ct), device=dev, dtype=torch.float32), 'norm': torch.zeros(2, device=dev, dtype=torch.float32)}
        self.entries, self.n, self.nchunks, self.dev = entries, n, len(ct), dev
        self.strides = [t.stride() for t, _, _ in entries]
        self.adam = [a for _, _, a in entries]
        self._key = self._state_key()

    @torch.no_grad()
    def step(self):
        """clip + AdamW + EMA on the current stream; returns the device tensor [grad norm, clip coefficient] (no host read-back)."""
        import ctypes
        from . import _lib
        from .hostio import stager
        if self._key is None or self._key != self._state_key():   # first step, or state / EMA tensors were replaced (load_state_dict)
            self._build()
        ptrs = [0] * self.n
        for i, (t, _, adam) in enumerate(self.entries):
            if adam:
                g = t.grad
                if g is not None:
                    if g.stride() != self.strides[i] or g.dtype != torch.float32:   # (never on this path: AccumulateGrad keeps the parameter's layout)
                        g = torch.empty_like(t, memory_format=torch.preserve_format).copy_(g)
                        t.grad = g
                    ptrs[i] = g.data_ptr()
        gptr = stager().h2d(torch.tensor(ptrs, dtype=torch.int64), self.dev)
        groups = self.opt.param_groups
        ng = len(groups)
        lr = (ctypes.c_float * ng)(*[float(g['lr']) for g in groups])
        wd = (ctypes.c_float * ng)(*[float(g['weight_decay']) for g in groups])
        b1, b2 = groups[0]['betas']
        if any(g['betas'] != groups[0]['betas'] or g['eps'] != groups[0]['eps'] for g in groups):
            raise ValueError('FusedOptimStep: parameter groups must share betas and eps')
        do_ema, d = 0, 0.0
        if self.ema is not None and self.ema.enabled:
            self.ema.updates += 1
            do_ema, d = 1, self.ema.decay(self.ema.updates)
        T = self.tab
        P = _lib.ptr
        _lib.call('tamtr_optim_step', P(T['p']), P(T['m']), P(T['v']), P(T['e']), P(T['sh']), P(T['step']), P(T['numel']), P(T['group']), P(T['ct']), P(T['co']),
                  P(gptr), self.n, self.nchunks, P(T['partial']), P(T['norm']), ctypes.cast(lr, ctypes.c_void_p), ctypes.cast(wd, ctypes.c_void_p), ng,
                  float(b1), float(b2), float(groups[0]['eps']), self.max_norm, float(d), do_ema, _lib.stream_ptr())
        if self.shadows:   # the kernel rewrote the copy of every tensor that had a gradient; anything else that changed under us is re-derived
            stale = []
            k = 0
            for i, (t, _, adam) in enumerate(self.entries):
                if not adam:
                    continue
                s = self.shadows[k][1]
                k += 1
                if s._tamtr_version != t._version:
                    if ptrs[i]:
                        s._tamtr_version = t._version
                    else:
                        stale.append((t, s))
            if stale:
                torch._foreach_copy_([s for _, s in stale], [t for t, _ in stale])
                for t, s in stale:
                    s._tamtr_version = t._version
        return T['norm']


def train_step(model, batch, optimizer, ema=None, max_norm=0.1):
    """One optimisation step as the reference trainer runs it for RT-DETR models (bf16 autocast replaces the fp16 scaler)."""
    loss, items = model(batch)
    loss.backward()
    torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.grad is not None], max_norm=max_norm)
    optimizer.step()
    optimizer.zero_grad(set_to_none=True)
    if ema is not None:
        ema.update(model)
    return loss.detach(), items   # (the three torch calls; engine.fit and bench.py run them as FusedOptimStep on the GPU)


# ------------------------------------------------------------------------------------------------ validation side
def xywh2xyxy(b):
    half = b[..., 2:] / 2
    return torch.cat([b[..., :2] - half, b[..., :2] + half], -1)


def box_iou(box1, box2, eps=1e-7):
    """[N,4] x [M,4] xyxy -> [N,M]."""
    (a1, a2), (b1, b2) = box1.unsqueeze(1).chunk(2, 2), box2.unsqueeze(0).chunk(2, 2)
    inter = (torch.min(a2, b2) - torch.max(a1, b1)).clamp_(0).prod(2)
    return inter / ((a2 - a1).prod(2) + (b2 - b1).prod(2) - inter + eps)


def nms(boxes, scores, iou_thres):
    """Greedy NMS (torchvision.ops.nms semantics: keep in descending score order, suppress IoU > thres); <= a few hundred boxes."""
    order = scores.argsort(descending=True)
    if order.numel() == 0:
        return order
    iou = box_iou(boxes[order], boxes[order], eps=0.0).cpu().numpy()
    alive = np.ones(len(order), dtype=bool)
    keep = []
    for i in range(len(order)):
        if alive[i]:
            keep.append(i)
            alive &= ~(iou[i] > iou_thres)
            alive[i] = False
    return order[torch.as_tensor(keep, dtype=torch.long, device=order.device)]


def postprocess(preds, imgsz, conf=0.001, iou=0.7, single_cls=False, max_wh=7680):
    """Eval output [B, nq, 4 + nc] (xywh in 0..1, sigmoid scores) -> list of [n, 6] (xyxy pixels, conf, cls), conf-sorted and
    class-aware NMS'ed, as RTDETRValidator.postprocess."""
    y = preds[0] if isinstance(preds, (list, tuple)) else preds
    nd = y.shape[-1]
    bboxes, scores = y.split((4, nd - 4), dim=-1)
    bboxes = bboxes * imgsz
    out = []
    for i, bbox in enumerate(bboxes):
        bbox = xywh2xyxy(bbox)
        score, cls = scores[i].max(-1)
        pred = torch.cat([bbox, score[..., None], cls[..., None].to(bbox.dtype)], -1)
        order = score.argsort(descending=True)
        pred = pred[order][score > conf]  # the reference applies the UNSORTED confidence mask to the sorted rows (val.py:113-121): kept
        c = pred[:, 5:6] * (0 if single_cls else max_wh)
        out.append(pred[nms(pred[:, :4] + c, pred[:, 4], iou)])
    return out


IOUV = torch.linspace(0.5, 0.95, 10)


def match_predictions(pred_classes, true_classes, iou, iouv=IOUV):
    """correct[N, 10]: detection n is a true positive at IoU threshold t (one detection per label, best IoU first)."""
    correct = np.zeros((pred_classes.shape[0], iouv.shape[0])).astype(bool)
    correct_class = true_classes[:, None] == pred_classes
    iou = (iou * correct_class).cpu().numpy()
    for i, threshold in enumerate(iouv.cpu().tolist()):
        matches = np.array(np.nonzero(iou >= threshold)).T
        if matches.shape[0]:
            if matches.shape[0] > 1:
                matches = matches[iou[matches[:, 0], matches[:, 1]].argsort()[::-1]]
                matches = matches[np.unique(matches[:, 1], return_index=True)[1]]
                matches = matches[np.unique(matches[:, 0], return_index=True)[1]]
            correct[matches[:, 1].astype(int), i] = True
    return torch.tensor(correct, dtype=torch.bool, device=pred_classes.device)


def process_batch(detections, labels, iouv=IOUV):
    """detections [N, 6] xyxy conf cls; labels [M, 5] cls xyxy."""
    return match_predictions(detections[:, 5], labels[:, 0], box_iou(labels[:, 1:], detections[:, :4]), iouv)


def smooth(y, f=0.05):
    """Centred moving average of y over an odd window of about 2*f*len(y) samples; beyond the ends y is held at y[0] / y[-1].
    (metrics.py:941-946 does the same with a padded convolution; here a running sum.)"""
    win = round(len(y) * f * 2) // 2 + 1
    half = win // 2
    held = np.concatenate((np.full(half, y[0]), y, np.full(half, y[-1])))
    run = np.concatenate(([0.0], np.cumsum(held)))
    return (run[win:win + len(held) - win + 1] - run[:len(held) - win + 1]) / win


_AP_GRID = np.linspace(0.0, 1.0, 101)   # COCO's 101 recall samples


def compute_ap(recall, precision):
    """Area under the precision envelope sampled on the 101-point recall grid (metrics.py:999-1029).  recall, precision: the
    cumulative curves of one class at one IoU threshold, in descending-confidence order.  Returns (ap, envelope, recall knots)."""
    knots_r = np.empty(len(recall) + 2)
    knots_p = np.empty(len(precision) + 2)
    knots_r[0], knots_r[1:-1], knots_r[-1] = 0.0, recall, 1.0
    knots_p[0], knots_p[1:-1], knots_p[-1] = 1.0, precision, 0.0
    envelope = np.maximum.accumulate(knots_p[::-1])[::-1]        # best precision attainable at recall >= r
    y = np.interp(_AP_GRID, knots_r, envelope)
    h = _AP_GRID[1] - _AP_GRID[0]
    area = h * (y.sum() - 0.5 * (y[0] + y[-1]))                   # trapezoid rule on the uniform grid
    return area, envelope, knots_r


def _curves_of_class(hit, score, n_gt, grid, eps):
    """hit [n, T] bool (descending score), score [n]: recall / precision of the first IoU threshold resampled on `grid` (confidence
    axis) and the AP of every threshold."""
    tp_run = hit.cumsum(0)
    fp_run = (1 - hit).cumsum(0)
    recall = tp_run / (n_gt + eps)
    precision = tp_run / (tp_run + fp_run)
    r_grid = np.interp(-grid, -score, recall[:, 0], left=0)          # confidence decreases along the arrays: negate to interpolate
    p_grid = np.interp(-grid, -score, precision[:, 0], left=1)
    ap = np.array([compute_ap(recall[:, t], precision[:, t])[0] for t in range(hit.shape[1])])
    return r_grid, p_grid, ap


def ap_per_class(tp, conf, pred_cls, target_cls, eps=1e-16):
    """-> tp, fp, p, r, f1 (at the max-F1 confidence), ap [nc, 10], unique_classes   (metrics.py:1032-1128, plots dropped).
    Classes are the ones that have ground truth; a class nobody predicted keeps zero curves."""
    order = np.argsort(-conf)
    tp, conf, pred_cls = tp[order], conf[order], pred_cls[order]
    classes, n_gt = np.unique(target_cls, return_counts=True)
    grid = np.linspace(0, 1, 1000)
    ap = np.zeros((len(classes), tp.shape[1]))
    p_curve, r_curve = np.zeros((len(classes), grid.size)), np.zeros((len(classes), grid.size))
    for row, (c, n) in enumerate(zip(classes, n_gt)):
        mine = pred_cls == c
        if n == 0 or not mine.any():
            continue
        r_curve[row], p_curve[row], ap[row] = _curves_of_class(tp[mine], conf[mine], n, grid, eps)
    f1_curve = 2 * p_curve * r_curve / (p_curve + r_curve + eps)
    best = smooth(f1_curve.mean(0), 0.1).argmax()                       # one operating confidence for all classes
    p, r, f1 = p_curve[:, best], r_curve[:, best], f1_curve[:, best]
    tpn = (r * n_gt).round()
    fpn = (tpn / (p + eps) - tpn).round()
    return tpn, fpn, p, r, f1, ap, classes.astype(int)


class Validator:
    """Accumulates (correct, conf, pred_cls, target_cls) over batches and reduces to mp, mr, mAP50, mAP50-95."""

    def __init__(self, imgsz=640, conf=0.001, iou=0.7, single_cls=False):
        self.imgsz, self.conf, self.iou, self.single_cls = imgsz, conf, iou, single_cls
        self.stats, self.seen = [], 0

    @torch.no_grad()
    def update(self, preds, batch):
        dev = preds[0].device if isinstance(preds, (list, tuple)) else preds.device
        iouv = IOUV.to(dev)
        for si, pred in enumerate(postprocess(preds, self.imgsz, self.conf, self.iou, self.single_cls)):
            idx = batch['batch_idx'].view(-1).to(dev) == si
            cls = batch['cls'].to(dev).view(-1, 1)[idx].float()
            bbox = batch['bboxes'].to(dev)[idx].float()
            shape = batch['ori_shape'][si] if 'ori_shape' in batch else (self.imgsz, self.imgsz)
            nl, npr = cls.shape[0], pred.shape[0]
            correct = torch.zeros(npr, iouv.numel(), dtype=torch.bool, device=dev)
            self.seen += 1
            if npr == 0:
                if nl:
                    self.stats.append((correct, torch.zeros(0, device=dev), torch.zeros(0, device=dev), cls.squeeze(-1)))
                continue
            if self.single_cls:
                pred[:, 5] = 0
            predn = pred.clone()
            predn[..., [0, 2]] *= shape[1] / self.imgsz
            predn[..., [1, 3]] *= shape[0] / self.imgsz
            if nl:
                tbox = xywh2xyxy(bbox)
                tbox[..., [0, 2]] *= shape[1]
                tbox[..., [1, 3]] *= shape[0]
                correct = process_batch(predn.float(), torch.cat((cls, tbox), 1), iouv)
            self.stats.append((correct, pred[:, 4], pred[:, 5], cls.squeeze(-1)))

    def results(self):
        if not self.stats:
            return {'precision': 0.0, 'recall': 0.0, 'mAP50': 0.0, 'mAP50-95': 0.0, 'seen': self.seen}
        stats = [torch.cat(x, 0).cpu().numpy() for x in zip(*self.stats)]
        if not stats[0].any():
            return {'precision': 0.0, 'recall': 0.0, 'mAP50': 0.0, 'mAP50-95': 0.0, 'seen': self.seen}
        _, _, p, r, _, ap, _ = ap_per_class(*stats)
        return {'precision': float(p.mean()), 'recall': float(r.mean()), 'mAP50': float(ap[:, 0].mean()),
                'mAP50-95': float(ap.mean()), 'seen': self.seen}


@torch.no_grad()
def validate(model, batches, imgsz=640, conf=0.001, iou=0.7, autocast_dtype=None):
    """model in eval mode over an iterable of batches -> metric dict (valTAMTR.py's flow without the dataset plumbing)."""
    was_training = model.training
    model.eval()
    v = Validator(imgsz, conf, iou)
    for batch in batches:
        img = batch['img']
        with torch.autocast(img.device.type, dtype=autocast_dtype or torch.bfloat16, enabled=autocast_dtype is not None):
            preds = model(img, txt_feats=batch.get('txt_feats'))
        v.update(preds, batch)
    model.train(was_training)
    return v.results()


# ------------------------------------------------------------------------------------------------ the epoch loop
def fitness(metrics):
    """0.1 * mAP50 + 0.9 * mAP50-95 (utils/metrics.py:1252-1256)."""
    return 0.1 * metrics['mAP50'] + 0.9 * metrics['mAP50-95']


def fit(model, train_loader, prepare, epochs, val_loader=None, lr0=1e-4, lrf=1.0, momentum=0.9, weight_decay=1e-4, optimizer='AdamW',
        warmup_iters=2000, warmup_bias_lr=0.1, warmup_momentum=0.8, close_mosaic=0, imgsz=640, reducer=None, rank=0, world=1,
        save_dir=None, max_steps=None, log=None, resume=None, static_graph=False):
    """Train `model` for `epochs` passes over train_loader; defaults are the reference's shipped hyper-parameters
    (cfg/default.yaml:23,84-90; this fork sets nbs = batch, so there is no gradient accumulation and weight decay is unscaled, and
    reads warmup_epochs as an iteration count: trainer.py:263-265,294).

    prepare(batch, training) -> batch on the device (data.preprocess_batch with the prompt table bound).  reducer: dist.GradReducer for
    world > 1 - gradients are SUMMED over ranks, which is the reference's mean-reduce of a loss pre-multiplied by world_size
    (trainer.py:346-347).  Rank 0 validates the EMA weights after every epoch and keeps last.pt / best.pt under save_dir.
    resume: a checkpoint dict written by an earlier fit() ({'epoch', 'best_fitness', 'model', 'ema', 'updates', 'optimizer'}; the caller
    has loaded 'model'): EMA weights and update count, optimizer state, the best fitness so far and the epoch counter continue from it
    (trainer.py:593-615), so the warm-up does not start over, best.pt is only replaced by a better epoch, and a run that resumes inside
    its last `close_mosaic` epochs starts with mosaic already closed.  static_graph: record trunk + VSS blocks + input projection as HIP graphs on the first batch
    (model.capture_static_part; batches of another shape, and evaluation, run eagerly).  Returns the per-epoch records."""
    nb = len(train_loader)
    if nb == 0:   # e.g. drop_last with fewer samples per rank than the batch size: the loop below would 'train' for zero steps without a word
        raise ValueError(f'fit(): the training loader yields no batches ({len(train_loader.dataset)} samples, batch size {train_loader.batch_size}, '
                         f'drop_last={getattr(train_loader, "drop_last", None)})')
    opt = build_optimizer(model, name=optimizer, lr=lr0, momentum=momentum, decay=weight_decay,
                          iterations=math.ceil(len(train_loader.dataset) / max(train_loader.batch_size or 1, 1)) * epochs)
    lf = linear_lr(epochs, lrf)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lf)
    ema = ModelEMA(model) if rank == 0 else None
    # clip + optimizer step + EMA as one table-driven launch group when the combination is served (AdamW, fp32 parameters on the GPU)
    stepper = FusedOptimStep.create(model, opt, ema, max_norm=0.1, shadows=getattr(model, 'autocast_dtype', None) == torch.bfloat16)
    history, best, steps, start = [], None, 0, 0
    if resume is not None:
        start = int(resume.get('epoch', -1)) + 1
        if 'optimizer' in resume:
            opt.load_state_dict(resume['optimizer'])
        if ema is not None and 'ema' in resume:
            ema.ema.load_state_dict(resume['ema'])
            ema.updates = int(resume.get('updates', 0))
        best = resume.get('best_fitness', resume.get('metrics', {}).get('fitness'))  # (second form: checkpoints written before round 3)
        for _ in range(start):
            sched.step()
    mosaic_open = bool(close_mosaic) and hasattr(train_loader.dataset, 'close_mosaic')

    def shut_mosaic():
        train_loader.dataset.close_mosaic()
        from .data import reset_workers
        reset_workers(train_loader)     # persistent workers keep their own copy of the dataset

    if mosaic_open and resume is not None and start > epochs - close_mosaic:   # RESUMED past the switch-over epoch (trainer.py:593-594,617: not a fresh run)
        shut_mosaic()
        mosaic_open = False
    for epoch in range(start, epochs):
        model.train()
        if hasattr(train_loader.sampler, 'set_epoch'):
            train_loader.sampler.set_epoch(epoch)
        if mosaic_open and epoch == epochs - close_mosaic:   # trainer.py:315 tests equality: a run shorter than close_mosaic never closes it
            shut_mosaic()
            mosaic_open = False
        t0, mean_items, waited, i = time.time(), None, 0.0, -1
        opt.zero_grad(set_to_none=True)
        batches = iter(train_loader)
        while True:
            t1 = time.perf_counter()
            batch = next(batches, None)
            waited += time.perf_counter() - t1        # host time blocked on the loader (0 when the workers keep ahead)
            if batch is None:
                break
            i += 1
            warmup(opt, i + nb * epoch, warmup_iters, lf(epoch), warmup_bias_lr, warmup_momentum, momentum)
            batch = prepare(batch, True)
            if static_graph and getattr(model, '_static', None) is None and batch['img'].is_cuda and hasattr(model, 'capture_static_part'):
                static_graph = False   # one attempt
                try:
                    model.capture_static_part(batch['img'], batch['txt_feats'], log=log)   # replays are checked against eager execution
                except Exception as e:  # noqa: BLE001 - training goes on eagerly; say so
                    model.release_static_part()
                    if log:
                        log(f'static part not captured ({type(e).__name__}: {e}); running eagerly')
            if reducer is not None:
                reducer.prepare()
            loss, items = model(batch)
            loss.backward()
            if reducer is not None:
                reducer.finish()
            if stepper is not None:  # clip + AdamW + EMA in four launches (csrc/optim.hip)
                stepper.step()
            else:
                torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.grad is not None], max_norm=0.1)
                opt.step()
            if reducer is None:      # (the reducer's gradients are views of its flat buckets, zeroed by prepare())
                opt.zero_grad(set_to_none=True)
            if ema is not None and stepper is None:
                ema.update(model)
            mean_items = items.detach() if mean_items is None else (mean_items * i + items.detach()) / (i + 1)   # no host sync
            steps += 1
            if max_steps is not None and steps >= max_steps:
                break
        rec = {'epoch': epoch, 'steps': steps, 'seconds': time.time() - t0, 'loader_wait': waited, 'lr': [g['lr'] for g in opt.param_groups],
               'loss_items': mean_items.float().cpu().tolist() if mean_items is not None else []}
        sched.step()
        if rank == 0:
            if val_loader is not None:
                rec.update(validate(ema.ema, (prepare(b, False) for b in val_loader), imgsz=imgsz,
                                    autocast_dtype=getattr(model, 'autocast_dtype', None)))
                rec['fitness'] = fitness(rec)
            if save_dir is not None:
                os.makedirs(save_dir, exist_ok=True)
                is_best = val_loader is not None and (best is None or rec['fitness'] >= best)
                if is_best:
                    best = rec['fitness']
                ckpt = {'epoch': epoch, 'best_fitness': best, 'model': model.state_dict(), 'ema': ema.ema.state_dict(), 'updates': ema.updates,
                        'optimizer': opt.state_dict(), 'metrics': {k: v for k, v in rec.items() if isinstance(v, float)}}
                torch.save(ckpt, os.path.join(save_dir, 'last.pt'))
                if is_best:
                    torch.save(ckpt, os.path.join(save_dir, 'best.pt'))
            if log:
                log(rec)
        history.append(rec)
        if max_steps is not None and steps >= max_steps:
            break
    return history
