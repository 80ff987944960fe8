"""Host-side training objective of the hot path (stays PyTorch by design): RIOU, Hungarian assignment,
contrastive-denoising query groups and the 12-term RT-DETR loss.

  bbox_iou(RIOU)        ultralytics/utils/metrics.py:71-130
  HungarianMatcher      ultralytics/models/utils/ops.py:12-119     (cost on device; assignment by the HIP solver
                        ops.lsap_assign for GPU tensors - no host round trip; scipy, as the reference, for CPU tensors)
  get_cdn_group         ultralytics/models/utils/ops.py:152-291
  RTDETRDetectionLoss   ultralytics/models/utils/loss.py:14-442 (VFL + L1 + RIOU, aux + dn branches)
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .hostio import stager

_SCIPY_MATCHER = os.environ.get('TAMTR_MATCHER') == 'scipy'  # A/B switch: the reference's host round trip on GPU runs too


import os as _os
_TORCH_LOSS = _os.environ.get('TAMTR_LOSS') == 'torch'   # A/B switch: the elementwise torch form of the loss terms and of the matcher's cost


class Matches(list):
    """list of per-image (query_idx, gt_idx) pairs as the reference returns them (models/utils/ops.py:117-119), plus
    `.flat` = (batch_idx, query_idx, gt_idx) device tensors over all images, which is what the loss terms consume."""
    flat = None


def flat_matches(match, dev):
    """One staged (non-synchronising) H2D of a host-side list of (query_idx, gt_idx) pairs -> (bi, si, gi) on `dev`."""
    if getattr(match, 'flat', None) is not None:
        return match.flat
    bi = torch.cat([torch.full_like(s, i) for i, (s, _) in enumerate(match)])
    idx = torch.stack([bi, torch.cat([s for s, _ in match]), torch.cat([g for _, g in match])]).long()
    idx = idx.to(dev) if idx.is_cuda else stager().h2d(idx, dev)
    return idx[0], idx[1], idx[2]


def xywh2xyxy(b):
    half = b[..., 2:] / 2
    return torch.cat([b[..., :2] - half, b[..., :2] + half], -1)


def xyxy2xywh(b):
    return torch.cat([(b[..., :2] + b[..., 2:]) / 2, b[..., 2:] - b[..., :2]], -1)


def bbox_iou(box1, box2, xywh=True, RIOU=False, eps=1e-7, **unused):
    """IoU or RIOU = IoU - (rho^2 / (max(w1,h1) + max(w2,h2) + rho + eps)^2 + v*alpha) of broadcastable box tensors."""
    if xywh:
        x1, y1, w1, h1 = box1.chunk(4, -1)
        x2, y2, w2, h2 = box2.chunk(4, -1)
        ax1, ax2, ay1, ay2 = x1 - w1 / 2, x1 + w1 / 2, y1 - h1 / 2, y1 + h1 / 2
        bx1, bx2, by1, by2 = x2 - w2 / 2, x2 + w2 / 2, y2 - h2 / 2, y2 + h2 / 2
    else:
        ax1, ay1, ax2, ay2 = box1.chunk(4, -1)
        bx1, by1, bx2, by2 = box2.chunk(4, -1)
        w1, h1 = ax2 - ax1, ay2 - ay1 + eps
        w2, h2 = bx2 - bx1, by2 - by1 + eps
    inter = (torch.minimum(ax2, bx2) - torch.maximum(ax1, bx1)).clamp(min=0) * \
            (torch.minimum(ay2, by2) - torch.maximum(ay1, by1)).clamp(min=0)
    iou = inter / (w1 * h1 + w2 * h2 - inter + eps)
    if not RIOU:
        return iou
    rho2 = ((bx1 + bx2 - ax1 - ax2) ** 2 + (by1 + by2 - ay1 - ay2) ** 2) / 4
    c2 = (torch.max(w1, h1) + torch.max(w2, h2) + torch.sqrt(rho2) + eps).pow(2)
    v = (4 / math.pi ** 2) * (torch.atan(w2 / h2) - torch.atan(w1 / h1)).pow(2)
    with torch.no_grad():
        alpha = v / (v - iou + (1 + eps))
    return iou - (rho2 / c2 + v * alpha)


class HungarianMatcher(nn.Module):
    def __init__(self, cost_gain=None, use_fl=True, with_mask=False, num_sample_points=12544, alpha=0.25, gamma=2.0):
        super().__init__()
        self.cost_gain = cost_gain or {'class': 1, 'bbox': 5, 'giou': 2, 'mask': 1, 'dice': 1}
        self.use_fl, self.alpha, self.gamma = use_fl, alpha, gamma
        if with_mask:
            raise NotImplementedError('mask costs are not part of TAM-TR')

    @torch.no_grad()
    def forward(self, pred_bboxes, pred_scores, gt_bboxes, gt_cls, gt_groups, masks=None, gt_mask=None):
        """pred_* [bs, nq, .] -> Matches (the reference's signature), or [layers, bs, nq, .] -> list of Matches: the cost matrices
        of all decoder layers come out of ONE set of elementwise kernels, only the assignment runs per layer."""
        from scipy.optimize import linear_sum_assignment
        layered = pred_scores.dim() == 4
        if not layered:
            pred_bboxes, pred_scores = pred_bboxes.unsqueeze(0), pred_scores.unsqueeze(0)
        Lr, bs, nq, nc = pred_scores.shape
        if sum(gt_groups) == 0:
            empty = [[(torch.zeros(0, dtype=torch.long), torch.zeros(0, dtype=torch.long)) for _ in range(bs)] for _ in range(Lr)]
            return empty if layered else empty[0]
        if pred_scores.is_cuda and self.use_fl and not _TORCH_LOSS:
            from . import ops   # one kernel for the cost matrices of all layers (csrc/detrloss.hip)
            C = ops.detr_match_cost(pred_scores.detach(), pred_bboxes.detach(), gt_bboxes, gt_cls,
                                    (self.cost_gain['class'], self.cost_gain['bbox'], self.cost_gain['giou']), self.alpha, self.gamma)
        else:
            ps = pred_scores.detach().float().reshape(-1, nc)
            ps = (ps.sigmoid() if self.use_fl else ps.softmax(-1))[:, gt_cls]
            pb = pred_bboxes.detach().float().reshape(-1, 4)
            if self.use_fl:
                neg = (1 - self.alpha) * ps ** self.gamma * (-(1 - ps + 1e-8).log())
                pos = self.alpha * (1 - ps) ** self.gamma * (-(ps + 1e-8).log())
                c_cls = pos - neg
            else:
                c_cls = -ps
            c_l1 = (pb.unsqueeze(1) - gt_bboxes.unsqueeze(0)).abs().sum(-1)
            c_iou = 1.0 - bbox_iou(pb.unsqueeze(1), gt_bboxes.unsqueeze(0), xywh=True, RIOU=True).squeeze(-1)
            C = self.cost_gain['class'] * c_cls + self.cost_gain['bbox'] * c_l1 + self.cost_gain['giou'] * c_iou
            C = torch.where(torch.isfinite(C), C, torch.zeros_like(C)).view(Lr, bs, nq, -1)
        groups = [int(n) for n in gt_groups]
        sizes = [min(nq, n) for n in groups]
        res = []
        if C.is_cuda and not _SCIPY_MATCHER:
            from . import ops
            for l in range(Lr):
                bi, si, gi = ops.lsap_assign(C[l], groups)  # HIP solver, scipy's pairs in scipy's order, nothing leaves the GPU
                out = Matches(zip(si.split(sizes), gi.split(sizes)))
                out.flat = (bi, si, gi)
                res.append(out)
            return res if layered else res[0]
        dev, C = C.device, C.cpu()  # reference behaviour: device->host sync, scipy per image
        for l in range(Lr):
            out, off = Matches(), 0
            for i, c in enumerate(C[l].split(groups, -1)):
                r, k = linear_sum_assignment(c[i].numpy())
                out.append((torch.as_tensor(r, dtype=torch.long), torch.as_tensor(k, dtype=torch.long) + off))
                off += groups[i]
            if dev.type != 'cpu':
                out.flat = tuple(t.to(dev) for t in flat_matches(out, 'cpu'))
            res.append(out)
        return res if layered else res[0]


def get_cdn_group(batch, num_classes, num_queries, class_embed, num_dn=100, cls_noise_ratio=0.5, box_noise_scale=1.0,
                  training=False):
    """Contrastive-denoising queries.  RNG draws are made on the CPU generator in the reference's order (rand,
    randint_like, randint_like, rand_like) so that a given torch.manual_seed reproduces the reference's groups."""
    if (not training) or num_dn <= 0 or batch is None:
        return None, None, None, None
    groups = batch['gt_groups']
    total, mx = sum(groups), max(groups)
    if mx == 0:
        return None, None, None, None
    dev = class_embed.device
    ng = max(num_dn // mx, 1)
    bs = len(groups)
    host = batch.get('host') or {k: batch[k].cpu() for k in ('cls', 'bboxes', 'batch_idx')}  # host copies (model.loss keeps them)
    cls = host['cls'].repeat(2 * ng)
    box = host['bboxes'].float().repeat(2 * ng, 1)
    bidx = host['batch_idx'].repeat(2 * ng).view(-1)
    neg = torch.arange(total * ng, dtype=torch.long) + ng * total
    if cls_noise_ratio > 0:
        idx = torch.nonzero(torch.rand(cls.shape) < cls_noise_ratio * 0.5).squeeze(-1)
        cls[idx] = torch.randint_like(idx, 0, num_classes, dtype=cls.dtype)
    if box_noise_scale > 0:
        known = xywh2xyxy(box)
        diff = (box[..., 2:] * 0.5).repeat(1, 2) * box_noise_scale
        sign = torch.randint_like(box, 0, 2) * 2.0 - 1.0
        part = torch.rand_like(box)
        part[neg] += 1.0
        known = (known + part * sign * diff).clip(0.0, 1.0)
        box = torch.logit(xyxy2xywh(known), eps=1e-6)
    n_dn = int(mx * 2 * ng)
    within = torch.cat([torch.arange(n, dtype=torch.long) for n in groups])
    pos_idx = torch.stack([within + mx * i for i in range(ng)], 0)
    slot = torch.cat([within + mx * i for i in range(2 * ng)])
    put = stager().h2d  # pinned, non-synchronising uploads
    cls, box, bidx, slot = put(cls, dev), put(box, dev), put(bidx, dev), put(slot, dev)
    if class_embed.is_cuda:
        from . import ops
        emb = ops.embed_rows(class_embed, cls)   # = class_embed[cls]; its backward is one small product instead of a sorted index_put
    else:
        emb = class_embed[cls]
    pad_c = torch.zeros(bs, n_dn, emb.shape[-1], device=dev, dtype=emb.dtype)
    pad_b = torch.zeros(bs, n_dn, 4, device=dev)
    pad_c[bidx, slot] = emb
    pad_b[bidx, slot] = box
    mask = _dn_attn_mask(n_dn, num_queries, mx, ng, str(dev))
    meta = {'dn_pos_idx': [p.reshape(-1) for p in pos_idx.split(list(groups), 1)], 'dn_num_group': ng,
            'dn_num_split': [n_dn, num_queries]}
    return pad_c, pad_b, mask, meta


_MASKS = {}


def _dn_attn_mask(n_dn, num_queries, mx, ng, dev):
    """Block mask of the denoising groups (ops.py:273-284); depends only on the four sizes, so it is built once per shape."""
    key = (n_dn, num_queries, mx, ng, dev)
    if key not in _MASKS:
        size = n_dn + num_queries
        mask = torch.zeros(size, size, dtype=torch.bool)
        mask[n_dn:, :n_dn] = True
        for i in range(ng):
            lo, hi = mx * 2 * i, mx * 2 * (i + 1)
            mask[lo:hi, hi:n_dn] = True
            mask[lo:hi, :lo] = True
        if len(_MASKS) > 64:
            _MASKS.clear()
        _MASKS[key] = mask.to(dev)
    return _MASKS[key]


def varifocal_loss(pred, gt_score, label, alpha=0.75, gamma=2.0):
    w = alpha * pred.sigmoid().pow(gamma) * (1 - label) + gt_score * label
    return (F.binary_cross_entropy_with_logits(pred.float(), gt_score.float(), reduction='none') * w).mean(1).sum()


def focal_loss(pred, label, gamma=1.5, alpha=0.25):
    loss = F.binary_cross_entropy_with_logits(pred, label, reduction='none')
    p = pred.sigmoid()
    pt = label * p + (1 - label) * (1 - p)
    return (loss * (1.0 - pt) ** gamma * (label * alpha + (1 - label) * (1 - alpha))).mean(1).sum()


class DETRLoss(nn.Module):
    def __init__(self, nc=80, loss_gain=None, aux_loss=True, use_fl=True, use_vfl=False, **unused):
        super().__init__()
        self.nc = nc
        self.loss_gain = loss_gain or {'class': 1, 'bbox': 5, 'giou': 2, 'no_object': 0.1, 'mask': 1, 'dice': 1}
        self.matcher = HungarianMatcher(cost_gain={'class': 2, 'bbox': 5, 'giou': 2})
        self.aux_loss, self.use_fl, self.use_vfl = aux_loss, use_fl, use_vfl
        # teacher forcing of the Hungarian assignment (parity measurements): per stacked layer [enc, dec 0..n-1] a list of per-image
        # (query_idx, gt_idx) pairs used INSTEAD of the matcher's; None (always, outside tests) = the matcher (loss.py:282-326)
        self.fixed_matches = None
        self.last_matches = None

    def _layers(self, pb, ps, gt_bboxes, gt_cls, gt_groups, match):
        """(class, bbox, giou) of ALL decoder layers at once: pb [layers, bs, nq, 4], ps [layers, bs, nq, nc] -> three [layers]
        tensors.  Same arithmetic per layer as loss.py:282-326 of the reference; stacking the layers turns eight passes of
        ~40 tiny kernels (and as many in the backward) into one."""
        dev = pb.device
        Lr, bs, nq = pb.shape[:3]
        if match is None and self.fixed_matches is not None:
            assert len(self.fixed_matches) == Lr, (len(self.fixed_matches), Lr)
            flats = [flat_matches(m if isinstance(m, Matches) else Matches(m), dev) for m in self.fixed_matches]
        elif match is None:
            self.last_matches = ms = self.matcher(pb, ps, gt_bboxes, gt_cls, gt_groups)   # (kept: parity measurements replay them through fixed_matches)
            flats = [flat_matches(m, dev) for m in ms]
        else:
            flats = [flat_matches(match, dev)] * Lr
        n = int(flats[0][0].shape[0])  # matched pairs per layer (every layer matches all boxes: the same count)
        li = torch.arange(Lr, device=dev).repeat_interleave(n)
        bi, si, gi = (torch.cat([f[j] for f in flats]) for j in range(3))
        if n and pb.is_cuda and self.use_vfl and not _TORCH_LOSS:
            from . import ops   # the three terms of all layers in three launches (+ two backward): csrc/detrloss.hip
            return ops.detr_layer_losses(pb, ps, gt_bboxes, gt_cls, li, bi, si, gi, n,
                                         (self.loss_gain['class'], self.loss_gain['bbox'], self.loss_gain['giou']))
        p_sel, g_sel = pb[li, bi, si].float(), gt_bboxes[gi]
        targets = torch.full((Lr, bs, nq), self.nc, device=dev, dtype=gt_cls.dtype)
        targets[li, bi, si] = gt_cls[gi]
        gt_scores = torch.zeros(Lr, bs, nq, device=dev)
        if n:
            gt_scores[li, bi, si] = bbox_iou(p_sel.detach(), g_sel, xywh=True).squeeze(-1)
        one_hot = F.one_hot(targets, self.nc + 1)[..., :-1]
        ps = ps.float()
        if n and self.use_vfl:  # varifocal_loss per layer: (...).mean(1).sum() over [bs, nq, nc]
            label, score = one_hot, gt_scores.unsqueeze(-1) * one_hot
            w = 0.75 * ps.sigmoid().pow(2.0) * (1 - label) + score * label
            l_cls = (F.binary_cross_entropy_with_logits(ps, score.float(), reduction='none') * w).mean(2).sum((1, 2))
        else:  # focal_loss per layer
            label = one_hot.float()
            bce = F.binary_cross_entropy_with_logits(ps, label, reduction='none')
            p = ps.sigmoid()
            pt = label * p + (1 - label) * (1 - p)
            l_cls = (bce * (1.0 - pt) ** 1.5 * (label * 0.25 + (1 - label) * 0.75)).mean(2).sum((1, 2))
        l_cls = l_cls / (max(n, 1) / nq) * self.loss_gain['class']
        if n == 0:
            z = torch.zeros(Lr, device=dev)
            return l_cls, z, z.clone()
        l_box = self.loss_gain['bbox'] * (p_sel - g_sel).abs().view(Lr, n, 4).sum((1, 2)) / n
        l_iou = self.loss_gain['giou'] * (1.0 - bbox_iou(p_sel, g_sel, xywh=True, RIOU=True)).view(Lr, n).sum(1) / n
        return l_cls, l_box, l_iou

    def forward(self, pred_bboxes, pred_scores, batch, postfix='', match_indices=None):
        gt_cls, gt_bboxes, gt_groups = batch['cls'], batch['bboxes'], batch['gt_groups']
        if match_indices is not None and getattr(match_indices, 'flat', None) is None:  # fixed (dn) pairs: upload once, not per layer
            fixed = Matches(match_indices)
            fixed.flat = flat_matches(match_indices, pred_bboxes.device)
            match_indices = fixed
        if not self.aux_loss:
            pred_bboxes, pred_scores = pred_bboxes[-1:], pred_scores[-1:]
        c, b, g = self._layers(pred_bboxes, pred_scores, gt_bboxes, gt_cls, gt_groups, match_indices)
        out = {f'loss_class{postfix}': c[-1], f'loss_bbox{postfix}': b[-1], f'loss_giou{postfix}': g[-1]}
        if self.aux_loss:
            out[f'loss_class_aux{postfix}'], out[f'loss_bbox_aux{postfix}'], out[f'loss_giou_aux{postfix}'] = c[:-1].sum(), b[:-1].sum(), g[:-1].sum()
        return out


class RTDETRDetectionLoss(DETRLoss):
    def forward(self, preds, batch, dn_bboxes=None, dn_scores=None, dn_meta=None):
        pred_bboxes, pred_scores = preds
        total = super().forward(pred_bboxes, pred_scores, batch)
        if dn_meta is not None:
            match = self.get_dn_match_indices(dn_meta['dn_pos_idx'], dn_meta['dn_num_group'], batch['gt_groups'])
            total.update(super().forward(dn_bboxes, dn_scores, batch, postfix='_dn', match_indices=match))
        else:
            total.update({f'{k}_dn': torch.zeros((), device=pred_bboxes.device) for k in list(total)})
        return total

    @staticmethod
    def get_dn_match_indices(dn_pos_idx, dn_num_group, gt_groups):
        out, off = [], 0
        for i, n in enumerate(gt_groups):
            if n > 0:
                gt_idx = (torch.arange(n, dtype=torch.long) + off).repeat(dn_num_group)
                assert len(dn_pos_idx[i]) == len(gt_idx)
                out.append((dn_pos_idx[i], gt_idx))
            else:
                out.append((torch.zeros(0, dtype=torch.long), torch.zeros(0, dtype=torch.long)))
            off += n
        return out
