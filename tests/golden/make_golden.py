#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the *reference* (read-only, /root/reference)
on CPU.  Runs only in the build container; the GPU box never sees the reference.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

What is stored: seeded inputs, outputs, input/parameter gradients (large ones as strided samples), and a
checksum of the name-keyed weights (tests/golden/weights.py) that were loaded into the reference modules.
No reference source is stored - fixtures are data only.

Import recipe = SURVEY.md Appendix B: third-party modules that the reference imports but never uses
arithmetically on this path (cv2, torchvision, timm, fvcore, thop, cpuinfo, seaborn, clip) are absent from the
image and are registered as inert placeholder modules so that `import ultralytics` succeeds.  The only
placeholder that sits inside a forward() is timm's DropPath, given as identity (= eval / p=0 behaviour).
The VMamba selective-scan CUDA extension cannot run here: VSSBlocks are either replaced by
identity (head / end-to-end fixtures; flag recorded) or driven through forward_corev2's own `SelectiveScan=`
argument with a closed-form surrogate (vss fixture) so that everything *around* the scan is pinned.
"""
import importlib.machinery
import math
import os
import sys
import types
import warnings

import numpy as np
import torch
import torch.nn as nn

warnings.filterwarnings('ignore')
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from weights import checksum, fill_state, rnd, summarize, urnd  # noqa: E402

REF = '/root/reference'


def _import_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)

    class _Any:
        def __init__(self, *a, **k): pass
        def __call__(self, *a, **k): return _Any()
        def __getattr__(self, n): return _Any()
        def __mro_entries__(self, b): return (object,)
        def __iter__(self): return iter(())

    class _Perm(types.ModuleType):
        def __getattr__(self, n):
            if n.startswith('__'):
                raise AttributeError(n)
            return _Any()

    def stub(name, **kw):
        m = _Perm(name)
        m.__dict__.update(kw)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        m.__path__ = []
        sys.modules[name] = m

    class DropPath(nn.Module):
        def __init__(self, p=0.):
            super().__init__()
            self.drop_prob = p

        def forward(self, x):
            return x

    for n in ['cv2', 'torchvision', 'torchvision.transforms', 'torchvision.datasets', 'torchvision.ops',
              'torchvision.models', 'thop', 'cpuinfo', 'seaborn', 'clip', 'timm', 'timm.models', 'fvcore']:
        stub(n)
    stub('timm.layers', DropPath=DropPath)
    stub('timm.models.layers', DropPath=DropPath, trunc_normal_=nn.init.trunc_normal_)
    stub('fvcore.nn', FlopCountAnalysis=None, flop_count_str=None, flop_count=None, parameter_count=None)


def set_bn(m):
    """What utils/torch_utils.py:303-313 does to every BatchNorm2d of a built model."""
    for mod in m.modules():
        if type(mod) is nn.BatchNorm2d:
            mod.eps = 1e-3
            mod.momentum = 0.03
    return m


def load_filled(module, seed):
    sd = module.state_dict()
    st = fill_state(sd, seed)
    module.load_state_dict(st)
    return checksum(st)


def pack(d, prefix, s):
    """flatten summarize() dict into d with prefix."""
    for k, v in s.items():
        d[f'{prefix}.{k}'] = np.asarray(v)


def grads_of(module, out, cot, inputs):
    """backward with cotangent; return dict of summarized grads for inputs and parameters."""
    module.zero_grad(set_to_none=True)
    for t in inputs.values():
        t.grad = None
    if isinstance(out, (tuple, list)):
        tot = sum((o * c).sum() for o, c in zip(out, cot))
    else:
        tot = (out * cot).sum()
    tot.backward()
    d = {}
    for k, t in inputs.items():
        if t.grad is not None:
            pack(d, f'gin.{k}', summarize(t.grad))
    for k, p in module.named_parameters():
        if p.grad is not None:
            pack(d, f'gpar.{k}', summarize(p.grad))
        else:
            d[f'gpar.{k}.none'] = np.asarray(1)
    return d


def save(name, d):
    path = os.path.join(HERE, name + '.npz')
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(path, **out)
    print(f'{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB  ({len(out)} arrays)')


# ------------------------------------------------------------------------------------------------------------
def gen_gate():
    """a-1 MaxSigmoidAttnBlock (extra_modules/block.py:194-226) and a-2 TIAGELAN (:171-186)."""
    from ultralytics.nn.extra_modules.block import MaxSigmoidAttnBlock, TIAGELAN
    d = {}
    cases = [('A', 2, 64, 2, 12, 12, 10, True), ('B', 2, 256, 8, 9, 7, 7, False), ('C', 3, 128, 4, 5, 20, 1, True)]
    for tag, B, c, nh, H, W, T, train in cases:
        m = set_bn(MaxSigmoidAttnBlock(c, c, nh=nh, ec=c))
        d[f'{tag}.wsum'] = load_filled(m, seed=11)
        m.train(train)
        x = rnd((B, c, H, W), 100 + ord(tag)).requires_grad_()
        g = torch.nn.functional.normalize(rnd((B, T, 512), 200 + ord(tag)), dim=-1).requires_grad_()
        out = m(x, g)
        d[f'{tag}.cfg'] = np.asarray([B, c, nh, H, W, T, int(train)])
        d[f'{tag}.x'], d[f'{tag}.guide'], d[f'{tag}.out'] = x, g, out
        if train:
            cot = rnd(out.shape, 300 + ord(tag))
            d[f'{tag}.cot'] = cot
            for k, v in grads_of(m, out, cot, {'x': x, 'guide': g}).items():
                d[f'{tag}.{k}'] = v
            d[f'{tag}.bn_mean'] = m.proj_conv.bn.running_mean
            d[f'{tag}.bn_var'] = m.proj_conv.bn.running_var
    save('gate', d)

    d = {}
    m = set_bn(TIAGELAN(96, 64, 128, 64, 1, 2))
    d['wsum'] = load_filled(m, seed=12)
    m.train()
    x = rnd((2, 96, 10, 8), 1).requires_grad_()
    g = torch.nn.functional.normalize(rnd((2, 10, 512), 2), dim=-1).requires_grad_()
    out = m(x, g)
    cot = rnd(out.shape, 3)
    d.update(x=x, guide=g, out=out, cot=cot)
    d.update(grads_of(m, out, cot, {'x': x, 'guide': g}))
    d['attn_bn_mean'] = m.attn.proj_conv.bn.running_mean
    d['attn_bn_var'] = m.attn.proj_conv.bn.running_var
    m.eval()
    d['out_eval'] = m(x, g)
    save('tiagelan', d)


def gen_msdeform():
    """a-6 multi_scale_deformable_attn_pytorch (modules/utils.py:42-89), a-5 MSDeformAttn (transformer.py:204-299)."""
    from ultralytics.nn.modules.utils import multi_scale_deformable_attn_pytorch
    from ultralytics.nn.modules.transformer import MSDeformAttn
    d = {}
    for tag, B, nh, D, shapes, Q, P in [('A', 2, 8, 32, [(12, 10), (6, 5), (3, 3)], 37, 4),
                                         ('B', 1, 8, 64, [(7, 9), (4, 4), (2, 3)], 19, 4),
                                         ('C', 2, 4, 32, [(5, 5)], 3, 2)]:
        L = sum(h * w for h, w in shapes)
        nl = len(shapes)
        value = rnd((B, L, nh, D), 1 + ord(tag)).requires_grad_()
        loc = urnd((B, Q, nh, nl, P, 2), 2 + ord(tag), -0.25, 1.25).requires_grad_()
        aw = torch.softmax(rnd((B, Q, nh, nl * P), 3 + ord(tag)), -1).view(B, Q, nh, nl, P).requires_grad_()
        out = multi_scale_deformable_attn_pytorch(value, shapes, loc, aw)
        cot = rnd(out.shape, 4 + ord(tag))
        (out * cot).sum().backward()
        d[f'{tag}.shapes'] = np.asarray(shapes)
        d[f'{tag}.value'], d[f'{tag}.loc'], d[f'{tag}.aw'] = value, loc, aw
        d[f'{tag}.out'], d[f'{tag}.cot'] = out, cot
        d[f'{tag}.g_value'], d[f'{tag}.g_loc'], d[f'{tag}.g_aw'] = value.grad, loc.grad, aw.grad
    save('msdeform_core', d)

    d = {}
    shapes = [[12, 10], [6, 5], [3, 3]]
    L = sum(h * w for h, w in shapes)
    m = MSDeformAttn(256, 3, 8, 4)
    d['wsum'] = load_filled(m, seed=21)
    q = rnd((2, 37, 256), 1).requires_grad_()
    ref = urnd((2, 37, 1, 4), 2, 0.05, 0.95).requires_grad_()
    val = rnd((2, L, 256), 3).requires_grad_()
    out = m(q, ref, val, shapes)
    cot = rnd(out.shape, 4)
    d.update(shapes=np.asarray(shapes), query=q, refer=ref, value=val, out=out, cot=cot)
    d.update(grads_of(m, out, cot, {'query': q, 'refer': ref, 'value': val}))
    # KAT: at the module's own init (transformer.py:234-250) attention weights are uniform 1/12 and the offsets ring-shaped
    m2 = MSDeformAttn(256, 3, 8, 4)
    d['init_offsets_bias'] = m2.sampling_offsets.bias
    save('msdeform_attn', d)


def gen_contrastive():
    """a-8 ContrastiveHeadMLP (modules/block.py:522-541)."""
    from ultralytics.nn.modules.block import ContrastiveHeadMLP
    d = {}
    m = ContrastiveHeadMLP()
    d['wsum'] = load_filled(m, seed=31)
    d['bias'], d['logit_scale'] = m.bias, m.logit_scale
    x = rnd((2, 37, 512), 1, 3.0).requires_grad_()
    w = rnd((2, 10, 512), 2).requires_grad_()  # deliberately NOT unit norm: the head re-normalises
    out = m(x, w)
    cot = rnd(out.shape, 3)
    d.update(x=x, w=w, out=out, cot=cot)
    d.update(grads_of(m, out, cot, {'x': x, 'w': w}))
    m0 = ContrastiveHeadMLP()
    d['init_bias'], d['init_logit_scale'] = m0.bias, m0.logit_scale
    save('contrastive', d)


def gen_decoder():
    """a-7 DeformableTransformerDecoderLayer (transformer.py:498-558), a-4 TextDeformableTransformerDecoder (:835-891)."""
    from ultralytics.nn.modules.transformer import (DeformableTransformerDecoderLayer, TextDeformableTransformerDecoder,
                                                    MLP)
    from ultralytics.nn.modules.block import ContrastiveHeadMLP
    shapes = [[12, 10], [6, 5], [3, 3]]
    L = sum(h * w for h, w in shapes)
    hd, nh, ffn, Q, B = 256, 8, 512, 37, 2
    d = {}
    layer = DeformableTransformerDecoderLayer(hd, nh, ffn, 0., nn.ReLU(), 3, 4)
    d['wsum'] = load_filled(layer, seed=41)
    emb = rnd((B, Q, hd), 1).requires_grad_()
    ref = urnd((B, Q, 4), 2, 0.05, 0.95).requires_grad_()
    feats = rnd((B, L, hd), 3).requires_grad_()
    pos = rnd((B, Q, hd), 4).requires_grad_()
    mask = torch.zeros(Q, Q, dtype=torch.bool)
    mask[12:, :12] = True
    mask[:6, 6:12] = True
    mask[6:12, :6] = True
    out = layer(emb, ref, feats, shapes, None, mask, pos)
    cot = rnd(out.shape, 5)
    d.update(shapes=np.asarray(shapes), embed=emb, refer=ref, feats=feats, pos=pos, mask=mask, out=out, cot=cot)
    d.update(grads_of(layer, out, cot, {'embed': emb, 'refer': ref, 'feats': feats, 'pos': pos}))
    d['out_nomask'] = layer(emb, ref, feats, shapes, None, None, pos)
    save('decoder_layer', d)

    d = {}
    layer = DeformableTransformerDecoderLayer(hd, nh, ffn, 0., nn.ReLU(), 3, 4)
    dec = TextDeformableTransformerDecoder(hd, layer, 3, -1)
    heads = nn.ModuleDict(dict(
        decoder=dec,
        dec_bbox_head=nn.ModuleList([MLP(hd, hd, 4, num_layers=3) for _ in range(3)]),
        dec_score_head=nn.ModuleList([ContrastiveHeadMLP() for _ in range(3)]),
        query_pos_head=MLP(4, 2 * hd, hd, num_layers=2)))
    d['wsum'] = load_filled(heads, seed=42)
    emb = rnd((B, Q, hd), 11).requires_grad_()
    refl = rnd((B, Q, 4), 12).requires_grad_()  # logits; decoder applies sigmoid
    feats = rnd((B, L, hd), 13).requires_grad_()
    text = torch.nn.functional.normalize(rnd((B, 10, hd), 14), dim=-1).requires_grad_()
    heads.train()
    bb, sc = dec(emb, refl, feats, shapes, text, heads['dec_bbox_head'], heads['dec_score_head'],
                 heads['query_pos_head'], attn_mask=mask)
    cb, cs = rnd(bb.shape, 15), rnd(sc.shape, 16)
    d.update(shapes=np.asarray(shapes), embed=emb, refer=refl, feats=feats, text=text, mask=mask,
             bboxes=bb, scores=sc, cot_b=cb, cot_s=cs)
    d.update(grads_of(heads, (bb, sc), (cb, cs), {'embed': emb, 'refer': refl, 'feats': feats, 'text': text}))
    heads.eval()
    bb, sc = dec(emb, refl, feats, shapes, text, heads['dec_bbox_head'], heads['dec_score_head'],
                 heads['query_pos_head'], attn_mask=None)
    d.update(bboxes_eval=bb, scores_eval=sc)
    save('text_decoder', d)


def make_targets(B, n_per, seed):
    g = torch.Generator().manual_seed(seed)
    n = sum(n_per)
    cls = torch.randint(0, 10, (n,), generator=g)
    xy = 0.2 + 0.6 * torch.rand(n, 2, generator=g)
    wh = 0.02 + 0.2 * torch.rand(n, 2, generator=g)
    bidx = torch.cat([torch.full((k,), i, dtype=torch.long) for i, k in enumerate(n_per)])
    return {'cls': cls, 'bboxes': torch.cat([xy, wh], 1), 'batch_idx': bidx, 'gt_groups': list(n_per)}


def gen_cdn():
    """get_cdn_group (models/utils/ops.py:152-291): RNG-dependent; seed is set immediately before the call."""
    from ultralytics.models.utils.ops import get_cdn_group
    d = {}
    for tag, n_per, nq, nd in [('A', [3, 1], 20, 100), ('B', [0, 5, 2], 10, 7), ('C', [8, 8], 100, 100)]:
        t = make_targets(len(n_per), n_per, 5)
        emb = rnd((11, 64), 6)
        torch.manual_seed(1234)
        e, b, m, meta = get_cdn_group(t, 10, nq, emb, nd, 0.5, 1.0, True)
        d[f'{tag}.n_per'] = np.asarray(n_per)
        d[f'{tag}.cfg'] = np.asarray([nq, nd])
        d[f'{tag}.cls'], d[f'{tag}.bboxes'], d[f'{tag}.batch_idx'] = t['cls'], t['bboxes'], t['batch_idx']
        d[f'{tag}.class_embed'] = emb
        d[f'{tag}.dn_embed'], d[f'{tag}.dn_bbox'], d[f'{tag}.mask'] = e, b, m
        d[f'{tag}.num_group'] = meta['dn_num_group']
        d[f'{tag}.split'] = np.asarray(meta['dn_num_split'])
        for i, p in enumerate(meta['dn_pos_idx']):
            d[f'{tag}.pos_idx{i}'] = p
    save('cdn', d)


def gen_loss():
    """a-10: bbox_iou RIOU (utils/metrics.py:71-130), HungarianMatcher (models/utils/ops.py:48-119),
    RTDETRDetectionLoss (models/utils/loss.py:376-416)."""
    from ultralytics.utils.metrics import bbox_iou
    from ultralytics.models.utils.ops import HungarianMatcher, get_cdn_group
    from ultralytics.models.utils.loss import RTDETRDetectionLoss
    d = {}
    b1 = torch.cat([urnd((64, 2), 1, 0.1, 0.9), urnd((64, 2), 2, 0.02, 0.4)], 1).requires_grad_()
    b2 = torch.cat([urnd((64, 2), 3, 0.1, 0.9), urnd((64, 2), 4, 0.02, 0.4)], 1).requires_grad_()
    r = bbox_iou(b1, b2, xywh=True, RIOU=True)
    r.sum().backward()
    d.update(b1=b1, b2=b2, riou=r, g_b1=b1.grad, g_b2=b2.grad, iou=bbox_iou(b1.detach(), b2.detach(), xywh=True))
    save('riou', d)

    d = {}
    B, nq, nc = 3, 30, 10
    t = make_targets(B, [4, 0, 7], 9)
    pb = torch.cat([urnd((B, nq, 2), 5, 0.1, 0.9), urnd((B, nq, 2), 6, 0.02, 0.4)], -1)
    ps = rnd((B, nq, nc), 7, 2.0)
    mt = HungarianMatcher(cost_gain={'class': 2, 'bbox': 5, 'giou': 2})
    idx = mt(pb, ps, t['bboxes'], t['cls'], t['gt_groups'])
    d.update(pred_bboxes=pb, pred_scores=ps, cls=t['cls'], bboxes=t['bboxes'], batch_idx=t['batch_idx'],
             n_per=np.asarray(t['gt_groups']))
    for i, (a, b) in enumerate(idx):
        d[f'match{i}.src'], d[f'match{i}.dst'] = a, b
    save('matcher', d)

    d = {}
    B, nq, nc, nl = 2, 20, 10, 4
    t = make_targets(B, [3, 5], 19)
    torch.manual_seed(77)
    _, _, _, meta = get_cdn_group(t, nc, nq, rnd((11, 8), 1), 100, 0.5, 1.0, True)
    num_dn = meta['dn_num_split'][0]
    Q = num_dn + nq
    db = torch.sigmoid(rnd((nl - 1, B, Q, 4), 21)).requires_grad_()  # decoder layers (dn + queries)
    ds = rnd((nl - 1, B, Q, nc), 22, 2.0).requires_grad_()
    eb = torch.sigmoid(rnd((B, nq, 4), 23)).requires_grad_()
    es = rnd((B, nq, nc), 24, 2.0).requires_grad_()
    dn_b, dec_b = torch.split(db, meta['dn_num_split'], dim=2)
    dn_s, dec_s = torch.split(ds, meta['dn_num_split'], dim=2)
    dec_b = torch.cat([eb.unsqueeze(0), dec_b])
    dec_s = torch.cat([es.unsqueeze(0), dec_s])
    crit = RTDETRDetectionLoss(nc=nc, use_vfl=True, use_sl=False, use_emasl=False, use_svfl=False, use_emasvfl=False)
    loss = crit((dec_b, dec_s), t, dn_bboxes=dn_b, dn_scores=dn_s, dn_meta=meta)
    tot = sum(loss.values())
    tot.backward()
    d.update(dec_bboxes=db, dec_scores=ds, enc_bboxes=eb, enc_scores=es, cls=t['cls'], bboxes=t['bboxes'],
             batch_idx=t['batch_idx'], n_per=np.asarray(t['gt_groups']), num_group=meta['dn_num_group'],
             split=np.asarray(meta['dn_num_split']), total=tot,
             g_dec_bboxes=db.grad, g_dec_scores=ds.grad, g_enc_bboxes=eb.grad, g_enc_scores=es.grad)
    for i, p in enumerate(meta['dn_pos_idx']):
        d[f'pos_idx{i}'] = p
    for k, v in loss.items():
        d[f'loss.{k}'] = v
    # no-denoising branch
    loss2 = crit((dec_b.detach(), dec_s.detach()), t, dn_bboxes=None, dn_scores=None, dn_meta=None)
    for k, v in loss2.items():
        d[f'loss_nodn.{k}'] = v
    # no-GT image batch
    t0 = make_targets(B, [0, 0], 19)
    loss3 = crit((dec_b.detach(), dec_s.detach()), t0, dn_bboxes=None, dn_scores=None, dn_meta=None)
    for k, v in loss3.items():
        d[f'loss_nogt.{k}'] = v
    save('loss', d)


class _Surrogate:
    """Closed-form, differentiable stand-in passed through forward_corev2's own `SelectiveScan=` argument
    (vmamba.py:919).  It mixes every operand so that the plumbing before and after the scan is pinned; the S6
    recurrence itself is NOT pinned by this (parity unpinned for a-9)."""
    record = {}

    @staticmethod
    def apply(u, delta, A, Bm, Cm, D, delta_bias, delta_softplus, nrows, backnrows, oflex):
        _Surrogate.record = dict(u=u, delta=delta, A=A, B=Bm, C=Cm, D=D, delta_bias=delta_bias)
        Bn, KD, L = u.shape
        K = Bm.shape[1]
        dt = torch.nn.functional.softplus(delta + delta_bias[None, :, None])
        bc = (Bm * Cm).sum(2)  # [B,K,L]
        bc = bc[:, :, None, :].expand(Bn, K, KD // K, L).reshape(Bn, KD, L)
        return u * D[None, :, None] + dt * bc * torch.exp(A.mean(1))[None, :, None]


def gen_vss():
    """a-9 VSSBlock (VManba/vmamba.py:1169-1256) around the scan; CrossScan/CrossMerge (csms6s.py:4-46)."""
    from functools import partial
    from ultralytics.nn.extra_modules.VManba.vmamba import VSSBlock
    from ultralytics.nn.extra_modules.VManba.csms6s import CrossScan, CrossMerge
    d = {}
    x = rnd((2, 6, 5, 7), 1).requires_grad_()
    xs = CrossScan.apply(x)
    cot = rnd(xs.shape, 2)
    (xs * cot).sum().backward()
    d.update(cs_x=x, cs_out=xs, cs_cot=cot, cs_gx=x.grad)
    ys = rnd((2, 4, 6, 5, 7), 3).requires_grad_()
    y = CrossMerge.apply(ys)
    cot = rnd(y.shape, 4)
    (y * cot).sum().backward()
    d.update(cm_ys=ys, cm_out=y, cm_cot=cot, cm_gys=ys.grad)

    blk = VSSBlock(hidden_dim=32, drop_path=0.1)
    d['wsum'] = load_filled(blk, seed=51)
    blk.op.forward_core = partial(blk.op.forward_corev2, force_fp32=True, SelectiveScan=_Surrogate)
    inp = rnd((2, 6, 5, 32), 5).requires_grad_()  # NHWC
    out = blk(inp)
    cot = rnd(out.shape, 6)
    d.update(x=inp, out=out, cot=cot)
    for k, v in _Surrogate.record.items():
        d[f'scan_in.{k}'] = v
    d.update(grads_of(blk, out, cot, {'x': inp}))
    save('vss', d)


def gen_head():
    """a-3 ManbaWorldDecoder (modules/head.py:1005-1290) with VSSBlocks := identity (flag vss_identity=1)."""
    from ultralytics.nn.modules.head import ManbaWorldDecoder
    d = {'vss_identity': 1}
    ch, hd, nq, nh, ndl, ffn, nc = [32, 64, 128], 128, 20, 4, 3, 256, 10
    m = ManbaWorldDecoder(nc, ch, hd, nq, 4, nh, ndl, ffn, dims=ch, embed=hd)
    m.VSSBlocks = nn.ModuleList([nn.Identity() for _ in ch])
    set_bn(m)
    d['wsum'] = load_filled(m, seed=61)
    B = 2
    # level 0 is 64x64: its border ring is "invalid" (eps=1e-2, head.py:1197-1199); level 1 is non-square (8x6) and
    # pins the reference's x/h, y/w normalisation (head.py:1188-1189: valid_WH=[h, w] divides (x, y)).
    # Inputs are NOT stored: x_i = rnd((B, c, h, w), 70 + i) (tests/golden/weights.py).
    sizes = [(64, 64), (8, 6), (3, 3)]
    xs = [rnd((B, c, h, w), 70 + i).requires_grad_() for i, (c, (h, w)) in enumerate(zip(ch, sizes))]
    text = torch.nn.functional.normalize(rnd((B, nc, hd), 80), dim=-1).requires_grad_()
    t = make_targets(B, [3, 2], 29)
    d.update(cfg=np.asarray([hd, nq, nh, ndl, ffn, nc]), ch=np.asarray(ch), sizes=np.asarray(sizes), text=text,
             cls=t['cls'], bboxes=t['bboxes'], batch_idx=t['batch_idx'], n_per=np.asarray(t['gt_groups']))
    m.train()
    torch.manual_seed(4321)
    db, ds, eb, es, meta = m(xs, text, t)
    cots = [rnd(o.shape, 90 + i) for i, o in enumerate((db, ds, eb, es))]
    d.update(dec_bboxes=db, dec_scores=ds, enc_bboxes=eb, enc_scores=es, split=np.asarray(meta['dn_num_split']),
             num_group=meta['dn_num_group'])
    for i, c in enumerate(cots):
        d[f'cot{i}'] = c
    ins = {f'x{i}': x for i, x in enumerate(xs)}
    ins['text'] = text
    d.update(grads_of(m, (db, ds, eb, es), cots, ins))
    for i in range(3):
        d[f'bn{i}_mean'] = m.input_proj[i][1].running_mean
        d[f'bn{i}_var'] = m.input_proj[i][1].running_var
    # tie check (SURVEY 8g "Top-k ties"): the invalid-anchor tie group must not straddle rank nq
    with torch.no_grad():
        m.eval()
        feats, shapes = m._get_encoder_input(xs)
        anchors, valid = m._generate_anchors(shapes, dtype=feats.dtype)
        sc = m.enc_score_head(m.enc_output(valid * feats)).max(-1).values
        srt = sc.sort(dim=1, descending=True).values
        assert (srt[:, nq - 1] - srt[:, nq]).abs().min() > 1e-4, 'top-k boundary tie: change seed'
        d['n_invalid'] = int((~valid).sum())
        y, _ = m(xs, text, None)
        d['y_eval'] = y
    save('head', d)


def _gen_e2e(wseed):
    """a-11: full RTDETRDetectionWorldModel (nn/tasks.py:518-672) on 256x256 images, VSSBlocks := identity.
    (64x64 was tried first: BatchNorm over 2x2x2 samples in the deepest layers makes loss and gradients chaotic under
    1e-6 input perturbations, so no cross-device comparison is meaningful there.)"""
    from ultralytics.nn.tasks import RTDETRDetectionWorldModel, yaml_model_load
    d = {'vss_identity': 1}
    torch.manual_seed(0)
    m = RTDETRDetectionWorldModel(yaml_model_load(REF + '/ultralytics/cfg/models/TAMTR/TAMTR.yaml'), nc=10,
                                  verbose=False)
    m.nc = 10
    head = m.model[-1]
    head.VSSBlocks = nn.ModuleList([nn.Identity() for _ in range(3)])
    d['n_params'] = sum(p.numel() for p in m.parameters())
    d['save'] = np.asarray(m.save)
    d['wseed'] = wseed
    d['wsum'] = load_filled(m, seed=wseed)
    B, S = 2, 256
    img = urnd((B, 3, S, S), 1)
    txt = torch.nn.functional.normalize(rnd((B, 10, 512), 2), dim=-1)
    t = make_targets(B, [3, 2], 39)
    batch = {'img': img, 'txt_feats': txt, 'cls': t['cls'].view(-1, 1).float(), 'bboxes': t['bboxes'],
             'batch_idx': t['batch_idx'].float()}
    d.update(S=S, txt=txt, cls=t['cls'], bboxes=t['bboxes'], batch_idx=t['batch_idx'],
             n_per=np.asarray(t['gt_groups']))
    scores = []
    hook = head.enc_score_head.register_forward_hook(lambda mod, i, o: scores.append(o.detach().max(-1).values))
    m.train()
    torch.manual_seed(999)
    loss, items = m(batch)
    d['loss'], d['loss_items'] = loss, items
    m.zero_grad(set_to_none=True)
    loss.backward()
    none = []
    for k, p in m.named_parameters():
        if p.grad is None:
            none.append(k)
        elif p.numel() <= 1024 or k.endswith('cv4.conv.weight') or 'value_proj.weight' in k:
            pack(d, f'g.{k}', summarize(p.grad, full_max=1024, n_sample=256))
    d['grad_none'] = np.asarray(none)
    d['bn0_mean'] = m.model[0].bn.running_mean
    d['attn16_bn_mean'] = m.model[16].attn.proj_conv.bn.running_mean
    # preds in train mode, same dn seed, for a direct output comparison
    torch.manual_seed(999)
    tg = {'cls': t['cls'], 'bboxes': t['bboxes'], 'batch_idx': t['batch_idx'], 'gt_groups': t['gt_groups']}
    with torch.no_grad():
        db, ds, eb, es, meta = m.predict(img, batch=tg, txt_feats=txt)
    d.update(dec_bboxes=db, dec_scores=ds, enc_bboxes=eb, enc_scores=es, split=np.asarray(meta['dn_num_split']))
    m.eval()
    with torch.no_grad():
        y, _ = m.predict(img, txt_feats=txt)
    d['y_eval'] = y
    hook.remove()
    gaps = []
    for sc in scores:  # SURVEY 8g "Top-k ties": the selection boundary (rank 100) must be clear of ties
        srt = sc.sort(dim=1, descending=True).values
        gaps.append(float((srt[:, 99] - srt[:, 100]).min()))
    d['topk_gap'] = np.asarray(gaps)
    if min(gaps) <= 1e-4:
        print(f'  weight seed {wseed}: top-k boundary nearly tied {gaps}, trying the next seed')
        return False
    save('e2e', d)
    return True


def gen_metrics():
    """Validation / training harness numbers from the reference's own functions: ap_per_class, match_predictions, box_iou
    (utils/metrics.py, engine/validator.py), ModelEMA (utils/torch_utils.py), build_optimizer grouping (engine/trainer.py)."""
    import types as _t
    from ultralytics.engine.trainer import BaseTrainer
    from ultralytics.engine.validator import BaseValidator
    from ultralytics.utils.metrics import ap_per_class, box_iou
    from ultralytics.utils.torch_utils import ModelEMA
    g = np.random.default_rng(11)
    d = {}
    n, ncls = 400, 6
    conf = g.random(n).astype(np.float32)
    pred_cls = g.integers(0, ncls - 1, n).astype(np.float32)          # class 5 never predicted
    target_cls = g.integers(0, ncls, 150).astype(np.float32)
    tp = (g.random((n, 10)) < np.linspace(0.7, 0.1, 10)[None] * conf[:, None] ** 0.3)
    tp = np.logical_and.accumulate(tp, 1)                              # a TP at a stricter IoU implies the looser ones
    out = ap_per_class(tp, conf, pred_cls, target_cls, plot=False, names={i: str(i) for i in range(ncls)})
    d.update({'ap.tp': tp, 'ap.conf': conf, 'ap.pred_cls': pred_cls, 'ap.target_cls': target_cls})
    for k, v in zip(('tpn', 'fpn', 'p', 'r', 'f1', 'ap', 'classes'), out[:7]):
        d['ap.out.' + k] = np.asarray(v)
    # matching
    b1 = torch.tensor(g.random((40, 4)).astype(np.float32)) * 300
    b1[:, 2:] = b1[:, :2] + 20 + b1[:, 2:] * 0.5
    b2 = b1[g.integers(0, 40, 90)] + torch.tensor(g.normal(0, 6, (90, 4)).astype(np.float32))
    pc, tc = torch.tensor(g.integers(0, 3, 90)).float(), torch.tensor(g.integers(0, 3, 40)).float()
    iou = box_iou(b1, b2)
    dummy = _t.SimpleNamespace(iouv=torch.linspace(0.5, 0.95, 10))
    correct = BaseValidator.match_predictions(dummy, pc, tc, iou)
    d.update({'match.labels_xyxy': b1.numpy(), 'match.dets_xyxy': b2.numpy(), 'match.pred_cls': pc.numpy(), 'match.true_cls': tc.numpy(),
              'match.iou': iou.numpy(), 'match.correct': correct.numpy()})
    # EMA
    torch.manual_seed(3)
    net = nn.Sequential(nn.Conv2d(3, 4, 3), nn.BatchNorm2d(4), nn.Flatten(), nn.Linear(4, 2))
    ema = ModelEMA(net, decay=0.9999, tau=2000)
    for step in range(3):
        with torch.no_grad():
            for q, p_ in enumerate(net.parameters()):
                p_.add_(0.1 * (step + 1) * (q + 1))
            net[1].running_mean.add_(0.5)
        ema.update(net)
    for k, v in ema.ema.state_dict().items():
        d['ema.' + k] = v.numpy()
    d['ema.updates'] = np.array(ema.updates)
    # optimizer groups on the TAMTR head-like mixture of layer types
    torch.manual_seed(4)
    mix = nn.Sequential(nn.Conv2d(3, 8, 3, bias=False), nn.BatchNorm2d(8), nn.Linear(8, 8), nn.LayerNorm(8), nn.Embedding(5, 8),
                        nn.MultiheadAttention(8, 2))
    tr = _t.SimpleNamespace(args=_t.SimpleNamespace(lr0=0.01, momentum=0.9, warmup_bias_lr=0.1))
    opt = BaseTrainer.build_optimizer(tr, mix, name='AdamW', lr=0.002, momentum=0.9, decay=5e-4)
    names = {id(p_): n_ for n_, p_ in mix.named_parameters()}
    for gi, grp in enumerate(opt.param_groups):
        d[f'opt.group{gi}.names'] = np.array([names[id(p_)] for p_ in grp['params']])
        d[f'opt.group{gi}.wd'] = np.array(grp['weight_decay'])
    save('metrics', d)


def gen_data():
    """Data path through the reference's own host functions: verify_image_label / img2label_paths
    (data/utils.py), Instances conversions (utils/instance.py), RandomPerspective's box path, RandomFlip, RandomLoadText, Mosaic,
    MixUp, Format (data/augment.py) and YOLODataset.collate_fn (data/dataset.py).

    cv2 is an inert placeholder in this container, so for RandomPerspective two of its entry points are given here:
    getRotationMatrix2D as OpenCV documents it for centre (0, 0) ([[a, b, 0], [-b, a, 0]], a = s cos, b = s sin) and a warpAffine
    that returns a blank canvas of the requested size - only the random draws, the matrix and the box path of that transform are
    stored, never its image."""
    import random
    import tempfile
    import types as _t
    import cv2
    from PIL import Image
    from ultralytics.data.augment import Format, MixUp, Mosaic, RandomFlip, RandomLoadText, RandomPerspective
    from ultralytics.data.dataset import YOLODataset
    from ultralytics.data.utils import img2label_paths, verify_image_label
    from ultralytics.utils.instance import Instances
    g = np.random.default_rng(21)
    d = {}

    # --- label files ------------------------------------------------------------------------------------------
    paths = ['/d/images/train/a.jpg', '/d/images/images/b.c.png', 'rel/images/x/y.jpeg']
    d['paths.in'], d['paths.out'] = np.array(paths), np.array(img2label_paths(paths))
    cases = {
        'plain': '3 0.5 0.5 0.2 0.1\n0 0.25 0.75 0.1 0.3\n9 0.9 0.1 0.05 0.05\n',
        'dups': '1 0.5 0.5 0.2 0.2\n4 0.3 0.3 0.1 0.1\n1 0.5 0.5 0.2 0.2\n0 0.1 0.2 0.05 0.06\n4 0.3 0.3 0.1 0.1\n',
        'blank_lines': '\n2 0.4 0.4 0.1 0.1\n\n5 0.6 0.6 0.2 0.2\n\n',
        'empty': '',
        'six_cols': '1 0.5 0.5 0.2 0.2 0.9\n',
        'four_cols': '1 0.5 0.5 0.2\n',
        'out_of_bounds': '1 0.5 1.5 0.2 0.2\n',
        'negative': '1 0.5 -0.1 0.2 0.2\n',
        'big_class': '11 0.5 0.5 0.2 0.2\n',
        'class_eq_nc': '10 0.5 0.5 0.2 0.2\n',
    }
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(f'{tmp}/images'), os.makedirs(f'{tmp}/labels')
        for name, text in list(cases.items()) + [('missing', None)]:
            Image.fromarray(g.integers(0, 255, (24, 40, 3), dtype=np.uint8)).save(f'{tmp}/images/{name}.png')
            if text is not None:
                with open(f'{tmp}/labels/{name}.txt', 'w') as f:
                    f.write(text)
            r = verify_image_label((f'{tmp}/images/{name}.png', f'{tmp}/labels/{name}.txt', '', False, 10, 0, 0))
            d[f'labels.{name}.text'] = np.array('' if text is None else text)
            d[f'labels.{name}.ok'] = np.array(r[0] is not None)
            if r[0] is not None:
                d[f'labels.{name}.lb'], d[f'labels.{name}.shape'] = r[1], np.array(r[2])
    d['labels.names'] = np.array(list(cases) + ['missing'])

    # --- box conversions --------------------------------------------------------------------------------------
    n = 12
    xywhn = np.concatenate([g.uniform(0.1, 0.9, (n, 2)), g.uniform(0.01, 0.2, (n, 2))], 1).astype(np.float32)
    ins = Instances(xywhn.copy(), bbox_format='xywh', normalized=True)
    ins.convert_bbox('xyxy')
    ins.denormalize(640, 480)
    d['box.xywhn'], d['box.xyxy_px'] = xywhn, ins.bboxes.copy()
    ins.add_padding(13, -7)
    ins.clip(600, 470)
    d['box.padded_clipped'] = ins.bboxes.copy()
    ins.convert_bbox('xywh')
    ins.normalize(600, 470)
    d['box.back_xywhn'] = ins.bboxes.copy()

    # --- RandomPerspective: draws, matrix, boxes, filter ------------------------------------------------------
    cv2.getRotationMatrix2D = lambda angle, center, scale: np.array(
        [[scale * math.cos(math.radians(angle)), scale * math.sin(math.radians(angle)), 0.0],
         [-scale * math.sin(math.radians(angle)), scale * math.cos(math.radians(angle)), 0.0]])
    cv2.warpAffine = lambda img, M, dsize, borderValue: np.zeros((dsize[1], dsize[0], img.shape[2]), np.uint8)
    for ci, (kw, border, seed) in enumerate([(dict(degrees=0.0, translate=0.1, scale=0.9, shear=0.0), None, 5),
                                               (dict(degrees=10.0, translate=0.2, scale=0.5, shear=2.0), None, 6),
                                               (dict(degrees=0.0, translate=0.1, scale=0.5, shear=0.0), (-32, -32), 7)]):
        h, w = (64, 64) if border is None else (128, 128)
        nb = 20
        b = np.concatenate([g.uniform(0.05, 0.95, (nb, 2)), g.uniform(0.02, 0.4, (nb, 2))], 1).astype(np.float32)
        cls = g.integers(0, 10, (nb, 1)).astype(np.float32)
        lab = {'img': np.zeros((h, w, 3), np.uint8), 'cls': cls.copy(), 'instances': Instances(b.copy(), bbox_format='xywh', normalized=True)}
        if border is not None:
            lab['mosaic_border'] = border
        t = RandomPerspective(pre_transform=None, **kw)
        random.seed(seed)
        captured = {}
        orig = t.affine_transform

        def spy(img, brd, _o=orig, _c=captured):
            out = _o(img, brd)
            _c['M'], _c['s'] = out[1].copy(), out[2]
            return out
        t.affine_transform = spy
        out = t(lab)
        p = f'affine{ci}.'
        d[p + 'kw'] = np.array([kw['degrees'], kw['translate'], kw['scale'], kw['shear']])
        d[p + 'border'] = np.array(border if border is not None else (0, 0))
        d[p + 'seed'], d[p + 'hw'] = np.array(seed), np.array((h, w))
        d[p + 'in.boxes'], d[p + 'in.cls'] = b, cls
        d[p + 'M'], d[p + 's'] = captured['M'], np.array(captured['s'])
        d[p + 'out.boxes'], d[p + 'out.cls'] = out['instances'].bboxes.copy(), out['cls'].copy()
        d[p + 'out.shape'] = np.array(out['img'].shape)
        d[p + 'next_draw'] = np.array(random.random())

    # --- RandomFlip -------------------------------------------------------------------------------------------
    img = g.integers(0, 255, (10, 14, 3), dtype=np.uint8)
    b = np.concatenate([g.uniform(0.1, 0.9, (6, 2)), g.uniform(0.02, 0.2, (6, 2))], 1).astype(np.float32)
    d['flip.img'], d['flip.boxes'] = img, b
    for direction in ('horizontal', 'vertical'):
        for normalized in (True, False):
            for seed in (0, 1, 2, 3):
                bb = b.copy() if normalized else b * np.array([14, 10, 14, 10], np.float32)
                lab = {'img': img.copy(), 'instances': Instances(bb.copy(), bbox_format='xywh', normalized=normalized)}
                random.seed(seed)
                out = RandomFlip(p=0.5, direction=direction)(lab)
                p = f'flip.{direction}.{int(normalized)}.{seed}.'
                d[p + 'img'], d[p + 'boxes'] = out['img'], out['instances'].bboxes.copy()

    # --- RandomLoadText ---------------------------------------------------------------------------------------
    names10 = ['pedestrian', 'people', 'bicycle', 'car', 'van', 'truck', 'tricycle', 'awning-tricycle', 'bus', 'motor']
    names_syn = ['person/human/pedestrian', 'car/automobile', 'bus', 'bike/bicycle/cycle', 'dog', 'cat/kitten']
    cfgs = [('visdrone', names10, dict(max_samples=10, padding=True), [3, 3, 0, 9, 4, 3]),
            ('visdrone_one', names10, dict(max_samples=10, padding=True), [5]),
            ('visdrone_none', names10, dict(max_samples=10, padding=True), []),
            ('syn_budget', names_syn, dict(max_samples=4, neg_samples=(1, 3), padding=True), [1, 4, 4]),
            ('syn_nopad', names_syn, dict(max_samples=80, neg_samples=(0, 2), padding=False), [0, 5, 2, 2]),
            ('syn_fmt', names_syn, dict(max_samples=3, neg_samples=(80, 80), padding=True, prompt_format='a photo of {}', padding_value='-'), [3])]
    d['text.cases'] = np.array([c[0] for c in cfgs])
    for name, names, kw, cl in cfgs:
        for seed in (0, 1, 2):
            cls = np.array(cl, np.float32).reshape(-1, 1)
            bx = g.uniform(0.1, 0.9, (len(cl), 4)).astype(np.float32)
            lab = {'texts': [v.split('/') for v in names], 'cls': cls.copy(), 'instances': Instances(bx.copy(), bbox_format='xywh', normalized=True)}
            random.seed(seed)
            out = RandomLoadText(**kw)(lab)
            p = f'text.{name}.{seed}.'
            d[p + 'in.cls'], d[p + 'in.boxes'] = cls, bx
            d[p + 'out.cls'] = np.asarray(out['cls']).reshape(-1).astype(np.int64)
            d[p + 'out.boxes'], d[p + 'out.texts'] = out['instances'].bboxes.copy(), np.array(out['texts'])
            d[p + 'next_draw'] = np.array(random.random())
    d['text.names10'], d['text.names_syn'] = np.array(names10), np.array(names_syn)

    # --- Mosaic (numpy only) and MixUp ------------------------------------------------------------------------
    s = 16

    class FakeSet:
        def __init__(self):
            self.buffer = list(range(5))
            self.imgs = [g.integers(0, 255, (s, s, 3), dtype=np.uint8) for _ in range(5)]
            self.boxes = [np.concatenate([g.uniform(0.1, 0.9, (k + 1, 2)), g.uniform(0.05, 0.5, (k + 1, 2))], 1).astype(np.float32) for k in range(5)]
            self.cls = [g.integers(0, 10, (k + 1, 1)).astype(np.float32) for k in range(5)]

        def __len__(self):
            return 5

        def get_image_and_label(self, i):
            return {'im_file': f'{i}.jpg', 'ori_shape': (s, s), 'resized_shape': (s, s), 'img': self.imgs[i].copy(), 'cls': self.cls[i].copy(),
                    'instances': Instances(self.boxes[i].copy(), bbox_format='xywh', normalized=True)}
    fs = FakeSet()
    for k in range(5):
        d[f'mosaic.src{k}.img'], d[f'mosaic.src{k}.boxes'], d[f'mosaic.src{k}.cls'] = fs.imgs[k], fs.boxes[k], fs.cls[k]
    for seed in (0, 1, 2, 3):
        random.seed(seed)
        out = Mosaic(fs, imgsz=s, p=1.0, n=4)(fs.get_image_and_label(seed))
        p = f'mosaic.{seed}.'
        d[p + 'img'], d[p + 'boxes'], d[p + 'cls'] = out['img'], out['instances'].bboxes.copy(), out['cls'].copy()
        d[p + 'border'] = np.array(out['mosaic_border'])
    random.seed(4)
    np.random.seed(4)
    base = fs.get_image_and_label(0)
    base['instances'].convert_bbox('xyxy')
    out = MixUp(fs, pre_transform=None, p=1.0)(base)
    d['mixup.img'], d['mixup.boxes'], d['mixup.cls'] = out['img'], out['instances'].bboxes.copy(), out['cls'].copy()

    # --- Format + collate -------------------------------------------------------------------------------------
    fmt = Format(bbox_format='xywh', normalize=True, return_mask=False, return_keypoint=False, batch_idx=True, mask_ratio=4, mask_overlap=True)
    samples = []
    for k, nb in enumerate((3, 0, 2)):
        img = g.integers(0, 255, (8, 12, 3), dtype=np.uint8)
        bx = (np.concatenate([g.uniform(1, 7, (nb, 2)), g.uniform(8, 11, (nb, 2))], 1)[:, [0, 1, 2, 3]]).astype(np.float32)
        bx[:, 2:] = bx[:, :2] + 1 + bx[:, 2:] * 0.1
        cls = np.arange(nb).reshape(-1, 1) if k != 2 else g.integers(0, 10, (nb, 1)).astype(np.float32)
        lab = {'im_file': f'{k}.jpg', 'ori_shape': (20, 30), 'resized_shape': (8, 12), 'img': img.copy(), 'cls': cls,
               'instances': Instances(bx.copy(), bbox_format='xyxy', normalized=False), 'texts': ['car', 'bus', '']}
        d[f'format.{k}.in.img'], d[f'format.{k}.in.boxes'], d[f'format.{k}.in.cls'] = img, bx, np.asarray(cls)
        out = fmt(lab)
        # the reference holds BGR and flips the channel axis; the build holds RGB: store the RGB view of the same picture
        d[f'format.{k}.out.img_bgr_flipped'] = out['img'].numpy()
        d[f'format.{k}.out.cls'], d[f'format.{k}.out.bboxes'] = out['cls'].numpy(), out['bboxes'].numpy()
        d[f'format.{k}.out.batch_idx'] = out['batch_idx'].numpy().copy()   # collate_fn adds the image index in place
        samples.append(out)
    bt = YOLODataset.collate_fn(samples)
    d['collate.img'], d['collate.cls'], d['collate.bboxes'] = bt['img'].numpy(), bt['cls'].numpy(), bt['bboxes'].numpy()
    d['collate.batch_idx'] = bt['batch_idx'].numpy()
    d['collate.im_file'], d['collate.ori_shape'] = np.array(bt['im_file']), np.array(bt['ori_shape'])
    save('data', d)


def gen_fuse():
    """The evaluation graph after fuse() (nn/tasks.py:121-152; SURVEY 8g "Fused eval graph"): fuse_conv_and_bn
    (utils/torch_utils.py:159-180), RepConvN.switch_to_deploy (extra_modules/block.py:53-124), and the full model's fused
    state_dict keys + eval predictions on the e2e fixture's weights and inputs (VSSBlocks := identity as there)."""
    from ultralytics.nn.extra_modules.block import RepConvN
    from ultralytics.nn.tasks import RTDETRDetectionWorldModel, yaml_model_load
    from ultralytics.utils.torch_utils import fuse_conv_and_bn
    d = {}
    for name, (c1, c2, k, s_, g_, bias) in {'plain': (6, 8, 3, 1, 1, False), 'biased_grouped': (8, 12, 1, 2, 4, True)}.items():
        torch.manual_seed(len(name))
        conv = nn.Conv2d(c1, c2, k, s_, k // 2, groups=g_, bias=bias)
        bn = nn.BatchNorm2d(c2, eps=1e-3, momentum=0.03)
        with torch.no_grad():
            bn.weight.copy_(urnd((c2,), 3, 0.5, 1.5)), bn.bias.copy_(rnd((c2,), 4, 0.3))
            bn.running_mean.copy_(rnd((c2,), 5, 0.5)), bn.running_var.copy_(urnd((c2,), 6, 0.2, 2.0))
        f = fuse_conv_and_bn(conv, bn)
        p = f'conv.{name}.'
        d[p + 'cfg'] = np.array([c1, c2, k, s_, g_, int(bias)])
        d[p + 'w'], d[p + 'b'] = conv.weight.detach(), (conv.bias.detach() if bias else np.zeros(0, np.float32))
        for kk in ('weight', 'bias', 'running_mean', 'running_var'):
            d[p + 'bn.' + kk] = getattr(bn, kk).detach()
        d[p + 'fused.w'], d[p + 'fused.b'] = f.weight.detach(), f.bias.detach()
    rep = set_bn(RepConvN(6, 6, 3, 1))
    d['rep.wsum'] = load_filled(rep, seed=41)
    rep.eval()
    x = rnd((2, 6, 9, 9), 7)
    with torch.no_grad():
        d['rep.x'], d['rep.y_before'] = x, rep(x)
        rep.switch_to_deploy()
        d['rep.w'], d['rep.b'], d['rep.y_after'] = rep.conv.weight.detach(), rep.conv.bias.detach(), rep(x)
    d['rep.keys'] = np.array(sorted(rep.state_dict()))
    # full model
    e2e = np.load(os.path.join(HERE, 'e2e.npz'))
    torch.manual_seed(0)
    m = RTDETRDetectionWorldModel(yaml_model_load(REF + '/ultralytics/cfg/models/TAMTR/TAMTR.yaml'), nc=10, verbose=False)
    m.nc = 10
    m.model[-1].VSSBlocks = nn.ModuleList([nn.Identity() for _ in range(3)])
    wsum = load_filled(m, seed=int(e2e['wseed']))
    assert abs(float(wsum) - float(e2e['wsum'])) < 1e-6 * abs(float(e2e['wsum']))
    m.eval()
    S = int(e2e['S'])
    img, txt = urnd((2, 3, S, S), 1), torch.from_numpy(e2e['txt'])
    with torch.no_grad():
        y0, _ = m.predict(img, txt_feats=txt)
        m.fuse(verbose=False)
        y1, _ = m.predict(img, txt_feats=txt)
    d['model.keys'] = np.array(sorted(m.state_dict()))
    d['model.n_bn'] = np.array(sum(isinstance(v, nn.BatchNorm2d) for v in m.modules()))
    d['model.y_eval'], d['model.y_eval_fused'] = y0, y1     # (fresh running statistics: e2e.npz's y_eval follows two training forwards)
    d['model.wseed'], d['model.S'], d['model.txt'] = e2e['wseed'], e2e['S'], e2e['txt']
    d['model.max_shift'] = np.array(float((y1 - y0).abs().max()))
    pack(d, 'model.w0', summarize(m.model[0].conv.weight))
    d['model.b0'] = m.model[0].conv.bias.detach()
    save('fuse', d)


def gen_detect():
    """Boundary class `Detect` (nn/modules/head.py:22-82): train-mode maps, eval output, bias_init, for nc 10 and two channel sets."""
    from ultralytics.nn.modules.head import Detect
    d = {}
    for tag, nc, ch, sizes in (('A', 10, (32, 64, 128), ((2, 8, 8), (2, 4, 4), (2, 2, 2))), ('B', 3, (16, 48), ((1, 6, 10), (1, 3, 5)))):
        m = set_bn(Detect(nc, ch))
        d[f'{tag}.wsum'] = load_filled(m, seed=21)
        m.stride = torch.tensor([8., 16., 32.][:len(ch)])
        xs = [rnd((b, c, h, w), 40 + i) for i, (c, (b, h, w)) in enumerate(zip(ch, sizes))]
        m.train()
        out = m([x.clone() for x in xs])
        for i, o in enumerate(out):
            d[f'{tag}.train{i}'] = o
        d[f'{tag}.bn_mean'] = m.cv2[0][0].bn.running_mean
        m.eval()
        y, raw = m([x.clone() for x in xs])
        d[f'{tag}.y'] = y
        d[f'{tag}.raw0'] = raw[0]
        m.bias_init()
        d[f'{tag}.bias_box'] = m.cv2[1][-1].bias
        d[f'{tag}.bias_cls'] = m.cv3[1][-1].bias
        d[f'{tag}.cfg'] = np.asarray([nc, len(ch), m.no, m.reg_max])
        d[f'{tag}.keys'] = np.asarray(sorted(m.state_dict()))
    save('detect', d)


def gen_e2e():
    for wseed in range(71, 91):
        if _gen_e2e(wseed):
            return
    raise RuntimeError('no weight seed with a clear top-k boundary')


if __name__ == '__main__':
    _import_reference()
    torch.set_num_threads(8)
    which = sys.argv[1:] or ['gate', 'msdeform', 'contrastive', 'decoder', 'cdn', 'loss', 'vss', 'head', 'e2e', 'metrics', 'data', 'fuse', 'detect']
    for w in which:
        globals()['gen_' + w]()
