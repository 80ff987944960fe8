"""CPU oracle for the TAM-TR text-image attention hot path (BTA-PAN gate + MEH) - TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product package
(tam-tr_amd/) never does, and it raises if its HIP library is missing instead of falling back to anything here.

What it is: a plain fp32 PyTorch-CPU restatement of the reference's arithmetic for every row of SURVEY.md section 8a,
written as *pure functions over a flat state_dict* (a functional interpreter, not nn.Modules), each citing the
reference file:line (relative to /root/reference) it follows.  It is pinned by tests/golden/*.npz, which were
produced by importing the reference itself on CPU (tests/golden/make_golden.py).

Parity status: every row is pinned by reference-generated fixtures EXCEPT the selective scan inside a-9
(`selective_scan`): the reference's scan is an external CUDA extension that is absent from the tree and cannot run
here, so that one function is "parity unpinned" - it restates the published S6 recurrence from the call contract
in VManba/vmamba.py:962-990 and is cross-checked only against its own C twin (oracle/selscan_ref.c).
"""
import math

import torch
import torch.nn.functional as F

BN_EPS, BN_MOM = 1e-3, 0.03  # utils/torch_utils.py:303-313 rewrites every BatchNorm2d of a built model


class View:
    """Prefix view over a flat state dict: V['a.b'] -> state[prefix + 'a.b']; V.sub('x.') narrows."""

    def __init__(self, state, prefix=''):
        self.s, self.p = state, prefix

    def __getitem__(self, k):
        return self.s[self.p + k]

    def has(self, k):
        return (self.p + k) in self.s

    def sub(self, k):
        return View(self.s, self.p + k)


# ------------------------------------------------------------------------------------------------ backbone bits
def _bn(x, P, train):
    return F.batch_norm(x, P['running_mean'], P['running_var'], P['weight'], P['bias'], train, BN_MOM, BN_EPS)


def conv(x, P, s=1, act=True, train=False, g=1):
    """Conv = conv(no bias, pad k//2) + BN + SiLU   (nn/modules/conv.py:23-40)."""
    w = P['conv.weight']
    k = w.shape[-1]
    y = _bn(F.conv2d(x, w, None, s, k // 2, 1, g), P.sub('bn.'), train)
    return F.silu(y) if act else y


def repconvn(x, P, train):
    """RepConvN training form: SiLU(conv3x3+BN + conv1x1+BN)   (extra_modules/block.py:26-50)."""
    return F.silu(conv(x, P.sub('conv1.'), act=False, train=train) + conv(x, P.sub('conv2.'), act=False, train=train))


def repncsp(x, P, train):
    """RepNCSP(n=1) with one RepNBottleneck(shortcut, e=1)   (extra_modules/block.py:124-148)."""
    a = conv(x, P.sub('cv1.'), train=train)
    b = P.sub('m.0.')
    a = a + conv(repconvn(a, b.sub('cv1.'), train), b.sub('cv2.'), train=train)
    return conv(torch.cat((a, conv(x, P.sub('cv2.'), train=train)), 1), P.sub('cv3.'), train=train)


def _elan_trunk(x, P, train):
    y = list(conv(x, P.sub('cv1.'), train=train).chunk(2, 1))
    for name in ('cv2.', 'cv3.'):
        q = P.sub(name)
        y.append(conv(repncsp(y[-1], q.sub('0.'), train), q.sub('1.'), train=train))
    return y


def repncspelan4(x, P, train):
    """extra_modules/block.py:150-163."""
    return conv(torch.cat(_elan_trunk(x, P, train), 1), P.sub('cv4.'), train=train)


def sppelan(x, P, train):
    """extra_modules/block.py:255-270: cv1, 3 chained maxpool5, cv5."""
    y = [conv(x, P.sub('cv1.'), train=train)]
    for _ in range(3):
        y.append(F.max_pool2d(y[-1], 5, 1, 2))
    return conv(torch.cat(y, 1), P.sub('cv5.'), train=train)


def cpam(x):
    """CPAM (extra_modules/block.py:271-308): channel gate sigmoid(up2(maxpool3s2(x)))*x, then per 8-chunk
    spatial gate sigmoid(max_c)."""
    c = torch.sigmoid(F.interpolate(F.max_pool2d(x, 3, 2, 1), scale_factor=2, mode='bilinear', align_corners=False)) * x
    return torch.cat([torch.sigmoid(s.max(1, keepdim=True).values) * s for s in c.chunk(8, 1)], 1)


# ------------------------------------------------------------------------------------------------ a-1 / a-2
def maxsigmoid_gate_weights(x, guide, P, nh):
    """aw of MaxSigmoidAttnBlock.forward (extra_modules/block.py:212-221): gl -> per-head dot with x -> max over
    text -> /sqrt(hc) -> + bias -> sigmoid (scale == 1.0; ec conv absent because c1 == ec)."""
    B, C, H, W = x.shape
    hc = C // nh
    g = F.linear(guide, P['gl.weight'], P['gl.bias']).view(B, -1, nh, hc)  # [B,T,nh,hc]
    e = x.view(B, nh, hc, H * W)
    aw = torch.einsum('bmcp,bnmc->bmpn', e, g).max(-1).values  # [B,nh,HW]
    aw = aw / (hc ** 0.5) + P['bias'][None, :, None]
    return torch.sigmoid(aw).view(B, nh, 1, H, W)


def maxsigmoid_attn_block(x, guide, P, nh, train):
    """a-1 MaxSigmoidAttnBlock.forward (extra_modules/block.py:208-226)."""
    B, C, H, W = x.shape
    aw = maxsigmoid_gate_weights(x, guide, P, nh)
    v = conv(x, P.sub('proj_conv.'), act=False, train=train)
    return (v.view(B, nh, -1, H, W) * aw).view(B, -1, H, W)


def tiagelan(x, guide, P, nh, train):
    """a-2 TIAGELAN.forward (extra_modules/block.py:182-186).  The attention result is computed and DISCARDED
    (SURVEY D2); only its BatchNorm running statistics are a side effect."""
    y = _elan_trunk(x, P, train)
    maxsigmoid_attn_block(y[-3], guide, P.sub('attn.'), nh, train)
    return conv(torch.cat(y, 1), P.sub('cv4.'), train=train)


# ------------------------------------------------------------------------------------------------ a-6 / a-5
def ms_deform_attn_core(value, shapes, loc, aw):
    """a-6 multi_scale_deformable_attn_pytorch (nn/modules/utils.py:42-89), restated without grid_sample:
    bilinear(align_corners=False, zero padding) == pixel coords x = loc*W - 0.5, 4 corners, OOB corners weigh 0.
    value [B,L,M,D], loc [B,Q,M,nl,P,2] (x,y in 0..1), aw [B,Q,M,nl,P] -> [B,Q,M*D]."""
    B, L, M, D = value.shape
    _, Q, _, nl, Pn, _ = loc.shape
    out = value.new_zeros(B, Q, M, D)
    start = 0
    bi = torch.arange(B).view(B, 1, 1, 1)
    mi = torch.arange(M).view(1, 1, M, 1)
    for l, (H, W) in enumerate(shapes):
        H, W = int(H), int(W)
        v = value[:, start:start + H * W]  # [B,HW,M,D]
        start += H * W
        x = loc[:, :, :, l, :, 0] * W - 0.5  # [B,Q,M,P]
        y = loc[:, :, :, l, :, 1] * H - 0.5
        x0, y0 = torch.floor(x), torch.floor(y)
        fx, fy = x - x0, y - y0
        for dy, wy in ((0, 1 - fy), (1, fy)):
            for dx, wx in ((0, 1 - fx), (1, fx)):
                xi, yi = (x0 + dx).long(), (y0 + dy).long()
                ok = ((xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)).to(value.dtype)
                idx = yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1)  # [B,Q,M,P]
                samp = v[bi, idx, mi]  # [B,Q,M,P,D]
                out = out + (samp * (wx * wy * ok * aw[:, :, :, l, :]).unsqueeze(-1)).sum(3)
    return out.reshape(B, Q, M * D)


def msdeform_attn(query, refer_bbox, value, shapes, P, nh, n_points=4):
    """a-5 MSDeformAttn.forward (nn/modules/transformer.py:252-299); refer_bbox [B,Q,1,4] or [B,Q,nl,2]."""
    B, Q, C = query.shape
    nl = len(shapes)
    v = F.linear(value, P['value_proj.weight'], P['value_proj.bias']).view(B, -1, nh, C // nh)
    off = F.linear(query, P['sampling_offsets.weight'], P['sampling_offsets.bias']).view(B, Q, nh, nl, n_points, 2)
    aw = F.linear(query, P['attention_weights.weight'], P['attention_weights.bias']).view(B, Q, nh, nl * n_points)
    aw = torch.softmax(aw, -1).view(B, Q, nh, nl, n_points)
    if refer_bbox.shape[-1] == 2:
        norm = torch.as_tensor(shapes, dtype=query.dtype).flip(-1)
        loc = refer_bbox[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    elif refer_bbox.shape[-1] == 4:
        loc = refer_bbox[:, :, None, :, None, :2] + off / n_points * refer_bbox[:, :, None, :, None, 2:] * 0.5
    else:
        raise ValueError(f'Last dim of reference_points must be 2 or 4, but got {refer_bbox.shape[-1]}.')
    out = ms_deform_attn_core(v, shapes, loc, aw)
    return F.linear(out, P['output_proj.weight'], P['output_proj.bias'])


# ------------------------------------------------------------------------------------------------ a-7 / a-8 / a-4
def mha_self_attn(q_in, k_in, v_in, P, nh, mask):
    """nn.MultiheadAttention(d, nh) as called at transformer.py:546 (q = k = embed+pos, v = embed, bool mask with
    True = blocked, dropout 0), written out batch-first: packed in_proj, softmax(QK^T/sqrt(dh)+mask)V, out_proj."""
    B, Q, C = q_in.shape
    dh = C // nh
    w, b = P['in_proj_weight'], P['in_proj_bias']
    q = F.linear(q_in, w[:C], b[:C]).view(B, Q, nh, dh).transpose(1, 2)
    k = F.linear(k_in, w[C:2 * C], b[C:2 * C]).view(B, Q, nh, dh).transpose(1, 2)
    v = F.linear(v_in, w[2 * C:], b[2 * C:]).view(B, Q, nh, dh).transpose(1, 2)
    s = (q * (dh ** -0.5)) @ k.transpose(-1, -2)
    if mask is not None:
        s = s.masked_fill(mask[None, None], float('-inf'))
    o = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, Q, C)
    return F.linear(o, P['out_proj.weight'], P['out_proj.bias'])


def _ln(x, P):
    return F.layer_norm(x, x.shape[-1:], P['weight'], P['bias'], 1e-5)


def decoder_layer(embed, refer_bbox, feats, shapes, P, nh, attn_mask=None, query_pos=None, n_points=4):
    """a-7 DeformableTransformerDecoderLayer.forward (transformer.py:539-558), post-norm, dropout 0, ReLU FFN."""
    qk = embed if query_pos is None else embed + query_pos
    embed = _ln(embed + mha_self_attn(qk, qk, embed, P.sub('self_attn.'), nh, attn_mask), P.sub('norm1.'))
    q = embed if query_pos is None else embed + query_pos
    t = msdeform_attn(q, refer_bbox.unsqueeze(2), feats, shapes, P.sub('cross_attn.'), nh, n_points)
    embed = _ln(embed + t, P.sub('norm2.'))
    t = F.linear(F.relu(F.linear(embed, P['linear1.weight'], P['linear1.bias'])), P['linear2.weight'], P['linear2.bias'])
    return _ln(embed + t, P.sub('norm3.'))


def mlp(x, P, n):
    """MLP (transformer.py:162-176): ReLU between layers, none after the last."""
    for i in range(n):
        x = F.linear(x, P[f'layers.{i}.weight'], P[f'layers.{i}.bias'])
        if i < n - 1:
            x = F.relu(x)
    return x


def contrastive_head(x, w, P):
    """a-8 ContrastiveHeadMLP.forward (nn/modules/block.py:534-541): L2-normalise both (eps 1e-12), dot,
    * exp(logit_scale) + bias.  x [B,Q,C], w [B,K,C] -> [B,Q,K]."""
    xn = x / x.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    wn = w / w.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    return torch.einsum('bqc,bkc->bqk', xn, wn) * P['logit_scale'].exp() + P['bias']


def inverse_sigmoid(x, eps=1e-5):
    """nn/modules/utils.py:34-39."""
    x = x.clamp(0, 1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def text_decoder(embed, refer_bbox, feats, shapes, text, P, nh, n_layers, train, attn_mask=None, eval_idx=-1):
    """a-4 TextDeformableTransformerDecoder.forward (transformer.py:850-891).  P is the head's view (needs
    decoder.layers.*, dec_bbox_head.*, dec_score_head.*, query_pos_head.*)."""
    eval_idx = eval_idx if eval_idx >= 0 else n_layers + eval_idx
    out, boxes, cls, last = embed, [], [], None
    refer = refer_bbox.sigmoid()
    for i in range(n_layers):
        pos = mlp(refer, P.sub('query_pos_head.'), 2)
        out = decoder_layer(out, refer, feats, shapes, P.sub(f'decoder.layers.{i}.'), nh, attn_mask, pos)
        bbox = mlp(out, P.sub(f'dec_bbox_head.{i}.'), 3)
        refined = torch.sigmoid(bbox + inverse_sigmoid(refer))
        if train:
            cls.append(contrastive_head(out, text, P.sub(f'dec_score_head.{i}.')))
            boxes.append(refined if i == 0 else torch.sigmoid(bbox + inverse_sigmoid(last)))
        elif i == eval_idx:
            cls.append(contrastive_head(out, text, P.sub(f'dec_score_head.{i}.')))
            boxes.append(refined)
            break
        last = refined
        refer = refined.detach() if train else refined
    return torch.stack(boxes), torch.stack(cls)


# ------------------------------------------------------------------------------------------------ a-9 (VSS)
def cross_scan(x):
    """CrossScan.forward (VManba/csms6s.py:4-14): [B,C,H,W] -> [B,4,C,HW]: row-major, column-major, and both reversed."""
    a = x.flatten(2)
    b = x.transpose(2, 3).flatten(2)
    return torch.stack([a, b, a.flip(-1), b.flip(-1)], 1)


def cross_merge(ys, H, W):
    """CrossMerge.forward (VManba/csms6s.py:26-34): [B,4,D,HW] -> [B,D,HW]."""
    B, K, D, L = ys.shape
    y = ys[:, 0:2] + ys[:, 2:4].flip(-1)
    return y[:, 0] + y[:, 1].view(B, D, W, H).transpose(2, 3).reshape(B, D, L)


def selective_scan(u, delta, A, Bm, Cm, D, delta_bias, delta_softplus=True):
    """S6 recurrence per the call contract at VManba/vmamba.py:962-990 / csms6s.py:252-258 (PARITY UNPINNED, see
    header): u, delta [B,KD,L]; A [KD,N]; Bm, Cm [B,K,N,L]; D, delta_bias [KD].
        dt = softplus(delta + bias); h_t = exp(dt_t A) h_{t-1} + dt_t B_t u_t; y_t = <C_t, h_t> + D u_t.
    Plain sequential loop (differentiable, small L only); the C twin handles big L."""
    Bn, KD, L = u.shape
    K, N = Bm.shape[1], Bm.shape[2]
    dt = delta + delta_bias[None, :, None]
    if delta_softplus:
        dt = F.softplus(dt)
    rep = KD // K
    Be = Bm.repeat_interleave(rep, 1)  # [B,KD,N,L]
    Ce = Cm.repeat_interleave(rep, 1)
    h = u.new_zeros(Bn, KD, N)
    ys = []
    for t in range(L):
        dA = torch.exp(dt[:, :, t, None] * A[None])
        h = dA * h + (dt[:, :, t] * u[:, :, t])[..., None] * Be[..., t]
        ys.append((h * Ce[..., t]).sum(-1))
    return torch.stack(ys, -1) + u * D[None, :, None]


def ss2d(x, P, scan_fn=None):
    """SS2D.forwardv2 + forward_corev2 (VManba/vmamba.py:898-1038), forward_type v2: d_state 16, ssm_ratio 2,
    dt_rank = ceil(d/16), dwconv3+SiLU, 4-direction cross scan, fp32 scan, LayerNorm, * SiLU(z), out_proj.  x is NHWC."""
    scan_fn = scan_fn or selective_scan
    B, H, W, _ = x.shape
    xz = F.linear(x, P['in_proj.weight'])
    xi, z = xz.chunk(2, -1)
    z = F.silu(z)
    xi = xi.permute(0, 3, 1, 2).contiguous()
    d_inner = xi.shape[1]
    xi = F.silu(F.conv2d(xi, P['conv2d.weight'], P['conv2d.bias'], 1, 1, 1, d_inner))
    xpw, dtw = P['x_proj_weight'], P['dt_projs_weight']  # [K,R+2N,D], [K,D,R]
    K, _, R = dtw.shape
    N = P['A_logs'].shape[1]
    L = H * W
    xs = cross_scan(xi)  # [B,4,D,L]
    x_dbl = torch.einsum('bkdl,kcd->bkcl', xs, xpw)
    dts, Bs, Cs = torch.split(x_dbl, [R, N, N], 2)
    dts = torch.einsum('bkrl,kdr->bkdl', dts, dtw)
    As = -torch.exp(P['A_logs'].float())
    ys = scan_fn(xs.reshape(B, -1, L), dts.reshape(B, -1, L), As, Bs.contiguous(), Cs.contiguous(), P['Ds'].float(),
                 P['dt_projs_bias'].reshape(-1).float(), True)
    y = cross_merge(ys.view(B, K, -1, L), H, W)  # [B,D,L]
    y = _ln(y.transpose(1, 2), P.sub('out_norm.')).view(B, H, W, -1)
    return F.linear(y * z, P['out_proj.weight'])


def vss_block(x, P, scan_fn=None):
    """a-9 VSSBlock._forward (VManba/vmamba.py:1237-1256), NHWC; DropPath treated as identity (eval / p=0)."""
    x = x + ss2d(_ln(x, P.sub('norm.')), P.sub('op.'), scan_fn)
    h = F.linear(F.gelu(F.linear(_ln(x, P.sub('norm2.')), P['mlp.fc1.weight'], P['mlp.fc1.bias'])),
                 P['mlp.fc2.weight'], P['mlp.fc2.bias'])
    return x + h


# ------------------------------------------------------------------------------------------------ a-10 loss side
def xywh2xyxy(b):
    xy, wh = b[..., :2], b[..., 2:] / 2
    return torch.cat([xy - wh, xy + wh], -1)


def xyxy2xywh(b):
    return torch.cat([(b[..., :2] + b[..., 2:]) / 2, b[..., 2:] - b[..., :2]], -1)


def box_iou_xywh(b1, b2, riou=False, eps=1e-7):
    """bbox_iou(xywh=True[, RIOU=True]) (utils/metrics.py:91-130).  Returns [...,1]."""
    x1, y1, w1, h1 = b1.chunk(4, -1)
    x2, y2, w2, h2 = b2.chunk(4, -1)
    l1, r1, t1, d1 = x1 - w1 / 2, x1 + w1 / 2, y1 - h1 / 2, y1 + h1 / 2
    l2, r2, t2, d2 = x2 - w2 / 2, x2 + w2 / 2, y2 - h2 / 2, y2 + h2 / 2
    inter = (torch.minimum(r1, r2) - torch.maximum(l1, l2)).clamp(0) * \
            (torch.minimum(d1, d2) - torch.maximum(t1, t2)).clamp(0)
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    if not riou:
        return iou
    rho2 = ((l2 + r2 - l1 - r1) ** 2 + (t2 + d2 - t1 - d1) ** 2) / 4
    c2 = (torch.max(w1, h1) + torch.max(w2, h2) + torch.sqrt(rho2) + eps).pow(2)
    v = (4 / math.pi ** 2) * (torch.atan(w2 / h2) - torch.atan(w1 / h1)).pow(2)
    with torch.no_grad():
        alpha = v / (v - iou + (1 + eps))
    return iou - (rho2 / c2 + v * alpha)


# Test hook: when a dict, meh_head stores the selected anchor indices under 'top' (one entry per call) and _layer_loss appends every
# Hungarian assignment it computes under 'matches' (call order of _detr_loss: last layer first, then the aux layers 0..n-2), so a test
# can hand the SAME discrete choices to the implementation under test and compare everything downstream of them elementwise.
TRACE = None


def hungarian_match(pred_bboxes, pred_scores, gt_bboxes, gt_cls, gt_groups, alpha=0.25, gamma=2.0,
                    gain=(2.0, 5.0, 2.0)):
    """HungarianMatcher.forward (models/utils/ops.py:48-119) with the matcher gains DETRLoss installs
    (loss.py:61: class 2, bbox 5, giou 2).  Returns per image (query idx, global gt idx)."""
    from scipy.optimize import linear_sum_assignment
    bs, nq, nc = pred_scores.shape
    if sum(gt_groups) == 0:
        return [(torch.zeros(0, dtype=torch.long), torch.zeros(0, dtype=torch.long)) for _ in range(bs)]
    ps = torch.sigmoid(pred_scores.detach().reshape(-1, nc))[:, gt_cls]
    pb = pred_bboxes.detach().reshape(-1, 4)
    neg = (1 - alpha) * ps ** gamma * (-(1 - ps + 1e-8).log())
    pos = alpha * (1 - ps) ** gamma * (-(ps + 1e-8).log())
    c_cls = pos - neg
    c_l1 = (pb.unsqueeze(1) - gt_bboxes.unsqueeze(0)).abs().sum(-1)
    c_iou = 1.0 - box_iou_xywh(pb.unsqueeze(1), gt_bboxes.unsqueeze(0), riou=True).squeeze(-1)
    C = gain[0] * c_cls + gain[1] * c_l1 + gain[2] * c_iou
    C = torch.where(C.isnan() | C.isinf(), torch.zeros_like(C), C).view(bs, nq, -1).cpu()
    out, off = [], 0
    for i, c in enumerate(C.split(list(gt_groups), -1)):
        r, k = linear_sum_assignment(c[i].numpy())
        out.append((torch.as_tensor(r, dtype=torch.long), torch.as_tensor(k, dtype=torch.long) + off))
        off += gt_groups[i]
    return out


def _varifocal(pred, gt_score, label, alpha=0.75, gamma=2.0):
    """VarifocalLoss (utils/loss.py:146-153)."""
    w = alpha * pred.sigmoid().pow(gamma) * (1 - label) + gt_score * label
    return (F.binary_cross_entropy_with_logits(pred.float(), gt_score.float(), reduction='none') * w).mean(1).sum()


def _focal(pred, label, gamma=1.5, alpha=0.25):
    """FocalLoss (utils/loss.py:157-178), used only when an image batch has no GT."""
    loss = F.binary_cross_entropy_with_logits(pred, label, reduction='none')
    p = pred.sigmoid()
    pt = label * p + (1 - label) * (1 - p)
    loss = loss * (1.0 - pt) ** gamma * (label * alpha + (1 - label) * (1 - alpha))
    return loss.mean(1).sum()


def _layer_loss(pb, ps, gt_bboxes, gt_cls, gt_groups, nc, match=None, gains=(1.0, 5.0, 2.0)):
    """DETRLoss._get_loss (models/utils/loss.py:282-326) for one decoder layer -> (class, bbox, giou)."""
    if match is None:
        match = hungarian_match(pb, ps, gt_bboxes, gt_cls, gt_groups)
        if TRACE is not None:
            TRACE.setdefault('matches', []).append(match)
    bi = torch.cat([torch.full_like(s, i) for i, (s, _) in enumerate(match)])
    si = torch.cat([s for s, _ in match])
    gi = torch.cat([g for _, g in match])
    bs, nq = pb.shape[:2]
    p_sel, g_sel = pb[bi, si], gt_bboxes[gi]
    targets = torch.full((bs, nq), nc, dtype=gt_cls.dtype)
    targets[bi, si] = gt_cls[gi]
    gt_scores = torch.zeros(bs, nq)
    n = len(g_sel)
    if n:
        gt_scores[bi, si] = box_iou_xywh(p_sel.detach(), g_sel).squeeze(-1)
    one_hot = F.one_hot(targets, nc + 1)[..., :-1]
    gts = gt_scores.view(bs, nq, 1) * one_hot
    l_cls = _varifocal(ps, gts, one_hot) if n else _focal(ps, one_hot.float())
    l_cls = l_cls / (max(n, 1) / nq) * gains[0]
    if n == 0:
        z = torch.tensor(0.)
        return l_cls, z, z.clone()
    l_box = gains[1] * F.l1_loss(p_sel, g_sel, reduction='sum') / n
    l_iou = gains[2] * (1.0 - box_iou_xywh(p_sel, g_sel, riou=True)).sum() / n
    return l_cls, l_box, l_iou


def _detr_loss(pb, ps, t, nc, postfix='', match=None):
    """DETRLoss.forward (loss.py:327-373): last layer + summed aux layers."""
    out = {}
    c, b, g = _layer_loss(pb[-1], ps[-1], t['bboxes'], t['cls'], t['gt_groups'], nc, match)
    out[f'loss_class{postfix}'], out[f'loss_bbox{postfix}'], out[f'loss_giou{postfix}'] = c, b, g
    aux = [torch.zeros(()), torch.zeros(()), torch.zeros(())]
    for i in range(len(pb) - 1):
        c, b, g = _layer_loss(pb[i], ps[i], t['bboxes'], t['cls'], t['gt_groups'], nc, match)
        aux = [aux[0] + c, aux[1] + b, aux[2] + g]
    out[f'loss_class_aux{postfix}'], out[f'loss_bbox_aux{postfix}'], out[f'loss_giou_aux{postfix}'] = aux
    return out


def rtdetr_loss(dec_bboxes, dec_scores, t, nc, dn_bboxes=None, dn_scores=None, dn_meta=None):
    """a-10 RTDETRDetectionLoss.forward (models/utils/loss.py:384-416) -> dict of 12 terms."""
    tot = _detr_loss(dec_bboxes, dec_scores, t, nc)
    if dn_meta is not None:
        ng, groups = dn_meta['dn_num_group'], t['gt_groups']
        off, match = 0, []
        for i, n in enumerate(groups):
            if n > 0:
                match.append((dn_meta['dn_pos_idx'][i], (torch.arange(n) + off).repeat(ng)))
            else:
                match.append((torch.zeros(0, dtype=torch.long), torch.zeros(0, dtype=torch.long)))
            off += n
        tot.update(_detr_loss(dn_bboxes, dn_scores, t, nc, '_dn', match))
    else:
        tot.update({f'{k}_dn': torch.tensor(0.) for k in list(tot)})
    return tot


def cdn_group(t, nc, nq, class_embed, num_dn=100, cls_noise=0.5, box_noise=1.0, train=True):
    """get_cdn_group (models/utils/ops.py:152-291).  Draws from the global torch RNG in the reference's order
    (rand, randint_like, randint_like, rand_like) so that a preceding torch.manual_seed reproduces its output."""
    if not train or num_dn <= 0:
        return None, None, None, None
    groups = t['gt_groups']
    total, mx = sum(groups), max(groups)
    if mx == 0:
        return None, None, None, None
    ng = max(num_dn // mx, 1)
    bs = len(groups)
    cls = t['cls'].repeat(2 * ng)
    box = t['bboxes'].repeat(2 * ng, 1)
    bidx = t['batch_idx'].repeat(2 * ng).view(-1)
    neg = torch.arange(total * ng, dtype=torch.long) + ng * total
    if cls_noise > 0:
        idx = torch.nonzero(torch.rand(cls.shape) < cls_noise * 0.5).squeeze(-1)
        cls[idx] = torch.randint_like(idx, 0, nc, dtype=cls.dtype)
    if box_noise > 0:
        known = xywh2xyxy(box)
        diff = (box[..., 2:] * 0.5).repeat(1, 2) * box_noise
        sign = torch.randint_like(box, 0, 2) * 2.0 - 1.0
        part = torch.rand_like(box)
        part[neg] += 1.0
        known = (known + part * sign * diff).clip(0.0, 1.0)
        box = torch.logit(xyxy2xywh(known), eps=1e-6)
    n_dn = int(mx * 2 * ng)
    emb = class_embed[cls]
    pad_c = torch.zeros(bs, n_dn, emb.shape[-1])
    pad_b = torch.zeros(bs, n_dn, 4)
    within = torch.cat([torch.arange(n, dtype=torch.long) for n in groups])
    pos_idx = torch.stack([within + mx * i for i in range(ng)], 0)
    slot = torch.cat([within + mx * i for i in range(2 * ng)])
    pad_c[bidx, slot] = emb
    pad_b[bidx, slot] = box
    size = n_dn + nq
    mask = torch.zeros(size, size, dtype=torch.bool)
    mask[n_dn:, :n_dn] = True
    for i in range(ng):
        lo, hi = mx * 2 * i, mx * 2 * (i + 1)
        mask[lo:hi, hi:n_dn] = True
        mask[lo:hi, :lo] = True
    meta = {'dn_pos_idx': [p.reshape(-1) for p in pos_idx.split(list(groups), 1)], 'dn_num_group': ng,
            'dn_num_split': [n_dn, nq]}
    return pad_c, pad_b, mask, meta


# ------------------------------------------------------------------------------------------------ a-3 MEH head
def generate_anchors(shapes, grid_size=0.05, eps=1e-2):
    """ManbaWorldDecoder._generate_anchors (nn/modules/head.py:1177-1200).  NOTE the reference divides (x, y) by
    [h, w] (valid_WH = [h, w], head.py:1188-1189) - reproduced as is; it only matters for non-square maps."""
    out = []
    for i, (h, w) in enumerate(shapes):
        gy, gx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing='ij')
        xy = (torch.stack([gx, gy], -1) + 0.5) / torch.tensor([h, w], dtype=torch.float32)
        wh = torch.full_like(xy, grid_size * 2.0 ** i)
        out.append(torch.cat([xy, wh], -1).view(1, h * w, 4))
    a = torch.cat(out, 1)
    valid = ((a > eps) & (a < 1 - eps)).all(-1, keepdim=True)
    a = torch.log(a / (1 - a)).masked_fill(~valid, float('inf'))
    return a, valid


def meh_head(xs, text, targets, P, nh, nq, ndl, nc, train, vss='real', scan_fn=None, num_dn=100):
    """a-3 ManbaWorldDecoder.forward (nn/modules/head.py:1130-1175).  vss='identity' reproduces the fixture flag."""
    if vss == 'real':
        xs = [vss_block(f.permute(0, 2, 3, 1), P.sub(f'VSSBlocks.{i}.'), scan_fn).permute(0, 3, 1, 2)
              for i, f in enumerate(xs)]
    feats, shapes = [], []
    for i, f in enumerate(xs):  # _get_encoder_input (head.py:1202-1219)
        q = P.sub(f'input_proj.{i}.')
        y = _bn(F.conv2d(f, q['0.weight']), q.sub('1.'), train)
        feats.append(y.flatten(2).permute(0, 2, 1))
        shapes.append([f.shape[2], f.shape[3]])
    feats = torch.cat(feats, 1)
    dn_embed, dn_bbox, mask, meta = cdn_group(targets, nc, nq, P['denoising_class_embed.weight'], num_dn, 0.5, 1.0, train) \
        if targets is not None else (None, None, None, None)
    # _get_decoder_input (head.py:1221-1265)
    B = feats.shape[0]
    anchors, valid = generate_anchors(shapes)
    memory = _ln(F.linear(valid * feats, P['enc_output.0.weight'], P['enc_output.0.bias']), P.sub('enc_output.1.'))
    scores = F.linear(memory, P['enc_score_head.weight'], P['enc_score_head.bias'])
    top = torch.topk(scores.max(-1).values, nq, dim=1).indices  # [B,nq]
    if TRACE is not None:
        TRACE.setdefault('top', []).append(top)
    bi = torch.arange(B).unsqueeze(-1)
    top_feat = memory[bi, top]
    refer = mlp(top_feat, P.sub('enc_bbox_head.'), 3) + anchors[0][top]
    enc_bboxes = refer.sigmoid()
    if dn_bbox is not None:
        refer = torch.cat([dn_bbox, refer], 1)
    enc_scores = scores[bi, top]
    embed = top_feat
    if train:
        refer, embed = refer.detach(), embed.detach()
    if dn_embed is not None:
        embed = torch.cat([dn_embed, embed], 1)
    db, ds = text_decoder(embed, refer, feats, shapes, text, P, nh, ndl, train, mask)
    if train:
        return db, ds, enc_bboxes, enc_scores, meta
    return torch.cat((db.squeeze(0), ds.squeeze(0).sigmoid()), -1)


# ------------------------------------------------------------------------------------------------ a-11 full graph
# (from, kind, stride-or-arg) per cfg/models/TAMTR/TAMTR.yaml:9-67; -1 = previous layer.
GRAPH = [(-1, 'conv', 2), (-1, 'conv', 2), (-1, 'elan', 0), (-1, 'conv', 2), (-1, 'elan', 0), (-1, 'conv', 2),
         (-1, 'elan', 0), (-1, 'conv', 2), (-1, 'elan', 0), (-1, 'spp', 0),
         (-1, 'conv', 1), (-1, 'up', 2.0), (6, 'conv', 1), (4, 'conv', 1), (-1, 'up', 0.5), ([-1, 12, 11], 'cat', 0),
         (-1, 'tia', 8), (-1, 'cpam', 0),
         (-1, 'conv', 1), (-1, 'up', 2.0), (4, 'conv', 1), (2, 'conv', 1), (-1, 'up', 0.5), ([-1, 20, 19], 'cat', 0),
         (-1, 'tia', 4), (-1, 'cpam', 0),
         (-1, 'conv', 1), (-1, 'up', 2.0), (2, 'conv', 1), (0, 'conv', 1), (-1, 'up', 0.5), ([-1, 28, 27], 'cat', 0),
         (-1, 'tia', 2), (-1, 'cpam', 0),
         (-1, 'conv', 2), ([-1, 24], 'cat', 0), (-1, 'tia', 4), (-1, 'cpam', 0),
         (-1, 'conv', 2), ([-1, 16], 'cat', 0), (-1, 'tia', 8)]
HEAD_FROM = (32, 36, 40)


def tamtr_predict(state, img, txt, targets=None, train=False, vss='real', scan_fn=None, nq=100, nh=8, ndl=3, nc=10):
    """a-11 RTDETRDetectionWorldModel.predict (nn/tasks.py:625-672): layer walk with `from` routing, TIAGELAN text
    dispatch, head call with the *cloned* text features."""
    if len(txt) != len(img):
        txt = txt.repeat(len(img), 1, 1)
    ys, x = [], img
    for i, (f, kind, arg) in enumerate(GRAPH):
        P = View(state, f'model.{i}.')
        if f != -1:
            x = ys[f] if isinstance(f, int) else [x if j == -1 else ys[j] for j in f]
        if kind == 'conv':
            x = conv(x, P, s=arg, train=train)
        elif kind == 'elan':
            x = repncspelan4(x, P, train)
        elif kind == 'spp':
            x = sppelan(x, P, train)
        elif kind == 'up':
            x = F.interpolate(x, scale_factor=arg, mode='nearest')
        elif kind == 'cat':
            x = torch.cat(x, 1)
        elif kind == 'tia':
            x = tiagelan(x, txt, P, arg, train)
        elif kind == 'cpam':
            x = cpam(x)
        ys.append(x)
    P = View(state, f'model.{len(GRAPH)}.')
    return meh_head([ys[j] for j in HEAD_FROM], txt.clone(), targets, P, nh, nq, ndl, nc, train, vss, scan_fn)


def tamtr_loss(state, batch, train=True, vss='real', scan_fn=None, nc=10):
    """RTDETRDetectionWorldModel.loss (nn/tasks.py:580-623) -> (sum of 12 terms, [giou, class, bbox], dict)."""
    img = batch['img']
    bidx = batch['batch_idx'].long().view(-1)
    t = {'cls': batch['cls'].long().view(-1), 'bboxes': batch['bboxes'], 'batch_idx': bidx,
         'gt_groups': [int((bidx == i).sum()) for i in range(len(img))]}
    db, ds, eb, es, meta = tamtr_predict(state, img, batch['txt_feats'], t, train, vss, scan_fn, nc=nc)
    dn_b = dn_s = None
    if meta is not None:
        dn_b, db = torch.split(db, meta['dn_num_split'], 2)
        dn_s, ds = torch.split(ds, meta['dn_num_split'], 2)
    db = torch.cat([eb.unsqueeze(0), db])
    ds = torch.cat([es.unsqueeze(0), ds])
    terms = rtdetr_loss(db, ds, t, nc, dn_b, dn_s, meta)
    return sum(terms.values()), torch.stack([terms[k].detach() for k in ('loss_giou', 'loss_class', 'loss_bbox')]), terms
