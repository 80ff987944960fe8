"""Hot-path modules of TAM-TR behind the reference's ultralytics plugin surface (same class names, constructor and
forward signatures, same state_dict keys), with the arithmetic routed to the hand-written gfx950 kernels in ops.py.

  MaxSigmoidAttnBlock   ultralytics/nn/extra_modules/block.py:194-226   -> ops.maxsigmoid_gate
  TIAGELAN              ultralytics/nn/extra_modules/block.py:171-192
  MSDeformAttn          ultralytics/nn/modules/transformer.py:204-299   -> ops.linear_bf16 (value_proj), ops.ms_deform_attn_core
  DeformableTransformerDecoderLayer  transformer.py:498-558             -> ops.self_attention
  TextDeformableTransformerDecoder   transformer.py:835-891
  ContrastiveHeadMLP    ultralytics/nn/modules/block.py:522-541         -> ops.contrastive_logits
  MLP                   transformer.py:162-176
  C2f / C2fAttn         ultralytics/nn/modules/block.py:189-212,620-646  (API surface named by north_star)

All modules need CUDA(HIP) tensors in forward; there is no CPU path (TamtrHipError otherwise).
"""
import copy
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .backbone import Conv, RepNCSPELAN4


class MaxSigmoidAttnBlock(nn.Module):
    """Text-guided max-sigmoid gate: out = proj_conv(x) * sigmoid(max_n <embed, gl(guide)_n>/sqrt(hc) + bias) * scale."""

    def __init__(self, c1, c2, nh=1, ec=128, gc=512, scale=False):
        super().__init__()
        self.nh = nh
        self.hc = c2 // nh
        self.ec = Conv(c1, ec, k=1, act=False) if c1 != ec else None
        self.gl = nn.Linear(gc, ec)
        self.bias = nn.Parameter(torch.zeros(nh))
        self.proj_conv = Conv(c1, c2, k=3, s=1, act=False)
        self.scale = nn.Parameter(torch.ones(1, nh, 1, 1)) if scale else 1.0

    def forward(self, x, guide):
        bs, _, h, w = x.shape
        gk = self.gl(guide)  # [B,T,ec]
        pc = self.proj_conv
        if (self.ec is None and not torch.is_grad_enabled() and not isinstance(self.scale, torch.Tensor) and 'bn' in pc._modules
                and pc.bn.training and pc.bn.affine and isinstance(pc.act, nn.Identity) and ops.gate_cl_ok(x, x.shape[1], self.nh)):
            # NHWC trunk, nothing to differentiate (TIAGELAN's discarded evaluation): the BatchNorm of proj_conv is applied
            # inside the gate kernel's load of the raw convolution output - no apply pass, no NCHW repacking (next-3)
            if ops.conv3x3_cl_ok(x, pc.conv) and ops.bn_cl_ok(pc.conv.out_channels, x.dtype):
                # bf16: the 3x3 convolution itself on the MFMA kernel of csrc/conv3x3.hip, batch statistics from its epilogue
                v_raw, stats = ops.conv3x3_cl_stats(x, pc.conv, pc.bn)
                return ops.maxsigmoid_gate_cl(x, gk, self.bias, v_raw, stats, pc.bn, self.nh, 1.0)
            v_raw = pc.conv(x)
            if ops.is_cl(v_raw) and v_raw.dtype == x.dtype and ops.bn_cl_ok(v_raw.shape[1], v_raw.dtype):
                stats = ops.bn_stats_cl(v_raw.permute(0, 2, 3, 1).reshape(bs * h * w, -1), pc.bn)
                return ops.maxsigmoid_gate_cl(x, gk, self.bias, v_raw, stats, pc.bn, self.nh, 1.0)
            v = pc.post(v_raw)
            return ops.maxsigmoid_gate(x if x.dtype == v.dtype else x.to(v.dtype), gk, self.bias, v, self.nh, 1.0)
        embed = self.ec(x) if self.ec is not None else x
        v = self.proj_conv(x)
        if embed.dtype != v.dtype:
            embed = embed.to(v.dtype)
        out = ops.maxsigmoid_gate(embed, gk, self.bias, v, self.nh, 1.0)
        if isinstance(self.scale, torch.Tensor):  # learnable per-head scale (never enabled by TAMTR.yaml)
            out = (out.view(bs, self.nh, -1, h, w) * self.scale.unsqueeze(2).to(out.dtype)).view(bs, -1, h, w)
        return out


class TIAGELAN(RepNCSPELAN4):
    """BTA-PAN fusion block = GELAN trunk + text gate on the second split branch.

    Reference behaviour is kept bit-for-bit by default: the gate IS evaluated (its BatchNorm running
    statistics update in train mode) and its result is NOT used.  `use_attn_output=True` (non-default, outside parity)
    feeds the gated branch into the fuse conv instead, i.e. what the paper describes."""
    use_attn_output = False

    def __init__(self, c1, c2, c3, c4, c5=1, nh=8):
        super().__init__(c1, c2, c3, c4, c5)
        self.attn = MaxSigmoidAttnBlock(c4, c4, nh=nh, ec=c4)

    def forward(self, x, guide):
        y = self.branches(x)
        if self.use_attn_output:
            y[-3] = self.attn(y[-3], guide)
        else:
            with torch.no_grad():  # result discarded by the reference: no graph is ever needed
                self.attn(y[-3].detach(), guide.detach())
        return self.cv4(ops.cat_channels(y))


class MLP(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        dims = [input_dim] + [hidden_dim] * (num_layers - 1) + [output_dim]
        self.layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))

    def forward(self, x):
        for i, lin in enumerate(self.layers):
            x = ops.linear(x, lin)
            if i + 1 < self.num_layers:
                x = F.relu(x)
        return x


class ContrastiveHeadMLP(nn.Module):
    """Region-text logits: cosine(x, w) * exp(logit_scale) + bias, fused normalise+dot kernel."""

    def __init__(self):
        super().__init__()
        self.bias = nn.Parameter(torch.tensor([-10.0]))
        self.logit_scale = nn.Parameter(torch.ones([]) * math.log(1 / 0.07))

    def forward(self, x, w):
        return ops.contrastive_logits(x, w, self.logit_scale, self.bias)


class MSDeformAttn(nn.Module):
    """Multi-scale deformable attention.  value_proj is the dominant dense contraction of the whole head
    (M = B*L rows); in bf16 mode it runs on the hand-written MFMA kernel and leaves `value` in [B, L, nh, dh] bf16,
    the layout the gather kernel consumes."""

    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError(f'd_model must be divisible by n_heads, but got {d_model} and {n_heads}')
        self.im2col_step = 64
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        nn.init.zeros_(self.sampling_offsets.weight)
        th = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        ring = torch.stack([th.cos(), th.sin()], -1)
        ring = ring / ring.abs().max(-1, keepdim=True).values
        ring = ring.view(self.n_heads, 1, 1, 2) * torch.arange(1, self.n_points + 1, dtype=torch.float32).view(1, 1, -1, 1)
        with torch.no_grad():
            self.sampling_offsets.bias.copy_(ring.expand(-1, self.n_levels, -1, -1).reshape(-1))
        nn.init.zeros_(self.attention_weights.weight)
        nn.init.zeros_(self.attention_weights.bias)
        for lin in (self.value_proj, self.output_proj):
            nn.init.xavier_uniform_(lin.weight)
            nn.init.zeros_(lin.bias)

    def project_value(self, value):
        """value_proj (+bias) -> [B, L, nh, dh]; bf16 activations go through the MFMA kernel."""
        bs, len_v = value.shape[:2]
        if value.dtype == torch.bfloat16:
            v = ops.linear_bf16(value, self.value_proj.weight, self.value_proj.bias)
        else:
            v = F.linear(value, self.value_proj.weight, self.value_proj.bias)
        return v.view(bs, len_v, self.n_heads, self.d_model // self.n_heads)

    def forward(self, query, refer_bbox, value, value_shapes, value_mask=None):
        bs, len_q = query.shape[:2]
        len_v = value.shape[1]
        assert sum(s[0] * s[1] for s in value_shapes) == len_v
        paired = value_mask is None and ops.value_proj_msda_ok(value, self.value_proj, self.n_heads, len_q, self.n_points)
        if paired:
            v = None   # projected inside the paired node (ops.value_proj_msda)
        elif value_mask is not None:
            v = F.linear(value, self.value_proj.weight, self.value_proj.bias).masked_fill(value_mask[..., None], 0.0)
            v = v.view(bs, len_v, self.n_heads, -1)
        else:
            v = self.project_value(value)
        q32 = ops.shared_bf16(query.float(), self.sampling_offsets, self.attention_weights)   # (bf16 mode: one cast for both projections)
        off = ops.linear(q32, self.sampling_offsets) if q32.dtype == torch.bfloat16 else \
            F.linear(q32, self.sampling_offsets.weight.float(), self.sampling_offsets.bias.float())
        off = off.float().view(bs, len_q, self.n_heads, self.n_levels, self.n_points, 2)
        aw = ops.linear(q32, self.attention_weights) if q32.dtype == torch.bfloat16 else \
            F.linear(q32, self.attention_weights.weight.float(), self.attention_weights.bias.float())
        aw = F.softmax(aw.float().view(bs, len_q, self.n_heads, -1), -1).view(bs, len_q, self.n_heads, self.n_levels, self.n_points)
        ref = refer_bbox.float()
        n = ref.shape[-1]
        if n == 2:
            norm = torch.as_tensor(value_shapes, dtype=torch.float32, device=query.device).flip(-1)
            loc = ref[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
        elif n == 4:
            # ref_xy + off / n_points * ref_wh * 0.5 as one fused multiply-add on the offsets (the small factor is formed first)
            loc = torch.addcmul(ref[:, :, None, :, None, :2], off, ref[:, :, None, :, None, 2:] * (0.5 / self.n_points))
        else:
            raise ValueError(f'Last dim of reference_points must be 2 or 4, but got {n}.')
        out = ops.value_proj_msda(value, self.value_proj, self.n_heads, value_shapes, loc, aw) if paired else ops.ms_deform_attn_core(v, value_shapes, loc, aw)
        return ops.linear(out.to(query.dtype), self.output_proj)


class _SelfAttention(nn.Module):
    """Parameter container with nn.MultiheadAttention's state_dict keys; forward = packed in-proj, fused masked
    softmax attention kernel, out-proj (batch-first, no transposes)."""

    def __init__(self, d_model, n_heads):
        super().__init__()
        self.embed_dim, self.num_heads = d_model, n_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d_model, d_model))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d_model))
        self.out_proj = nn.Linear(d_model, d_model)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)

    def forward(self, qk, v, attn_mask=None):
        B, Q, C = qk.shape
        w, b = self.in_proj_weight, self.in_proj_bias
        qk_p = ops.linear_rows(qk, w, b, 0, 2 * C)  # q and k share their input: one GEMM
        v_p = ops.linear_rows(v, w, b, 2 * C, 3 * C)
        q, k = qk_p[..., :C], qk_p[..., C:]
        o = ops.self_attention(q, k, v_p, self.num_heads, attn_mask)
        return ops.linear(o, self.out_proj)


class DeformableTransformerDecoderLayer(nn.Module):
    def __init__(self, d_model=256, n_heads=8, d_ffn=1024, dropout=0., act=nn.ReLU(), n_levels=4, n_points=4):
        super().__init__()
        if dropout != 0.:
            raise NotImplementedError('TAM-TR trains with dropout=0 (head.py:1026); non-zero dropout is not built')
        self.self_attn = _SelfAttention(d_model, n_heads)
        self.norm1 = nn.LayerNorm(d_model)
        self.cross_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.norm2 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.act = act
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.norm3 = nn.LayerNorm(d_model)

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else tensor + pos

    def forward_ffn(self, tgt):
        return ops.layer_norm_module(self.norm3, tgt + ops.linear(self.act(ops.linear(tgt, self.linear1)), self.linear2))

    def forward(self, embed, refer_bbox, feats, shapes, padding_mask=None, attn_mask=None, query_pos=None):
        qk = self.with_pos_embed(embed, query_pos)
        embed = ops.layer_norm_module(self.norm1, embed + self.self_attn(qk, embed, attn_mask))
        t = self.cross_attn(self.with_pos_embed(embed, query_pos), refer_bbox.unsqueeze(2), feats, shapes, padding_mask)
        embed = ops.layer_norm_module(self.norm2, embed + t)
        return self.forward_ffn(embed)


def inverse_sigmoid(x, eps=1e-5):
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


class TextDeformableTransformerDecoder(nn.Module):
    def __init__(self, hidden_dim, decoder_layer, num_layers, eval_idx=-1):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(decoder_layer) for _ in range(num_layers)])
        self.num_layers = num_layers
        self.hidden_dim = hidden_dim
        self.eval_idx = eval_idx if eval_idx >= 0 else num_layers + eval_idx

    def forward(self, embed, refer_bbox, feats, shapes, text, bbox_head, score_head, pos_mlp, attn_mask=None,
                padding_mask=None):
        output, dec_bboxes, dec_cls, last_refined = embed, [], [], None
        refer_bbox = refer_bbox.sigmoid()
        for i, layer in enumerate(self.layers):
            f_i = feats[i] if isinstance(feats, (list, tuple)) else feats   # per-layer handle on the token memory (ops.fanout)
            output = layer(output, refer_bbox, f_i, shapes, padding_mask, attn_mask, pos_mlp(refer_bbox))
            bbox = bbox_head[i](output)
            refined = ops.box_refine(bbox, refer_bbox)       # sigmoid(bbox + inverse_sigmoid(refer_bbox)), one kernel
            if self.training:
                dec_cls.append(score_head[i](output, text))
                dec_bboxes.append(refined if i == 0 else ops.box_refine(bbox, last_refined))
            elif i == self.eval_idx:
                dec_cls.append(score_head[i](output, text))
                dec_bboxes.append(refined)
                break
            last_refined = refined
            refer_bbox = refined.detach() if self.training else refined
        return torch.stack(dec_bboxes), torch.stack(dec_cls)


# ---------------------------------------------------------------------------------------------- API-surface modules
class Bottleneck(nn.Module):
    def __init__(self, c1, c2, shortcut=True, g=1, k=(3, 3), e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1, self.cv2 = Conv(c1, c_, k[0], 1), Conv(c_, c2, k[1], 1, g=g)
        self.add = shortcut and c1 == c2

    def forward(self, x):
        y = self.cv2(self.cv1(x))
        return x + y if self.add else y


class C2f(nn.Module):
    """ultralytics C2f (nn/modules/block.py:189-212): signature kept for drop-in; not instantiated by TAMTR.yaml."""

    def __init__(self, c1, c2, n=1, shortcut=False, g=1, e=0.5):
        super().__init__()
        self.c = int(c2 * e)
        self.cv1 = Conv(c1, 2 * self.c, 1, 1)
        self.cv2 = Conv((2 + n) * self.c, c2, 1)
        self.m = nn.ModuleList(Bottleneck(self.c, self.c, shortcut, g, k=((3, 3), (3, 3)), e=1.0) for _ in range(n))

    def forward(self, x):
        y = list(self.cv1(x).chunk(2, 1))
        y.extend(m(y[-1]) for m in self.m)
        return self.cv2(torch.cat(y, 1))


class C2fAttn(nn.Module):
    """YOLO-World C2fAttn (nn/modules/block.py:620-646): C2f whose last branch is the text gate (result USED here)."""

    def __init__(self, c1, c2, n=1, ec=128, nh=1, gc=512, shortcut=False, g=1, e=0.5):
        super().__init__()
        self.c = int(c2 * e)
        self.cv1 = Conv(c1, 2 * self.c, 1, 1)
        self.cv2 = Conv((3 + n) * self.c, c2, 1)
        self.m = nn.ModuleList(Bottleneck(self.c, self.c, shortcut, g, k=((3, 3), (3, 3)), e=1.0) for _ in range(n))
        self.attn = MaxSigmoidAttnBlock(self.c, self.c, gc=gc, ec=ec, nh=nh)

    def forward(self, x, guide):
        y = list(self.cv1(x).chunk(2, 1))
        y.extend(m(y[-1]) for m in self.m)
        y.append(self.attn(y[-1], guide))
        return self.cv2(torch.cat(y, 1))
