"""Multi-modal Encoder-decoder Head (MEH).  Reference: ManbaWorldDecoder, ultralytics/nn/modules/head.py:1005-1290.

Data layout in HBM: the three input maps go VSSBlock (NHWC) -> 1x1 conv+BN -> flattened and concatenated once into
`feats` [B, L, hd] (L = 33 600 tokens at 640^2), which stays resident for the whole decoder: enc_output, the three
value projections and the gather kernels all read that one buffer.
"""
import math

import torch
import torch.nn as nn

from .backbone import batchnorm
from . import ops
from .loss import get_cdn_group
from .modules import ContrastiveHeadMLP, DeformableTransformerDecoderLayer, MLP, TextDeformableTransformerDecoder
from .vss import TallLinear, VSSBlock


class ManbaWorldDecoder(nn.Module):
    export = False

    def __init__(self, nc=80, ch=(512, 1024, 2048), hd=512, nq=300, ndp=4, nh=8, ndl=6, d_ffn=1024, eval_idx=-1, dropout=0.,
                 act=nn.ReLU(), nd=100, label_noise_ratio=0.5, box_noise_scale=1.0, learnt_init_query=False,
                 dims=(128, 256, 512), drop_path=(0.1, 0.1, 0.1), embed=512, with_bn=False):
        super().__init__()
        if with_bn:
            # (nn/modules/block.py:544-570: its forward hands a 3-D [B, C, Q] tensor to nn.BatchNorm2d, which raises for anything but 4-D
            # input - the reference's with_bn=True head cannot run either; TAMTR.yaml leaves it False)
            raise NotImplementedError('with_bn=True (BNContrastiveHeadMLP) is not built: not used by TAMTR.yaml, and not runnable in the reference')
        self.hidden_dim, self.nhead, self.nl, self.nc = hd, nh, len(ch), nc
        self.num_queries, self.num_decoder_layers = nq, ndl
        self.input_proj = nn.ModuleList(nn.Sequential(nn.Conv2d(c, hd, 1, bias=False), batchnorm(hd)) for c in ch)
        self.VSSBlocks = nn.ModuleList(VSSBlock(hidden_dim=d, drop_path=p) for d, p in zip(dims, drop_path))
        self.num_Blocks = len(dims)
        layer = DeformableTransformerDecoderLayer(hd, nh, d_ffn, dropout, act, self.nl, ndp)
        self.decoder = TextDeformableTransformerDecoder(hd, layer, ndl, eval_idx)
        self.denoising_class_embed = nn.Embedding(nc + 1, hd)
        self.num_denoising, self.label_noise_ratio, self.box_noise_scale = nd, label_noise_ratio, box_noise_scale
        self.learnt_init_query = learnt_init_query
        if learnt_init_query:
            self.tgt_embed = nn.Embedding(nq, hd)
        self.query_pos_head = MLP(4, 2 * hd, hd, num_layers=2)
        self.enc_output = nn.Sequential(nn.Linear(hd, hd), nn.LayerNorm(hd))
        self.enc_score_head = TallLinear(hd, nc)  # nn.Linear whose weight gradient over the B*L = 537 600 tokens is a split-K GEMM
        self.enc_bbox_head = MLP(hd, hd, 4, num_layers=3)
        self.dec_score_head = nn.ModuleList(ContrastiveHeadMLP() for _ in range(ndl))
        self.dec_bbox_head = nn.ModuleList(MLP(hd, hd, 4, num_layers=3) for _ in range(ndl))
        self._anchor_cache = {}
        # teacher forcing of the one discrete choice of the head (parity measurements): a LongTensor [B, nq] of anchor positions used
        # INSTEAD of torch.topk's picks in _get_decoder_input; None (always, outside tests) = the reference's top-k (head.py:1237)
        self.fixed_topk = None
        self.last_topk = None
        self._reset_parameters()

    def forward(self, x, text, batch=None):
        return self.decode(*self.encode(x), text, batch)

    def draw_drop_scales(self, n, device):
        """[num_Blocks, 2, n]: this step's DropPath factors of the VSS blocks (vss.VSSBlock.draw_drop_scales), drawn outside encode()
        so that a recorded graph of encode() is a deterministic function of its inputs."""
        if not all(hasattr(blk, 'draw_drop_scales') for blk in self.VSSBlocks):   # blocks swapped out (tests put nn.Identity here)
            return None
        return torch.stack([blk.draw_drop_scales(n, device) for blk in self.VSSBlocks])

    def encode(self, x, drop_scales=None):
        """The three trunk maps -> the token memory `feats` [B, L, hd] and the level shapes.  Shapes depend on the image size only.
        (Round 3 tried the three independent levels on two streams, to fill the wave slots the finest level's scans leave idle: as
        parallel branches of the recorded graphs, back-to-back replays of consecutive steps never returned from the runtime - the bench
        hung in its timed loop after three synchronised warm-up steps had passed - so the levels run one after the other.)"""
        # VSS blocks run channels-last ([B,H,W,C], head.py:1136-1140); their outputs stay token-major: the 1x1 input projection
        # is a GEMM over tokens and its result is already in the [B, L, hd] layout of the token memory
        toks = []
        for i, (blk, f) in enumerate(zip(self.VSSBlocks, x)):
            f = f.permute(0, 2, 3, 1)
            toks.append(blk(f) if drop_scales is None else blk(f, drop_scales[i]))
        return self._get_encoder_input(toks)

    def decode(self, feats, shapes, text, batch=None):
        dn_embed, dn_bbox, attn_mask, dn_meta = get_cdn_group(batch, self.nc, self.num_queries,
                                                              self.denoising_class_embed.weight, self.num_denoising,
                                                              self.label_noise_ratio, self.box_noise_scale, self.training)
        # the token memory has 1 + num_layers heavy consumers (enc_output, every layer's value_proj): their gradients are added in one
        # pass instead of pairwise by autograd
        if self.training and ops.enc_select_ok(feats, self.enc_output[0], self.enc_output[1], self.enc_score_head):
            # query selection as one autograd node that also hands out the decoder layers' handles: its backward works on the picked rows only
            f_enc, f_dec = feats, self.decoder.num_layers
        else:
            f_enc, *f_dec = ops.fanout(feats, 1 + self.decoder.num_layers) if self.training else (feats,) * (1 + self.decoder.num_layers)
        embed, refer_bbox, enc_bboxes, enc_scores, f_dec = self._get_decoder_input(f_enc, shapes, dn_embed, dn_bbox, handles=f_dec)
        dec_bboxes, dec_scores = self.decoder(embed, refer_bbox, f_dec, shapes, text, self.dec_bbox_head, self.dec_score_head,
                                              self.query_pos_head, attn_mask=attn_mask)
        x = dec_bboxes, dec_scores, enc_bboxes, enc_scores, dn_meta
        if self.training:
            return x
        y = torch.cat((dec_bboxes.squeeze(0), dec_scores.squeeze(0).sigmoid()), -1)
        return y if self.export else (y, x)

    def _generate_anchors(self, shapes, grid_size=0.05, dtype=torch.float32, device='cpu', eps=1e-2):
        """Grid-centre anchors in logit space.  The reference normalises (x, y) by [h, w] (head.py:1188-1189) - kept."""
        key = (tuple(map(tuple, shapes)), str(device))
        if key not in self._anchor_cache:
            out = []
            for i, (h, w) in enumerate(shapes):
                gy, gx = torch.meshgrid(torch.arange(h, dtype=torch.float32, device=device),
                                        torch.arange(w, dtype=torch.float32, device=device), indexing='ij')
                xy = (torch.stack([gx, gy], -1) + 0.5) / torch.tensor([h, w], dtype=torch.float32, device=device)
                out.append(torch.cat([xy, torch.full_like(xy, grid_size * 2.0 ** i)], -1).view(1, h * w, 4))
            a = torch.cat(out, 1)
            valid = ((a > eps) & (a < 1 - eps)).all(-1, keepdim=True)
            self._invalid_rows = (~valid.view(-1)).nonzero().view(-1)   # positions along L (cached with the anchors: no per-step sync)
            a = torch.log(a / (1 - a)).masked_fill(~valid, float('inf'))
            self._anchor_cache = {key: (a, valid)}
        a, valid = self._anchor_cache[key]
        return a.to(dtype), valid

    def _get_encoder_input(self, toks):
        """input_proj on channels-last tokens + concatenation over the levels (head.py:1202-1219).  toks: list of [B, H, W, C].
        On the GPU in training mode the three BatchNorms write straight into their segments of the token memory (ops.bn_cat_cl): the
        550 MB concatenation copy and the slice copies of its gradient do not exist."""
        ys = [self._project_linear(i, t) for i, t in enumerate(toks)]
        bns = [p[1] for p in self.input_proj]
        shapes = [[t.shape[1], t.shape[2]] for t in toks]
        if ops.bn_cat_cl_ok(ys, bns):
            return ops.bn_cat_cl(ys, bns, toks[0].shape[0]), shapes
        outs = [self._project_level(i, t) for i, t in enumerate(toks)]
        return torch.cat([o[0] for o in outs], 1), [o[1] for o in outs]

    def _project_linear(self, i, t):
        """The 1x1 convolution of input_proj[i] as a per-token linear map: t [B, H, W, C] -> [B*H*W, hd] (bf16: the MFMA kernel)."""
        conv = self.input_proj[i][0]
        B, H, W, C = t.shape
        t2 = t.reshape(B * H * W, C)
        w = conv.weight.view(conv.out_channels, C)
        if t2.is_cuda and t2.dtype == torch.bfloat16 and C % 64 == 0 and conv.out_channels % 128 == 0:
            return ops.linear_bf16(t2, w, None)
        return torch.nn.functional.linear(t2, w.to(t2.dtype))

    def _project_level(self, i, t):
        """input_proj[i] (Conv1x1 no bias + BatchNorm, head.py:1087) on channels-last tokens t [B, H, W, C] -> ([B, H*W, hd], [H, W]).
        A 1x1 convolution IS a per-token linear map, so it runs as a [B*H*W, C] x [C, hd] GEMM (bf16: the MFMA kernel) and BatchNorm
        takes its batch statistics over the token axis - no NCHW round trip, no transposing concatenation."""
        bn = self.input_proj[i][1]
        B, H, W, C = t.shape
        y = self._project_linear(i, t)
        C2 = y.shape[1]
        if y.is_cuda and bn.training and ops.bn_cl_ok(C2, y.dtype):
            y = ops.bn_act(y, bn, False)  # channels-last BatchNorm kernels (csrc/bn.hip)
        else:
            if bn.training and bn.track_running_stats:
                bn.num_batches_tracked += 1
            y = torch.nn.functional.batch_norm(y, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.training, bn.momentum, bn.eps)
        return y.view(B, H * W, -1), [H, W]

    def _get_decoder_input(self, feats, shapes, dn_embed=None, dn_bbox=None, text=None, handles=None):
        """Query selection (reference head.py:1205-1245).  handles: the decoder layers' handles on the token memory made by the caller, or
        their NUMBER when this call is to make them (the one-node selection, ops.enc_select); returned as the fifth value."""
        bs = feats.shape[0]
        anchors, valid = self._generate_anchors(shapes, dtype=torch.float32, device=feats.device)
        lin, norm = self.enc_output[0], self.enc_output[1]
        if isinstance(handles, int):
            top_feat, enc_scores, top, handles = ops.enc_select(feats, lin, norm, self.enc_score_head, self._invalid_rows, self.num_queries,
                                                                self.fixed_topk, handles)
        else:
            if feats.is_cuda and feats.dtype == torch.bfloat16:
                # same [B*L, 512] x [512, 512] contraction as the value projection on the MFMA kernel; `valid * feats` (head.py:1213) is not
                # materialised: the invalid-anchor rows of the product are set to the bias instead (two 550 MB multiply passes saved)
                y = ops.linear_bf16_zero_rows(feats, lin.weight, lin.bias, self._invalid_rows)
            else:
                y = lin(valid.to(feats.dtype) * feats)
            memory = VSSBlock._ln(norm, y)  # LayerNorm kernel in the activation dtype
            scores = self.enc_score_head(memory)
            top = torch.topk(scores.max(-1).values, self.num_queries, dim=1).indices if self.fixed_topk is None else self.fixed_topk.to(feats.device)
            bi = torch.arange(bs, device=feats.device).unsqueeze(-1)
            top_feat, enc_scores = memory[bi, top], scores[bi, top]
        self.last_topk = top.detach()   # this forward's picks (parity measurements replay them through fixed_topk)
        refer = self.enc_bbox_head(top_feat).float() + anchors[0][top]
        enc_bboxes = refer.sigmoid()
        if dn_bbox is not None:
            refer = torch.cat([dn_bbox, refer], 1)
        embed = self.tgt_embed.weight.unsqueeze(0).repeat(bs, 1, 1) if self.learnt_init_query else top_feat
        if self.training:
            refer = refer.detach()
            if not self.learnt_init_query:
                embed = embed.detach()
        if dn_embed is not None:
            embed = torch.cat([dn_embed.to(embed.dtype), embed], 1)
        return embed, refer, enc_bboxes, enc_scores, handles

    def _reset_parameters(self):
        bias_cls = float(-math.log((1 - 0.01) / 0.01)) / 80 * self.nc
        nn.init.constant_(self.enc_score_head.bias, bias_cls)
        for m in [self.enc_bbox_head, *self.dec_bbox_head]:
            nn.init.zeros_(m.layers[-1].weight)
            nn.init.zeros_(m.layers[-1].bias)
        nn.init.xavier_uniform_(self.enc_output[0].weight)
        bound = 1 / math.sqrt(self.hidden_dim)
        nn.init.uniform_(self.enc_output[0].bias, -bound, bound)
        if self.learnt_init_query:
            nn.init.xavier_uniform_(self.tgt_embed.weight)
        nn.init.xavier_uniform_(self.query_pos_head.layers[0].weight)
        nn.init.xavier_uniform_(self.query_pos_head.layers[1].weight)
        for layer in self.input_proj:
            nn.init.xavier_uniform_(layer[0].weight)
