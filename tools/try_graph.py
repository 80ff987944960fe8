#!/usr/bin/env python3
"""Experiment: hipGraph-capture trunk+head forward and its backward (torch.cuda.make_graphed_callables)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch
from tamtr_amd.model import RTDETRDetectionWorldModel
from tamtr_amd.loss import get_cdn_group
from tamtr_amd.modules import TIAGELAN
import torch.nn as nn

torch.manual_seed(0)
model = RTDETRDetectionWorldModel(nc=10).cuda().train()
model.autocast_dtype = torch.bfloat16
batch = synth_batch(16, 640, 1, 'cuda')
head = model.model[-1]

class Core(nn.Module):
    def __init__(self, m):
        super().__init__()
        self.m = m
    def forward(self, img, txt, dn_embed, dn_bbox, attn_mask):
        m = self.m
        with torch.autocast('cuda', dtype=torch.bfloat16):
            y, x = [], img
            for l in m.model[:-1]:
                if l.f != -1:
                    x = y[l.f] if isinstance(l.f, int) else [x if j == -1 else y[j] for j in l.f]
                x = l(x, txt) if isinstance(l, TIAGELAN) else l(x)
                y.append(x if l.i in m.save else None)
            h = m.model[-1]
            xs = [blk(f.permute(0, 2, 3, 1)).permute(0, 3, 1, 2) for blk, f in zip(h.VSSBlocks, [y[j] for j in h.f])]
            feats, shapes = h._get_encoder_input(xs)
            embed, refer, eb, es = h._get_decoder_input(feats, shapes, dn_embed, dn_bbox)
            db, ds = h.decoder(embed, refer, feats, shapes, txt.clone(), h.dec_bbox_head, h.dec_score_head, h.query_pos_head, attn_mask=attn_mask)
        return db.float(), ds.float(), eb.float(), es.float()

core = Core(model)
bidx = batch['batch_idx'].long()
targets = {'cls': batch['cls'].long().view(-1), 'bboxes': batch['bboxes'], 'batch_idx': bidx, 'gt_groups': [8] * 16}
dn_embed, dn_bbox, mask, meta = get_cdn_group(targets, 10, 100, head.denoising_class_embed.weight, 100, 0.5, 1.0, True)
args = (batch['img'], batch['txt_feats'], dn_embed.detach().requires_grad_(), dn_bbox, mask)
crit = model.init_criterion()
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True)

def loss_of(outs):
    db, ds, eb, es = outs
    dn_b, dec_b = torch.split(db, meta['dn_num_split'], 2)
    dn_s, dec_s = torch.split(ds, meta['dn_num_split'], 2)
    t = crit((torch.cat([eb.unsqueeze(0), dec_b]), torch.cat([es.unsqueeze(0), dec_s])), targets, dn_bboxes=dn_b, dn_scores=dn_s, dn_meta=meta)
    return sum(t.values())

def step(fn):
    opt.zero_grad(set_to_none=False)
    loss = loss_of(fn(*args))
    loss.backward()
    opt.step()
    return loss

for _ in range(3): step(core)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): l = step(core)
torch.cuda.synchronize()
print(f'eager: {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms/step loss {float(l):.4f}', flush=True)
g = torch.cuda.make_graphed_callables(core, args, num_warmup_iters=3, allow_unused_input=True)
print('captured', flush=True)
for _ in range(2): step(g)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): l = step(g)
torch.cuda.synchronize()
print(f'graphed: {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms/step loss {float(l):.4f}', flush=True)
