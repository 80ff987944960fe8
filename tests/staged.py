"""Stage-by-stage bf16 (parity instrumentation shared by tools/bf16_attribution.py and tests/test_gpu_fullsize.py): one training-mode
forward + 12-term loss of the TAMTR graph with bf16 autocast enabled for a chosen set of stages and fp32 everywhere else, the oracle's
discrete choices (top-k picks, Hungarian pairs) injected.

    trunk      model.model[:-1]            GELAN + BTA-PAN: library convolutions, BatchNorm + SiLU kernels, gates
    vss        head.VSSBlocks              in_proj / out_proj / MLP GEMMs in bf16 (the scan itself is fp32 in both modes)
    proj       head.input_proj             1x1 projection GEMM + BatchNorm -> the token memory
    enc        head._get_decoder_input     enc_output GEMM + LayerNorm, enc_score_head, top-k gather, enc_bbox_head
    decoder    head.decoder + heads        self-attention, value_proj GEMM, deformable gather, FFN, bbox heads, contrastive head
"""
import torch

STAGES = ['trunk', 'vss', 'proj', 'enc', 'decoder']
HIP_PATH = ['vss', 'proj', 'enc', 'decoder']       # everything of the hot path behind the trunk


def staged_forward(model, state, batch, tg_host, choices, on, seed=5):
    """-> (loss float, {term: float}, dec_bboxes, dec_scores, enc_bboxes, enc_scores) as fp32 CPU tensors; `on`: set of STAGES in bf16."""
    from tamtr_amd import ops
    from tamtr_amd.loss import get_cdn_group
    from tamtr_amd.modules import TIAGELAN
    head = model.model[-1]
    if not hasattr(model, 'criterion'):
        model.criterion = model.init_criterion()

    def ac(s):
        return torch.autocast('cuda', dtype=torch.bfloat16, enabled=s in on, cache_enabled=False)

    def to(t, s):
        return t.bfloat16() if s in on else t.float()
    model.load_state_dict(state)
    model.train()
    head.fixed_topk, model.criterion.fixed_matches = choices['top'], choices['matches']
    try:
        tg = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in tg_host.items()}
        tg['host'] = {k: tg_host[k] for k in ('cls', 'bboxes', 'batch_idx')}
        img, txt = batch['img'].cuda(), batch['txt_feats'].cuda().float()
        if model.channels_last:
            img = img.contiguous(memory_format=torch.channels_last)
        torch.manual_seed(seed)
        with torch.no_grad():
            x, y = img, []
            ops.begin_bn_counter_batch()
            with ac('trunk'):
                for m in model.model[:-1]:
                    if m.f != -1:
                        x = y[m.f] if isinstance(m.f, int) else [x if j == -1 else y[j] for j in m.f]
                    x = m(x, txt) if isinstance(m, TIAGELAN) else m(x)
                    y.append(x if m.i in model.save else None)
            ops.end_bn_counter_batch()
            outs = []
            for i, (blk, j) in enumerate(zip(head.VSSBlocks, head.f)):
                with ac('vss'):
                    tok = blk(to(y[j], 'vss').permute(0, 2, 3, 1))
                with ac('proj'):
                    outs.append(head._project_level(i, to(tok, 'proj')))
            feats, shapes = torch.cat([o[0] for o in outs], 1), [o[1] for o in outs]
            dn_embed, dn_bbox, attn_mask, dn_meta = get_cdn_group(tg, head.nc, head.num_queries, head.denoising_class_embed.weight, head.num_denoising,
                                                                  head.label_noise_ratio, head.box_noise_scale, True)
            with ac('enc'):
                embed, refer, enc_b, enc_s, _ = head._get_decoder_input(to(feats, 'enc'), shapes, dn_embed, dn_bbox)
            with ac('decoder'):
                dec_b, dec_s = head.decoder(to(embed, 'decoder'), refer, to(feats, 'decoder'), shapes, txt.clone(), head.dec_bbox_head,
                                            head.dec_score_head, head.query_pos_head, attn_mask=attn_mask)
            dn_b, db = torch.split(dec_b, dn_meta['dn_num_split'], dim=2)
            dn_s, ds = torch.split(dec_s, dn_meta['dn_num_split'], dim=2)
            allb = torch.cat([enc_b.unsqueeze(0).to(db.dtype), db])
            alls = torch.cat([enc_s.unsqueeze(0).to(ds.dtype), ds])
            terms = model.criterion((allb, alls), tg, dn_bboxes=dn_b, dn_scores=dn_s, dn_meta=dn_meta)
            loss = float(torch.stack(list(terms.values())).sum())
    finally:
        head.fixed_topk = model.criterion.fixed_matches = None
    return (loss, {k: float(v) for k, v in terms.items()}) + tuple(t.float().cpu() for t in (dec_b, dec_s, enc_b, enc_s))


def errors(run, ref):
    """run / ref: (loss, terms, dec_bboxes, dec_scores, enc_bboxes, enc_scores); ref = the fp32 oracle's."""
    loss, terms, db, ds, eb, es = run
    rloss, rterms, rdb, rds, reb, res = ref
    tr = {k: abs(v - float(rterms[k])) / max(abs(float(rterms[k])), 1e-6) for k, v in terms.items()}
    worst = max(tr, key=tr.get)
    e_box, e_cls = (db - rdb).abs(), (ds - rds).abs()
    return {'loss': loss, 'loss_rel': abs(loss - float(rloss)) / abs(float(rloss)), 'term_rel_max': tr[worst], 'worst_term': worst,
            'box_abs_max': float(e_box.max()), 'box_abs_mean': float(e_box.mean()), 'cls_logit_abs_max': float(e_cls.max()),
            'cls_logit_abs_mean': float(e_cls.mean()), 'enc_box_abs_max': float((eb - reb).abs().max()),
            'enc_score_abs_max': float((es - res).abs().max()), 'enc_score_abs_mean': float((es - res).abs().mean())}
