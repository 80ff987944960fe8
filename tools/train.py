#!/usr/bin/env python3
"""Train TAM-TR on a YOLO-format dataset (the reference's trainTAMTR.py flow on tam-tr_amd/engine.py + data.py).

    python tools/train.py --data dataset.yaml --text-feats clip_vitb32.npz --epochs 300 --batch 6 --save-dir runs/train/TAMTR
    python tools/train.py --gpus 8 ...                         # one rank per GPU: the script starts its own torch.distributed.run child
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/train.py --gpus 8 ...   # or under a launcher

dataset.yaml: `path`, `train`, `val` (image directories or list files) and `names` (index -> class name, `a/b` = synonyms), as
dataset/visdrone.yaml in the reference.  --text-feats: .npz {texts [n], feats [n, 512]} or a torch-saved {text: vector} with one
CLIP ViT-B/32 text embedding per prompt, every synonym and the padding prompt "" included (the text encoder itself is not part of
this repo).  --synthetic N writes N random images + labels into a temporary directory instead, to exercise the loop end to end.
"""
import argparse
import json
import os
import sys
import tempfile


import numpy as np
import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VISDRONE = ['pedestrian', 'people', 'bicycle', 'car', 'van', 'truck', 'tricycle', 'awning-tricycle', 'bus', 'motor']


def synthetic_dataset(root, n, size=(540, 960), seed=0):
    from PIL import Image
    g = np.random.default_rng(seed)
    os.makedirs(f'{root}/images'), os.makedirs(f'{root}/labels')
    for i in range(n):
        Image.fromarray(g.integers(0, 255, (*size, 3), dtype=np.uint8)).save(f'{root}/images/{i:05d}.jpg', quality=90)
        with open(f'{root}/labels/{i:05d}.txt', 'w') as f:
            for _ in range(int(g.integers(1, 9))):
                f.write(f'{int(g.integers(0, 10))} {g.uniform(0.2, 0.8):.6f} {g.uniform(0.2, 0.8):.6f} {g.uniform(0.02, 0.22):.6f} {g.uniform(0.02, 0.22):.6f}\n')
    return {'train': f'{root}/images', 'val': f'{root}/images', 'names': dict(enumerate(VISDRONE))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--data')
    ap.add_argument('--gpus', type=int, default=None, help="GPUs of this node (the reference's device='0,1,..'): N > 1 without a launcher "
                                                           'in the environment starts N ranks (trainer.py:161-189); default: WORLD_SIZE or 1')
    ap.add_argument('--text-feats')
    ap.add_argument('--synthetic', type=int, default=0)
    ap.add_argument('--epochs', type=int, default=300)
    ap.add_argument('--batch', type=int, default=6, help='images per GPU')
    ap.add_argument('--imgsz', type=int, default=640)
    ap.add_argument('--workers', type=int, default=8)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--lr0', type=float, default=1e-4)
    ap.add_argument('--close-mosaic', type=int, default=0)
    ap.add_argument('--mosaic', type=float, default=0.0)
    ap.add_argument('--device-augment', action='store_true', help='affine warp / HSV / flips / float conversion on the GPU (needs --mosaic 0)')
    ap.add_argument('--max-steps', type=int)
    ap.add_argument('--weights', help='last.pt / best.pt of an earlier run: weights, EMA, optimizer state and epoch are restored')
    ap.add_argument('--save-dir', default='runs/train/TAMTR')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--grad-dtype', default='fp32', choices=['fp32', 'bf16'], help='dtype of the gradient buckets on the wire (N > 1)')
    ap.add_argument('--static-part', default='graph', choices=['graph', 'eager'], help='replay trunk + VSS + input projection as HIP graphs (model.capture_static_part)')
    ap.add_argument('--drop-tail', action='store_true', help="leave out each epoch's incomplete last batch (NOT the reference's behaviour - its "
                                                             'build_dataloader has no drop_last and trains on the tail; the tail is a shape outside the tuned '
                                                             'tables and the recorded graphs: it runs kernel by kernel and its convolutions are searched once per machine)')
    ap.add_argument('--conv-tuning', default='shipped', choices=['shipped', 'search', 'off'], help='MIOpen solver choice (tam-tr_amd/tuning.py)')
    args = ap.parse_args()

    import tamtr_amd  # noqa: F401  (raises if the HIP library is missing)
    from tamtr_amd import data as D, dist as tdist, engine as E
    from tamtr_amd.model import RTDETRDetectionWorldModel
    plan = tdist.launch_plan(args.gpus or 1, os.environ, sys.argv[1:], __file__)
    if plan is not None:   # become the launcher of the ranks (a child process; nothing here has touched the GPU)
        raise SystemExit(tdist.self_launch(plan))
    rank, local, world = tdist.init_from_env()
    if args.gpus is not None and args.gpus != world:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    try:       # host side = kernel launches + a few tiny CPU ops: a thread pool sized for the whole host only adds wake-up latency
        torch.set_num_threads(min(8, len(os.sched_getaffinity(0))))
    except AttributeError:
        torch.set_num_threads(8)
    local = local % max(torch.cuda.device_count(), 1)      # (rehearsals with more ranks than GPUs share a device)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    from tamtr_amd.tuning import use_tuned_convolutions_ranked
    # before the first convolution (tables cover 640 px / 16 images; other shapes are searched once); rank 0 seeds the shared directory first
    conv_tuning = use_tuned_convolutions_ranked(args.conv_tuning, log=(print if rank == 0 else None), rank=rank, world=world)
    if rank == 0:
        print(f'convolution tuning: {conv_tuning}', flush=True)
    tmp = None
    if args.synthetic:
        tmp = tempfile.TemporaryDirectory()
        spec = synthetic_dataset(tmp.name, args.synthetic)
    else:
        with open(args.data) as f:
            spec = yaml.safe_load(f)
        root = spec.get('path', os.path.dirname(os.path.abspath(args.data)))
        spec = {**spec, **{k: os.path.normpath(os.path.join(root, spec[k])) for k in ('train', 'val') if k in spec}}
    names = spec['names'] if isinstance(spec['names'], dict) else dict(enumerate(spec['names']))
    prompts = sorted({p for v in names.values() for p in v.split('/')} | {''})
    tf = D.TextFeatures.load(args.text_feats) if args.text_feats else D.TextFeatures.synthetic(prompts, 512, seed=0)
    if not args.text_feats and rank == 0:
        print('no --text-feats: random prompt embeddings (loop exercise only)', file=sys.stderr)

    train = D.PromptDetDataset(spec['train'], names, args.imgsz, augment=True, hyp={'mosaic': args.mosaic}, batch_size=args.batch,
                               device_augment=args.device_augment)
    tl = D.build_dataloader(train, args.batch, args.workers, shuffle=True, rank=rank if world > 1 else -1, drop_last=args.drop_tail)
    vl = None
    if rank == 0 and 'val' in spec:
        vl = D.build_dataloader(D.PromptDetDataset(spec['val'], names, args.imgsz, augment=False), args.batch * 2, args.workers, shuffle=False)

    torch.manual_seed(0)
    model = RTDETRDetectionWorldModel(nc=len(names)).to(dev).train()
    resume = None
    if args.weights:   # a checkpoint written by engine.fit: {'epoch', 'model', 'ema', 'updates', 'optimizer', 'metrics'} (state_dicts only;
        resume = torch.load(args.weights, map_location=dev)   # a reference .pt pickles modules: export its state_dict first)
        model.load_state_dict(resume['model'])
    torch.manual_seed(args.seed + 1 + rank)    # per-rank streams for DropPath / denoising noise (reference: seed + 1 + RANK)
    model.autocast_dtype = torch.bfloat16 if args.dtype == 'bf16' else None
    model.names = names
    model.set_text_features(tf.encode([v.split('/')[0] for v in names.values()])[None].to(dev))     # validation vocabulary
    reducer = tdist.GradReducer(model.named_parameters(), skip=lambda n: '.attn.' in n, late=lambda n: 'denoising_class_embed' in n,
                                grad_dtype=torch.bfloat16 if args.grad_dtype == 'bf16' else None) if world > 1 else None

    def prepare(batch, training):
        return D.preprocess_batch(batch, tf if training else None, dev)

    def log(rec):
        if not isinstance(rec, dict):
            return print(rec, flush=True)
        print(json.dumps({k: (round(v, 5) if isinstance(v, float) else v) for k, v in rec.items()}), flush=True)
    E.fit(model, tl, prepare, args.epochs, val_loader=vl, lr0=args.lr0, close_mosaic=args.close_mosaic, imgsz=args.imgsz, reducer=reducer,
          rank=rank, world=world, save_dir=args.save_dir if rank == 0 else None, max_steps=args.max_steps, log=log, resume=resume,
          static_graph=args.static_part == 'graph')
    if tmp is not None:
        tmp.cleanup()


if __name__ == '__main__':
    main()
