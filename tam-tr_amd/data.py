"""Data path of the train / validate harness: YOLO-format detection data with class-name prompts.

What the reference does on this path and where (all host-side numpy / PIL / torch there too):

  label files, image list      ultralytics/data/utils.py:32-35,84-152, data/base.py:98-124
  stretch-resize load, buffer  ultralytics/data/base.py:146-178, models/rtdetrworld/val.py:28-31
  train transforms             ultralytics/data/augment.py:1018-1048 with stretch=True (models/rtdetrworld/val.py:33-41):
                               Mosaic 155-306, RandomPerspective 329-566, MixUp 308-326, RandomHSV 569-609, RandomFlip 612-666,
                               RandomLoadText 942-1016, Format 858-926
  batching                     ultralytics/data/dataset.py:191-206 (collate), data/build.py:71-115 (loader, worker seeding)
  prompts -> text features     models/rtdetrworld/train.py:134-159, nn/tasks.py:552-571

Built MI355X-side as plain host code feeding the sync-free step: samples stay numpy until `Format`, the batch keeps its
labels on the host (model.loss uploads them through the pinned ring) and only `img` / `txt_feats` go to the device.
With `device_augment=True` the workers only decode, stretch and draw the random parameters; the pixel work of the affine warp,
HSV jitter, flips and the uint8 -> float conversion runs on the GPU after the upload (tamtr_img_augment_u8, one kernel per batch,
bit-identical to the host kernels), which halves the per-image host time.

Pinned against the reference's own functions (tests/golden/data.npz): label parsing, box conversions, the affine box path and
candidate filter, flips, RandomLoadText's draw sequence, Format, collate.  NOT pinned - the reference calls OpenCV for these and
cv2 is absent from the image: `resize_linear_u8`, `warp_affine_u8`, `hsv_jitter_u8` run in libtamtr_host.so (C, include/tamtr_host.h;
per-image cost is what bounds the loader) and restate OpenCV's published 8-bit algorithms (fixed-point bilinear
resize, 1/32-pixel affine remap, integer RGB->HSV / float HSV->RGB); tests hold them bit-exact to a numpy twin
(oracle/imgproc_np.py), to closed forms, and within rounding of torch's float sampling and colorsys.  The CLIP text encoder is out of scope: prompts are looked up in a table of precomputed embeddings.
Albumentations (absent from the reference's requirements -> a no-op there) and CopyPaste (needs segments) are not built.
"""
import glob
import math
import os
import random
from dataclasses import dataclass, field
from itertools import chain

import numpy as np
import torch
from PIL import Image, ImageOps

from . import _hostlib

IMG_FORMATS = ('bmp', 'dng', 'jpeg', 'jpg', 'mpo', 'png', 'tif', 'tiff', 'webp', 'pfm')


# ------------------------------------------------------------------------------------------------ files and labels
def img2label_paths(img_paths):
    """.../images/x.jpg -> .../labels/x.txt (last `/images/` component only)."""
    sa, sb = f'{os.sep}images{os.sep}', f'{os.sep}labels{os.sep}'
    return [sb.join(p.rsplit(sa, 1)).rsplit('.', 1)[0] + '.txt' for p in img_paths]


def list_images(img_path):
    """A directory (recursive), a list file, or a list of either -> sorted image paths."""
    found = []
    for p in img_path if isinstance(img_path, (list, tuple)) else [img_path]:
        p = str(p)
        if os.path.isdir(p):
            found += glob.glob(os.path.join(p, '**', '*.*'), recursive=True)
        elif os.path.isfile(p):
            parent = os.path.dirname(p) + os.sep
            with open(p) as f:
                found += [ln.replace('./', parent) if ln.startswith('./') else ln for ln in f.read().strip().splitlines()]
        else:
            raise FileNotFoundError(f'{p} does not exist')
    files = sorted(x for x in found if x.rsplit('.', 1)[-1].lower() in IMG_FORMATS)
    if not files:
        raise FileNotFoundError(f'no images found in {img_path}')
    return files


def image_shape(path):
    """(h, w) as the decoder will return it (EXIF orientations 6 / 8 swap the axes of a JPEG)."""
    with Image.open(path) as im:
        w, h = im.size
        if im.format == 'JPEG':
            try:
                if im.getexif().get(274) in (6, 8):
                    w, h = h, w
            except Exception:
                pass
        fmt = (im.format or '').lower()
    if h <= 9 or w <= 9:
        raise ValueError(f'image size {(h, w)} <10 pixels')
    if fmt not in IMG_FORMATS:
        raise ValueError(f'invalid image format {fmt}')
    return h, w


def parse_labels(text, num_cls):
    """Rows `cls cx cy w h` (normalised) -> float32 [n, 5]; the reference's acceptance rules, duplicates dropped in sorted order."""
    rows = [ln.split() for ln in text.strip().splitlines() if len(ln)]
    if not rows:
        return np.zeros((0, 5), np.float32)
    if any(len(r) != 5 for r in rows):
        raise ValueError(f'labels require 5 columns, {max(len(r) for r in rows)} columns detected')
    lb = np.array(rows, dtype=np.float32)
    if lb[:, 1:].max() > 1:
        raise ValueError('non-normalized or out of bounds coordinates')
    if lb.min() < 0:
        raise ValueError('negative label values')
    if lb[:, 0].max() > num_cls:
        raise ValueError(f'label class {int(lb[:, 0].max())} exceeds dataset class count {num_cls}')
    _, keep = np.unique(lb, axis=0, return_index=True)
    return lb[keep] if len(keep) < len(lb) else lb


def scan_labels(im_files, num_cls, log=None):
    """[{im_file, shape, cls [n,1], bboxes [n,4] xywh normalised}] for every readable image; unreadable pairs are skipped."""
    out = []
    for im_file, lb_file in zip(im_files, img2label_paths(im_files)):
        try:
            shape = image_shape(im_file)
            if os.path.isfile(lb_file):
                with open(lb_file) as f:
                    lb = parse_labels(f.read(), num_cls)
            else:
                lb = np.zeros((0, 5), np.float32)
        except Exception as e:  # corrupt image / label: the reference drops the pair and carries on
            if log:
                log(f'{im_file}: ignoring corrupt image/label: {e}')
            continue
        out.append({'im_file': im_file, 'shape': shape, 'cls': lb[:, 0:1], 'bboxes': lb[:, 1:]})
    return out


def decode_image(path):
    """RGB uint8 [h, w, 3], EXIF orientation applied as cv2.imread does."""
    with Image.open(path) as im:
        im = ImageOps.exif_transpose(im)
        return np.asarray(im.convert('RGB'))


# ------------------------------------------------------------------------------------------------ 8-bit image kernels (host, C)
def _u8_image(a):
    a = np.ascontiguousarray(a)
    if a.dtype != np.uint8 or a.ndim != 3:
        raise TypeError(f'expected a uint8 [h, w, c] image, got {a.dtype} {a.shape}')
    return a


def resize_linear_u8(src, dw, dh):
    """cv2.resize(..., INTER_LINEAR) for 8-bit images (tamtr_resize_linear_u8).  src [h, w, c] -> [dh, dw, c]."""
    src = _u8_image(src)
    out = np.empty((dh, dw, src.shape[2]), np.uint8)
    _hostlib.check(_hostlib.lib().tamtr_resize_linear_u8(src.ctypes.data, src.shape[0], src.shape[1], src.shape[2], out.ctypes.data, dh, dw),
                   'tamtr_resize_linear_u8')
    return out


def warp_affine_u8(src, M, dw, dh, border=114):
    """cv2.warpAffine(src, M[:2], (dw, dh), borderValue=border) for 8-bit images (tamtr_warp_affine_u8)."""
    src = _u8_image(src)
    m = np.ascontiguousarray(np.asarray(M, np.float64)[:2])
    out = np.empty((dh, dw, src.shape[2]), np.uint8)
    _hostlib.check(_hostlib.lib().tamtr_warp_affine_u8(src.ctypes.data, src.shape[0], src.shape[1], src.shape[2], m.ctypes.data, out.ctypes.data,
                                                      dh, dw, int(border)), 'tamtr_warp_affine_u8')
    return out


def invert_affine(M):
    """The six doubles both warps evaluate: destination -> source map (m0 m1 b1 / m3 m4 b2) of the 2x3 source -> destination M."""
    M = np.asarray(M, np.float64)
    det = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    det = 1.0 / det if det != 0 else 0.0
    m0, m1, m3, m4 = M[1, 1] * det, -M[0, 1] * det, -M[1, 0] * det, M[0, 0] * det
    return np.array([m0, m1, -m0 * M[0, 2] - m1 * M[1, 2], m3, m4, -m3 * M[0, 2] - m4 * M[1, 2]], np.float64)


def hsv_luts(gains):
    """[3, 256] uint8 hue / saturation / value tables of RandomHSV for one draw of gains."""
    x = np.arange(0, 256, dtype=np.float64)
    return np.stack([((x * gains[0]) % 180).astype(np.uint8), np.clip(x * gains[1], 0, 255).astype(np.uint8),
                     np.clip(x * gains[2], 0, 255).astype(np.uint8)])


def hsv_jitter_u8(img, gains):
    """Per-channel gain look-up tables on the HSV planes (hue wraps at 180, saturation / value clip at 255), RGB in, RGB out."""
    luts = hsv_luts(gains)
    out = _u8_image(img).copy()
    if out.shape[2] != 3:
        raise TypeError('hsv_jitter_u8 needs 3 channels')
    _hostlib.check(_hostlib.lib().tamtr_hsv_lut_u8(out.ctypes.data, out.shape[0] * out.shape[1], *(t.ctypes.data for t in luts)), 'tamtr_hsv_lut_u8')
    return out


# ------------------------------------------------------------------------------------------------ samples and boxes
@dataclass
class Sample:
    """One image with its boxes on the way through the transforms.  boxes float32 [n, 4] in `fmt`, pixels unless `normalized`."""
    img: np.ndarray
    cls: np.ndarray
    boxes: np.ndarray
    fmt: str = 'xywh'
    normalized: bool = True
    texts: list = field(default_factory=list)
    im_file: str = ''
    ori_shape: tuple = (0, 0)
    resized_shape: tuple = (0, 0)
    ratio_pad: tuple = None
    mosaic_border: tuple = None
    defer: dict = None        # device_augment: the pixel work recorded instead of done (inverse affine, HSV tables, flips, output size)

    def to_xyxy(self):
        if self.fmt == 'xywh':
            b, o = self.boxes, np.empty_like(self.boxes)
            dw, dh = b[..., 2] / 2, b[..., 3] / 2
            o[..., 0], o[..., 1], o[..., 2], o[..., 3] = b[..., 0] - dw, b[..., 1] - dh, b[..., 0] + dw, b[..., 1] + dh
            self.boxes, self.fmt = o, 'xyxy'
        return self

    def to_xywh(self):
        if self.fmt == 'xyxy':
            b, o = self.boxes, np.empty_like(self.boxes)
            o[..., 0], o[..., 1] = (b[..., 0] + b[..., 2]) / 2, (b[..., 1] + b[..., 3]) / 2
            o[..., 2], o[..., 3] = b[..., 2] - b[..., 0], b[..., 3] - b[..., 1]
            self.boxes, self.fmt = o, 'xywh'
        return self

    def _mul(self, sx, sy):
        for k, f in enumerate((sx, sy, sx, sy)):
            self.boxes[:, k] *= f

    def denormalize(self, w, h):
        if self.normalized:
            self._mul(w, h)
            self.normalized = False
        return self

    def normalize(self, w, h):
        if not self.normalized:
            self._mul(1 / w, 1 / h)
            self.normalized = True
        return self

    def shift(self, dx, dy):
        for k, o in enumerate((dx, dy, dx, dy)):
            self.boxes[:, k] += o

    def clip(self, w, h):
        fmt = self.fmt
        self.to_xyxy()
        self.boxes[:, [0, 2]] = self.boxes[:, [0, 2]].clip(0, w)
        self.boxes[:, [1, 3]] = self.boxes[:, [1, 3]].clip(0, h)
        if fmt == 'xywh':
            self.to_xywh()

    def select(self, keep):
        self.boxes, self.cls = self.boxes[keep], self.cls[keep]


def affine_boxes(boxes, M):
    """Axis-aligned hull of the four corners of each xyxy box under the 3x3 map M (affine: last row ignored)."""
    n = len(boxes)
    if n == 0:
        return boxes
    pts = np.ones((n * 4, 3), dtype=boxes.dtype)
    pts[:, :2] = boxes[:, [0, 1, 2, 3, 0, 3, 2, 1]].reshape(n * 4, 2)
    pts = (pts @ M.T)[:, :2].reshape(n, 8)
    xs, ys = pts[:, 0::2], pts[:, 1::2]
    return np.concatenate((xs.min(1), ys.min(1), xs.max(1), ys.max(1)), dtype=boxes.dtype).reshape(4, n).T


def box_candidates(before, after, wh_thr=2, ar_thr=100, area_thr=0.1, eps=1e-16):
    """Boxes [4, n] that survive an augmentation: still >2 px on a side, aspect <100, kept >10 % of their (scaled) area."""
    w1, h1 = before[2] - before[0], before[3] - before[1]
    w2, h2 = after[2] - after[0], after[3] - after[1]
    ar = np.maximum(w2 / (h2 + eps), h2 / (w2 + eps))
    return (w2 > wh_thr) & (h2 > wh_thr) & (w2 * h2 / (w1 * h1 + eps) > area_thr) & (ar < ar_thr)


# ------------------------------------------------------------------------------------------------ transforms
class RandomAffine:
    """Rotation / scale / shear / translation about the image centre (the reference's RandomPerspective with perspective = 0,
    its default).  Draw order: two perspective draws (consumed to keep the stream aligned), angle, scale, 2 shears, 2 shifts."""

    def __init__(self, degrees=0.0, translate=0.1, scale=0.5, shear=0.0, perspective=0.0, border=(0, 0)):
        if perspective:
            raise NotImplementedError('perspective warps are not built (reference default: 0.0)')
        self.degrees, self.translate, self.scale, self.shear, self.border = degrees, translate, scale, shear, border

    def matrix(self, w, h, size):
        C = np.eye(3, dtype=np.float32)
        C[0, 2], C[1, 2] = -w / 2, -h / 2
        random.uniform(0, 0), random.uniform(0, 0)
        a = random.uniform(-self.degrees, self.degrees)
        s = random.uniform(1 - self.scale, 1 + self.scale)
        R = np.eye(3, dtype=np.float32)
        ca, sa = s * math.cos(math.radians(a)), s * math.sin(math.radians(a))
        R[:2] = [[ca, sa, 0.0], [-sa, ca, 0.0]]
        S = np.eye(3, dtype=np.float32)
        S[0, 1] = math.tan(random.uniform(-self.shear, self.shear) * math.pi / 180)
        S[1, 0] = math.tan(random.uniform(-self.shear, self.shear) * math.pi / 180)
        T = np.eye(3, dtype=np.float32)
        T[0, 2] = random.uniform(0.5 - self.translate, 0.5 + self.translate) * size[0]
        T[1, 2] = random.uniform(0.5 - self.translate, 0.5 + self.translate) * size[1]
        return T @ S @ R @ C, s

    def __call__(self, smp):
        smp.ratio_pad = None
        h, w = smp.img.shape[:2]
        smp.to_xyxy().denormalize(w, h)
        border = smp.mosaic_border if smp.mosaic_border is not None else self.border
        smp.mosaic_border = None
        size = (w + border[1] * 2, h + border[0] * 2)
        M, s = self.matrix(w, h, size)
        if smp.defer is not None:
            smp.defer.update(inv=invert_affine(M[:2]), size=(size[1], size[0]))
        elif border[0] != 0 or border[1] != 0 or (M != np.eye(3)).any():
            smp.img = warp_affine_u8(smp.img, M[:2], size[0], size[1], 114)
        old = smp.boxes
        smp.boxes = affine_boxes(old, M)
        smp.clip(*size)
        for k in range(4):
            old[:, k] *= s
        smp.select(box_candidates(old.T, smp.boxes.T, area_thr=0.10))
        smp.resized_shape = (size[1], size[0]) if smp.defer is not None else smp.img.shape[:2]
        return smp


class Mosaic4:
    """Four stretched images around a random centre on a 2s x 2s canvas (grey 114); RandomAffine crops it back to s x s."""

    def __init__(self, dataset, imgsz=640, p=1.0):
        self.dataset, self.imgsz, self.p, self.border = dataset, imgsz, p, (-imgsz // 2, -imgsz // 2)

    def __call__(self, smp):
        if random.uniform(0, 1) > self.p:
            return smp
        parts = [smp] + [self.dataset.load_sample(i) for i in random.choices(list(self.dataset.buffer), k=3)]
        s = self.imgsz
        yc, xc = (int(random.uniform(-b, 2 * s + b)) for b in self.border)
        canvas = np.full((2 * s, 2 * s, smp.img.shape[2]), 114, dtype=np.uint8)
        for i, part in enumerate(parts):
            h, w = part.resized_shape
            if i == 0:
                x1a, y1a, x2a, y2a = max(xc - w, 0), max(yc - h, 0), xc, yc
                x1b, y1b, x2b, y2b = w - (x2a - x1a), h - (y2a - y1a), w, h
            elif i == 1:
                x1a, y1a, x2a, y2a = xc, max(yc - h, 0), min(xc + w, s * 2), yc
                x1b, y1b, x2b, y2b = 0, h - (y2a - y1a), min(w, x2a - x1a), h
            elif i == 2:
                x1a, y1a, x2a, y2a = max(xc - w, 0), yc, xc, min(s * 2, yc + h)
                x1b, y1b, x2b, y2b = w - (x2a - x1a), 0, w, min(y2a - y1a, h)
            else:
                x1a, y1a, x2a, y2a = xc, yc, min(xc + w, s * 2), min(s * 2, yc + h)
                x1b, y1b, x2b, y2b = 0, 0, min(w, x2a - x1a), min(y2a - y1a, h)
            canvas[y1a:y2a, x1a:x2a] = part.img[y1b:y2b, x1b:x2b]
            part.to_xyxy().denormalize(w, h)
            part.shift(x1a - x1b, y1a - y1b)
        out = Sample(canvas, np.concatenate([p.cls for p in parts], 0), np.concatenate([p.boxes for p in parts], 0), 'xyxy', False,
                     smp.texts, smp.im_file, smp.ori_shape, (2 * s, 2 * s), None, self.border)
        out.clip(2 * s, 2 * s)
        out.select((out.boxes[:, 2] - out.boxes[:, 0]) * (out.boxes[:, 3] - out.boxes[:, 1]) > 0)
        return out


class MixUp:
    """Blend with one more (already pre-transformed) sample, Beta(32, 32) ratio; boxes of both are kept."""

    def __init__(self, dataset, pre_transform, p=0.0):
        self.dataset, self.pre, self.p = dataset, pre_transform, p

    def __call__(self, smp):
        if random.uniform(0, 1) > self.p:
            return smp
        other = self.pre(self.dataset.load_sample(random.randint(0, len(self.dataset) - 1)))
        r = np.random.beta(32.0, 32.0)
        smp.img = (smp.img * r + other.img * (1 - r)).astype(np.uint8)
        smp.boxes, smp.cls = np.concatenate([smp.boxes, other.boxes], 0), np.concatenate([smp.cls, other.cls], 0)
        return smp


class RandomHSV:
    def __init__(self, hgain=0.5, sgain=0.5, vgain=0.5):
        self.gains = [hgain, sgain, vgain]

    def __call__(self, smp):
        if any(self.gains):
            gains = np.random.uniform(-1, 1, 3) * self.gains + 1
            if smp.defer is not None:
                smp.defer['luts'] = hsv_luts(gains)
                smp.defer['flags'] &= ~4
            else:
                smp.img = hsv_jitter_u8(smp.img, gains)
        return smp


class RandomFlip:
    def __init__(self, p=0.5, direction='horizontal'):
        assert direction in ('horizontal', 'vertical') and 0 <= p <= 1.0
        self.p, self.axis = p, 1 if direction == 'horizontal' else 0

    def __call__(self, smp):
        smp.to_xywh()
        h, w = smp.defer['size'] if smp.defer is not None else smp.img.shape[:2]
        extent = 1 if smp.normalized else (w if self.axis else h)
        if random.random() < self.p:
            if smp.defer is not None:
                smp.defer['flags'] ^= 2 if self.axis else 1
            else:
                smp.img = np.flip(smp.img, self.axis)
            k = 0 if self.axis else 1
            smp.boxes[:, k] = extent - smp.boxes[:, k]
        smp.img = np.ascontiguousarray(smp.img)
        return smp


class RandomLoadText:
    """Sample the prompts of one image: every class present (at most `max_samples`), negatives up to the budget, shuffled; class
    ids are re-indexed into that order and boxes of unsampled classes dropped; one synonym per class; padded with `padding_value`."""

    def __init__(self, prompt_format='{}', neg_samples=(80, 80), max_samples=80, padding=False, padding_value=''):
        self.prompt_format, self.neg_samples, self.max_samples = prompt_format, neg_samples, max_samples
        self.padding, self.padding_value = padding, padding_value

    def __call__(self, smp):
        class_texts = smp.texts
        nc = len(class_texts)
        cls = np.asarray(smp.cls, dtype=int)
        pos = np.unique(cls).tolist()
        if len(pos) > self.max_samples:   # (the reference keeps a set here and fails on the concatenation below)
            pos = random.sample(pos, k=self.max_samples)
        n_neg = min(min(nc, self.max_samples) - len(pos), random.randint(*self.neg_samples))
        neg = random.sample([i for i in range(nc) if i not in pos], k=n_neg)
        sampled = pos + neg
        random.shuffle(sampled)
        new_id = {label: i for i, label in enumerate(sampled)}
        labels = cls.reshape(-1).tolist()
        keep = np.array([label in new_id for label in labels], dtype=bool)
        smp.boxes = smp.boxes[keep]
        smp.cls = np.array([[new_id[label]] for label in labels if label in new_id])
        texts = []
        for label in sampled:
            prompts = class_texts[label]
            texts.append(self.prompt_format.format(prompts[random.randrange(len(prompts))]))
        if self.padding:
            texts += [self.padding_value] * max(self.max_samples - len(sampled), 0)
        smp.texts = texts
        return smp


class Format:
    """Sample -> dict of tensors for `collate`: img uint8 [3, h, w] RGB, cls [n, 1], bboxes [n, 4] xywh normalised, batch_idx [n]."""

    def __call__(self, smp):
        h, w = smp.defer['size'] if smp.defer is not None else smp.img.shape[:2]
        smp.to_xywh().denormalize(w, h).normalize(w, h)
        n = len(smp.boxes)
        out = {'im_file': smp.im_file, 'ori_shape': smp.ori_shape, 'resized_shape': smp.resized_shape}
        if smp.ratio_pad is not None:
            out['ratio_pad'] = smp.ratio_pad
        out['texts'] = smp.texts
        if smp.defer is not None:      # the untransformed picture travels; ops.img_augment finishes it on the device
            d = smp.defer
            out['src'] = torch.from_numpy(np.ascontiguousarray(smp.img))
            out['aug_inv'], out['aug_luts'] = torch.from_numpy(d['inv']), torch.from_numpy(np.ascontiguousarray(d['luts']))
            out['aug_flags'] = torch.tensor(d['flags'], dtype=torch.int32)
        else:
            out['img'] = torch.from_numpy(np.ascontiguousarray(smp.img.transpose(2, 0, 1)))
        out['cls'] = torch.from_numpy(smp.cls) if n else torch.zeros(n)
        out['bboxes'] = torch.from_numpy(smp.boxes) if n else torch.zeros((n, 4))
        out['batch_idx'] = torch.zeros(n)
        return out


def collate(samples):
    """Stack images, concatenate labels with the image index written into batch_idx, keep everything else as tuples."""
    batch = {k: [s[k] for s in samples] for k in samples[0]}
    for k in ('img', 'src', 'aug_inv', 'aug_luts', 'aug_flags'):
        if k in batch:
            batch[k] = torch.stack(batch[k], 0)
    batch['batch_idx'] = torch.cat([b + i for i, b in enumerate(batch['batch_idx'])], 0)
    for k in ('bboxes', 'cls'):
        batch[k] = torch.cat(batch[k], 0)
    return {k: (tuple(v) if isinstance(v, list) else v) for k, v in batch.items()}


# ------------------------------------------------------------------------------------------------ dataset and loader
AUG_DEFAULTS = dict(mosaic=0.0, mixup=0.0, degrees=0.0, translate=0.1, scale=0.9, shear=0.0, perspective=0.0, hsv_h=0.015, hsv_s=0.7,
                    hsv_v=0.4, flipud=0.0, fliplr=0.5)      # ultralytics/cfg/default.yaml:98-110 as shipped by the reference


class PromptDetDataset(torch.utils.data.Dataset):
    """Images stretched to imgsz x imgsz with YOLO labels and the class-name prompts (`a/b` = synonyms).  augment=True runs the
    training transforms, otherwise the sample is only formatted (the reference's RTDETRDataset, models/rtdetrworld/val.py:15-58)."""

    def __init__(self, img_path, names, imgsz=640, augment=False, hyp=None, batch_size=16, log=None, device_augment=False):
        self.imgsz, self.augment = imgsz, augment
        self.device_augment = bool(device_augment and augment)
        self.names = dict(enumerate(names)) if isinstance(names, (list, tuple)) else dict(names)
        self.prompts = [v.split('/') for _, v in self.names.items()]
        self.labels = scan_labels(list_images(img_path), len(self.names), log)
        self.im_files = [lb['im_file'] for lb in self.labels]
        n = len(self.labels)
        self.cache, self.buffer = [None] * n, []
        self.max_buffer = min(n, batch_size * 8, 1000) if augment else 0
        self.hyp = dict(AUG_DEFAULTS, **(hyp or {}))
        self.transforms = self._build()

    def _build(self):
        if not self.augment:
            return [Format()]
        h = self.hyp
        if self.device_augment and (h['mosaic'] or h['mixup']):
            raise ValueError('device_augment needs one source image per sample: mosaic and mixup must be 0 (the reference\'s defaults)')
        pre = [Mosaic4(self, self.imgsz, h['mosaic']),
               RandomAffine(h['degrees'], h['translate'], h['scale'], h['shear'], h['perspective'])]

        def run_pre(smp):
            for t in pre:
                smp = t(smp)
            return smp
        return pre + [MixUp(self, run_pre, h['mixup']), RandomHSV(h['hsv_h'], h['hsv_s'], h['hsv_v']),
                      RandomFlip(h['flipud'], 'vertical'), RandomFlip(h['fliplr'], 'horizontal'),
                      RandomLoadText(max_samples=min(len(self.names), 80), padding=True), Format()]

    def close_mosaic(self):
        self.hyp.update(mosaic=0.0, mixup=0.0)
        self.transforms = self._build()

    def __len__(self):
        return len(self.labels)

    def load_image(self, i):
        """(stretched RGB image, original hw).  Training keeps the last few decoded images for the mosaic to draw from."""
        if self.cache[i] is not None:
            return self.cache[i]
        im = decode_image(self.im_files[i])
        h0, w0 = im.shape[:2]
        if not (h0 == w0 == self.imgsz):
            im = resize_linear_u8(im, self.imgsz, self.imgsz)
        if self.augment:
            self.cache[i] = (im, (h0, w0))
            self.buffer.append(i)
            if len(self.buffer) >= self.max_buffer:
                self.cache[self.buffer.pop(0)] = None
        return im, (h0, w0)

    def load_sample(self, i):
        lb = self.labels[i]
        im, hw0 = self.load_image(i)
        rs = im.shape[:2]
        smp = Sample(im, lb['cls'].copy(), lb['bboxes'].copy(), 'xywh', True, [list(p) for p in self.prompts], lb['im_file'], hw0, rs,
                     (rs[0] / hw0[0], rs[1] / hw0[1]))
        if self.device_augment:     # workers only decode, stretch and draw; tamtr_img_augment_u8 does the pixels after the upload
            smp.defer = {'flags': 4, 'luts': np.zeros((3, 256), np.uint8), 'inv': invert_affine(np.eye(3)[:2]), 'size': rs}
        return smp

    def __getitem__(self, i):
        smp = self.load_sample(i)
        for t in self.transforms:
            smp = t(smp)
        return smp


def seed_worker(worker_id):
    seed = torch.initial_seed() % 2 ** 32
    np.random.seed(seed)
    random.seed(seed)


def build_dataloader(dataset, batch, workers=4, shuffle=True, rank=-1, drop_last=False):
    """torch DataLoader with the reference's conventions: DistributedSampler when ranked, fixed generator seed + rank, workers
    re-seeded from torch's per-worker seed, pinned batches.  drop_last (not in the reference, default off): leave out an epoch's
    incomplete tail batch - a shape of its own, i.e. a MIOpen search of its own and an eager (un-replayed) step."""
    batch = min(batch, len(dataset))
    nw = min(os.cpu_count() // max(torch.cuda.device_count(), 1), batch if batch > 1 else 0, workers)
    sampler = None if rank == -1 else torch.utils.data.distributed.DistributedSampler(dataset, shuffle=shuffle)
    gen = torch.Generator()
    gen.manual_seed(6148914691236517205 + max(rank, -1))
    return torch.utils.data.DataLoader(dataset, batch_size=batch, shuffle=shuffle and sampler is None, num_workers=nw, sampler=sampler,
                                       pin_memory=torch.cuda.is_available(), collate_fn=collate, worker_init_fn=seed_worker,
                                       generator=gen, persistent_workers=nw > 0, drop_last=drop_last)


def reset_workers(loader):
    """Make the loader's next epoch start from fresh worker processes (the reference's `train_loader.reset()`,
    engine/trainer.py:319-321, data/build.py:44-50): persistent workers hold a pickled copy of the dataset, so a change made to
    `loader.dataset` in the main process (close_mosaic) would otherwise never reach them."""
    it = getattr(loader, '_iterator', None)
    if it is not None:
        if hasattr(it, '_shutdown_workers'):
            it._shutdown_workers()
        loader._iterator = None


# ------------------------------------------------------------------------------------------------ prompts -> text features
class TextFeatures:
    """Prompt -> embedding table standing in for the frozen CLIP ViT-B/32 text tower (computed offline; the encoder is out of
    scope).  `encode` returns unit-norm rows like the reference's preprocess_batch / set_classes."""

    def __init__(self, table):
        self.table = {k: torch.as_tensor(v, dtype=torch.float32) for k, v in table.items()}
        dims = {v.numel() for v in self.table.values()}
        if len(dims) != 1:
            raise ValueError(f'embeddings of different sizes: {sorted(dims)}')
        self.dim = dims.pop()

    @classmethod
    def load(cls, path):
        """.npz with arrays `texts` [n] (str) and `feats` [n, d], or a torch-saved {text: vector} dict."""
        if str(path).endswith('.npz'):
            z = np.load(path, allow_pickle=False)
            return cls({str(t): f for t, f in zip(z['texts'], z['feats'])})
        return cls(torch.load(path, map_location='cpu'))

    @classmethod
    def synthetic(cls, texts, dim=512, seed=0):
        g = torch.Generator().manual_seed(seed)
        return cls({t: torch.randn(dim, generator=g) for t in texts})

    def encode(self, texts):
        missing = sorted({t for t in texts if t not in self.table})
        if missing:
            raise KeyError(f'no embedding for prompts {missing}')
        f = torch.stack([self.table[t] for t in texts], 0)
        return f / f.norm(p=2, dim=-1, keepdim=True)


def preprocess_batch(batch, text_features, device):
    """img -> device float in [0, 1]; sampled prompts -> txt_feats [B, T, d] (training batches only: a validation batch carries the
    synonym lists and the model uses the features set in advance); labels stay on the host for the sync-free loss."""
    out = dict(batch)
    if 'src' in batch:      # device_augment: finish the transforms' pixel work on the GPU (csrc/imgaug.hip)
        from . import ops
        a = [out.pop(k).to(device, non_blocking=True) for k in ('src', 'aug_inv', 'aug_luts', 'aug_flags')]
        out['img'] = ops.img_augment(*a, out_hw=tuple(batch['resized_shape'][0]))
    else:
        out['img'] = batch['img'].to(device, non_blocking=True).float() / 255
    if text_features is None or not all(isinstance(t, str) for t in batch['texts'][0]):
        return out
    texts = list(chain(*batch['texts']))
    feats = text_features.encode(texts).to(dtype=out['img'].dtype)
    out['txt_feats'] = feats.reshape(len(batch['texts']), -1, feats.shape[-1]).to(device, non_blocking=True)
    return out
