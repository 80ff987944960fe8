"""GPU box: where does tamtr_img_augment_u8 differ from the host kernels?  Warp alone (HSV off), then with HSV; prints the pixels."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import imgproc_np as NP
from tamtr_amd import data as D, ops
g = np.random.default_rng(12)
B, SH, SW, H, W = 8, 70, 90, 48, 64
src = g.integers(0, 256, (B, SH, SW, 3), dtype=np.uint8)
mats = [np.array([[1, 0, 0], [0, 1, 0]], np.float32), np.array([[1, 0, -13], [0, 1, -11]], np.float32),
        np.array([[0.61, 0.07, 4.3], [-0.05, 0.66, 2.9]], np.float32), np.array([[1.9, 0, -50.2], [0, 1.9, -30.7]], np.float32),
        np.array([[0.8, 0.6, 10], [-0.6, 0.8, 30]], np.float32), np.array([[0.11, 0, 20], [0, 0.11, 20]], np.float32),
        np.array([[1, 0, 0.5], [0, 1, 0.25]], np.float32), np.array([[-1, 0, 63], [0, -1, 47]], np.float32)]
gains = [np.array([1 + 0.015 * g.uniform(-1, 1), 1 + 0.7 * g.uniform(-1, 1), 1 + 0.4 * g.uniform(-1, 1)]) for _ in range(B)]
luts = np.stack([D.hsv_luts(gn) for gn in gains])
inv = np.stack([D.invert_affine(m) for m in mats])
cu = lambda a: torch.from_numpy(a).cuda()
for hsv_on in (False, True):
    flags = np.full(B, 0 if hsv_on else 4, np.int32)
    got = (ops.img_augment(cu(src), cu(inv), cu(luts), cu(flags), (H, W)) * 255).round().to(torch.uint8).cpu().numpy().transpose(0, 2, 3, 1)
    for b in range(B):
        warped = D.warp_affine_u8(src[b], mats[b], W, H, 114)
        want = NP.hsv_lut_u8(warped, *luts[b]) if hsv_on else warped
        bad = np.argwhere((got[b] != want).any(-1))
        print(f'hsv={hsv_on} image {b}: {len(bad)} pixels differ')
        for (y, x) in bad[:6]:
            h, s, v = NP.rgb_to_hsv_u8(warped[y:y + 1, x:x + 1])
            print(f'   (y={y}, x={x}) warped rgb {warped[y, x].tolist()} hsv {int(h[0,0]), int(s[0,0]), int(v[0,0])} -> lut {int(luts[b][0][h[0,0]]), int(luts[b][1][s[0,0]]), int(luts[b][2][v[0,0]])}'
                  f'  want {want[y, x].tolist()} got {got[b][y, x].tolist()}')
