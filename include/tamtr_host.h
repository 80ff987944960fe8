/* tamtr_host.h - C ABI of libtamtr_host.so: the 8-bit image kernels of the data path, host side.
 *
 * The reference calls OpenCV for these from its dataset / augmentation code; each entry point names the call it stands in for.
 * Plain pointers and sizes, row-major interleaved uint8 images [h, w, c]; return 0 = ok, -1 = bad argument.  Thread-safe (no
 * globals beyond one table built on first use under a once-flag); the calls release nothing and allocate at most O(width) scratch.
 * Restated from OpenCV's published 8-bit algorithms; cv2 is not in the image, so parity with it is unpinned - the numpy twin
 * under oracle/imgproc_np.py (same arithmetic, test infrastructure) is the bit-exact checker. */
#ifndef TAMTR_HOST_H
#define TAMTR_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int tamtr_host_abi_version(void);

/* cv2.resize(im, (dw, dh), interpolation=cv2.INTER_LINEAR) - ultralytics/data/base.py:156-164 (stretch load for RT-DETR).
 * Pixel-centre mapping, 11-bit fixed-point taps, horizontal then vertical pass; an exact 2x decimation is the rounded 2x2 mean. */
int tamtr_resize_linear_u8(const uint8_t* src, int sh, int sw, int c, uint8_t* dst, int dh, int dw);

/* cv2.warpAffine(img, M[:2], dsize=(dw, dh), borderValue=(b, b, b)) - ultralytics/data/augment.py:415-420 (RandomPerspective).
 * M[6] = row-major 2x3 source->destination map; inverse evaluated in 10-bit fixed point, positions quantised to 1/32 pixel,
 * four taps blended with 15-bit weights, constant border. */
int tamtr_warp_affine_u8(const uint8_t* src, int sh, int sw, int c, const double* M, uint8_t* dst, int dh, int dw, int border);

/* cv2.cvtColor(BGR2HSV) -> cv2.LUT per plane -> cv2.cvtColor(HSV2BGR) - ultralytics/data/augment.py:590-609 (RandomHSV).
 * In place on n RGB pixels; lut_* are the three 256-entry tables (hue table values < 180). */
int tamtr_hsv_lut_u8(uint8_t* rgb, long long n, const uint8_t* lut_h, const uint8_t* lut_s, const uint8_t* lut_v);

#ifdef __cplusplus
}
#endif
#endif
