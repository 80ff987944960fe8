// ASan/UBSan run of libtamtr_host's kernels over awkward shapes (CPU only):
//   gcc -g -O1 -fsanitize=address,undefined -fno-sanitize-recover -ffp-contract=off tools/host_sanitize.c tam-tr_amd/csrc/host/imgproc.c -lm -lpthread -o build/host_sanitize && build/host_sanitize
#include "../include/tamtr_host.h"
#include <stdio.h>
#include <stdlib.h>

int main(void) {
  unsigned seed = 1;
  const int shapes[][4] = {{1, 1, 5, 5}, {2, 3, 1, 1}, {37, 53, 64, 64}, {64, 64, 32, 32}, {9, 200, 9, 50}, {50, 3, 20, 7}, {765, 1360, 640, 640}};
  for (unsigned i = 0; i < sizeof(shapes) / sizeof(shapes[0]); ++i)
    for (int c = 1; c <= 4; ++c) {
      const int sh = shapes[i][0], sw = shapes[i][1], dh = shapes[i][2], dw = shapes[i][3];
      uint8_t* s = malloc((size_t)sh * sw * c);
      uint8_t* d = malloc((size_t)dh * dw * c);
      for (size_t k = 0; k < (size_t)sh * sw * c; ++k) s[k] = (uint8_t)(seed = seed * 1664525u + 1013904223u) >> 3;
      if (tamtr_resize_linear_u8(s, sh, sw, c, d, dh, dw)) return 1;
      const double Ms[][6] = {{1, 0, 0, 0, 1, 0}, {0.3, 0.2, -500, -0.2, 0.3, 900}, {1e-9, 0, 0, 0, 1e-9, 0}, {0, 0, 1, 0, 0, 1}, {3, 0, -2.5 * sw, 0, 3, 1.5 * sh}};
      for (unsigned m = 0; m < 5; ++m)
        if (tamtr_warp_affine_u8(s, sh, sw, c, Ms[m], d, dh, dw, 114)) return 2;
      free(s); free(d);
    }
  uint8_t lut_h[256], lut_s[256], lut_v[256];
  for (int k = 0; k < 256; ++k) { lut_h[k] = (uint8_t)((k * 7) % 180); lut_s[k] = (uint8_t)(255 - k); lut_v[k] = (uint8_t)k; }
  uint8_t* all = malloc(3u << 24);
  for (unsigned k = 0; k < (1u << 24); ++k) { all[3 * k] = (uint8_t)k; all[3 * k + 1] = (uint8_t)(k >> 8); all[3 * k + 2] = (uint8_t)(k >> 16); }
  if (tamtr_hsv_lut_u8(all, 1 << 24, lut_h, lut_s, lut_v)) return 3;
  free(all);
  puts("host kernels: sanitizers clean");
  return 0;
}
