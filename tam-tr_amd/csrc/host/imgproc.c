// imgproc.c - libtamtr_host.so: 8-bit resize / affine warp / HSV look-up for the data path (include/tamtr_host.h).
// Built with -ffp-contract=off: every float operation rounds separately, which is what the numpy twin (oracle/imgproc_np.py) does.
#include "../../../include/tamtr_host.h"
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

int tamtr_host_abi_version(void) { return 1; }

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline uint8_t sat_u8(long long v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// ------------------------------------------------------------------------------------------------ resize
// tap position of destination index d on a source axis of sn samples: integer part and fraction
static void linear_taps(int sn, int dn, int* idx, float* frac) {
  const double scale = (double)sn / dn;
  for (int d = 0; d < dn; ++d) {
    const float f = (float)((d + 0.5) * scale - 0.5);
    const int s = (int)floorf(f);
    idx[d] = s;
    frac[d] = f - (float)s;
  }
}

int tamtr_resize_linear_u8(const uint8_t* src, int sh, int sw, int c, uint8_t* dst, int dh, int dw) {
  if (!src || !dst || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || c <= 0 || c > 4) return -1;
  if (sh == dh && sw == dw) {
    memcpy(dst, src, (size_t)sh * sw * c);
    return 0;
  }
  if (sw == 2 * dw && sh == 2 * dh) {
    for (int y = 0; y < dh; ++y) {
      const uint8_t *r0 = src + (size_t)(2 * y) * sw * c, *r1 = r0 + (size_t)sw * c;
      uint8_t* o = dst + (size_t)y * dw * c;
      for (int x = 0; x < dw; ++x)
        for (int k = 0; k < c; ++k)
          o[x * c + k] = (uint8_t)((r0[2 * x * c + k] + r0[(2 * x + 1) * c + k] + r1[2 * x * c + k] + r1[(2 * x + 1) * c + k] + 2) >> 2);
    }
    return 0;
  }
  int* sx = (int*)malloc(sizeof(int) * (size_t)(2 * dw + dh));
  float* fr = (float*)malloc(sizeof(float) * (size_t)(dw > dh ? dw : dh));
  int* line = (int*)malloc(sizeof(int) * (size_t)2 * dw * c);     // horizontal passes of the two source rows in use
  if (!sx || !fr || !line) { free(sx); free(fr); free(line); return -1; }
  int *a0 = sx + dw, *sy = sx + 2 * dw;
  linear_taps(sw, dw, sx, fr);
  for (int x = 0; x < dw; ++x) {       // a1 = 2048 - ... is NOT assumed: both taps are rounded on their own
    float f = fr[x];
    if (sx[x] < 0 || sx[x] >= sw - 1) f = 0.f;
    sx[x] = clampi(sx[x], 0, sw - 1);
    a0[x] = (int)lrintf((1.f - f) * 2048.f) | ((int)lrintf(f * 2048.f) << 16);
  }
  linear_taps(sh, dh, sy, fr);
  int have[2] = {-1, -1};            // source row held by each line buffer; the two taps are adjacent rows -> distinct parity
  for (int y = 0; y < dh; ++y) {
    const int rows[2] = {clampi(sy[y], 0, sh - 1), clampi(sy[y] + 1, 0, sh - 1)};
    const int b0 = (int)lrintf((1.f - fr[y]) * 2048.f), b1 = (int)lrintf(fr[y] * 2048.f);
    for (int t = 0; t < 2; ++t) {
      const int slot = rows[t] & 1;
      if (have[slot] == rows[t]) continue;
      have[slot] = rows[t];
      const uint8_t* r = src + (size_t)rows[t] * sw * c;
      int* o = line + (size_t)slot * dw * c;
      for (int x = 0; x < dw; ++x) {
        const int w0 = a0[x] & 0xffff, w1 = a0[x] >> 16, x0 = sx[x] * c, x1 = (sx[x] + 1 < sw ? sx[x] + 1 : sw - 1) * c;
        for (int k = 0; k < c; ++k) o[x * c + k] = r[x0 + k] * w0 + r[x1 + k] * w1;
      }
    }
    const int *top = line + (size_t)(rows[0] & 1) * dw * c, *bot = line + (size_t)(rows[1] & 1) * dw * c;
    uint8_t* o = dst + (size_t)y * dw * c;
    for (int i = 0; i < dw * c; ++i) o[i] = sat_u8((((b0 * (top[i] >> 4)) >> 16) + ((b1 * (bot[i] >> 4)) >> 16) + 2) >> 2);
  }
  free(sx); free(fr); free(line);
  return 0;
}

// ------------------------------------------------------------------------------------------------ affine warp
static int g_wtab[32 * 32][4];          // 15-bit weights of the four taps per 1/32-pixel offset (fy * 32 + fx); each row sums to 32768
static pthread_once_t g_wtab_once = PTHREAD_ONCE_INIT;

static void build_wtab(void) {
  for (int iy = 0; iy < 32; ++iy)
    for (int ix = 0; ix < 32; ++ix) {
      const float ty = (float)iy / 32.f, tx = (float)ix / 32.f;
      const float wy[2] = {1.f - ty, ty}, wx[2] = {1.f - tx, tx};
      int* w = g_wtab[iy * 32 + ix];
      int sum = 0, big = 0;
      for (int k = 0; k < 4; ++k) {
        const long v = lrintf(wy[k >> 1] * wx[k & 1] * 32768.f);
        w[k] = (int)(v > 32767 ? 32767 : v);
        sum += w[k];
        if (w[k] > w[big]) big = k;
      }
      w[big] += 32768 - sum;
    }
}

int tamtr_warp_affine_u8(const uint8_t* src, int sh, int sw, int c, const double* M, uint8_t* dst, int dh, int dw, int border) {
  if (!src || !dst || !M || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || c <= 0 || c > 4 || border < 0 || border > 255) return -1;
  pthread_once(&g_wtab_once, build_wtab);
  double det = M[0] * M[4] - M[1] * M[3];
  det = det != 0 ? 1.0 / det : 0.0;
  const double m0 = M[4] * det, m1 = -M[1] * det, m3 = -M[3] * det, m4 = M[0] * det;
  const double b1 = -m0 * M[2] - m1 * M[5], b2 = -m3 * M[2] - m4 * M[5];
  const double big = 1e12 / ((dw > dh ? dw : dh) + 1.0);      // keeps every fixed-point coordinate far inside 63 bits
  if (!(fabs(m0) < big && fabs(m1) < big && fabs(m3) < big && fabs(m4) < big && fabs(b1) < 1e12 && fabs(b2) < 1e12)) return -1;
  long long* ad = (long long*)malloc(sizeof(long long) * (size_t)2 * dw);
  if (!ad) return -1;
  long long* bd = ad + dw;
  for (int x = 0; x < dw; ++x) {
    ad[x] = llrint(m0 * x * 1024);
    bd[x] = llrint(m3 * x * 1024);
  }
  for (int y = 0; y < dh; ++y) {
    const long long X0 = llrint((m1 * y + b1) * 1024) + 16, Y0 = llrint((m4 * y + b2) * 1024) + 16;
    uint8_t* o = dst + (size_t)y * dw * c;
    for (int x = 0; x < dw; ++x) {
      const long long X = (X0 + ad[x]) >> 5, Y = (Y0 + bd[x]) >> 5;
      const long long px = X >> 5, py = Y >> 5;
      const int fx = (int)(X & 31), fy = (int)(Y & 31);
      if (px >= sw || px + 1 < 0 || py >= sh || py + 1 < 0) {
        for (int k = 0; k < c; ++k) o[x * c + k] = (uint8_t)border;
        continue;
      }
      const int inx[2] = {px >= 0, px + 1 < sw}, iny[2] = {py >= 0, py + 1 < sh};
      const int* w = g_wtab[fy * 32 + fx];
      int acc[4] = {0, 0, 0, 0};
      for (int t = 0; t < 4; ++t) {
        if (inx[t & 1] && iny[t >> 1]) {
          const uint8_t* p = src + ((size_t)(py + (t >> 1)) * sw + (size_t)(px + (t & 1))) * c;
          for (int k = 0; k < c; ++k) acc[k] += p[k] * w[t];
        } else {
          for (int k = 0; k < c; ++k) acc[k] += border * w[t];
        }
      }
      for (int k = 0; k < c; ++k) o[x * c + k] = sat_u8((acc[k] + (1 << 14)) >> 15);
    }
  }
  free(ad);
  return 0;
}

// ------------------------------------------------------------------------------------------------ HSV look-up
static int g_sdiv[256], g_hdiv[256];
static pthread_once_t g_div_once = PTHREAD_ONCE_INIT;

static void build_div(void) {
  g_sdiv[0] = g_hdiv[0] = 0;
  for (int i = 1; i < 256; ++i) {
    g_sdiv[i] = (int)llrint((255 << 12) / (double)i);
    g_hdiv[i] = (int)llrint((180 << 12) / (6.0 * i));
  }
}

int tamtr_hsv_lut_u8(uint8_t* rgb, long long n, const uint8_t* lut_h, const uint8_t* lut_s, const uint8_t* lut_v) {
  if (!rgb || !lut_h || !lut_s || !lut_v || n < 0) return -1;
  pthread_once(&g_div_once, build_div);
  static const int pick[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};   // (b, g, r) per sextant
  const float hscale = (float)(6.0 / 180.0), inv255 = (float)(1 / 255.0);
  for (long long i = 0; i < n; ++i) {
    uint8_t* px = rgb + 3 * i;
    const int r = px[0], g = px[1], b = px[2];
    int v = r > g ? r : g; v = v > b ? v : b;
    int mn = r < g ? r : g; mn = mn < b ? mn : b;
    const int diff = v - mn;
    const int s = (diff * g_sdiv[v] + (1 << 11)) >> 12;
    int h = v == r ? g - b : (v == g ? b - r + 2 * diff : r - g + 4 * diff);
    h = (h * g_hdiv[diff] + (1 << 11)) >> 12;
    if (h < 0) h += 180;
    const int H = lut_h[(uint8_t)h], S = lut_s[(uint8_t)s], V = lut_v[(uint8_t)v];
    float hf = (float)H * hscale;
    const float sf = (float)S * inv255, vf = (float)V * inv255;
    float out[3];
    if (S == 0) {
      out[0] = out[1] = out[2] = vf;
    } else {
      if (hf >= 6.f) hf -= 6.f;
      int sector = (int)floorf(hf);
      float fr = hf - (float)sector;
      if (sector >= 6) { sector = 0; fr = 0.f; }
      const float tab[4] = {vf, vf * (1.f - sf), vf * (1.f - sf * fr), vf * (1.f - sf * (1.f - fr))};
      out[0] = tab[pick[sector][2]];   // r
      out[1] = tab[pick[sector][1]];   // g
      out[2] = tab[pick[sector][0]];   // b
    }
    for (int k = 0; k < 3; ++k) {
      const float q = rintf(out[k] * 255.f);
      px[k] = (uint8_t)(q < 0.f ? 0 : (q > 255.f ? 255 : (int)q));
    }
  }
  return 0;
}
