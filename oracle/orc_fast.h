/*
 * orc_fast.h -- TEST INFRASTRUCTURE ONLY (see svo_oracle.h).
 * Restatement of the FAST-10 corner detector (E. Rosten & T. Drummond, "Machine learning
 * for high-speed corner detection", ECCV 2006) as packaged in uzh-rpg/fast, the un-vendored,
 * un-pinned library svo::feature_detection::FastDetector calls
 * (svo/src/feature_detection.cpp:76-93): fast_corner_detect_10, fast_corner_score_10,
 * fast_nonmax_3x3; and of vk::shiTomasiScore (rpg_vikit vision.cpp) used at :101.
 * Shared by the dependency shim oracle/shim/fast/fast.h (so the reference's own
 * feature_detection.cpp runs on it) and by the tests of the GPU detector.  UNPINNED.
 */
#ifndef ORC_FAST_H_
#define ORC_FAST_H_

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Bresenham circle of radius 3, clockwise from 12 o'clock (dx, dy) */
static const int orc_fast_ring_dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int orc_fast_ring_dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

/* segment test: >= 10 contiguous ring pixels all brighter than p+b or all darker than p-b */
static inline int orc_fast10_is_corner(const uint8_t* p, int stride, int b) {
  int cb = *p + b, c_b = *p - b;
  unsigned bright = 0, dark = 0;
  for (int k = 0; k < 16; ++k) {
    int v = p[orc_fast_ring_dy[k] * stride + orc_fast_ring_dx[k]];
    if (v > cb) bright |= 1u << k;
    if (v < c_b) dark |= 1u << k;
  }
  /* a run of 10 in a circular 16-bit mask */
  unsigned m;
  m = bright | (bright << 16);
  { unsigned r = m; for (int i = 1; i < 10; ++i) r &= m >> i; if (r & 0xffffu) return 1; }
  m = dark | (dark << 16);
  { unsigned r = m; for (int i = 1; i < 10; ++i) r &= m >> i; if (r & 0xffffu) return 1; }
  return 0;
}

/* fast_corner_detect_10: raster order, 3-pixel border.  Returns the count; xy[2*i] = x, y. */
static inline int orc_fast10_detect(const uint8_t* img, int w, int h, int stride, int b, short* xy, int max_corners) {
  int n = 0;
  for (int y = 3; y < h - 3; ++y)
    for (int x = 3; x < w - 3; ++x)
      if (orc_fast10_is_corner(img + y * stride + x, stride, b)) {
        if (n < max_corners) { xy[2 * n] = (short)x; xy[2 * n + 1] = (short)y; }
        ++n;
      }
  return n;
}

/* fast_corner_score_10: largest threshold for which the pixel is still a corner (bisection) */
static inline int orc_fast10_score(const uint8_t* p, int stride, int b) {
  int bmin = b, bmax = 255;
  int t = (bmax + bmin) / 2;
  for (;;) {
    if (orc_fast10_is_corner(p, stride, t)) bmin = t; else bmax = t;
    if (bmin == bmax - 1 || bmin == bmax) return bmin;
    t = (bmin + bmax) / 2;
  }
}

/* fast_nonmax_3x3: corner i survives unless one of its 8 neighbours is a corner with
 * score >= its own.  keep[] receives the surviving indices in input order. */
static inline int orc_fast_nonmax_3x3(const short* xy, const int* scores, int n, int w, int h, int* keep) {
  int* map = (int*)malloc(sizeof(int) * (size_t)w * h);
  for (size_t i = 0; i < (size_t)w * h; ++i) map[i] = -1;
  for (int i = 0; i < n; ++i) map[(size_t)xy[2 * i + 1] * w + xy[2 * i]] = scores[i];
  int m = 0;
  for (int i = 0; i < n; ++i) {
    int x = xy[2 * i], y = xy[2 * i + 1], s = scores[i], ok = 1;
    for (int dy = -1; dy <= 1 && ok; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        if (!dx && !dy) continue;
        int xx = x + dx, yy = y + dy;
        if (xx < 0 || yy < 0 || xx >= w || yy >= h) continue;
        if (map[(size_t)yy * w + xx] >= s) { ok = 0; break; }
      }
    if (ok) keep[m++] = i;
  }
  free(map);
  return m;
}

/* vk::shiTomasiScore: smaller eigenvalue of the 8x8 gradient matrix (central differences) */
static inline float orc_shi_tomasi_score(const uint8_t* data, int cols, int rows, int stride, int u, int v) {
  float dXX = 0.0f, dYY = 0.0f, dXY = 0.0f;
  const int halfbox_size = 4, box_size = 8, box_area = 64;
  const int x_min = u - halfbox_size, x_max = u + halfbox_size;
  const int y_min = v - halfbox_size, y_max = v + halfbox_size;
  if (x_min < 1 || x_max >= cols - 1 || y_min < 1 || y_max >= rows - 1) return 0.0f;
  for (int y = y_min; y < y_max; ++y) {
    const uint8_t* l = data + stride * y + x_min - 1;
    const uint8_t* r = data + stride * y + x_min + 1;
    const uint8_t* t = data + stride * (y - 1) + x_min;
    const uint8_t* bt = data + stride * (y + 1) + x_min;
    for (int x = 0; x < box_size; ++x, ++l, ++r, ++t, ++bt) {
      float dx = (float)(*r - *l);
      float dy = (float)(*bt - *t);
      dXX += dx * dx;
      dYY += dy * dy;
      dXY += dx * dy;
    }
  }
  dXX = dXX / (2.0 * box_area);
  dYY = dYY / (2.0 * box_area);
  dXY = dXY / (2.0 * box_area);
  /* vikit is C++: sqrt(float) resolves to the float overload; written out so that C and C++
   * translation units agree */
  return 0.5 * (dXX + dYY - sqrtf((dXX + dYY) * (dXX + dYY) - 4 * (dXX * dYY - dXY * dXY)));
}

#endif /* ORC_FAST_H_ */
