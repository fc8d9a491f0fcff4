/*
 * orc_vikit.h -- TEST INFRASTRUCTURE ONLY (see svo_oracle.h).
 * Restatements, from the published upstream sources, of the small rpg_vikit
 * (vikit_common) and boost::math routines the reference's hot path calls but does not
 * vendor.  Shared by the C oracle (svo_oracle*.c) and by the dependency shims under
 * oracle/shim/ that oracle/_ref is compiled against, so both sides execute the same
 * third-party arithmetic.  UNPINNED: there is no copy of rpg_vikit / boost in this image.
 */
#ifndef ORC_VIKIT_H_
#define ORC_VIKIT_H_

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* vk::halfSample (vikit vision.cpp).  mode: 0 scalar (a+b+c+d)/4 truncating;
 * 1 = halfSampleSSE2 (avg_epu8 of the two rows, then avg_epu16 of column pairs, both
 * rounding up); 2 = what an x86 build dispatches to: SSE2 iff in_w % 16 == 0
 * (aligned_mem::is_aligned16(in.data) && is_aligned16(out.data) && (in.cols % 16)==0). */
static inline void orc_half_sample_impl(const uint8_t* in, int in_w, int in_h, int in_stride,
                                        uint8_t* out, int out_stride, int mode) {
  const int out_w = in_w / 2;
  const int out_h = in_h / 2;
  if (mode == 2) mode = (in_w % 16 == 0) ? 1 : 0;
  for (int y = 0; y < out_h; ++y) {
    const uint8_t* top = in + (size_t)(2 * y) * in_stride;
    const uint8_t* bottom = top + in_stride;
    uint8_t* p = out + (size_t)y * out_stride;
    if (mode == 0) {
      for (int j = 0; j < out_w; ++j)
        p[j] = (uint8_t)(((uint16_t)top[2 * j] + top[2 * j + 1] + bottom[2 * j] + bottom[2 * j + 1]) / 4);
    } else {
      for (int j = 0; j < out_w; ++j) {
        unsigned a = ((unsigned)top[2 * j] + bottom[2 * j] + 1u) >> 1;
        unsigned b = ((unsigned)top[2 * j + 1] + bottom[2 * j + 1] + 1u) >> 1;
        p[j] = (uint8_t)((a + b + 1u) >> 1);
      }
    }
  }
}

/* vk::interpolateMat_8u (vikit vision.h): bilinear sample of an 8-bit image, float weights,
 * w11 = 1 - w00 - w01 - w10. */
static inline float orc_interpolate_mat_8u(const uint8_t* data, int stride, float u, float v) {
  int x = (int)floorf(u);
  int y = (int)floorf(v);
  float subpix_x = u - x;
  float subpix_y = v - y;
  float w00 = (1.0f - subpix_x) * (1.0f - subpix_y);
  float w01 = (1.0f - subpix_x) * subpix_y;
  float w10 = subpix_x * (1.0f - subpix_y);
  float w11 = 1.0f - w00 - w01 - w10;
  const uint8_t* ptr = data + y * stride + x;
  return w00 * ptr[0] + w01 * ptr[stride] + w10 * ptr[1] + w11 * ptr[stride + 1];
}

/* vk::robust_cost::TukeyWeightFunction::value (vikit robust_cost.cpp), b = 4.6851 */
static inline float orc_tukey_weight(float x) {
  const float b_square = 4.6851f * 4.6851f;
  const float x_square = x * x;
  if (x_square <= b_square) {
    const float tmp = 1.0f - x_square / b_square;
    return tmp * tmp;
  }
  return 0.0f;
}

/* vk::getMedian: nth_element at floor(size/2); the value is algorithm independent */
static int orc_cmp_float_(const void* a, const void* b) {
  float x = *(const float*)a, y = *(const float*)b;
  return (x > y) - (x < y);
}
static int orc_cmp_double_(const void* a, const void* b) {
  double x = *(const double*)a, y = *(const double*)b;
  return (x > y) - (x < y);
}
static inline float orc_median_float(float* v, int n) { /* sorts v in place */
  qsort(v, (size_t)n, sizeof(float), orc_cmp_float_);
  return v[n / 2];
}
static inline double orc_median_double(double* v, int n) {
  qsort(v, (size_t)n, sizeof(double), orc_cmp_double_);
  return v[n / 2];
}
#define ORC_MAD_NORMALIZER 1.48f /* vk::robust_cost::MADScaleEstimator::NORMALIZER */

/* vk::patch_score::ZMSSD<4> (vikit patch_score.h): zero-mean SSD of 8x8 u8 patches,
 * threshold() = 2000*patch_area. */
#define ORC_ZMSSD_THRESHOLD (2000 * 64)
static inline void orc_zmssd_init(const uint8_t ref_patch[64], int* sumA, int* sumAA) {
  uint32_t a = 0, aa = 0;
  for (int r = 0; r < 64; ++r) {
    uint8_t n = ref_patch[r];
    a += n;
    aa += n * n;
  }
  *sumA = (int)a;
  *sumAA = (int)aa;
}
static inline int orc_zmssd_score(const uint8_t ref_patch[64], int sumA, int sumAA,
                                  const uint8_t* cur_patch, int stride) {
  uint32_t sumB_uint = 0, sumBB_uint = 0, sumAB_uint = 0;
  for (int y = 0; y < 8; ++y) {
    const uint8_t* c = cur_patch + y * stride;
    for (int x = 0; x < 8; ++x) {
      const uint8_t cur = c[x];
      sumB_uint += cur;
      sumBB_uint += cur * cur;
      sumAB_uint += cur * ref_patch[y * 8 + x];
    }
  }
  const int sumB = (int)sumB_uint, sumBB = (int)sumBB_uint, sumAB = (int)sumAB_uint;
  return sumAA - 2 * sumAB + sumBB - (sumA * sumA - 2 * sumA * sumB + sumB * sumB) / 64;
}

/* boost::math::pdf(normal_distribution<float>(mean, sd), x), all in float */
static inline float orc_normal_pdff(float x, float mean, float sd) {
  float exponent = x - mean;
  exponent *= -exponent;
  exponent /= 2 * sd * sd;
  float result = expf(exponent);
  result /= sd * sqrtf(2 * 3.14159265358979323846f);
  return result;
}

#endif /* ORC_VIKIT_H_ */
