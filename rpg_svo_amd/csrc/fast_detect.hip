// fast_detect.hip -- K7: seed initialisation on the device.
//
// Replaces svo::feature_detection::FastDetector::detect (svo/src/feature_detection.cpp:66-114):
// per pyramid level FAST-10 segment test (threshold 20), FAST score by bisection, 3x3 non-max
// suppression, then per grid cell the corner with the largest Shi-Tomasi score over all
// levels (vk::shiTomasiScore), skipping occupied cells.  The FAST routines live in the
// un-vendored uzh-rpg/fast library; the published algorithm is followed (Rosten & Drummond,
// ECCV 2006), integer for integer, and the Shi-Tomasi sums are accumulated in the CPU's
// order with contraction off, so the selected corners and scores are bit-identical to the
// restatement the reference's own feature_detection.cpp is run on in oracle/_ref.
//
//   fast_score_kernel   one lane per pixel and level: ring test at b = 20, bisection score
//                       -> score map (u8, 0 = no corner), laid out like the pyramid store (pyr_addr.h);
//                       a workgroup covers 16 x 16 pixels = two 16 x 8 tiles of the store
//   fast_select_kernel  one lane per pixel: 3x3 non-max on the score map; survivors compute
//                       Shi-Tomasi and race for their cell with a 64-bit atomicMax on
//                       (score bits | ~(level, y, x)), which reproduces "strictly greater
//                       wins, first in level/raster order keeps ties"
//   fast_emit_kernel    one lane per cell: decode the winner
#pragma clang fp contract(off)
#include "capi_common.h"

using namespace svo_capi;

namespace {

struct FastArgs {
  svo_hip_pyr_layout L;
  const uint8_t* store;
  uint8_t* score;        // [n_frames] slots of L.slot_bytes
  unsigned long long* keys;  // [n_frames][n_cells]
  const int32_t* slot;   // [n_frames]
  const uint8_t* occupancy;  // [n_frames][n_cells] or NULL
  int n_frames, n_levels, threshold, cell_size, grid_n_cols, n_cells;
  float detection_threshold;
  int32_t* out_xy;
  int32_t* out_level;
  float* out_score;
};

__device__ __forceinline__ bool run10(unsigned m) {
  m |= m << 16;
  unsigned r = m;
#pragma unroll
  for (int i = 1; i < 10; ++i) r &= m >> i;
  return (r & 0xffffu) != 0;
}

__device__ __forceinline__ bool is_corner(const int ring[16], int p, int b) {
  const int cb = p + b, c_b = p - b;
  unsigned bright = 0, dark = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    bright |= (ring[k] > cb ? 1u : 0u) << k;
    dark |= (ring[k] < c_b ? 1u : 0u) << k;
  }
  return run10(bright) || run10(dark);
}

__global__ void __launch_bounds__(256) fast_score_kernel(const FastArgs a) {
  const int f = blockIdx.z / a.n_levels, l = blockIdx.z % a.n_levels;
  const int w = a.L.w[l], h = a.L.h[l], pitch = a.L.pitch[l];
  const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (x >= w || y >= h) return;
  uint8_t* smap = a.score + (int64_t)f * a.L.slot_bytes + a.L.offset[l];
  uint8_t s = 0;
  if (x >= 3 && y >= 3 && x < w - 3 && y < h - 3) {
    const uint8_t* img = a.store + (int64_t)a.slot[f] * a.L.slot_bytes + a.L.offset[l];
    // Bresenham circle r = 3, clockwise from 12 o'clock
    const int dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
    const int dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
    uint32_t ro[7], co[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      ro[k] = svo_pyr::row_off(y + k - 3, pitch);
      co[k] = svo_pyr::col_off(x + k - 3);
    }
    int ring[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) ring[k] = img[ro[dy[k] + 3] + co[dx[k] + 3]];
    const int p = img[ro[3] + co[3]];
    if (is_corner(ring, p, a.threshold)) {
      int bmin = a.threshold, bmax = 255, t = (bmax + bmin) / 2;  // fast_corner_score_10
      for (;;) {
        if (is_corner(ring, p, t)) bmin = t; else bmax = t;
        if (bmin == bmax - 1 || bmin == bmax) break;
        t = (bmin + bmax) / 2;
      }
      s = (uint8_t)bmin;
    }
  }
  smap[svo_pyr::px_off(x, y, pitch)] = s;
}

// vk::shiTomasiScore: float sums in the CPU's order
__device__ __forceinline__ float shi_tomasi(const uint8_t* data, int cols, int rows, int pitch, int u, int v) {
  float dXX = 0.0f, dYY = 0.0f, dXY = 0.0f;
  const int x_min = u - 4, x_max = u + 4, y_min = v - 4, y_max = v + 4;
  if (x_min < 1 || x_max >= cols - 1 || y_min < 1 || y_max >= rows - 1) return 0.0f;
  uint32_t co[10];  // columns x_min-1 .. x_min+8
#pragma unroll
  for (int k = 0; k < 10; ++k) co[k] = svo_pyr::col_off(x_min - 1 + k);
  for (int y = y_min; y < y_max; ++y) {
    const uint32_t rm = svo_pyr::row_off(y - 1, pitch), r0 = svo_pyr::row_off(y, pitch), rp = svo_pyr::row_off(y + 1, pitch);
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      const float dx = (float)((int)data[r0 + co[x + 2]] - (int)data[r0 + co[x]]);
      const float dy = (float)((int)data[rp + co[x + 1]] - (int)data[rm + co[x + 1]]);
      dXX += dx * dx;
      dYY += dy * dy;
      dXY += dx * dy;
    }
  }
  dXX = dXX / 128.0f;  // / (2.0 * box_area): exact scaling by a power of two
  dYY = dYY / 128.0f;
  dXY = dXY / 128.0f;
  const float tr = dXX + dYY;
  return 0.5f * (tr - sqrtf(tr * tr - 4 * (dXX * dYY - dXY * dXY)));
}

__global__ void __launch_bounds__(256) fast_select_kernel(const FastArgs a) {
  const int f = blockIdx.z / a.n_levels, l = blockIdx.z % a.n_levels;
  const int w = a.L.w[l], h = a.L.h[l], pitch = a.L.pitch[l];
  const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (x < 3 || y < 3 || x >= w - 3 || y >= h - 3) return;
  const uint8_t* smap = a.score + (int64_t)f * a.L.slot_bytes + a.L.offset[l];
  const uint32_t r0 = svo_pyr::row_off(y, pitch), rm = svo_pyr::row_off(y - 1, pitch), rp = svo_pyr::row_off(y + 1, pitch);
  const uint32_t c0 = svo_pyr::col_off(x), cm = svo_pyr::col_off(x - 1), cp = svo_pyr::col_off(x + 1);
  const int s = smap[r0 + c0];
  if (s == 0) return;
  // fast_nonmax_3x3: suppressed by any neighbouring corner with score >= own
  if (smap[r0 + cm] >= s || smap[r0 + cp] >= s || smap[rm + cm] >= s || smap[rm + c0] >= s || smap[rm + cp] >= s ||
      smap[rp + cm] >= s || smap[rp + c0] >= s || smap[rp + cp] >= s)
    return;
  const int scale = 1 << l;
  const int k = ((y * scale) / a.cell_size) * a.grid_n_cols + (x * scale) / a.cell_size;
  if (k < 0 || k >= a.n_cells) return;
  if (a.occupancy && a.occupancy[(int64_t)f * a.n_cells + k]) return;
  const uint8_t* img = a.store + (int64_t)a.slot[f] * a.L.slot_bytes + a.L.offset[l];
  const float score = shi_tomasi(img, w, h, pitch, x, y);
  if (!(score > a.detection_threshold)) return;
  const unsigned order = ((unsigned)l << 28) | ((unsigned)y << 14) | (unsigned)x;
  const unsigned long long key = ((unsigned long long)__float_as_uint(score) << 32) | (unsigned long long)(~order);
  atomicMax(a.keys + (int64_t)f * a.n_cells + k, key);
}

__global__ void __launch_bounds__(256) fast_emit_kernel(const FastArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n_frames * a.n_cells) return;
  const unsigned long long key = a.keys[i];
  if (key == 0ull) {
    a.out_xy[2 * i] = a.out_xy[2 * i + 1] = -1;
    a.out_level[i] = -1;
    a.out_score[i] = a.detection_threshold;
    return;
  }
  const unsigned order = ~(unsigned)(key & 0xffffffffu);
  const int l = (int)(order >> 28), y = (int)((order >> 14) & 0x3fffu), x = (int)(order & 0x3fffu);
  a.out_xy[2 * i] = x << l;
  a.out_xy[2 * i + 1] = y << l;
  a.out_level[i] = l;
  a.out_score[i] = __uint_as_float((unsigned)(key >> 32));
}

}  // namespace

extern "C" {

size_t svo_hip_fast_workspace_bytes(const svo_hip_pyr_layout* L, int n_frames, int n_cells) {
  if (!layout_ok(L) || n_frames < 0 || n_cells < 0) return 0;
  return (size_t)n_frames * (size_t)L->slot_bytes + SVO_HIP_STORE_TAIL_PAD + (size_t)n_frames * n_cells * 8 + 256;
}

int svo_hip_fast_detect(const svo_hip_pyr_layout* L, const uint8_t* d_store, int n_frames, const int32_t* d_slot,
                        int n_levels, int fast_threshold, int cell_size, int grid_n_cols, int grid_n_rows,
                        const uint8_t* d_occupancy, double detection_threshold, int32_t* d_corner_xy,
                        int32_t* d_corner_level, float* d_corner_score, void* d_workspace, size_t workspace_bytes,
                        void* stream) {
  if (!layout_ok(L) || !d_store || !d_slot || n_frames < 0 || n_levels < 1 || n_levels > L->n_levels || n_levels > 8 ||
      cell_size < 1 || grid_n_cols < 1 || grid_n_rows < 1 || !d_corner_xy || !d_corner_level || !d_corner_score ||
      fast_threshold < 1 || fast_threshold > 254 || detection_threshold < 0.0)
    return SVO_HIP_EINVAL;
  if (L->w[0] >= 16384 || L->h[0] >= 16384) return SVO_HIP_ERANGE;
  if (n_frames == 0) return SVO_HIP_OK;
  const int n_cells = grid_n_cols * grid_n_rows;
  if (!d_workspace || workspace_bytes < svo_hip_fast_workspace_bytes(L, n_frames, n_cells)) return SVO_HIP_ERANGE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  FastArgs a;
  a.L = *L;
  a.store = d_store;
  a.score = static_cast<uint8_t*>(d_workspace);
  const size_t keys_off = (((size_t)n_frames * (size_t)L->slot_bytes + SVO_HIP_STORE_TAIL_PAD) + 255) & ~(size_t)255;
  a.keys = reinterpret_cast<unsigned long long*>(static_cast<uint8_t*>(d_workspace) + keys_off);
  a.slot = d_slot;
  a.occupancy = d_occupancy;
  a.n_frames = n_frames; a.n_levels = n_levels; a.threshold = fast_threshold; a.cell_size = cell_size;
  a.grid_n_cols = grid_n_cols; a.n_cells = n_cells;
  a.detection_threshold = (float)detection_threshold;
  a.out_xy = d_corner_xy; a.out_level = d_corner_level; a.out_score = d_corner_score;
  SVO_HIP_TRY(hipMemsetAsync(a.keys, 0, (size_t)n_frames * n_cells * 8, s));
  const dim3 block(256, 1, 1);
  int done = 0;
  while (done < n_frames) {  // grid.z limit
    const int chunk = min(n_frames - done, 65535 / n_levels);
    FastArgs c = a;
    c.n_frames = chunk;
    c.score = a.score + (size_t)done * L->slot_bytes;
    c.keys = a.keys + (size_t)done * n_cells;
    c.slot = a.slot + done;
    c.occupancy = a.occupancy ? a.occupancy + (size_t)done * n_cells : nullptr;
    const dim3 grid((L->w[0] + 15) / 16, (L->h[0] + 15) / 16, chunk * n_levels);
    hipLaunchKernelGGL(fast_score_kernel, grid, block, 0, s, c);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(fast_select_kernel, grid, block, 0, s, c);
    rc = check_launch();
    if (rc) return rc;
    done += chunk;
  }
  hipLaunchKernelGGL(fast_emit_kernel, dim3((n_frames * n_cells + 255) / 256), dim3(256), 0, s, a);
  return check_launch();
}

}  // extern "C"
