// pyramid.hip -- K0: image-pyramid construction in the pyramid store.
//
// Replaces frame_utils::createImgPyramid -> vk::halfSample
// (svo/src/frame.cpp:156-165; halfSample lives in the un-vendored rpg_vikit,
// vision.cpp).  Integer work, bit-exact with the CPU routine for either of its
// two flavours (scalar truncating mean / SSE2 avg-of-avg).
//
// HBM-bound streaming kernel.  The store is TILED (pyr_addr.h: 16 bytes x 8 rows = one 128-byte line per
// tile, because every reader gathers small 2-D windows); K0 is the only writer, so the tiling costs nothing but
// address arithmetic here: a lane's 16-byte (level 0), 8-byte (level 1) or 4-byte (levels 2+) piece never
// crosses a tile row, and the 64 lanes of a wave together still fill whole 128-byte lines.
#include "capi_common.h"

using namespace svo_capi;
using svo_pyr::px_off;

namespace {

__device__ __forceinline__ uint32_t half4(uint2 t, uint2 b, bool sse2) {
  // t, b: 8 consecutive pixels of the top / bottom row (little endian bytes)
  uint32_t out = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t tw = (k < 2) ? t.x : t.y;
    const uint32_t bw = (k < 2) ? b.x : b.y;
    const int sh = (k & 1) * 16;
    const uint32_t t0 = (tw >> sh) & 0xffu, t1 = (tw >> (sh + 8)) & 0xffu;
    const uint32_t b0 = (bw >> sh) & 0xffu, b1 = (bw >> (sh + 8)) & 0xffu;
    uint32_t v;
    if (sse2) {
      const uint32_t a = (t0 + b0 + 1u) >> 1;  // _mm_avg_epu8(here, next)
      const uint32_t c = (t1 + b1 + 1u) >> 1;
      v = (a + c + 1u) >> 1;                   // _mm_avg_epu16(even, odd)
    } else {
      v = (t0 + t1 + b0 + b1) >> 2;            // (uint16 sum)/4, truncating
    }
    out |= v << (8 * k);
  }
  return out;
}

// one launch per level transition, all slots: grid = (x-blocks, out_h, slots)
__global__ void __launch_bounds__(256) half_sample_kernel(uint8_t* __restrict__ store, int64_t slot_bytes,
                                                         int first_slot, int64_t in_off, int in_pitch,
                                                         int64_t out_off, int out_pitch, int out_w,
                                                         int out_h, int sse2) {
  const int x4 = blockIdx.x * blockDim.x + threadIdx.x;  // output dword index in the row
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x4 * 4 >= out_w || y >= out_h) return;
  uint8_t* slot = store + (int64_t)(first_slot + blockIdx.z) * slot_bytes;
  const uint2 t = *reinterpret_cast<const uint2*>(slot + in_off + px_off(x4 * 8, 2 * y, in_pitch));
  const uint2 b = *reinterpret_cast<const uint2*>(slot + in_off + px_off(x4 * 8, 2 * y + 1, in_pitch));
  *reinterpret_cast<uint32_t*>(slot + out_off + px_off(x4 * 4, y, out_pitch)) = half4(t, b, sse2 != 0);
}

// ---- fused builder ---------------------------------------------------------
// One pass over level 0 builds EVERY level: a workgroup owns a 128x64 tile of level 0
// (256 lanes, each 16 pixels x 2 rows = two dwordx4 loads), forms its 64x32 tile of level 1
// in registers, parks it in LDS, and 128 / 32 / 8 lanes then reduce 32x16, 16x8 and 8x4 tiles
// of levels 2..4 from LDS.  HBM traffic per level-0 pixel: 1 B read + 0.33 B written (plus
// 1 B written when level 0 itself is being filled from packed images), instead of the
// 1 + 0.25 + 2*(0.25 + 0.0625 + ...) of one launch per level.
constexpr int FUSED_MAX_LEVELS = 5;

struct FusedArgs {
  uint8_t* store;
  int64_t slot_bytes;
  int first_slot;
  int n_levels;                       // levels built here (<= FUSED_MAX_LEVELS)
  int w[FUSED_MAX_LEVELS], h[FUSED_MAX_LEVELS], pitch[FUSED_MAX_LEVELS];
  int64_t off[FUSED_MAX_LEVELS];
  int sse2[FUSED_MAX_LEVELS];         // flavour of the transition INTO level l
  const uint8_t* images;              // packed source images, or nullptr: level 0 is in the store
  int64_t image_stride;
  int row_stride;
};

__device__ __forceinline__ void store4(uint8_t* dst, uint32_t v, int x0, int w) {
  if (x0 + 4 <= w) {
    *reinterpret_cast<uint32_t*>(dst) = v;
  } else {
    for (int k = 0; k < w - x0; ++k) dst[k] = (uint8_t)(v >> (8 * k));
  }
}

__device__ __forceinline__ uint4 load16(const uint8_t* src, int x0, int w, bool aligned) {
  uint4 v = make_uint4(0, 0, 0, 0);
  if (x0 + 16 <= w && aligned) {
    v = *reinterpret_cast<const uint4*>(src);
  } else if (x0 < w) {
    uint32_t t[4] = {0, 0, 0, 0};
    const int n = min(16, w - x0);
    for (int k = 0; k < n; ++k) t[k >> 2] |= (uint32_t)src[k] << (8 * (k & 3));
    v = make_uint4(t[0], t[1], t[2], t[3]);
  }
  return v;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ntload16(const uint8_t* p) {
  const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}

// Level-0 tile per workgroup: TW x THT pixels, THT = NS * 8192 / TW; each of the 256 lanes owns
// NS blocks of 16 px x 2 rows (all 2*NS dwordx4 loads are issued before anything is consumed,
// so a lane keeps 32*NS bytes in flight).  <128,1> = 128x64, <256,1> = 256x32, <256,2> = 256x64.
template <int TW, int NS, bool NT = false>
__global__ void __launch_bounds__(256) pyramid_fused_kernel(const FusedArgs a) {
  constexpr int TH = 8192 / TW;          // rows covered by one sub-tile
  constexpr int THT = NS * TH;
  constexpr int WX = TW / 16;            // lanes across the tile
  constexpr int W1 = TW / 2, W2 = TW / 4, W3 = TW / 8;
  constexpr int N2 = (THT / 4) * (W2 / 4), N3 = (THT / 8) * (W3 / 4), N4 = (THT / 16) * (TW / 64);
  static_assert(N2 <= 256, "tile too large");
  __shared__ uint8_t l1[(THT / 2) * W1];
  __shared__ uint8_t l2[(THT / 4) * W2];
  __shared__ uint8_t l3[(THT / 8) * W3];
  const int tid = threadIdx.x;
  const int tx = tid % WX, ty = tid / WX;
  uint8_t* slot = a.store + (int64_t)(a.first_slot + blockIdx.z) * a.slot_bytes;
  const int x0 = blockIdx.x * TW + tx * 16;
  uint4 top[NS], bot[NS];
  const uint8_t* img = a.images ? a.images + (int64_t)blockIdx.z * a.image_stride : nullptr;
  const bool al = img && ((reinterpret_cast<uintptr_t>(img) | (uintptr_t)a.row_stride) & 15) == 0;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int y0 = blockIdx.y * THT + s * TH + ty * 2;
    top[s] = make_uint4(0, 0, 0, 0);
    bot[s] = top[s];
    if (img) {
      if (NT && al && x0 + 16 <= a.w[0]) {  // streaming source: read once, never again
        if (y0 < a.h[0]) top[s] = ntload16(img + (int64_t)y0 * a.row_stride + x0);
        if (y0 + 1 < a.h[0]) bot[s] = ntload16(img + (int64_t)(y0 + 1) * a.row_stride + x0);
      } else {
        if (y0 < a.h[0]) top[s] = load16(img + (int64_t)y0 * a.row_stride + x0, x0, a.w[0], al);
        if (y0 + 1 < a.h[0]) bot[s] = load16(img + (int64_t)(y0 + 1) * a.row_stride + x0, x0, a.w[0], al);
      }
    } else if (x0 < a.w[0]) {
      if (y0 < a.h[0]) top[s] = *reinterpret_cast<const uint4*>(slot + px_off(x0, y0, a.pitch[0]));
      if (y0 + 1 < a.h[0]) bot[s] = *reinterpret_cast<const uint4*>(slot + px_off(x0, y0 + 1, a.pitch[0]));
    }
  }
  if (img && x0 < a.w[0]) {  // fill level 0 on the way; x0 is a multiple of 16: one tile row per store
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int y0 = blockIdx.y * THT + s * TH + ty * 2;
      // plain stores: a wave's two store instructions each cover every other 16-byte row of its tiles, and
      // L2 merges them into whole lines; non-temporal stores of such half-lines measured 35 % slower
      if (y0 < a.h[0]) *reinterpret_cast<uint4*>(slot + px_off(x0, y0, a.pitch[0])) = top[s];
      if (y0 + 1 < a.h[0]) *reinterpret_cast<uint4*>(slot + px_off(x0, y0 + 1, a.pitch[0])) = bot[s];
    }
  }
  if (a.n_levels < 2) return;
#pragma unroll
  for (int s = 0; s < NS; ++s) {  // level 1: 8 px per lane and sub-tile
    const uint32_t lo = half4(make_uint2(top[s].x, top[s].y), make_uint2(bot[s].x, bot[s].y), a.sse2[1] != 0);
    const uint32_t hi = half4(make_uint2(top[s].z, top[s].w), make_uint2(bot[s].z, bot[s].w), a.sse2[1] != 0);
    const int r1 = s * (TH / 2) + ty;
    *reinterpret_cast<uint2*>(&l1[r1 * W1 + tx * 8]) = make_uint2(lo, hi);
    const int ox = blockIdx.x * W1 + tx * 8, oy = blockIdx.y * (THT / 2) + r1;
    if (oy < a.h[1] && ox < a.w[1]) {
      uint8_t* dst = slot + a.off[1] + px_off(ox, oy, a.pitch[1]);
      if (ox + 8 <= a.w[1]) {
        *reinterpret_cast<uint2*>(dst) = make_uint2(lo, hi);
      } else {
        store4(dst, lo, ox, a.w[1]);
        if (ox + 4 < a.w[1]) store4(dst + 4, hi, ox + 4, a.w[1]);
      }
    }
  }
  if (a.n_levels < 3) return;
  __syncthreads();
  if (tid < N2) {  // level 2: 4 px per lane
    const int r = tid / (W2 / 4), c = tid % (W2 / 4);
    const uint2 t = *reinterpret_cast<const uint2*>(&l1[(2 * r) * W1 + c * 8]);
    const uint2 b = *reinterpret_cast<const uint2*>(&l1[(2 * r + 1) * W1 + c * 8]);
    const uint32_t v = half4(t, b, a.sse2[2] != 0);
    *reinterpret_cast<uint32_t*>(&l2[r * W2 + c * 4]) = v;
    const int ox = blockIdx.x * W2 + c * 4, oy = blockIdx.y * (THT / 4) + r;
    if (oy < a.h[2] && ox < a.w[2]) store4(slot + a.off[2] + px_off(ox, oy, a.pitch[2]), v, ox, a.w[2]);
  }
  if (a.n_levels < 4) return;
  __syncthreads();
  if (tid < N3) {  // level 3
    const int r = tid / (W3 / 4), c = tid % (W3 / 4);
    const uint2 t = *reinterpret_cast<const uint2*>(&l2[(2 * r) * W2 + c * 8]);
    const uint2 b = *reinterpret_cast<const uint2*>(&l2[(2 * r + 1) * W2 + c * 8]);
    const uint32_t v = half4(t, b, a.sse2[3] != 0);
    *reinterpret_cast<uint32_t*>(&l3[r * W3 + c * 4]) = v;
    const int ox = blockIdx.x * W3 + c * 4, oy = blockIdx.y * (THT / 8) + r;
    if (oy < a.h[3] && ox < a.w[3]) store4(slot + a.off[3] + px_off(ox, oy, a.pitch[3]), v, ox, a.w[3]);
  }
  if (a.n_levels < 5) return;
  __syncthreads();
  if (tid < N4) {  // level 4
    const int r = tid / (TW / 64), c = tid % (TW / 64);
    const uint2 t = *reinterpret_cast<const uint2*>(&l3[(2 * r) * W3 + c * 8]);
    const uint2 b = *reinterpret_cast<const uint2*>(&l3[(2 * r + 1) * W3 + c * 8]);
    const uint32_t v = half4(t, b, a.sse2[4] != 0);
    const int ox = blockIdx.x * (TW / 16) + c * 4, oy = blockIdx.y * (THT / 16) + r;
    if (oy < a.h[4] && ox < a.w[4]) store4(slot + a.off[4] + px_off(ox, oy, a.pitch[4]), v, ox, a.w[4]);
  }
}

// packed images -> one level of the slots; 4 bytes per lane
__global__ void __launch_bounds__(256) load_level_kernel(uint8_t* __restrict__ store, int64_t slot_bytes,
                                                        int first_slot, int64_t level_off, int pitch, int w, int h,
                                                        const uint8_t* __restrict__ images,
                                                        int64_t image_stride, int row_stride) {
  const int x4 = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x4 * 4 >= w || y >= h) return;
  const uint8_t* src = images + (int64_t)blockIdx.z * image_stride + (int64_t)y * row_stride + x4 * 4;
  uint8_t* dst = store + (int64_t)(first_slot + blockIdx.z) * slot_bytes + level_off + px_off(x4 * 4, y, pitch);
  const int nb = min(4, w - x4 * 4);
  if (nb == 4 && ((reinterpret_cast<uintptr_t>(src) & 3) == 0)) {
    *reinterpret_cast<uint32_t*>(dst) = *reinterpret_cast<const uint32_t*>(src);
  } else {
    for (int k = 0; k < nb; ++k) dst[k] = src[k];
  }
}

// one level of one slot -> packed rows (w bytes each); one byte per lane (tests / debugging only)
__global__ void __launch_bounds__(256) unload_level_kernel(const uint8_t* __restrict__ level, int pitch, int w, int h,
                                                          uint8_t* __restrict__ out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= w || y >= h) return;
  out[(int64_t)y * w + x] = level[px_off(x, y, pitch)];
}

// Device scratch of the upload entry points: the caller's d_staging, or (d_staging == NULL) a stream-ordered
// temporary that is released again behind the work that reads it.
struct Staging {
  uint8_t* p = nullptr;
  bool owned = false;
  hipStream_t s;
  explicit Staging(hipStream_t stream) : s(stream) {}
  int get(void* d_staging, size_t bytes) {
    if (d_staging) {
      p = static_cast<uint8_t*>(d_staging);
      return SVO_HIP_OK;
    }
    void* t = nullptr;
    SVO_HIP_TRY(hipMallocAsync(&t, bytes, s));
    p = static_cast<uint8_t*>(t);
    owned = true;
    return SVO_HIP_OK;
  }
  ~Staging() {
    if (owned && p) (void)hipFreeAsync(p, s);
  }
};

// h rows of w bytes from a host image into the packed staging buffer.  Packed rows (the usual cv::Mat) go as ONE linear
// copy: from pageable memory the runtime stages a 2-D copy through a blit kernel of its own, which a linear copy avoids.
hipError_t copy_rows_h2d(uint8_t* d_dst, const uint8_t* image, int w, int h, int row_stride, hipStream_t s) {
  if (row_stride == w) return hipMemcpyAsync(d_dst, image, (size_t)w * h, hipMemcpyHostToDevice, s);
  return hipMemcpy2DAsync(d_dst, w, image, row_stride, w, h, hipMemcpyHostToDevice, s);
}

}  // namespace

extern "C" {

int svo_hip_pyramid_load_level0(const svo_hip_pyr_layout* L, uint8_t* d_store, int first_slot, int n_slots,
                                const uint8_t* d_images, int64_t image_stride, int row_stride, void* stream) {
  if (!layout_ok(L) || !d_store || !d_images || first_slot < 0 || n_slots < 0 || row_stride < L->w[0])
    return SVO_HIP_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 block(64, 4, 1);
  int done = 0;
  while (done < n_slots) {
    const int chunk = min(n_slots - done, 32768);
    const dim3 grid((L->w[0] / 4 + 1 + 63) / 64, (L->h[0] + 3) / 4, chunk);
    hipLaunchKernelGGL(load_level_kernel, grid, block, 0, s, d_store, L->slot_bytes, first_slot + done, (int64_t)0,
                       L->pitch[0], L->w[0], L->h[0], d_images + (int64_t)done * image_stride, image_stride,
                       row_stride);
    int rc = check_launch();
    if (rc) return rc;
    done += chunk;
  }
  return SVO_HIP_OK;
}

// host image -> packed device scratch -> one (tiled) level of one slot
static int upload_one_level(const svo_hip_pyr_layout* L, uint8_t* d_store, int slot, int level, const uint8_t* image,
                            int row_stride, void* d_staging, hipStream_t s) {
  const int w = L->w[level], h = L->h[level];
  Staging st(s);
  int rc = st.get(d_staging, (size_t)w * h);
  if (rc) return rc;
  SVO_HIP_TRY(copy_rows_h2d(st.p, image, w, h, row_stride, s));
  const dim3 block(64, 4, 1), grid((w / 4 + 1 + 63) / 64, (h + 3) / 4, 1);
  hipLaunchKernelGGL(load_level_kernel, grid, block, 0, s, d_store, L->slot_bytes, slot, L->offset[level], L->pitch[level],
                     w, h, st.p, (int64_t)0, w);
  return check_launch();
}

int svo_hip_pyramid_upload_level0(const svo_hip_pyr_layout* L, uint8_t* d_store, int slot, const uint8_t* image,
                                  int row_stride, void* d_staging, void* stream) {
  if (!layout_ok(L) || !d_store || !image || slot < 0 || row_stride < L->w[0]) return SVO_HIP_EINVAL;
  return upload_one_level(L, d_store, slot, 0, image, row_stride, d_staging, static_cast<hipStream_t>(stream));
}

int svo_hip_pyramid_upload_level(const svo_hip_pyr_layout* L, uint8_t* d_store, int slot, int level, const uint8_t* image,
                                 int row_stride, void* d_staging, void* stream) {
  if (!layout_ok(L) || !d_store || !image || slot < 0 || level < 0 || level >= L->n_levels || row_stride < L->w[level])
    return SVO_HIP_EINVAL;
  return upload_one_level(L, d_store, slot, level, image, row_stride, d_staging, static_cast<hipStream_t>(stream));
}

static bool tile_ok(int tile_width) {
  return tile_width == 0 || tile_width == 128 || tile_width == 256 || tile_width == 257 || tile_width == 512;
}

static int build_impl(const svo_hip_pyr_layout* L, uint8_t* d_store, int first_slot, int n_slots, const uint8_t* d_images,
                      int64_t image_stride, int row_stride, int halfsample_mode, int tile_width, hipStream_t s) {
  auto flavour = [&](int lvl) {
    if (halfsample_mode == SVO_HIP_HALFSAMPLE_AUTO) return (L->w[lvl - 1] % 16) == 0 ? 1 : 0;
    return halfsample_mode == SVO_HIP_HALFSAMPLE_SSE2 ? 1 : 0;
  };
  FusedArgs a;
  a.store = d_store;
  a.slot_bytes = L->slot_bytes;
  a.n_levels = min(L->n_levels, FUSED_MAX_LEVELS);
  for (int l = 0; l < FUSED_MAX_LEVELS; ++l) {
    const bool on = l < a.n_levels;
    a.w[l] = on ? L->w[l] : 0; a.h[l] = on ? L->h[l] : 0; a.pitch[l] = on ? L->pitch[l] : 0; a.off[l] = on ? L->offset[l] : 0;
    a.sse2[l] = (on && l > 0) ? flavour(l) : 0;
  }
  a.image_stride = image_stride;
  a.row_stride = row_stride;
  int done = 0;
  while (done < n_slots) {
    const int chunk = min(n_slots - done, 32768);
    a.first_slot = first_slot + done;
    a.images = d_images ? d_images + (int64_t)done * image_stride : nullptr;
    // tile selection: 0 = by image size; 128 -> 128x64, 256 -> 256x32, 512 -> 256x64 (two blocks per lane)
    const int tw = tile_width ? tile_width : (L->w[0] >= 256 ? 257 : 128);  // 257: 256x32 tile, non-temporal level 0
    if (tw == 512) {
      const dim3 grid((L->w[0] + 255) / 256, (L->h[0] + 63) / 64, chunk);
      hipLaunchKernelGGL((pyramid_fused_kernel<256, 2>), grid, dim3(256), 0, s, a);
    } else if (tw == 257) {  // non-temporal source loads (the packed images are read once)
      const dim3 grid((L->w[0] + 255) / 256, (L->h[0] + 31) / 32, chunk);
      hipLaunchKernelGGL((pyramid_fused_kernel<256, 1, true>), grid, dim3(256), 0, s, a);
    } else if (tw == 256) {
      const dim3 grid((L->w[0] + 255) / 256, (L->h[0] + 31) / 32, chunk);
      hipLaunchKernelGGL((pyramid_fused_kernel<256, 1>), grid, dim3(256), 0, s, a);
    } else {
      const dim3 grid((L->w[0] + 127) / 128, (L->h[0] + 63) / 64, chunk);
      hipLaunchKernelGGL((pyramid_fused_kernel<128, 1>), grid, dim3(256), 0, s, a);
    }
    int rc = check_launch();
    if (rc) return rc;
    done += chunk;
  }
  // levels beyond the fused ones (pyramids deeper than 5 levels): one launch per level
  const dim3 block(64, 4, 1);
  for (int lvl = FUSED_MAX_LEVELS; lvl < L->n_levels; ++lvl) {
    const int sse2 = flavour(lvl);
    const int out_w = L->w[lvl], out_h = L->h[lvl];
    done = 0;
    while (done < n_slots) {
      const int chunk = min(n_slots - done, 32768);
      const dim3 grid(((out_w + 3) / 4 + 63) / 64, (out_h + 3) / 4, chunk);
      hipLaunchKernelGGL(half_sample_kernel, grid, block, 0, s, d_store, L->slot_bytes, first_slot + done,
                         L->offset[lvl - 1], L->pitch[lvl - 1], L->offset[lvl], L->pitch[lvl], out_w, out_h, sse2);
      int rc = check_launch();
      if (rc) return rc;
      done += chunk;
    }
  }
  return SVO_HIP_OK;
}

int svo_hip_pyramid_build(const svo_hip_pyr_layout* L, uint8_t* d_store, int first_slot, int n_slots,
                          int halfsample_mode, void* stream) {
  if (!layout_ok(L) || !d_store || first_slot < 0 || n_slots < 0) return SVO_HIP_EINVAL;
  if (halfsample_mode < SVO_HIP_HALFSAMPLE_SCALAR || halfsample_mode > SVO_HIP_HALFSAMPLE_AUTO)
    return SVO_HIP_EINVAL;
  return build_impl(L, d_store, first_slot, n_slots, nullptr, 0, 0, halfsample_mode, 0, static_cast<hipStream_t>(stream));
}

int svo_hip_pyramid_build_from_images(const svo_hip_pyr_layout* L, uint8_t* d_store, int first_slot, int n_slots,
                                      const uint8_t* d_images, int64_t image_stride, int row_stride,
                                      int halfsample_mode, void* stream) {
  if (!layout_ok(L) || !d_store || !d_images || first_slot < 0 || n_slots < 0 || row_stride < L->w[0]) return SVO_HIP_EINVAL;
  if (halfsample_mode < SVO_HIP_HALFSAMPLE_SCALAR || halfsample_mode > SVO_HIP_HALFSAMPLE_AUTO)
    return SVO_HIP_EINVAL;
  return build_impl(L, d_store, first_slot, n_slots, d_images, image_stride, row_stride, halfsample_mode, 0,
                    static_cast<hipStream_t>(stream));
}

int svo_hip_pyramid_build_tiled(const svo_hip_pyr_layout* L, uint8_t* d_store, int first_slot, int n_slots,
                                const uint8_t* d_images, int64_t image_stride, int row_stride, int halfsample_mode,
                                int tile_width, void* stream) {
  if (!layout_ok(L) || !d_store || first_slot < 0 || n_slots < 0 || !tile_ok(tile_width)) return SVO_HIP_EINVAL;
  if (d_images && row_stride < L->w[0]) return SVO_HIP_EINVAL;
  if (halfsample_mode < SVO_HIP_HALFSAMPLE_SCALAR || halfsample_mode > SVO_HIP_HALFSAMPLE_AUTO)
    return SVO_HIP_EINVAL;
  return build_impl(L, d_store, first_slot, n_slots, d_images, image_stride, row_stride, halfsample_mode, tile_width,
                    static_cast<hipStream_t>(stream));
}

// the level-by-level builder the fused kernel replaced; kept for A/B timing (bench.py --pyramid-ab)
int svo_hip_pyramid_build_per_level(const svo_hip_pyr_layout* L, uint8_t* d_store, int first_slot, int n_slots,
                                    int halfsample_mode, void* stream) {
  if (!layout_ok(L) || !d_store || first_slot < 0 || n_slots < 0) return SVO_HIP_EINVAL;
  if (halfsample_mode < SVO_HIP_HALFSAMPLE_SCALAR || halfsample_mode > SVO_HIP_HALFSAMPLE_AUTO)
    return SVO_HIP_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 block(64, 4, 1);
  for (int lvl = 1; lvl < L->n_levels; ++lvl) {
    int sse2 = halfsample_mode == SVO_HIP_HALFSAMPLE_SSE2;
    if (halfsample_mode == SVO_HIP_HALFSAMPLE_AUTO) sse2 = (L->w[lvl - 1] % 16) == 0;
    const int out_w = L->w[lvl], out_h = L->h[lvl];
    int done = 0;
    while (done < n_slots) {
      const int chunk = min(n_slots - done, 32768);
      const dim3 grid(((out_w + 3) / 4 + 63) / 64, (out_h + 3) / 4, chunk);
      hipLaunchKernelGGL(half_sample_kernel, grid, block, 0, s, d_store, L->slot_bytes, first_slot + done,
                         L->offset[lvl - 1], L->pitch[lvl - 1], L->offset[lvl], L->pitch[lvl], out_w, out_h, sse2);
      int rc = check_launch();
      if (rc) return rc;
      done += chunk;
    }
  }
  return SVO_HIP_OK;
}

// A new camera frame in one go: H2D copy of the image into packed scratch, then ONE kernel that writes level 0
// and every further level of the slot (Frame::initFrame -> createImgPyramid, svo/src/frame.cpp:48-59,156-165).
int svo_hip_pyramid_upload_build(const svo_hip_pyr_layout* L, uint8_t* d_store, int slot, const uint8_t* image,
                                 int row_stride, int halfsample_mode, void* d_staging, void* stream) {
  if (!layout_ok(L) || !d_store || !image || slot < 0 || row_stride < L->w[0]) return SVO_HIP_EINVAL;
  if (halfsample_mode < SVO_HIP_HALFSAMPLE_SCALAR || halfsample_mode > SVO_HIP_HALFSAMPLE_AUTO)
    return SVO_HIP_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int w = L->w[0], h = L->h[0];
  Staging st(s);
  int rc = st.get(d_staging, (size_t)w * h);
  if (rc) return rc;
  SVO_HIP_TRY(copy_rows_h2d(st.p, image, w, h, row_stride, s));
  return build_impl(L, d_store, slot, 1, st.p, (int64_t)w * h, w, halfsample_mode, 0, s);
}

int svo_hip_pyramid_download_level(const svo_hip_pyr_layout* L, const uint8_t* d_store, int slot, int level,
                                   uint8_t* out, void* stream) {
  if (!layout_ok(L) || !d_store || !out || slot < 0 || level < 0 || level >= L->n_levels) return SVO_HIP_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int w = L->w[level], h = L->h[level];
  void* tmp = nullptr;
  SVO_HIP_TRY(hipMalloc(&tmp, (size_t)w * h));
  const dim3 block(64, 4, 1), grid((w + 63) / 64, (h + 3) / 4, 1);
  hipLaunchKernelGGL(unload_level_kernel, grid, block, 0, s, d_store + (int64_t)slot * L->slot_bytes + L->offset[level],
                     L->pitch[level], w, h, static_cast<uint8_t*>(tmp));
  int rc = check_launch();
  if (rc == SVO_HIP_OK) {
    hipError_t e = hipMemcpyAsync(out, tmp, (size_t)w * h, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) rc = hip_fail(e);
  }
  (void)hipFree(tmp);
  return rc;
}

}  // extern "C"
