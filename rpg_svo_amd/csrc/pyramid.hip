// pyramid.hip -- K0: image-pyramid construction in the pyramid store.
//
// Replaces frame_utils::createImgPyramid -> vk::halfSample
// (svo/src/frame.cpp:156-165; halfSample lives in the un-vendored rpg_vikit,
// vision.cpp).  Integer work, bit-exact with the CPU routine for either of its
// two flavours (scalar truncating mean / SSE2 avg-of-avg).
//
// HBM-bound streaming kernel: per output dword one lane reads 2 x 8 contiguous
// bytes (rows are 64-byte aligned in the store, so wave loads are fully
// coalesced 512-byte segments) and writes 4 bytes.
#include "capi_common.h"

using namespace svo_capi;

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
  const uint8_t* top = slot + in_off + (int64_t)(2 * y) * in_pitch + x4 * 8;
  const uint2 t = *reinterpret_cast<const uint2*>(top);
  const uint2 b = *reinterpret_cast<const uint2*>(top + in_pitch);
  *reinterpret_cast<uint32_t*>(slot + out_off + (int64_t)y * out_pitch + x4 * 4) = half4(t, b, sse2 != 0);
}

// packed images -> level 0 of the slots; 4 bytes per lane
__global__ void __launch_bounds__(256) load_level0_kernel(uint8_t* __restrict__ store, int64_t slot_bytes,
                                                         int first_slot, int pitch, int w, int h,
                                                         const uint8_t* __restrict__ images,
                                                         int64_t image_stride, int row_stride) {
  const int x4 = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x4 * 4 >= w || y >= h) return;
  const uint8_t* src = images + (int64_t)blockIdx.z * image_stride + (int64_t)y * row_stride + x4 * 4;
  uint8_t* dst = store + (int64_t)(first_slot + blockIdx.z) * slot_bytes + (int64_t)y * pitch + x4 * 4;
  const int nb = min(4, w - x4 * 4);
  if (nb == 4 && ((reinterpret_cast<uintptr_t>(src) & 3) == 0)) {
    *reinterpret_cast<uint32_t*>(dst) = *reinterpret_cast<const uint32_t*>(src);
  } else {
    for (int k = 0; k < nb; ++k) dst[k] = src[k];
  }
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
    hipLaunchKernelGGL(load_level0_kernel, grid, block, 0, s, d_store, L->slot_bytes, first_slot + done,
                       L->pitch[0], L->w[0], L->h[0], d_images + (int64_t)done * image_stride, image_stride,
                       row_stride);
    int rc = check_launch();
    if (rc) return rc;
    done += chunk;
  }
  return SVO_HIP_OK;
}

int svo_hip_pyramid_upload_level0(const svo_hip_pyr_layout* L, uint8_t* d_store, int slot, const uint8_t* image,
                                  int row_stride, void* stream) {
  if (!layout_ok(L) || !d_store || !image || slot < 0 || row_stride < L->w[0]) return SVO_HIP_EINVAL;
  SVO_HIP_TRY(hipMemcpy2DAsync(d_store + (int64_t)slot * L->slot_bytes, L->pitch[0], image, row_stride, L->w[0],
                               L->h[0], hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
  return SVO_HIP_OK;
}

int svo_hip_pyramid_build(const svo_hip_pyr_layout* L, uint8_t* d_store, int first_slot, int n_slots,
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

int svo_hip_pyramid_download_level(const svo_hip_pyr_layout* L, const uint8_t* d_store, int slot, int level,
                                   uint8_t* out, void* stream) {
  if (!layout_ok(L) || !d_store || !out || slot < 0 || level < 0 || level >= L->n_levels) return SVO_HIP_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  SVO_HIP_TRY(hipMemcpy2DAsync(out, L->w[level], d_store + (int64_t)slot * L->slot_bytes + L->offset[level],
                               L->pitch[level], L->w[level], L->h[level], hipMemcpyDeviceToHost, s));
  SVO_HIP_TRY(hipStreamSynchronize(s));
  return SVO_HIP_OK;
}

}  // extern "C"
