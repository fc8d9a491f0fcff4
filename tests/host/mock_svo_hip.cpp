// tests/host/mock_svo_hip.cpp -- TEST INFRASTRUCTURE ONLY.
//
// A stand-in for the handful of C-ABI entry points (include/svo_hip.h) that the product's C++ host layer
// (rpg_svo_amd/host/svo_hip_device.cpp) calls, so that its LOGIC -- arena addressing, slot cache / LRU / pinning,
// prediction and deferred-call bookkeeping, lane set-up -- can be unit-tested where there is no GPU
// (tests/test_host_device_gpu.py::test_host_logic_against_the_mock_abi).  "Device memory" is host memory, streams
// run synchronously, an event has fired as soon as it is recorded, the pyramid store is row-major with the plain
// half-sampler.  Nothing of the product links this file; on the GPU box the same test binary is linked against
// libsvo_hip.so instead.
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include <svo_hip.h>

extern "C" {

const char* svo_hip_strerror(int code) { return code == SVO_HIP_EINVAL ? "invalid argument (mock)" : "error (mock)"; }
int svo_hip_last_hip_error(void) { return 0; }
int svo_hip_device_count(void) { return 1; }
int svo_hip_set_device(int) { return SVO_HIP_OK; }
int svo_hip_pin_calling_thread(void) { return 0; }  // (no device: nothing to be next to)

int svo_hip_malloc(void** p, size_t bytes) {  // page aligned, like the real allocations
  if (!p) return SVO_HIP_EINVAL;
  *p = NULL;
  if (posix_memalign(p, 4096, bytes ? bytes : 1) != 0) return SVO_HIP_ENOMEM;
  return SVO_HIP_OK;
}
int svo_hip_free(void* p) { std::free(p); return SVO_HIP_OK; }
int svo_hip_host_alloc(void** p, size_t bytes) { return svo_hip_malloc(p, bytes); }
int svo_hip_host_free(void* p) { std::free(p); return SVO_HIP_OK; }
int svo_hip_memcpy_h2d(void* d, const void* s, size_t n, void*) { std::memcpy(d, s, n); return SVO_HIP_OK; }
int svo_hip_memcpy_d2h(void* d, const void* s, size_t n, void*) { std::memcpy(d, s, n); return SVO_HIP_OK; }
int svo_hip_memcpy_d2d(void* d, const void* s, size_t n, void*) { std::memmove(d, s, n); return SVO_HIP_OK; }
int svo_hip_memset(void* d, int v, size_t n, void*) { std::memset(d, v, n); return SVO_HIP_OK; }

int svo_hip_stream_create(void** s) { *s = std::malloc(1); return SVO_HIP_OK; }
int svo_hip_stream_destroy(void* s) { std::free(s); return SVO_HIP_OK; }
int svo_hip_stream_sync(void*) { return SVO_HIP_OK; }
int svo_hip_event_create(void** e) { *e = std::calloc(1, sizeof(int)); return SVO_HIP_OK; }
int svo_hip_event_destroy(void* e) { std::free(e); return SVO_HIP_OK; }
int svo_hip_event_record(void* e, void*) { *static_cast<int*>(e) = 1; return SVO_HIP_OK; }
int svo_hip_event_sync(void*) { return SVO_HIP_OK; }
int svo_hip_event_query(void* e) { return *static_cast<int*>(e); }  // 1 once recorded: the mock's streams are synchronous
int svo_hip_stream_wait_event(void*, void*) { return SVO_HIP_OK; }

size_t svo_hip_match_workspace_bytes(int M) { return (size_t)(M > 0 ? M : 1) * 1024; }

int svo_hip_pyr_layout_init(int width, int height, int n_levels, svo_hip_pyr_layout* L) {
  if (!L || width < 1 || height < 1 || n_levels < 1 || n_levels > SVO_HIP_MAX_LEVELS) return SVO_HIP_EINVAL;
  std::memset(L, 0, sizeof(*L));
  L->n_levels = n_levels;
  L->tile = SVO_HIP_PYR_ROWMAJOR;
  int64_t off = 0;
  for (int l = 0; l < n_levels; ++l) {
    L->w[l] = l ? L->w[l - 1] / 2 : width;
    L->h[l] = l ? L->h[l - 1] / 2 : height;
    L->pitch[l] = L->w[l];
    L->offset[l] = off;
    off += (int64_t)L->pitch[l] * L->h[l];
  }
  L->slot_bytes = off;
  return SVO_HIP_OK;
}
int64_t svo_hip_pyr_store_bytes(const svo_hip_pyr_layout* L, int n_slots) { return L->slot_bytes * n_slots + 256; }

static uint8_t* level_ptr(const svo_hip_pyr_layout* L, uint8_t* store, int slot, int level) {
  return store + (int64_t)slot * L->slot_bytes + L->offset[level];
}

int svo_hip_pyramid_upload_level(const svo_hip_pyr_layout* L, uint8_t* store, int slot, int level, const uint8_t* image,
                                 int row_stride, void*, void*) {
  for (int y = 0; y < L->h[level]; ++y)
    std::memcpy(level_ptr(L, store, slot, level) + (size_t)y * L->pitch[level], image + (size_t)y * row_stride, L->w[level]);
  return SVO_HIP_OK;
}

// level 0 from the host image, the further levels by vk::halfSample's SSE2 flavour (avg of avgs, rounding up) where the
// source width is a multiple of 16, the plain mean of four otherwise -- what SVO_HIP_HALFSAMPLE_AUTO does
int svo_hip_pyramid_upload_build(const svo_hip_pyr_layout* L, uint8_t* store, int slot, const uint8_t* image, int row_stride,
                                 int, void* st, void* stream) {
  svo_hip_pyramid_upload_level(L, store, slot, 0, image, row_stride, st, stream);
  for (int l = 1; l < L->n_levels; ++l) {
    const uint8_t* src = level_ptr(L, store, slot, l - 1);
    uint8_t* dst = level_ptr(L, store, slot, l);
    const int sp = L->pitch[l - 1];
    const bool sse = L->w[l - 1] % 16 == 0;
    for (int y = 0; y < L->h[l]; ++y)
      for (int x = 0; x < L->w[l]; ++x) {
        const uint8_t* p = src + (size_t)(2 * y) * sp + 2 * x;
        if (sse) {
          const unsigned a = (p[0] + p[sp] + 1u) >> 1, b = (p[1] + p[sp + 1] + 1u) >> 1;
          dst[(size_t)y * L->pitch[l] + x] = (uint8_t)((a + b + 1u) >> 1);
        } else {
          dst[(size_t)y * L->pitch[l] + x] = (uint8_t)((p[0] + p[1] + p[sp] + p[sp + 1]) / 4);
        }
      }
  }
  return SVO_HIP_OK;
}

int svo_hip_pyramid_download_level(const svo_hip_pyr_layout* L, const uint8_t* store, int slot, int level, uint8_t* out, void*) {
  for (int y = 0; y < L->h[level]; ++y)
    std::memcpy(out + (size_t)y * L->w[level], level_ptr(L, const_cast<uint8_t*>(store), slot, level) + (size_t)y * L->pitch[level],
                L->w[level]);
  return SVO_HIP_OK;
}

}  // extern "C"
