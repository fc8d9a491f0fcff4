// capi_util.hip -- error strings, raw device helpers and the pyramid-store
// layout of the C ABI (include/svo_hip.h).  No kernels here.
#include <sched.h>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>

#include <cstring>

#include <cmath>

#include "capi_common.h"

namespace svo_capi {
thread_local int g_last_hip_error = 0;
}

using namespace svo_capi;

extern "C" {

const char* svo_hip_strerror(int code) {
  switch (code) {
    case SVO_HIP_OK: return "ok";
    case SVO_HIP_EINVAL: return "invalid argument";
    case SVO_HIP_ERANGE: return "size beyond a documented limit";
    case SVO_HIP_EHIP: return "HIP runtime call failed (see svo_hip_last_hip_error)";
    case SVO_HIP_ENODEV: return "no usable HIP device";
    case SVO_HIP_ENOMEM: return "out of memory";
    default: return "unknown svo_hip error";
  }
}

int svo_hip_last_hip_error(void) { return g_last_hip_error; }

const char* svo_hip_version(void) { return "svo_hip 0.1 (gfx950)"; }

int svo_hip_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    g_last_hip_error = static_cast<int>(e);
    return SVO_HIP_ENODEV;
  }
  return n;
}

// The CPUs next to the current device (its PCI function's local_cpulist in sysfs), intersected with what the calling
// thread may run on; the calling thread is bound to them.  See the header.
int svo_hip_pin_calling_thread(void) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  char bdf[32] = {0};
  if (e == hipSuccess) e = hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), dev);
  if (e != hipSuccess) {
    g_last_hip_error = static_cast<int>(e);
    return SVO_HIP_EHIP;
  }
  for (char* c = bdf; *c; ++c) *c = (char)tolower(*c);
  char path[128];
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/local_cpulist", bdf);
  FILE* f = fopen(path, "r");
  if (!f) return 0;  // (no sysfs, no NUMA information: nothing to do)
  char list[4096] = {0};
  const size_t n_read = fread(list, 1, sizeof(list) - 1, f);
  fclose(f);
  if (n_read == 0) return 0;
  cpu_set_t allowed, want;
  CPU_ZERO(&want);
  if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return 0;
  // "0-63,128-191"
  int n_want = 0;
  for (const char* c = list; *c;) {
    while (*c && !isdigit((unsigned char)*c)) ++c;
    if (!*c) break;
    char* end;
    long lo = strtol(c, &end, 10), hi = lo;
    if (*end == '-') hi = strtol(end + 1, &end, 10);
    c = end;
    for (long k = lo; k <= hi && k < CPU_SETSIZE; ++k)
      if (k >= 0 && CPU_ISSET((int)k, &allowed)) { CPU_SET((int)k, &want); ++n_want; }
  }
  if (n_want == 0 || n_want == CPU_COUNT(&allowed)) return 0;  // nothing in common, or nothing to narrow
  if (sched_setaffinity(0, sizeof(want), &want) != 0) return 0;
  return n_want;
}

int svo_hip_set_device(int device) {
  SVO_HIP_TRY(hipSetDevice(device));
  return SVO_HIP_OK;
}

int svo_hip_malloc(void** d_ptr, size_t bytes) {
  if (!d_ptr) return SVO_HIP_EINVAL;
  hipError_t e = hipMalloc(d_ptr, bytes);
  if (e == hipErrorOutOfMemory) {
    g_last_hip_error = static_cast<int>(e);
    return SVO_HIP_ENOMEM;
  }
  SVO_HIP_TRY(e);
  return SVO_HIP_OK;
}

int svo_hip_free(void* d_ptr) {
  SVO_HIP_TRY(hipFree(d_ptr));
  return SVO_HIP_OK;
}

int svo_hip_host_alloc(void** ptr, size_t bytes) {
  if (!ptr) return SVO_HIP_EINVAL;
  hipError_t e = hipHostMalloc(ptr, bytes, hipHostMallocDefault);
  if (e == hipErrorOutOfMemory) {
    g_last_hip_error = static_cast<int>(e);
    return SVO_HIP_ENOMEM;
  }
  SVO_HIP_TRY(e);
  return SVO_HIP_OK;
}

int svo_hip_host_free(void* ptr) {
  SVO_HIP_TRY(hipHostFree(ptr));
  return SVO_HIP_OK;
}

int svo_hip_memcpy_h2d(void* d_dst, const void* src, size_t bytes, void* stream) {
  SVO_HIP_TRY(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
  return SVO_HIP_OK;
}

int svo_hip_memcpy_d2h(void* dst, const void* d_src, size_t bytes, void* stream) {
  SVO_HIP_TRY(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
  return SVO_HIP_OK;
}

int svo_hip_memcpy_d2d(void* d_dst, const void* d_src, size_t bytes, void* stream) {
  // (hipMemcpyDefault: either side may also be device-mapped pinned host memory, e.g. an output block of a hybrid arena)
  SVO_HIP_TRY(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDefault, static_cast<hipStream_t>(stream)));
  return SVO_HIP_OK;
}

int svo_hip_memset(void* d_dst, int value, size_t bytes, void* stream) {
  SVO_HIP_TRY(hipMemsetAsync(d_dst, value, bytes, static_cast<hipStream_t>(stream)));
  return SVO_HIP_OK;
}

int svo_hip_stream_create(void** stream_out) {
  if (!stream_out) return SVO_HIP_EINVAL;
  hipStream_t s;
  SVO_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream_out = s;
  return SVO_HIP_OK;
}

int svo_hip_stream_destroy(void* stream) {
  SVO_HIP_TRY(hipStreamDestroy(static_cast<hipStream_t>(stream)));
  return SVO_HIP_OK;
}

int svo_hip_stream_sync(void* stream) {
  SVO_HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  return SVO_HIP_OK;
}

int svo_hip_event_create(void** event_out) {
  if (!event_out) return SVO_HIP_EINVAL;
  hipEvent_t e;
  SVO_HIP_TRY(hipEventCreate(&e));
  *event_out = e;
  return SVO_HIP_OK;
}

int svo_hip_event_destroy(void* event) {
  SVO_HIP_TRY(hipEventDestroy(static_cast<hipEvent_t>(event)));
  return SVO_HIP_OK;
}

int svo_hip_event_record(void* event, void* stream) {
  SVO_HIP_TRY(hipEventRecord(static_cast<hipEvent_t>(event), static_cast<hipStream_t>(stream)));
  return SVO_HIP_OK;
}

int svo_hip_event_sync(void* event) {
  SVO_HIP_TRY(hipEventSynchronize(static_cast<hipEvent_t>(event)));
  return SVO_HIP_OK;
}

int svo_hip_event_query(void* event) {
  const hipError_t e = hipEventQuery(static_cast<hipEvent_t>(event));
  if (e == hipSuccess) return 1;
  if (e == hipErrorNotReady) {
    (void)hipGetLastError();  // not an error: the work before the event is still running
    return 0;
  }
  SVO_HIP_TRY(e);
  return SVO_HIP_OK;
}

int svo_hip_stream_wait_event(void* stream, void* event) {
  SVO_HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(event), 0));
  return SVO_HIP_OK;
}

int svo_hip_event_elapsed_ms(void* start, void* stop, float* ms_out) {
  if (!ms_out) return SVO_HIP_EINVAL;
  SVO_HIP_TRY(hipEventSynchronize(static_cast<hipEvent_t>(stop)));
  SVO_HIP_TRY(hipEventElapsedTime(ms_out, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)));
  return SVO_HIP_OK;
}

// ---- HIP graphs: capture a fixed chain of entry-point calls once, replay it per step ----------
int svo_hip_graph_begin_capture(void* stream) {
  SVO_HIP_TRY(hipStreamBeginCapture(static_cast<hipStream_t>(stream), hipStreamCaptureModeThreadLocal));
  return SVO_HIP_OK;
}

int svo_hip_graph_end_capture(void* stream, void** graph_exec_out) {
  if (!graph_exec_out) return SVO_HIP_EINVAL;
  hipGraph_t g = nullptr;
  SVO_HIP_TRY(hipStreamEndCapture(static_cast<hipStream_t>(stream), &g));
  hipGraphExec_t e = nullptr;
  hipError_t rc = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  SVO_HIP_TRY(rc);
  *graph_exec_out = e;
  return SVO_HIP_OK;
}

int svo_hip_graph_launch(void* graph_exec, void* stream) {
  SVO_HIP_TRY(hipGraphLaunch(static_cast<hipGraphExec_t>(graph_exec), static_cast<hipStream_t>(stream)));
  return SVO_HIP_OK;
}

int svo_hip_graph_destroy(void* graph_exec) {
  SVO_HIP_TRY(hipGraphExecDestroy(static_cast<hipGraphExec_t>(graph_exec)));
  return SVO_HIP_OK;
}

// vk::PinholeCamera::PinholeCamera (vikit pinhole_camera.cpp): distortion_ = fabs(d0) > 0.0000001
int svo_hip_camera_pinhole(int width, int height, double fx, double fy, double cx, double cy, double d0, double d1,
                           double d2, double d3, double d4, svo_hip_camera* out) {
  if (!out || width < 1 || height < 1) return SVO_HIP_EINVAL;
  out->fx = fx; out->fy = fy; out->cx = cx; out->cy = cy;
  out->width = width; out->height = height;
  out->reserved = 0;
  const bool distortion = fabs(d0) > 0.0000001;
  out->model = distortion ? SVO_HIP_CAM_PINHOLE_RADTAN : SVO_HIP_CAM_PINHOLE;
  const double d[5] = {d0, d1, d2, d3, d4};
  for (int i = 0; i < 5; ++i) out->d[i] = distortion ? d[i] : 0.0;
  return SVO_HIP_OK;
}

// vk::ATANCamera::ATANCamera (vikit atan_camera.cpp): fx_ = width*fx, cx_ = cx*width - 0.5, ...;
// s != 0: tans_ = 2 tan(s/2), tans_inv_ = 1/tans_, s_inv_ = 1/s
int svo_hip_camera_atan(int width, int height, double fx, double fy, double cx, double cy, double s,
                        svo_hip_camera* out) {
  if (!out || width < 1 || height < 1) return SVO_HIP_EINVAL;
  out->fx = (double)width * fx;
  out->fy = (double)height * fy;
  out->cx = cx * (double)width - 0.5;
  out->cy = cy * (double)height - 0.5;
  out->width = width; out->height = height;
  out->model = SVO_HIP_CAM_ATAN;
  out->reserved = 0;
  for (int i = 0; i < 5; ++i) out->d[i] = 0.0;
  if (s != 0.0) {
    const double tans = 2.0 * tan(s / 2.0);
    out->d[0] = s;
    out->d[1] = 1.0 / s;
    out->d[2] = tans;
    out->d[3] = 1.0 / tans;
  }
  return SVO_HIP_OK;
}

int svo_hip_pyr_layout_init(int width, int height, int n_levels, svo_hip_pyr_layout* out) {
  if (!out || width < 1 || height < 1 || n_levels < 1 || n_levels > SVO_HIP_MAX_LEVELS) return SVO_HIP_EINVAL;
  std::memset(out, 0, sizeof(*out));
  out->n_levels = n_levels;
  out->tile = SVO_HIP_PYR_TILED;
  int w = width, h = height;
  int64_t off = 0;
  for (int i = 0; i < n_levels; ++i) {
    if (w < 1 || h < 1) return SVO_HIP_EINVAL;  // pyramid deeper than the image allows
    out->w[i] = w;
    out->h[i] = h;
    out->pitch[i] = (w + 15) & ~15;  // 16-byte tile columns
    out->offset[i] = off;
    off += svo_pyr::level_bytes(out->pitch[i], h);
    off = (off + 255) & ~static_cast<int64_t>(255);
    w /= 2;  // cv::Mat(rows/2, cols/2), svo/src/frame.cpp:162
    h /= 2;
  }
  out->slot_bytes = (off + 4095) & ~static_cast<int64_t>(4095);
  return SVO_HIP_OK;
}

int64_t svo_hip_pyr_store_bytes(const svo_hip_pyr_layout* layout, int n_slots) {
  if (!layout_ok(layout) || n_slots < 0) return SVO_HIP_EINVAL;
  return layout->slot_bytes * n_slots + SVO_HIP_STORE_TAIL_PAD;
}

}  // extern "C"
