// solver_xcheck.hip -- the 6x6 Gauss-Newton solve through hipSOLVER (batched Cholesky), the form the
// north star names ("... J'J / J'r into a single hipSolver solve").
//
// The tracking kernels do NOT use it: a library call per Gauss-Newton iteration is a launch (and a
// host round trip for the stop / rollback rules) inside the innermost loop, whereas the persistent
// kernels solve the 36-number system in registers/LDS in well under a microsecond.  It is exported
// for (a) cross-checking the in-kernel solvers against an independent implementation
// (tests/test_hipsolver_xcheck_gpu.py) and (b) hosts that want x = H^-1 b or the covariance H^-1 for a
// batch of frames after the fact (SparseImgAlign::getFisherInformation consumers).
#include <hipsolver/hipsolver.h>

#include "capi_common.h"

using namespace svo_capi;

namespace {

__global__ void __launch_bounds__(256) xcheck_prepare_kernel(int B, const double* H, const double* b, double* A, double* x,
                                                            double** Ap, double** xp) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B) return;
  for (int k = 0; k < 36; ++k) A[36 * (size_t)i + k] = H[36 * (size_t)i + k];  // symmetric: row- == column-major
  for (int k = 0; k < 6; ++k) x[6 * (size_t)i + k] = b[6 * (size_t)i + k];
  Ap[i] = A + 36 * (size_t)i;
  xp[i] = x + 6 * (size_t)i;
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct Handle {
  hipsolverHandle_t h = nullptr;
  ~Handle() { if (h) hipsolverDestroy(h); }
};
thread_local Handle g_handle;

}  // namespace

extern "C" {

size_t svo_hip_solve6_hipsolver_workspace_bytes(int B) {
  if (B < 0) return 0;
  const size_t b = (size_t)B;
  // copy of H, two pointer arrays, info, and the library's own scratch (a few KB for n = 6)
  return align256(b * 36 * sizeof(double)) + 2 * align256(b * sizeof(double*)) + align256(b * sizeof(int)) + ((size_t)1 << 20);
}

int svo_hip_solve6_hipsolver(int B, const double* d_H, const double* d_b, double* d_x, int32_t* d_info, void* d_workspace,
                             size_t workspace_bytes, void* stream) {
  if (B < 0) return SVO_HIP_EINVAL;
  if (B == 0) return SVO_HIP_OK;
  if (!d_H || !d_b || !d_x || !d_workspace) return SVO_HIP_EINVAL;
  if (workspace_bytes < svo_hip_solve6_hipsolver_workspace_bytes(B)) return SVO_HIP_ERANGE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!g_handle.h && hipsolverCreate(&g_handle.h) != HIPSOLVER_STATUS_SUCCESS) return SVO_HIP_EHIP;
  if (hipsolverSetStream(g_handle.h, s) != HIPSOLVER_STATUS_SUCCESS) return SVO_HIP_EHIP;
  uint8_t* w = static_cast<uint8_t*>(d_workspace);
  double* A = reinterpret_cast<double*>(w);
  w += align256((size_t)B * 36 * sizeof(double));
  double** Ap = reinterpret_cast<double**>(w);
  w += align256((size_t)B * sizeof(double*));
  double** xp = reinterpret_cast<double**>(w);
  w += align256((size_t)B * sizeof(double*));
  int* info = reinterpret_cast<int*>(w);
  w += align256((size_t)B * sizeof(int));
  double* work = reinterpret_cast<double*>(w);
  hipLaunchKernelGGL(xcheck_prepare_kernel, dim3((B + 255) / 256), dim3(256), 0, s, B, d_H, d_b, A, d_x, Ap, xp);
  int rc = check_launch();
  if (rc) return rc;
  int lwork = 0;
  if (hipsolverDpotrfBatched_bufferSize(g_handle.h, HIPSOLVER_FILL_MODE_LOWER, 6, Ap, 6, &lwork, B) != HIPSOLVER_STATUS_SUCCESS)
    return SVO_HIP_EHIP;
  if ((size_t)lwork * sizeof(double) > ((size_t)1 << 20)) return SVO_HIP_ERANGE;
  if (hipsolverDpotrfBatched(g_handle.h, HIPSOLVER_FILL_MODE_LOWER, 6, Ap, 6, work, lwork, info, B) != HIPSOLVER_STATUS_SUCCESS)
    return SVO_HIP_EHIP;
  if (d_info) SVO_HIP_TRY(hipMemcpyAsync(d_info, info, (size_t)B * sizeof(int), hipMemcpyDeviceToDevice, s));
  int lwork2 = 0;
  if (hipsolverDpotrsBatched_bufferSize(g_handle.h, HIPSOLVER_FILL_MODE_LOWER, 6, 1, Ap, 6, xp, 6, &lwork2, B) != HIPSOLVER_STATUS_SUCCESS)
    return SVO_HIP_EHIP;
  if ((size_t)lwork2 * sizeof(double) > ((size_t)1 << 20)) return SVO_HIP_ERANGE;
  if (hipsolverDpotrsBatched(g_handle.h, HIPSOLVER_FILL_MODE_LOWER, 6, 1, Ap, 6, xp, 6, work, lwork2, info, B) != HIPSOLVER_STATUS_SUCCESS)
    return SVO_HIP_EHIP;
  return SVO_HIP_OK;
}

}  // extern "C"
