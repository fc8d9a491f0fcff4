/*
 * svo_hip.h -- C ABI of libsvo_hip.so, the MI355X (gfx950) implementation of
 * SVO's tracking hot path.  Plain pointers and sizes only; no C++/torch types.
 *
 * The reference (uzh-rpg/rpg_svo) has no FFI: the seam is a set of C++ methods
 * called by svo::FrameHandlerMono::processFrame (svo/src/frame_handler_mono.cpp
 * :129-235).  Each entry point below replaces the arithmetic behind one of
 * those methods; the API-compatible host classes that marshal into these calls
 * live in rpg_svo_amd/host/ (C++) and the rpg_svo_amd Python modules (mirror).
 *
 * Conventions
 *  - every function returns 0 on success or a negative SVO_HIP_E* code;
 *    svo_hip_strerror() gives text, svo_hip_last_hip_error() the raw HIP code.
 *  - pointers named d_* are DEVICE pointers (HBM); others are host pointers.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); all
 *    *_async style work is enqueued on it and NOT synchronised.
 *  - poses are 12 doubles: R row-major [9] followed by t [3]  (T = [R|t]).
 *  - image pyramids live in a "pyramid store": n_slots equally sized slots,
 *    one slot per frame, levels at fixed byte offsets with 64-byte-aligned
 *    row pitch (svo_hip_pyr_layout).  The store needs SVO_HIP_STORE_TAIL_PAD
 *    readable bytes after the last slot.
 */
#ifndef SVO_HIP_H_
#define SVO_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVO_HIP_MAX_LEVELS 8
#define SVO_HIP_STORE_TAIL_PAD 256
#define SVO_HIP_MAX_PATCHES 1024 /* per frame, sparse align (one patch per lane) */

/* error codes */
#define SVO_HIP_OK 0
#define SVO_HIP_EINVAL (-1)   /* bad argument                         */
#define SVO_HIP_ERANGE (-2)   /* size beyond a documented limit       */
#define SVO_HIP_EHIP (-3)     /* a HIP runtime call failed            */
#define SVO_HIP_ENODEV (-4)   /* no gfx950-compatible device visible  */
#define SVO_HIP_ENOMEM (-5)

/* vk::halfSample flavour (rpg_vikit vision.cpp), see svo_hip_pyramid_build */
#define SVO_HIP_HALFSAMPLE_SCALAR 0 /* (a+b+c+d)/4 truncating                    */
#define SVO_HIP_HALFSAMPLE_SSE2 1   /* avg_epu8 of rows, then avg_epu16 of cols  */
#define SVO_HIP_HALFSAMPLE_AUTO 2   /* x86 reference behaviour: SSE2 iff w%16==0 */

const char* svo_hip_strerror(int code);
int svo_hip_last_hip_error(void);
const char* svo_hip_version(void);
/* number of visible HIP devices (<0 on error) */
int svo_hip_device_count(void);

/* ---- raw device helpers for non-HIP host code (C++ host classes) -------- */
int svo_hip_set_device(int device);
int svo_hip_malloc(void** d_ptr, size_t bytes);
int svo_hip_free(void* d_ptr);
int svo_hip_memcpy_h2d(void* d_dst, const void* src, size_t bytes, void* stream);
int svo_hip_memcpy_d2h(void* dst, const void* d_src, size_t bytes, void* stream);
int svo_hip_memset(void* d_dst, int value, size_t bytes, void* stream);
int svo_hip_stream_create(void** stream_out);
int svo_hip_stream_destroy(void* stream);
int svo_hip_stream_sync(void* stream);
/* HIP-event timing on an arbitrary stream (used by bench.py) */
int svo_hip_event_create(void** event_out);
int svo_hip_event_destroy(void* event);
int svo_hip_event_record(void* event, void* stream);
int svo_hip_event_elapsed_ms(void* start, void* stop, float* ms_out); /* syncs on stop */

/* ---- pyramid store ------------------------------------------------------ */
typedef struct svo_hip_pyr_layout {
  int32_t n_levels;
  int32_t w[SVO_HIP_MAX_LEVELS];
  int32_t h[SVO_HIP_MAX_LEVELS];
  int32_t pitch[SVO_HIP_MAX_LEVELS];  /* bytes per row, multiple of 64          */
  int64_t offset[SVO_HIP_MAX_LEVELS]; /* byte offset of the level inside a slot */
  int64_t slot_bytes;                 /* distance between consecutive slots     */
} svo_hip_pyr_layout;

/* Level sizes follow frame_utils::createImgPyramid (svo/src/frame.cpp:156-165):
 * level i is (rows/2, cols/2) of level i-1.  Host-only, no GPU needed. */
int svo_hip_pyr_layout_init(int width, int height, int n_levels, svo_hip_pyr_layout* out);
/* bytes to allocate for n_slots (includes the tail pad) */
int64_t svo_hip_pyr_store_bytes(const svo_hip_pyr_layout* layout, int n_slots);

/* Copy n images (8-bit, `row_stride` bytes per row, `image_stride` bytes between
 * images; device memory) into level 0 of slots first_slot.. */
int svo_hip_pyramid_load_level0(const svo_hip_pyr_layout* layout, uint8_t* d_store, int first_slot,
                                int n_slots, const uint8_t* d_images, int64_t image_stride,
                                int row_stride, void* stream);
/* Same from host memory (pinned or pageable): H2D copy straight into level 0. */
int svo_hip_pyramid_upload_level0(const svo_hip_pyr_layout* layout, uint8_t* d_store, int slot,
                                  const uint8_t* image, int row_stride, void* stream);
/* K0: build levels 1..n_levels-1 of slots [first_slot, first_slot+n_slots) from
 * their level 0.  Replaces frame_utils::createImgPyramid -> vk::halfSample
 * (svo/src/frame.cpp:156-165).  Bit-exact with the selected flavour. */
int svo_hip_pyramid_build(const svo_hip_pyr_layout* layout, uint8_t* d_store, int first_slot,
                          int n_slots, int halfsample_mode, void* stream);
/* Download one level of one slot into a tightly packed host buffer (tests). */
int svo_hip_pyramid_download_level(const svo_hip_pyr_layout* layout, const uint8_t* d_store,
                                   int slot, int level, uint8_t* out, void* stream);

/* ---- K1: sparse image alignment ----------------------------------------- */
typedef struct svo_hip_sia_params {
  double fx, fy, cx, cy; /* vk::PinholeCamera without distortion            */
  int32_t max_level;     /* SparseImgAlign ctor, sparse_img_align.cpp:29-41 */
  int32_t min_level;
  int32_t n_iter;        /* 30 in the pipeline, frame_handler_mono.cpp:136  */
  int32_t reserved;
  double eps;            /* 1e-6, sparse_img_align.cpp:40                   */
} svo_hip_sia_params;

/* status bits written per problem */
#define SVO_HIP_SIA_STOP 1 /* vk::NLLSSolver::stop_ (solve produced NaN) */

/*
 * Batched SparseImgAlign::run (svo/src/sparse_img_align.cpp:43-75) for B
 * independent (reference frame, current frame) problems; one workgroup per
 * problem, one lane per 4x4 patch.
 *
 *   d_ref_slot/d_cur_slot [B]   pyramid-store slots of the two frames
 *   d_n [B]                     features of problem b (<= n_stride <= SVO_HIP_MAX_PATCHES)
 *   d_px      [B][n_stride][2]  Feature::px  (level-0 pixels, f64)
 *   d_xyz_ref [B][n_stride][3]  Feature::f * |point.pos - ref.pos()| (f64),
 *                               i.e. sparse_img_align.cpp:107-108
 *   d_valid   [B][n_stride]     0 where Feature::point == NULL (may be NULL = all 1)
 *   d_T_in    [B][12]           T_cur_from_ref prior  (cur.T_f_w * ref.T_f_w^-1, :59)
 *   d_T_out   [B][12]           optimised T_cur_from_ref
 *   d_H_out   [B][36]           H_ of the last evaluated iteration (getFisherInformation, :77-82); may be NULL
 *   d_n_tracked [B]             n_meas_/patch_area_  (:74)
 *   d_iters   [B][SVO_HIP_MAX_LEVELS]  residual evaluations per level; may be NULL
 *   d_chi2    [B]               chi2_ after the run; may be NULL
 *   d_status  [B]               SVO_HIP_SIA_* bits; may be NULL
 */
int svo_hip_sparse_align(const svo_hip_pyr_layout* layout, const uint8_t* d_store, int B,
                         const int32_t* d_ref_slot, const int32_t* d_cur_slot, const int32_t* d_n,
                         int n_stride, const double* d_px, const double* d_xyz_ref,
                         const uint8_t* d_valid, const svo_hip_sia_params* params,
                         const double* d_T_in, double* d_T_out, double* d_H_out,
                         int32_t* d_n_tracked, int32_t* d_iters, double* d_chi2,
                         int32_t* d_status, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SVO_HIP_H_ */
