/*
 * svo_hip.h -- C ABI of libsvo_hip.so, the MI355X (gfx950) implementation of
 * SVO's tracking hot path.  Plain pointers and sizes only; no C++/torch types.
 *
 * The reference (uzh-rpg/rpg_svo) has no FFI: the seam is a set of C++ methods
 * called by svo::FrameHandlerMono::processFrame (svo/src/frame_handler_mono.cpp
 * :129-235).  Each entry point below replaces the arithmetic behind one of
 * those methods; the API-compatible host classes that marshal into these calls
 * live in rpg_svo_amd/host/ (C++: svo_hip_device + the dropin/ sources, bodies for the
 * reference's own classes) and the rpg_svo_amd Python modules (batched mirror).
 *
 * Conventions
 *  - every function returns 0 on success or a negative SVO_HIP_E* code;
 *    svo_hip_strerror() gives text, svo_hip_last_hip_error() the raw HIP code.
 *  - pointers named d_* are DEVICE pointers (HBM); others are host pointers.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); all
 *    *_async style work is enqueued on it and NOT synchronised.
 *  - poses are 12 doubles: R row-major [9] followed by t [3]  (T = [R|t]).
 *  - image pyramids live in a "pyramid store": n_slots equally sized slots,
 *    one slot per frame, levels at fixed byte offsets (svo_hip_pyr_layout), each
 *    level cut into tiles of 16 bytes x 8 rows = one 128-byte line (the readers
 *    gather small 2-D windows).  The store is filled and read through the entry
 *    points below only; it needs SVO_HIP_STORE_TAIL_PAD readable bytes after the
 *    last slot.
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

/* Binds the CALLING THREAD to the CPUs next to the current device (the PCI function's local_cpulist in sysfs, intersected
 * with the thread's present affinity mask).  Returns the number of CPUs of the new mask, 0 when nothing was changed (no
 * NUMA information, nothing in common, already there), < 0 when the device could not be asked.  A single camera's frame
 * is a dozen host <-> device hand-overs through pinned memory: on a two-socket host the same 600-frame sequence takes
 * 0.263 ms per frame with the PROCESS bound to the GPU's own node (taskset), 0.268 on the other node and 0.280 wherever
 * the scheduler puts and moves it (profiles/r06aa_*).  Binding only the calling thread -- what this function can do from
 * inside -- did not reproduce that (profiles/r06ab_*: the runtime's own threads stay where they are), so nothing in the
 * library calls it by default (the drop-in's host layer does with SVO_HIP_PIN_HOST=1): a host that wants the gain
 * starts its process under `numactl --cpunodebind` / `taskset`, or calls this first thing in main(). */
int svo_hip_pin_calling_thread(void);

/* ---- raw device helpers for non-HIP host code (C++ host classes) -------- */
int svo_hip_set_device(int device);
int svo_hip_malloc(void** d_ptr, size_t bytes);
int svo_hip_free(void* d_ptr);
/* page-locked host memory (staging buffers of the host classes: async copies need it) */
int svo_hip_host_alloc(void** ptr, size_t bytes);
int svo_hip_host_free(void* ptr);
int svo_hip_memcpy_h2d(void* d_dst, const void* src, size_t bytes, void* stream);
int svo_hip_memcpy_d2h(void* dst, const void* d_src, size_t bytes, void* stream);
int svo_hip_memcpy_d2d(void* d_dst, const void* d_src, size_t bytes, void* stream);
int svo_hip_memset(void* d_dst, int value, size_t bytes, void* stream);
int svo_hip_stream_create(void** stream_out);
int svo_hip_stream_destroy(void* stream);
int svo_hip_stream_sync(void* stream);
/* HIP-event timing on an arbitrary stream (used by bench.py) */
int svo_hip_event_create(void** event_out);
int svo_hip_event_destroy(void* event);
int svo_hip_event_record(void* event, void* stream);
int svo_hip_event_elapsed_ms(void* start, void* stop, float* ms_out); /* syncs on stop */
int svo_hip_event_sync(void* event);                      /* host waits for the work recorded before `event` */
int svo_hip_event_query(void* event);                     /* 1: that work is done, 0: still running, <0: error */
int svo_hip_stream_wait_event(void* stream, void* event); /* later work of `stream` waits for it on the device */

/* HIP graphs: every entry point below only enqueues kernels / memsets on `stream`, so a fixed
 * chain of calls (same pointers, same sizes: e.g. one tracked frame of every camera of a rig) can
 * be captured once and replayed with a single launch per step -- for small batches the chain is
 * launch-bound.  begin_capture .. <entry-point calls on `stream`> .. end_capture, then launch. */
int svo_hip_graph_begin_capture(void* stream);
int svo_hip_graph_end_capture(void* stream, void** graph_exec_out);
int svo_hip_graph_launch(void* graph_exec, void* stream);
int svo_hip_graph_destroy(void* graph_exec);

/* ---- pyramid store ------------------------------------------------------ */
typedef struct svo_hip_pyr_layout {
  int32_t n_levels;
  int32_t w[SVO_HIP_MAX_LEVELS];
  int32_t h[SVO_HIP_MAX_LEVELS];
  int32_t pitch[SVO_HIP_MAX_LEVELS];  /* bytes per image row: w rounded up to 16 */
  int32_t tile;                       /* SVO_HIP_PYR_*: how (x, y) maps to bytes  */
  int64_t offset[SVO_HIP_MAX_LEVELS]; /* byte offset of the level inside a slot   */
  int64_t slot_bytes;                 /* distance between consecutive slots       */
} svo_hip_pyr_layout;

/* svo_hip_pyr_layout::tile.  TILED (what svo_hip_pyr_layout_init produces): byte offset of pixel (x, y) inside
 * its level = (y / 8) * 8 * pitch + (x / 16) * 128 + (y % 8) * 16 + x % 16, i.e. 16 x 8 pixel tiles of one
 * 128-byte line each, the tiles of an 8-row band consecutive; a level occupies pitch * roundup8(h) bytes.
 * ROWMAJOR (y * pitch + x) names the layout of rounds 1-2, which no kernel of this library addresses any more: every
 * entry point rejects a layout that is not TILED with SVO_HIP_EINVAL. */
#define SVO_HIP_PYR_ROWMAJOR 0
#define SVO_HIP_PYR_TILED 1

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
/* Same from host memory (pinned or pageable).  The store is tiled, so the image is copied H2D into packed device
 * scratch and re-tiled from there by a kernel: d_staging = w[level]*h[level] bytes of device memory owned by the
 * caller that nothing else touches until `stream` has passed this call (one buffer per stream, reused call after
 * call), or NULL: a stream-ordered temporary (hipMallocAsync / hipFreeAsync) per call. */
int svo_hip_pyramid_upload_level0(const svo_hip_pyr_layout* layout, uint8_t* d_store, int slot,
                                  const uint8_t* image, int row_stride, void* d_staging, void* stream);
/* One pyramid level of one slot from host memory (an image that is not level 0 of a frame, e.g. the
 * cv::Mat a direct caller hands to feature_alignment::align2D). */
int svo_hip_pyramid_upload_level(const svo_hip_pyr_layout* layout, uint8_t* d_store, int slot, int level,
                                 const uint8_t* image, int row_stride, void* d_staging, void* stream);
/* A new camera frame in one go (Frame::initFrame, svo/src/frame.cpp:48-59: the image and its pyramid): H2D copy
 * into the scratch, then ONE kernel that writes level 0 and every further level of the slot. */
int svo_hip_pyramid_upload_build(const svo_hip_pyr_layout* layout, uint8_t* d_store, int slot, const uint8_t* image,
                                 int row_stride, int halfsample_mode, void* d_staging, void* stream);
/* K0: build levels 1..n_levels-1 of slots [first_slot, first_slot+n_slots) from
 * their level 0.  Replaces frame_utils::createImgPyramid -> vk::halfSample
 * (svo/src/frame.cpp:156-165).  Bit-exact with the selected flavour. */
int svo_hip_pyramid_build(const svo_hip_pyr_layout* layout, uint8_t* d_store, int first_slot,
                          int n_slots, int halfsample_mode, void* stream);
/* N1: level 0 filled from packed device images AND every further level built in
 * the same single pass over the pixels (one read of the source, 1.33 B written per pixel). */
int svo_hip_pyramid_build_from_images(const svo_hip_pyr_layout* layout, uint8_t* d_store, int first_slot,
                                      int n_slots, const uint8_t* d_images, int64_t image_stride,
                                      int row_stride, int halfsample_mode, void* stream);
/* The two builders above with the level-0 tile of the fused kernel chosen by the caller instead of
 * by image size: 128 (128x64), 256 (256x32), 257 (256x32 with non-temporal source loads), 512
 * (256x64, two row blocks per lane), 0 = automatic (257 for widths >= 256).  d_images may be NULL
 * (level 0 already in the store).  Results do not depend on the tile; there is no global state. */
int svo_hip_pyramid_build_tiled(const svo_hip_pyr_layout* layout, uint8_t* d_store, int first_slot, int n_slots,
                                const uint8_t* d_images, int64_t image_stride, int row_stride,
                                int halfsample_mode, int tile_width, void* stream);
/* The one-launch-per-level builder svo_hip_pyramid_build used before the fused kernel; same
 * results, kept for A/B timing. */
int svo_hip_pyramid_build_per_level(const svo_hip_pyr_layout* layout, uint8_t* d_store, int first_slot,
                                    int n_slots, int halfsample_mode, void* stream);
/* Download one level of one slot into a tightly packed host buffer (tests). */
int svo_hip_pyramid_download_level(const svo_hip_pyr_layout* layout, const uint8_t* d_store,
                                   int slot, int level, uint8_t* out, void* stream);

/* ---- K1: sparse image alignment ----------------------------------------- */
/* camera models: the vk::AbstractCamera implementations rpg_vikit ships and SVO's launch files use
 * (svo_ros/param/camera_pinhole.yaml, camera_atan.yaml) */
#define SVO_HIP_CAM_PINHOLE 0        /* vk::PinholeCamera, all distortion coefficients zero            */
#define SVO_HIP_CAM_PINHOLE_RADTAN 1 /* vk::PinholeCamera, d = {d0, d1, d2, d3, d4} = k1 k2 p1 p2 k3    */
#define SVO_HIP_CAM_ATAN 2           /* vk::ATANCamera; d = {s, 1/s, 2 tan(s/2), 1/(2 tan(s/2)), 0} and
                                        fx, fy, cx, cy hold the constructor's fx_, fy_, cx_, cy_
                                        (width*fx, height*fy, cx*width - 0.5, cy*height - 0.5): fill
                                        the struct with svo_hip_camera_atan()                         */

typedef struct svo_hip_sia_params {
  double fx, fy, cx, cy; /* camera intrinsics (see svo_hip_camera)                  */
  int32_t max_level;     /* SparseImgAlign ctor, sparse_img_align.cpp:29-41         */
  int32_t min_level;
  int32_t n_iter;        /* 30 in the pipeline, frame_handler_mono.cpp:136          */
  int32_t cam_model;     /* SVO_HIP_CAM_*: world2cam of sparse_img_align.cpp:183    */
  double eps;            /* 1e-6, sparse_img_align.cpp:40                           */
  double d[5];           /* distortion parameters of cam_model (see svo_hip_camera) */
} svo_hip_sia_params;

/* status bits written per problem */
#define SVO_HIP_SIA_STOP 1 /* vk::NLLSSolver::stop_ (solve produced NaN) */
#define SVO_HIP_SIA_EXCHANGE_TIMEOUT 2 /* a frame split over four workgroups (frames of > 512 patches in batches of <= 128):
                                        * a part never heard from its siblings -- the GPU was too busy with other work for
                                        * the frame's workgroups to be resident together; the frame ends like a NaN solve.
                                        * SVO_HIP_K1_SPLIT=0 keeps every frame on one workgroup */

/*
 * Batched SparseImgAlign::run (svo/src/sparse_img_align.cpp:43-75) for B
 * independent (reference frame, current frame) problems.  One workgroup per
 * problem, one lane per 4x4 patch (sparse_align.hip); batches of >= 1024 problems
 * with n_stride <= 192 (<= 64 under a distorted camera model) run one WAVE per
 * problem, a lane carrying up to three patches (sparse_align_wave.hip).  Same arithmetic per patch; the order of the
 * (tolerance-mode, f32) sums differs between the two.
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

/* The workgroup-per-problem kernel whatever B and n_stride, exported so tests and the bench can put both
 * kernels on the same problems.  Arguments as svo_hip_sparse_align. */
int svo_hip_sparse_align_workgroup(const svo_hip_pyr_layout* layout, const uint8_t* d_store, int B,
                                   const int32_t* d_ref_slot, const int32_t* d_cur_slot, const int32_t* d_n,
                                   int n_stride, const double* d_px, const double* d_xyz_ref,
                                   const uint8_t* d_valid, const svo_hip_sia_params* params,
                                   const double* d_T_in, double* d_T_out, double* d_H_out,
                                   int32_t* d_n_tracked, int32_t* d_iters, double* d_chi2,
                                   int32_t* d_status, void* stream);

/* The 6x6 solve x = H^-1 b of a batch of problems through hipSOLVER (batched Cholesky: potrf +
 * potrs), e.g. on the H_out of svo_hip_sparse_align.  NOT used by the kernels (they solve in
 * registers/LDS inside the persistent loop); exported as an independent cross-check of those
 * solvers and for hosts that want covariances after the fact.  d_info [B]: potrf status per
 * problem (0 = positive definite), may be NULL. */
size_t svo_hip_solve6_hipsolver_workspace_bytes(int B);
int svo_hip_solve6_hipsolver(int B, const double* d_H, const double* d_b, double* d_x, int32_t* d_info,
                             void* d_workspace, size_t workspace_bytes, void* stream);

/* ======================================================================== */
/* Rows a8-a13: the steps FrameHandlerMono::processFrame runs after sparse   */
/* alignment (svo/src/frame_handler_mono.cpp:145-235) and the depth-filter    */
/* update of the mapping thread.  All arrays are device pointers.            */
/* ======================================================================== */

typedef struct svo_hip_camera { /* a vk::AbstractCamera: world2cam / cam2world / errorMultiplier2 = |fx| */
  double fx, fy, cx, cy;
  int32_t width, height;
  int32_t model; /* SVO_HIP_CAM_* */
  int32_t reserved;
  double d[5];   /* distortion parameters of the model (zeros for SVO_HIP_CAM_PINHOLE) */
} svo_hip_camera;

/* Host-only constructors mirroring the vikit ones (no GPU needed).
 *  svo_hip_camera_pinhole: vk::PinholeCamera(width, height, fx, fy, cx, cy, d0..d4); the model is
 *      SVO_HIP_CAM_PINHOLE when |d0| <= 1e-7 (vikit's `distortion_` test), else ..._RADTAN.
 *  svo_hip_camera_atan:    vk::ATANCamera(width, height, fx, fy, cx, cy, s) with the NORMALISED
 *      parameters of camera_atan.yaml. */
int svo_hip_camera_pinhole(int width, int height, double fx, double fy, double cx, double cy, double d0, double d1,
                           double d2, double d3, double d4, svo_hip_camera* out);
int svo_hip_camera_atan(int width, int height, double fx, double fy, double cx, double cy, double s,
                        svo_hip_camera* out);

#define SVO_HIP_FTR_CORNER 0  /* svo::Feature::FeatureType (feature.h:30-33) */
#define SVO_HIP_FTR_EDGELET 1

/* A batch refers to frames through a FRAME TABLE (device mirror of the svo::Frame objects
 * involved): d_frame_slot[F] = pyramid-store slot, d_frame_T[F][12] = Frame::T_f_w_. */
typedef struct svo_hip_frames {
  int32_t n_frames;
  int32_t reserved;
  const int32_t* d_slot;
  const double* d_T_f_w;
} svo_hip_frames;

/* A set of svo::Feature records, SoA (feature.h:26-71). */
typedef struct svo_hip_features {
  const int32_t* d_frame; /* [n] index into the frame table (Feature::frame) */
  const int32_t* d_level; /* [n] Feature::level                             */
  const uint8_t* d_type;  /* [n] SVO_HIP_FTR_*; NULL = all corners          */
  const double* d_px;     /* [n][2] Feature::px                             */
  const double* d_f;      /* [n][3] Feature::f                              */
  const double* d_grad;   /* [n][2] Feature::grad; may be NULL when d_type is NULL */
} svo_hip_features;

/*
 * K3: batched feature_alignment::align2D / align1D (svo/src/feature_alignment.cpp:30-277,
 * the float paths an x86 build runs).  One lane per trial, pixels visited in the
 * reference's order, no contraction: refined pixel and verdict carry the reference's bits
 * (checked against the reference's own translation unit, tests/test_tracking_gpu.py).
 *   d_slot/d_level [M]   image = level d_level[t] of pyramid slot d_slot[t]
 *   d_patch_with_border [M][100]  Matcher::patch_with_border_ (10x10 u8, row-major); the
 *                        8x8 ref_patch is its interior (Matcher::createPatchFromPatchWithBorder)
 *   d_dir [M][2] f32     align1D direction; used where d_use_1d[t] != 0 (both may be NULL)
 *   d_px [M][2]          in: estimate, out: refined position, in pixels OF THAT LEVEL
 *   d_ok [M]             1 = converged (the functions' bool result)
 *   d_h_inv [M]          align1D's h_inv (may be NULL)
 */
int svo_hip_align_batch(const svo_hip_pyr_layout* layout, const uint8_t* d_store, int M,
                        const int32_t* d_slot, const int32_t* d_level,
                        const uint8_t* d_patch_with_border, const float* d_dir,
                        const uint8_t* d_use_1d, int n_iter, double* d_px, int32_t* d_ok,
                        double* d_h_inv, void* stream);

/* Same, plus d_evaluations [M]: residual evaluations (9x9 windows read) per trial.  Instrumented
 * variant for roofline accounting (bytes per trial = 100 + 81 * evaluations); results are
 * identical to svo_hip_align_batch. */
int svo_hip_align_batch_counted(const svo_hip_pyr_layout* layout, const uint8_t* d_store, int M,
                                const int32_t* d_slot, const int32_t* d_level,
                                const uint8_t* d_patch_with_border, const float* d_dir,
                                const uint8_t* d_use_1d, int n_iter, double* d_px, int32_t* d_ok,
                                double* d_h_inv, int32_t* d_evaluations, void* stream);

/* svo_hip_align_batch[_counted] in PHASES, for large batches.  A wave of the one-lane-per-trial kernel runs as long as
 * its slowest trial; given scratch, the iterations are run in three launches (0-2, 3-5, 6...) and the trials still
 * iterating are compacted in between, so later launches run dense waves.  Same results bit for bit (a resumed trial
 * continues from its parked loop state).  svo_hip_align_workspace_bytes(M) is 0 for batches too small for the extra
 * launches to pay: the call is then a single launch.  d_evaluations may be NULL.  svo_hip_find_match_direct and
 * svo_hip_update_seeds do the same inside their own workspace. */
size_t svo_hip_align_workspace_bytes(int M);
int svo_hip_align_batch_phased(const svo_hip_pyr_layout* layout, const uint8_t* d_store, int M,
                               const int32_t* d_slot, const int32_t* d_level,
                               const uint8_t* d_patch_with_border, const float* d_dir,
                               const uint8_t* d_use_1d, int n_iter, double* d_px, int32_t* d_ok,
                               double* d_h_inv, int32_t* d_evaluations, void* d_workspace,
                               size_t workspace_bytes, void* stream);

/* bytes of scratch the matcher / depth-filter entry points need for M trials */
size_t svo_hip_match_workspace_bytes(int M);

/*
 * K2+K3: batched Matcher::findMatchDirect (svo/src/matcher.cpp:135-177) including
 * Point::getCloseViewObs (svo/src/point.cpp:97-117), warp::getWarpMatrixAffine /
 * getBestSearchLevel / warpAffine (matcher.cpp:33-105) and the feature alignment.
 * Candidate m is a map point seen from frame d_cur_frame[m]:
 *   d_pt_pos [M][3]      Point::pos_
 *   d_obs_ptr [M+1]      CSR offsets into `obs`: Point::obs_ of candidate m, in list order
 *   d_px_cur [M][2]      in: Candidate::px (projection, level-0 pixels); out: refined px
 *   d_ok [M]             findMatchDirect's result
 *   d_ref_obs [M]        index (into obs) of the observation chosen as ref_ftr_ (-1: none)
 *   d_search_level [M]   Matcher::search_level_
 *   d_A_cur_ref [M][4]   Matcher::A_cur_ref_, row-major 2x2 (may be NULL)
 *   d_patch_out [M][100] Matcher::patch_with_border_ (may be NULL)
 *   n_pyr_levels         Config::nPyrLevels(): search level <= n_pyr_levels-1
 *   align_max_iter       Matcher::Options::align_max_iter (10)
 */
int svo_hip_find_match_direct(const svo_hip_pyr_layout* layout, const uint8_t* d_store,
                              const svo_hip_camera* cam, const svo_hip_frames* frames, int M,
                              const int32_t* d_cur_frame, const double* d_pt_pos,
                              const int32_t* d_obs_ptr, const svo_hip_features* obs,
                              int n_pyr_levels, int align_max_iter, double* d_px_cur,
                              int32_t* d_ok, int32_t* d_ref_obs, int32_t* d_search_level,
                              double* d_A_cur_ref, uint8_t* d_patch_out, void* d_workspace,
                              size_t workspace_bytes, void* stream);

/* Reprojector::reprojectPoint (svo/src/reprojector.cpp:206-217): d_cell[m] = grid cell of the
 * projection of point m into frame d_cur_frame[m] (or -1 when outside the 8 px border),
 * d_px[m] = the projection. */
int svo_hip_reproject_points(const svo_hip_camera* cam, const svo_hip_frames* frames, int M,
                             const int32_t* d_cur_frame, const double* d_pt_pos, int cell_size,
                             int grid_n_cols, int32_t* d_cell, double* d_px, void* stream);

/* The cell loop of Reprojector::reprojectMap (svo/src/reprojector.cpp:131-139) with reprojectCell's "first success
 * of the cell" (:150-200), applied on the device to the results of a batch of findMatchDirect trials, so that pose
 * refinement can be enqueued behind the match kernels without the host in between.
 *   The M trials are given in the order the reference visits them: cells in grid_.cell_order, the candidates of a
 *   cell in the order of its sorted list; d_cell[m] identifies the cell (trials of one cell are adjacent), d_ok /
 *   d_px / d_level are the outputs of svo_hip_find_match_direct, d_pos the points' positions.
 *   A trial is selected when it matched and no earlier trial of its cell did; selection ends after max_fts + 1
 *   selected trials (:137-138).  Selected trial number i (its position in Frame::fts_) yields the observation
 *   pose_optimizer::optimizeGaussNewton reads from the new Feature (reprojector.cpp:182-187, feature.h:44-52):
 *     d_sel[i] = trial index, d_f[i] = cam2world(px), d_level_out[i], d_pos_out[i], d_has_point[i] = 1;
 *   d_n[0] = number selected.  Output arrays hold min(M, max_fts + 1) entries.
 *   d_signal (may be NULL): *d_signal = signal_value is stored (release, system scope) as the kernel starts, i.e. once
 *   everything enqueued on the stream before this call -- the match kernels -- has completed.  With d_signal and the
 *   match outputs in host-mapped pinned memory a host thread can poll it instead of synchronising the stream, and
 *   read the match results while this call and the pose refinement behind it are still running. */
int svo_hip_select_matches(const svo_hip_camera* cam, int M, const int32_t* d_cell, const int32_t* d_ok,
                           const double* d_px, const int32_t* d_level, const double* d_pos, int max_fts,
                           int32_t* d_n, int32_t* d_sel, double* d_f, int32_t* d_level_out, double* d_pos_out,
                           uint8_t* d_has_point, int32_t* d_signal, int32_t signal_value, void* stream);

/* ---- Row N2: a device-resident mirror of the map for Reprojector::reprojectMap ------------------------------------
 * reprojectMap (svo/src/reprojector.cpp:64-142) walks the pointer graph Map -> keyframes -> Frame::fts_ -> Point (and
 * MapPointCandidates::candidates_) on the host for every frame: ~2000 reprojectPoint calls into std::list cells, a
 * list sort per cell, Point::getCloseViewObs per candidate.  The mirror keeps what that walk reads resident in HBM as
 * SoA -- svo::Point records (point.h:35-62) with their observation lists, svo::Feature records (feature.h:26-71) --
 * updated incrementally by the host (svo_hip_map_patch: the entries that changed since the last call), so that one
 * kernel does the walk: projection, grid binning, the per-cell order, the close-view test and the list of
 * findMatchDirect trials in visiting order, and the match kernels follow on the stream without the host in between.
 * All arrays are device memory owned by the caller. */
typedef struct svo_hip_map {
  int32_t n_points;      /* entries [0, n_points), <= 8192: the points of the map's keyframes and the depth filter's candidates.
                          * (API CHANGE in round 5: the limit was 16384 up to round 4; a larger map now gets SVO_HIP_ERANGE here --
                          * the drop-in's Reprojector then takes its list-walking path, other callers split the map or keep the
                          * keyframe window below ~60 keyframes x 120 features.) */
  int32_t n_obs;         /* observation records [0, n_obs) */
  double* d_pos;         /* [P][3] Point::pos_ */
  int32_t* d_type;       /* [P] Point::type_ (point.h:38-43): 0 deleted = the entry is dead, 1 candidate, 2 unknown, 3 good */
  int32_t* d_order;      /* [P] candidates: position in MapPointCandidates::candidates_ (any key increasing along the list,
                                < 65536); other points: unused */
  int32_t* d_obs_begin;  /* [P] Point::obs_ in list order = records [d_obs_begin[p], d_obs_begin[p] + d_obs_count[p]) */
  int32_t* d_obs_count;  /* [P] */
  int32_t* d_obs_frame;  /* [O] Feature::frame as an index into the frame table of the call */
  int32_t* d_obs_order;  /* [O] written by the patch step as Feature::frame << 16 | position of the Feature in its frame's
                                fts_ list (< 4096; 0xffff: in no keyframe's list -- the Feature of a candidate,
                                map.cpp:215-218); the patch itself carries the position, -1 for "in no list" */
  int32_t* d_obs_level;  /* [O] Feature::level */
  uint8_t* d_obs_type;   /* [O] SVO_HIP_FTR_* */
  double* d_obs_px;      /* [O][2] Feature::px */
  double* d_obs_f;       /* [O][3] Feature::f */
  double* d_obs_grad;    /* [O][2] Feature::grad */
} svo_hip_map;

/* Entries the host rewrites before the map is read (applied by svo_hip_reproject_map itself, on its stream): whole
 * point records, whole observation records.  n = 0 with NULL arrays is "no change". */
typedef struct svo_hip_map_patch {
  int32_t n_points;
  int32_t n_obs;
  const int32_t* d_index;      /* [n_points] entry written */
  const double* d_pos;         /* [n_points][3] */
  const int32_t* d_type;       /* [n_points] */
  const int32_t* d_order;      /* [n_points] */
  const int32_t* d_obs_begin;  /* [n_points] */
  const int32_t* d_obs_count;  /* [n_points] */
  const int32_t* d_obs_index;  /* [n_obs] record written */
  const int32_t* d_obs_order;  /* [n_obs] */
  svo_hip_features obs;        /* [n_obs] frame (index into the frame table), level, type, px, f, grad: none NULL */
} svo_hip_map_patch;

/* Reprojector::Grid (reprojector.h:79-86): d_cell_rank[k] = position of cell k in grid_.cell_order, the order
 * reprojectMap visits the cells in (:131-139). */
typedef struct svo_hip_grid {
  int32_t cell_size, n_cols, n_rows, n_cells;
  const int32_t* d_cell_rank; /* [n_cells] */
} svo_hip_grid;

#define SVO_HIP_REPROJ_MAX_IN_FRAME 4096 /* points inside the frame one call can order (status 1 beyond) */
#define SVO_HIP_REPROJ_MAX_CELLS 2048
#define SVO_HIP_REPROJ_HEADER 8
typedef struct svo_hip_reprojection {
  int32_t* d_header;        /* [SVO_HIP_REPROJ_HEADER]: 0 status (0 ok; 1 a capacity was exceeded: nothing below is valid),
                               1 points inside the frame, 2 V = visits, 3 M = trials, 4 end_cell */
  int32_t* d_point_cell;    /* [P] grid cell of the projection (reprojectPoint, :206-217); -1: outside the frame;
                               -2: not projected (dead entry, or none of the point's keyframes is among the overlapping ones) */
  double* d_point_px;       /* [P][2] the projection (defined where d_point_cell >= -1) */
  int32_t* d_kf_count;      /* [n_frames] overlap_kfs[i].second (:87-102): points first met through that keyframe that
                               fell inside the frame */
  /* the candidates of cells [first_cell, end_cell) of the visiting order, in the order reprojectCell walks them
     (:151-153: per cell good before unknown before candidate points, otherwise in the order they were binned) */
  int32_t* d_visit_point;   /* [max_visits] entry in the map */
  int32_t* d_visit_cell;    /* [max_visits] position of its cell in the visiting order */
  int32_t* d_visit_trial;   /* [max_visits] its findMatchDirect trial, or -1: no close view, findMatchDirect returns false at once
                               (matcher.cpp:137-138) */
  /* the trials: the inputs of svo_hip_find_match_direct_indirect / svo_hip_select_matches_indirect */
  int32_t* d_trial_cur;       /* [max_trials] = cur_frame */
  double* d_trial_pos;        /* [max_trials][3] */
  int32_t* d_trial_obs_begin; /* [max_trials] the observation Point::getCloseViewObs chose ... */
  int32_t* d_trial_obs_end;   /* [max_trials] ... + 1 */
  int32_t* d_trial_cell;      /* [max_trials] = d_visit_cell of the visit */
  double* d_trial_px;         /* [max_trials][2] Candidate::px: the projection (in), refined by the match kernels (out) */
} svo_hip_reprojection;

/* Reprojector::reprojectMap up to the first findMatchDirect (svo/src/reprojector.cpp:64-142, 151-153, 206-217;
 * Point::getCloseViewObs, svo/src/point.cpp:97-117) for ONE frame, on the mirror:
 *   - applies `patch` (may be NULL);
 *   - every live map point (type >= 2) that has an observation in a keyframe with d_kf_rank[frame] >= 0 is projected
 *     into frame `cur_frame` once, at the place the reference meets it first: the keyframe of smallest rank, the
 *     smallest position in that keyframe's fts_ (:85-101, last_projected_kf_id_); every candidate (type 1) is
 *     projected, in list order, after them (:108-123);
 *   - points inside the frame (8 px border) fall into their grid cell; per cell the order is the reference's stable
 *     sort by type, descending (:153);
 *   - cells are taken in visiting order from `first_cell` until `max_cells_with_trials` of them hold at least one
 *     candidate with a close view (or the cells run out): end_cell.  Their candidates become the visit list, those
 *     with a close view the trials.
 * d_kf_rank [frames->n_frames]: rank among the overlapping keyframes, closest = 0, < 16; -1: not one of them.
 * One workgroup; results are complete when the stream reaches the next command. */
int svo_hip_reproject_map(const svo_hip_camera* cam, const svo_hip_frames* frames, int cur_frame,
                          const int32_t* d_kf_rank, const svo_hip_map* map, const svo_hip_map_patch* patch,
                          const svo_hip_grid* grid, int first_cell, int max_cells_with_trials, int max_visits,
                          int max_trials, const svo_hip_reprojection* out, void* stream);

/* svo_hip_find_match_direct / svo_hip_select_matches for a batch whose size is known on the device only: the number
 * of trials is read from d_M[0] by the kernels (clamped to M_cap, the capacity of the arrays; launches are sized for
 * M_cap <= 65536), and the observations of trial m are the records [d_obs_begin[m], d_obs_end[m]) of `obs`. */
int svo_hip_find_match_direct_indirect(const svo_hip_pyr_layout* layout, const uint8_t* d_store,
                                       const svo_hip_camera* cam, const svo_hip_frames* frames, int M_cap,
                                       const int32_t* d_M, const int32_t* d_cur_frame, const double* d_pt_pos,
                                       const int32_t* d_obs_begin, const int32_t* d_obs_end,
                                       const svo_hip_features* obs, int n_pyr_levels, int align_max_iter,
                                       double* d_px_cur, int32_t* d_ok, int32_t* d_ref_obs, int32_t* d_search_level,
                                       double* d_A_cur_ref, uint8_t* d_patch_out, void* d_workspace,
                                       size_t workspace_bytes, void* stream);
int svo_hip_select_matches_indirect(const svo_hip_camera* cam, int M_cap, const int32_t* d_M, const int32_t* d_cell,
                                    const int32_t* d_ok, const double* d_px, const int32_t* d_level,
                                    const double* d_pos, int max_fts, int32_t* d_n, int32_t* d_sel, double* d_f,
                                    int32_t* d_level_out, double* d_pos_out, uint8_t* d_has_point, int32_t* d_signal,
                                    int32_t signal_value, void* stream);

/* Frame glue that the reference does inline on the host, kept on the device so a tracked
 * frame never leaves HBM between kernels:
 *  - svo_hip_compose_poses: d_out[i] = d_A[i] * d_B[i] (Sophus SE3 product), e.g.
 *    cur_frame_->T_f_w_ = T_cur_from_ref * ref_frame_->T_f_w_ (sparse_img_align.cpp:70);
 *    d_out rows may be scattered through d_out_index (NULL = identity).
 *  - svo_hip_cam2world: d_f[i] = cam->cam2world(d_px[i]), the unit bearing a new Feature is
 *    constructed with (feature.h:44-52, reprojector.cpp:182). */
int svo_hip_compose_poses(int n, const double* d_A, const double* d_B, double* d_out,
                          const int32_t* d_out_index, void* stream);
int svo_hip_cam2world(const svo_hip_camera* cam, int n, const double* d_px, double* d_f,
                      void* stream);
/* The new frame's pose as SparseImgAlign::run leaves it (sparse_img_align.cpp:70: cur_frame_->T_f_w_ =
 * T_cur_from_ref * ref_frame_->T_f_w_), formed ON THE STREAM behind svo_hip_sparse_align so that the kernels of the
 * frame's next steps (svo_hip_reproject_map, the match kernels, a predicted svo_hip_pose_optimize_deferred) can be
 * enqueued behind it without the host in between.  The product is Sophus' SE3 product as the host forms it: the left
 * factor's unit quaternion from d_T_cur_ref's rotation matrix (SE3(R, t)), the right factor's unit quaternion AS THE
 * HOST HOLDS IT (d_q_ref: w, x, y, z; d_t_ref) -- not re-derived from a rotation matrix --, the result normalised and
 * converted to (R, t): bit for bit what `poseToRt(poseFromRt(T_cur_ref) * T_ref)` gives on the host, which is how a
 * caller verifies that the chain it enqueued ran on the pose it later computes itself.
 *   d_frame_T   [n_frames][12] the frame table the following kernels read; entry cur_frame is overwritten
 *   d_T_copy    (may be NULL) [12] a second copy, e.g. the in/out pose block of svo_hip_pose_optimize_deferred
 *   d_T_out     (may be NULL) [12] a third copy for the host to compare (device-mapped host memory in the drop-in)
 *   d_signal    (may be NULL) signal_value is stored there (system scope, released after the copies) when the kernel
 *               is through: a host polling it knows that svo_hip_sparse_align's results AND the composed pose are
 *               in memory
 * With d_rank != NULL the kernel also RANKS THE OVERLAPPING KEYFRAMES with the pose it has just formed -- what
 * Reprojector::reprojectMap does first (reprojector.cpp:78-84) -- so that svo_hip_reproject_map can follow on the
 * stream: entries [0, n_kf) of the frame table are the map's keyframes in Map::keyframes_ order; keyframe i is "close"
 * when the first of its key points (d_key_pos [n_kf][5][3] = Frame::key_pts_[k]->point->pos_, d_key_valid [n_kf][5] =
 * key_pts_[k] != NULL) that Frame::isVisible accepts exists (svo/src/map.cpp:106-131, frame.cpp:115-123), its
 * distance is the norm of the difference of the two T_f_w translations; the close keyframes are ranked closest first
 * (equal distances in map order, like the stable list sort) and cut at max_n_kfs.  d_rank [n_frames] (read by
 * svo_hip_reproject_map as d_kf_rank) and d_rank_out [n_frames] (may be NULL: the host's copy) receive the rank, or
 * -1; n_frames <= 64.  `cam` is needed for the ranking only. */
int svo_hip_frame_pose_compose(const double* d_T_cur_ref, const double* d_q_ref, const double* d_t_ref, double* d_frame_T,
                               int cur_frame, double* d_T_copy, double* d_T_out, const svo_hip_camera* cam, int n_frames,
                               int n_kf, const double* d_key_pos, const uint8_t* d_key_valid, int max_n_kfs, int32_t* d_rank,
                               int32_t* d_rank_out, int32_t* d_signal, int32_t signal_value, void* stream);

/*
 * K4: batched pose_optimizer::optimizeGaussNewton (svo/src/pose_optimizer.cpp:28-161).
 * Observations of frame b are rows [b*n_stride, b*n_stride+d_n[b]); d_n[b] <= n_stride is the
 * caller's contract (the kernels clamp it, they never read past a frame's row).
 *   svo_hip_pose_optimize          one wave per frame, f64 sums reduced by a wave butterfly:
 *                                  pose within 1e-9 (SE(3) log norm) of the reference on a frame of
 *                                  20 or more observations (measured: 1e-15; 2e-9 where the stop
 *                                  decision at convergence falls the other way and leaves the last
 *                                  step; a frame of five to seven observations can amplify the
 *                                  summation order to 6e-9), medians exact
 *                                  order statistics of that pose's residuals, pruning as
 *                                  e2 > thresh^2 on them (the reference: e.norm() > thresh) -- the
 *                                  reference's decisions on every frame of the tests.  n_stride > 256
 *                                  runs the ordered kernel.
 *   svo_hip_pose_optimize_ordered  one workgroup per frame, sums in the reference's observation
 *                                  order (same normal equations as the reference, pose <= 1e-12;
 *                                  Cov_ to 1e-6 relative: pivoted LDLT here, Matrix6d::inverse()
 *                                  there): the checker.
 *   d_f [B][n_stride][3]   Feature::f
 *   d_level [B][n_stride]  Feature::level
 *   d_pos [B][n_stride][3] Feature::point->pos_
 *   d_has_point [B][n_stride] in: 0 where Feature::point == NULL; out: also 0 where the
 *                          observation was pruned (:139-144)
 *   d_T_f_w [B][12]        in/out Frame::T_f_w_
 *   d_Cov [B][36]          Frame::Cov_ (may be NULL)
 *   d_stats [B][4]         estimated_scale, error_init, error_final, (double)num_obs
 *   d_ran [B]              0 where no observation had a point (:57-58: nothing is touched)
 */
int svo_hip_pose_optimize(const svo_hip_camera* cam, int B, const int32_t* d_n, int n_stride,
                          const double* d_f, const int32_t* d_level, const double* d_pos,
                          uint8_t* d_has_point, double reproj_thresh, int n_iter,
                          double* d_T_f_w, double* d_Cov, double* d_stats, int32_t* d_ran,
                          void* stream);
int svo_hip_pose_optimize_ordered(const svo_hip_camera* cam, int B, const int32_t* d_n, int n_stride,
                                  const double* d_f, const int32_t* d_level, const double* d_pos,
                                  uint8_t* d_has_point, double reproj_thresh, int n_iter,
                                  double* d_T_f_w, double* d_Cov, double* d_stats, int32_t* d_ran,
                                  void* stream);

/* svo_hip_pose_optimize without its second launch: frames the wave kernel cannot take (normal equations
 * singular to working precision, n_iter == 0) are left UNTOUCHED and reported with d_ran[b] == 2; the caller
 * finishes them with svo_hip_pose_optimize_ordered after looking at d_ran.  For single-stream hosts that
 * synchronise after the call anyway (the drop-in): the fix-up launch of svo_hip_pose_optimize is 5 us of an
 * otherwise 20 us call and almost never has work. */
int svo_hip_pose_optimize_deferred(const svo_hip_camera* cam, int B, const int32_t* d_n, int n_stride,
                                   const double* d_f, const int32_t* d_level, const double* d_pos,
                                   uint8_t* d_has_point, double reproj_thresh, int n_iter, double* d_T_f_w,
                                   double* d_Cov, double* d_stats, int32_t* d_ran, void* stream);

/*
 * K6: batched Point::optimize (svo/src/point.cpp:119-177).  Point p has observations
 * [d_obs_ptr[p], d_obs_ptr[p+1]) in (d_obs_frame -> frame table, d_obs_f); d_pos in/out.
 */
int svo_hip_point_optimize(const svo_hip_frames* frames, int P, const int32_t* d_obs_ptr,
                           const int32_t* d_obs_frame, const double* d_obs_f, int n_iter,
                           double* d_pos, void* stream);

/* ---- K5: depth filter ---------------------------------------------------- */
/* svo::Seed state, SoA (depth_filter.h:35-51): a, b, mu, z_range, sigma2 */
typedef struct svo_hip_seeds {
  float* d_a;
  float* d_b;
  float* d_mu;
  float* d_z_range;
  float* d_sigma2;
  const int32_t* d_batch_id;
} svo_hip_seeds;

#define SVO_HIP_SEED_ERASED_OLD 1   /* too old (depth_filter.cpp:216-219): erase           */
#define SVO_HIP_SEED_BEHIND 2       /* behind the camera (:225-228): untouched             */
#define SVO_HIP_SEED_NOT_IN_FRAME 3 /* projects outside the image (:229-232): untouched    */
#define SVO_HIP_SEED_NO_MATCH 4     /* findEpipolarMatchDirect failed: b++ (:238-245)      */
#define SVO_HIP_SEED_UPDATED 5      /* updateSeed ran, seed kept                           */
#define SVO_HIP_SEED_CONVERGED 6    /* updateSeed ran and converged -> new Point, erase    */
#define SVO_HIP_SEED_NAN 7          /* updateSeed ran, z_inv_min NaN: erase (:283-287)     */

typedef struct svo_hip_depth_filter_options {
  int32_t max_n_kfs;     /* DepthFilter::Options::max_n_kfs (3)                  */
  int32_t batch_counter; /* Seed::batch_counter                                  */
  double seed_convergence_sigma2_thresh; /* 200                                  */
  /* Matcher::Options (matcher.h:76-93) */
  int32_t align_1d;
  int32_t align_max_iter;
  int32_t max_epi_search_steps;
  int32_t subpix_refinement;
  int32_t epi_search_edgelet_filtering;
  int32_t n_pyr_levels;
  double epi_search_edgelet_max_angle;
} svo_hip_depth_filter_options;

/*
 * Batched DepthFilter::updateSeeds (svo/src/depth_filter.cpp:197-291) with
 * Matcher::findEpipolarMatchDirect (matcher.cpp:179-321), computeTau and updateSeed.
 * Seed s belongs to feature s of `ftr` and is updated with frame d_cur_frame[s].
 *   d_status [S]        SVO_HIP_SEED_*  (the list surgery -- erase / new Point -- stays on the host)
 *   d_xyz_world [S][3]  position of the new Point where status == CONVERGED
 *   d_px_cur [S][2]     Matcher::px_cur_ of successful matches (may be NULL)
 * static DepthFilter::updateSeed / computeTau are also exported on their own.
 */
int svo_hip_update_seeds(const svo_hip_pyr_layout* layout, const uint8_t* d_store,
                         const svo_hip_camera* cam, const svo_hip_frames* frames, int S,
                         const int32_t* d_cur_frame, const svo_hip_features* ftr,
                         const svo_hip_seeds* seeds, const svo_hip_depth_filter_options* opt,
                         int32_t* d_status, double* d_xyz_world, double* d_px_cur,
                         void* d_workspace, size_t workspace_bytes, void* stream);

/*
 * Row N2, seeds: DepthFilter::seeds_ (depth_filter.h:140, a std::list<Seed> of the struct at :35-51) kept RESIDENT in
 * HBM between frames instead of being flattened and shipped both ways by every updateSeeds call.  The store is a set
 * of caller-owned SoA columns (`svo_hip_features` for Seed::ftr, `svo_hip_seeds` for the state) indexed by SLOT; a seed
 * keeps its slot for life.  Feature::frame is stored as a key into the frame table of the call (the caller keeps the
 * keys of its keyframes stable; the current frame is entry `cur_frame` of the table).
 *   svo_hip_seed_store_patch     writes n new records (SoA in `src_*`, record i) to slots d_slot[i]: what
 *                                DepthFilter::initializeSeeds appended since the last call (depth_filter.cpp:114-132)
 *   svo_hip_update_seeds_resident  svo_hip_update_seeds for the S seeds at slots d_slot_of[s], s in list order: the
 *                                state is updated in place in the store; d_status / d_xyz_world / d_px_cur are dense
 *                                (index s) as above, and d_state_out [4][S] receives a, b, mu, sigma2 after the update
 *                                (dense, for the host's std::list<Seed>; may be NULL).  Same arithmetic, same results.
 */
typedef struct svo_hip_seed_patch {
  int32_t n;
  int32_t reserved;
  const int32_t* d_slot;   /* [n] destination slots */
  svo_hip_features src_ftr; /* [n] records (d_frame = the caller's frame key) */
  svo_hip_seeds src_seeds;  /* [n] records */
} svo_hip_seed_patch;
int svo_hip_seed_store_patch(const svo_hip_seed_patch* patch, const svo_hip_features* store_ftr,
                             const svo_hip_seeds* store_seeds, void* stream);
int svo_hip_update_seeds_resident(const svo_hip_pyr_layout* layout, const uint8_t* d_store,
                                  const svo_hip_camera* cam, const svo_hip_frames* frames, int cur_frame, int S,
                                  const int32_t* d_slot_of, const svo_hip_features* store_ftr,
                                  const svo_hip_seeds* store_seeds, const svo_hip_depth_filter_options* opt,
                                  int32_t* d_status, double* d_xyz_world, double* d_px_cur, float* d_state_out,
                                  void* d_workspace, size_t workspace_bytes, void* stream);
/* The same with the pose of frame `cur_frame` handed over BY VALUE: T_cur_f_w is a HOST pointer to the 12 doubles of its
 * frame-table row (R row-major, then t), read before the call returns; row `cur_frame` of frames->d_T_f_w is not read.  For
 * a host that uploads the call's tables BEFORE the frame's pose is known -- the depth filter's drop-in marshals and uploads
 * while pose_optimizer::optimizeGaussNewton is still running on the device and launches when its result has arrived
 * (frame_handler_mono.cpp:166-198; dropin/depth_filter.cpp, EarlyUpdate).  Same arithmetic, same results. */
int svo_hip_update_seeds_resident_pose(const svo_hip_pyr_layout* layout, const uint8_t* d_store,
                                       const svo_hip_camera* cam, const svo_hip_frames* frames, int cur_frame,
                                       const double* T_cur_f_w, int S, const int32_t* d_slot_of,
                                       const svo_hip_features* store_ftr, const svo_hip_seeds* store_seeds,
                                       const svo_hip_depth_filter_options* opt, int32_t* d_status, double* d_xyz_world,
                                       double* d_px_cur, float* d_state_out, void* d_workspace, size_t workspace_bytes,
                                       void* stream);

/*
 * Batched Matcher::findEpipolarMatchDirect (svo/src/matcher.cpp:179-321; matcher.h:113-123) on its own:
 * query s searches along the epipolar segment of feature s of `ftr` (a feature of frame ftr->d_frame[s])
 * in frame d_cur_frame[s], for depths d_d_min[s] .. d_d_max[s] around d_d_estimate[s].
 *   d_ok [S]            the function's result
 *   d_depth [S]         `depth` (triangulated, 0 where no match)
 *   d_px_cur [S][2]     Matcher::px_cur_ (may be NULL)
 *   d_search_level [S]  Matcher::search_level_ (may be NULL); -1 where the function returned before computing it
 *                       (an edgelet rejected by the angle filter, matcher.cpp:204-212: search_level_ keeps its old value)
 * Matcher::Options are taken from `opt` (the DepthFilter fields of the struct are ignored).
 */
int svo_hip_find_epipolar_match_direct(const svo_hip_pyr_layout* layout, const uint8_t* d_store,
                                       const svo_hip_camera* cam, const svo_hip_frames* frames, int S,
                                       const int32_t* d_cur_frame, const svo_hip_features* ftr,
                                       const double* d_d_estimate, const double* d_d_min, const double* d_d_max,
                                       const svo_hip_depth_filter_options* opt, int32_t* d_ok, double* d_depth,
                                       double* d_px_cur, int32_t* d_search_level, void* d_workspace,
                                       size_t workspace_bytes, void* stream);

/* Device pointer to n_steps [S] of the last svo_hip_update_seeds call that used this workspace: the
 * number of epipolar-line positions scanned per seed (matcher.cpp:248, 0 where no scan ran).  For
 * roofline accounting (64 B scanned per step). */
const int32_t* svo_hip_update_seeds_scan_steps(const void* d_workspace);

/* Roofline accounting of the depth filter's sub-pixel alignment (feature_alignment.cpp:30-277 inside
 * Matcher::findEpipolarMatchDirect, matcher.cpp:295-315): with counting switched on (process-wide, off by default;
 * returns the previous setting) svo_hip_update_seeds* and svo_hip_find_epipolar_match_direct run the instrumented
 * alignment kernel, which also stores the number of residual evaluations (9 x 9 windows read) of every seed;
 * svo_hip_update_seeds_align_evaluations is the device pointer to that array [S] inside the workspace of the last
 * such call with S seeds (0 for a seed that did not reach the alignment; undefined while counting was off).
 * Results do not depend on the switch. */
int svo_hip_update_seeds_count_evaluations(int on);
const int32_t* svo_hip_update_seeds_align_evaluations(const void* d_workspace, int S);

/* DepthFilter::updateSeed(x, tau2, seed) for S independent (x, tau2) measurements */
int svo_hip_update_seed_batch(int S, const float* d_x, const float* d_tau2,
                              const svo_hip_seeds* seeds, void* stream);

/* ---- K7: seed initialisation --------------------------------------------- */
/*
 * Batched feature_detection::FastDetector::detect (svo/src/feature_detection.cpp:66-114) on the
 * pyramids of n_frames store slots: FAST-10 (threshold fast_threshold = 20 in the reference),
 * FAST score, 3x3 non-max, Shi-Tomasi score, best corner per grid cell over levels
 * 0..n_levels-1 (Config::nPyrLevels()).
 *   d_occupancy [n_frames][cells]  AbstractDetector::grid_occupancy_ (setExistingFeatures /
 *                                  setGridOccpuancy); NULL = all free
 *   detection_threshold            Config::triangMinCornerScore()
 *   d_corner_xy [n_frames][cells][2]  Corner::x, y (level-0 pixels); -1 where the cell stays empty
 *   d_corner_level / d_corner_score   Corner::level / score (score == threshold where empty)
 * A Feature is created for every cell with score > detection_threshold, in cell order (:107-110).
 */
size_t svo_hip_fast_workspace_bytes(const svo_hip_pyr_layout* layout, int n_frames, int n_cells);
int svo_hip_fast_detect(const svo_hip_pyr_layout* layout, const uint8_t* d_store, int n_frames,
                        const int32_t* d_slot, int n_levels, int fast_threshold, int cell_size,
                        int grid_n_cols, int grid_n_rows, const uint8_t* d_occupancy,
                        double detection_threshold, int32_t* d_corner_xy, int32_t* d_corner_level,
                        float* d_corner_score, void* d_workspace, size_t workspace_bytes, void* stream);

/* static DepthFilter::computeTau(T_ref_cur, f, z, px_error_angle) (depth_filter.cpp:334-350) for S
 * independent measurements: d_t_ref_cur [S][3] = T_ref_cur.translation(), d_f [S][3], d_z [S].
 * Arithmetic (since round 5, here and inside svo_hip_update_seeds*): the ALGEBRAIC form -- alpha and beta enter only
 * through their cosines, which the reference forms as dot products before it calls acos (:339-340); their sines are
 * sqrt(1 - cos^2) and sin(pi - alpha - beta_plus) follows from the angle-sum formulas with the sine / cosine of
 * px_error_angle (a per-launch constant, computed on the host by libm).  No acos / sin runs on the device.  The value
 * differs from the reference's acos / sin expression by rounding only: <= 7e-13 relative on usable geometry
 * (tests/test_device_math_host.py::test_compute_tau_algebraic_form measures it; test_compute_tau_and_triangulation pins the acos / sin statement against the oracle's acos / sin form), i.e.
 * tau is NOT bit-identical to DepthFilter::computeTau; the f32 seed update that consumes it (1 / tau^2 rounded to float)
 * is, in every case the GPU suite compares (tests/test_tracking_gpu.py prints the measured deviation of a, b, sigma2: 0). */
int svo_hip_compute_tau_batch(int S, const double* d_t_ref_cur, const double* d_f, const double* d_z,
                              double px_error_angle, double* d_tau, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SVO_HIP_H_ */
