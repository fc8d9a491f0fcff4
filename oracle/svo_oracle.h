/*
 * svo_oracle.h -- CPU restatement ("oracle") of the SVO tracking hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / reported baseline.  The
 * product path (rpg_svo_amd/, include/svo_hip.h) never links or calls it.
 *
 * What it restates (file:line are relative to the reference checkout,
 * uzh-rpg/rpg_svo):
 *   svo/src/sparse_img_align.cpp:43-258      SparseImgAlign (run, precompute,
 *                                            computeResiduals, solve, update)
 *   svo/include/svo/frame.h:116-138          Frame::jacobian_xyz2uv
 *   svo/src/frame.cpp:156-165                createImgPyramid
 *   svo/src/feature_alignment.cpp:30-277     align1D / align2D (float paths)
 *   svo/src/matcher.cpp:33-177               affine warp + findMatchDirect
 *   svo/src/pose_optimizer.cpp:28-161        optimizeGaussNewton
 *   svo/src/depth_filter.cpp:309-350         updateSeed / computeTau
 *
 * Third-party arithmetic that is NOT in the reference tree and is restated
 * here from the published upstream sources (un-vendored, un-pinned in the
 * reference: svo/package.xml:16-26, svo/CMakeLists.txt:48-55):
 *   rpg_vikit   vk::NLLSSolver<6,SE3> Gauss-Newton loop, vk::halfSample,
 *               vk::PinholeCamera (zero distortion), vk::project2d,
 *               vk::norm_max, robust_cost Tukey / MAD, vk::getMedian,
 *               vk::interpolateMat_8u
 *   Sophus      old non-templated SE3/SO3 (unit quaternion storage)
 *   Eigen3      LDLT (pivoted, lower, unblocked), Quaternion<->Matrix3
 *   boost::math normal pdf
 *
 * PARITY PINNING: the reference's own tests pin no value for this path that
 * can be evaluated without the (absent) sin2_tex2_h1_v8_d dataset.  The
 * restatement is pinned instead against the reference's own translation
 * units compiled in place (oracle/_ref, see oracle/Makefile and
 * oracle/shim/): same inputs, outputs compared in tests/test_oracle_vs_ref.py.
 * The third-party pieces above are shims in both, so they remain
 * "restated from upstream, unpinned".
 */
#ifndef SVO_ORACLE_H_
#define SVO_ORACLE_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LEVELS 8

/* half-sampling flavours of vk::halfSample (rpg_vikit vision.cpp) */
#define ORC_HALFSAMPLE_SCALAR 0 /* (tl+tr+bl+br)/4, integer truncation            */
#define ORC_HALFSAMPLE_SSE2   1 /* avg_epu8 rows then avg_epu16 cols (round-up)   */
#define ORC_HALFSAMPLE_AUTO   2 /* what an x86 build does: SSE2 iff in_w % 16 == 0 */

/* ---- SE(3) helpers (pose = 12 doubles: R row-major [9], t [3]) ---------- */
void orc_se3_exp(const double xi[6], double T_out[12]);     /* Sophus SE3::exp, xi=[v,w] */
void orc_se3_log(const double T[12], double xi_out[6]);     /* Sophus SE3::log           */
void orc_se3_mul(const double A[12], const double B[12], double out[12]);
void orc_se3_inv(const double A[12], double out[12]);
int  orc_ldlt6_solve(const double H[36], const double b[6], double x[6]);
int  orc_ldlt_solve_n(int n, const double* H, const double* b, double* x);

/* ---- pyramid ------------------------------------------------------------ */
void orc_half_sample(const uint8_t* in, int in_w, int in_h, int in_stride,
                     uint8_t* out, int out_stride, int mode);
/* Builds levels 1..n_levels-1 below level 0 (contiguous rows, stride = w). */
void orc_create_img_pyramid(const uint8_t* lvl0, int w, int h, int n_levels, int mode,
                            uint8_t* const* levels_out /* [n_levels], [0] is copied */);

/* ---- SparseImgAlign ------------------------------------------------------ */
typedef struct {
  int n_levels;
  int w[ORC_MAX_LEVELS];
  int h[ORC_MAX_LEVELS];
  const uint8_t* data[ORC_MAX_LEVELS]; /* stride == w, like a continuous cv::Mat */
} orc_pyramid;

typedef struct {
  double fx, fy, cx, cy;
  int width, height;
} orc_pinhole;

typedef struct {
  int max_level;
  int min_level;
  int n_iter;
  double eps;     /* 1e-6, sparse_img_align.cpp:40 */
} orc_sia_options;

typedef struct {
  int n_tracked;                    /* n_meas_/patch_area_, sparse_img_align.cpp:74 */
  int stop;                         /* vk::NLLSSolver::stop_ after the run           */
  int iters[ORC_MAX_LEVELS];        /* residual evaluations executed, per level      */
  double chi2;                      /* chi2_ after the run                           */
  double H[36];                     /* H_ of the last evaluated iteration            */
  double T_cur_from_ref[12];        /* final relative pose                           */
} orc_sia_result;

/*
 * Restates SparseImgAlign::run(ref_frame, cur_frame).
 *   T_ref_w / T_cur_w : frame poses T_f_w (cur: in = prior, out = estimate)
 *   features of the reference frame in list order: px (level-0 pixels),
 *   f (unit bearing), has_point, pos (world position of the point).
 * visible_out (optional, n bytes) receives visible_fts_.
 */
int orc_sparse_img_align_run(const orc_pyramid* ref_pyr, const orc_pyramid* cur_pyr,
                             const orc_pinhole* cam,
                             const double T_ref_w[12], double T_cur_w[12],
                             int n, const double* px, const double* f,
                             const uint8_t* has_point, const double* pos,
                             const orc_sia_options* opt,
                             orc_sia_result* res, uint8_t* visible_out);

/* Batch driver used by tests / cpu_baseline: problems b=0..B-1 share cam and
 * options; pyramids are addressed through per-problem slot indices.         */
int orc_sparse_img_align_batch(int B, const orc_pyramid* pyrs, const int* ref_slot,
                               const int* cur_slot, const orc_pinhole* cam,
                               const double* T_ref_w /*[B][12]*/, double* T_cur_w /*[B][12]*/,
                               const int* n /*[B]*/, int n_stride,
                               const double* px, const double* f,
                               const uint8_t* has_point, const double* pos,
                               const orc_sia_options* opt, orc_sia_result* res /*[B]*/,
                               int n_threads);

#ifdef __cplusplus
}
#endif
#endif /* SVO_ORACLE_H_ */
