// matcher_device.h -- device functions of the warp:: namespace shared by matcher.hip and
// depth_filter.hip (no relocatable device code: they must be inline).
#pragma once
#include "track_math.h"

namespace svo_track {
using namespace svo_dev;

// warp::getWarpMatrixAffine (matcher.cpp:33-55); A row-major 2x2
__device__ inline void warp_matrix_affine(const Cam& cam, const double px_ref[2], const double f_ref[3], double depth_ref,
                                   const Se3& T_cur_ref, int level_ref, double A[4]) {
  const int halfpatch_size = 5;
  const double xyz_ref[3] = {f_ref[0] * depth_ref, f_ref[1] * depth_ref, f_ref[2] * depth_ref};
  double xyz_du_ref[3], xyz_dv_ref[3];
  const double s = (double)(1 << level_ref);
  cam2world(cam, px_ref[0] + (double)halfpatch_size * s, px_ref[1] + 0.0 * s, xyz_du_ref);
  cam2world(cam, px_ref[0] + 0.0 * s, px_ref[1] + (double)halfpatch_size * s, xyz_dv_ref);
  const double ku = xyz_ref[2] / xyz_du_ref[2];
  xyz_du_ref[0] *= ku; xyz_du_ref[1] *= ku; xyz_du_ref[2] *= ku;
  const double kv = xyz_ref[2] / xyz_dv_ref[2];
  xyz_dv_ref[0] *= kv; xyz_dv_ref[1] *= kv; xyz_dv_ref[2] *= kv;
  double p[3], px_cur[2], px_du[2], px_dv[2];
  se3_apply(T_cur_ref, xyz_ref, p);
  world2cam(cam, p, px_cur);
  se3_apply(T_cur_ref, xyz_du_ref, p);
  world2cam(cam, p, px_du);
  se3_apply(T_cur_ref, xyz_dv_ref, p);
  world2cam(cam, p, px_dv);
  A[0] = (px_du[0] - px_cur[0]) / halfpatch_size;
  A[2] = (px_du[1] - px_cur[1]) / halfpatch_size;
  A[1] = (px_dv[0] - px_cur[0]) / halfpatch_size;
  A[3] = (px_dv[1] - px_cur[1]) / halfpatch_size;
}

// warp::getBestSearchLevel (matcher.cpp:57-70)
__device__ inline int best_search_level(const double A[4], int max_level) {
  int search_level = 0;
  double D = det2<double>(A);
  while (D > 3.0 && search_level < max_level) {
    search_level += 1;
    D *= 0.25;
  }
  return search_level;
}

}  // namespace svo_track
