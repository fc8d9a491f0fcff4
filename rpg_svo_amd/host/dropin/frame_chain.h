// frame_chain.h -- a frame's NEXT steps enqueued behind its sparse alignment (VERDICT r05 "one completion per frame").
//
// FrameHandlerMono::processFrame (frame_handler_mono.cpp:129-176) calls SparseImgAlign::run, Reprojector::reprojectMap and
// pose_optimizer::optimizeGaussNewton one after the other, each on the result of the one before.  Up to round 5 every one
// of them was a call of its own on the lane's stream with the host in between: the GPU idled from the end of K1 until the
// host had woken up, formed the frame's pose, found the overlapping keyframes, patched the map mirror and launched again
// (33 us of 290, profiles/r05x_dropin_frame_timeline_600.txt).  Now the drop-in of reprojectMap registers a FrameChain on
// the tracking lane, and the drop-in of SparseImgAlign::run lets it add its inputs, outputs and launches to the SAME call:
//
//   H2D (K1's inputs) -> K0 (new frame) -> K1 | meanwhile on the host: the map patch, the key points of the keyframes, the
//   frame table -> H2D (the chain's inputs, second arena) -> frame_pose_compose (cur.T_f_w = T_cur_ref * ref.T_f_w on the
//   device: frame table, K4's pose block, a copy for the host; the overlapping keyframes ranked with that pose; signal 1)
//   -> reproject_map -> match kernels -> selection (signal 2) -> K4
//
// run() returns at signal 1 with the stream still busy; reprojectMap VERIFIES what was assumed when the chain was built --
// the pose the device formed is bit for bit the pose the host formed from K1's result, the device ranked the overlapping
// keyframes as Map::getCloseKeyframes and the reprojector's sort rank them on the host, the map has not changed in
// between -- waits for signal 2 and continues with its bookkeeping exactly as after a batch of
// its own; a failed check drains the stream and takes the ordinary path (same results either way: the chain only ever
// replaces work that is a pure function of what was verified).
#ifndef SVO_HIP_DROPIN_FRAME_CHAIN_H_
#define SVO_HIP_DROPIN_FRAME_CHAIN_H_

#include <svo/frame.h>

#include "marshal.h"

namespace svo {
namespace hip_dropin {

class FrameChain {
 public:
  virtual ~FrameChain() {}
  // Called by SparseImgAlign::run's drop-in with the lane locked and K1 ALREADY ENQUEUED (its inputs left with the lane's
  // arena): the host work of the chain -- bringing the map's shadow up to date, the key points, the frame table -- runs
  // while K1 does.  false: no chain for this frame (nothing was allocated or enqueued; the shadow of the map may have
  // been brought up to date, which the ordinary path would have done anyway).
  virtual bool prepare(const FramePtr& ref, const FramePtr& cur, svo_hip::Device& dev, svo_hip::Lane& lane) = 0;
  // the chain's input blocks, into the lane's second arena (uploaded by the caller right after), and its outputs, behind
  // K1's in the lane's arena
  virtual void allocInputs(svo_hip::Arena& inputs, const FramePtr& ref) = 0;
  virtual void allocOutputs(svo_hip::Arena& a) = 0;
  // behind svo_hip_sparse_align on the lane's stream; d_T_cur_ref: its pose block (device address)
  virtual void enqueue(const double* d_T_cur_ref) = 0;
  // device-mapped host word that reads 1 once K1's results and the composed pose are in host memory
  virtual const volatile int32_t* k1Signal() const = 0;
  virtual void abandon() = 0;  // something threw between prepare() and enqueue(): forget the frame
  // upper bound of what allocOutputs() takes from the lane's arena (reserved before K1's own blocks are carved)
  virtual size_t outputBytesBound() const = 0;
};

// Sophus::SE3 -> unit quaternion (w, x, y, z) and translation, as the object holds them (NOT via the rotation matrix: the
// device's product must start from the same bits the host's operator* reads)
inline void poseToQt(const SE3& T, double q[4], double t[3]) {
  q[0] = T.so3().unit_quaternion().w(); q[1] = T.so3().unit_quaternion().x();
  q[2] = T.so3().unit_quaternion().y(); q[3] = T.so3().unit_quaternion().z();
  const Vector3d tr = T.translation();
  t[0] = tr[0]; t[1] = tr[1]; t[2] = tr[2];
}

}  // namespace hip_dropin
}  // namespace svo
#endif  // SVO_HIP_DROPIN_FRAME_CHAIN_H_
