// svo_hip_device.h -- host-side device context shared by the C++ drop-in bodies
// (rpg_svo_amd/host/dropin/*.cpp).  Plain C++11 over the C ABI of include/svo_hip.h: no HIP
// headers, no Eigen, no reference headers, so it compiles with the g++ that builds libsvo.
//
// What it owns (one per process, see Device::instance()):
//   * the pyramid store in HBM, one slot per live svo::Frame, keyed by Frame::id_; a frame is
//     uploaded (level 0 H2D into the lane's packed staging buffer + K0, which writes the tiled
//     level 0 and every further level in one kernel, replacing the host pyramid of
//     frame_utils::createImgPyramid, svo/src/frame.cpp:156-165, for every device consumer)
//     the first time a kernel needs it; the pool is an LRU cache (a frame that was evicted
//     while its host object still lives is simply uploaded again on its next use);
//   * "lanes": per host thread and role (tracking / mapping -- depth_filter.cpp:64-67) a HIP
//     stream, a pinned host arena and its device mirror, so one call is: fill the arena ->
//     ONE H2D copy -> kernels -> ONE D2H copy -> stream sync.  Several FrameHandlers driven
//     from several threads (a camera rig on one GPU) therefore run concurrently on separate
//     streams; only the pyramid cache is shared.
#ifndef SVO_HIP_DEVICE_H_
#define SVO_HIP_DEVICE_H_

#include <cstddef>
#include <cstdint>
#include <functional>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include <svo_hip.h>

namespace svo_hip {

struct Error : std::runtime_error {
  explicit Error(const std::string& what) : std::runtime_error(what) {}
};
void check(int code, const char* what);  // throws Error on a negative svo_hip status
// Spins until *flag == value.  `flag` lies in device-mapped pinned host memory and a kernel on `stream` stores the
// value (svo_hip_select_matches' d_signal): the host gets there without a runtime call, and with work still running
// behind the signalling kernel.  After a second without the value the stream is synchronised instead, so that a
// failed launch surfaces as an Error rather than a hang.
void spinUntil(const volatile int32_t* flag, int32_t value, void* stream);

// Bump allocator over a pinned host buffer and its same-sized device mirror.  Input blocks
// are carved from the front, output blocks after them; both sides share offsets.
// In MAPPED mode there is no mirror: the kernels read and write the pinned (coherent,
// device-mapped) host buffer over the host link directly, so a call is kernels + one stream sync
// with no copy commands at all -- the shorter round trip for the few-KB payloads of a single
// camera (DESIGN.md "single-stream latency"); upload()/download() are then no-ops.
// HYBRID mirrors the inputs and maps the outputs: blocks allocated before endInputs() live in the device mirror and
// travel in the one H2D copy, blocks allocated after it ARE the pinned host buffer -- kernels write results (a few KB,
// posted writes) straight into host memory and the D2H copy of a call disappears.
class Arena {
 public:
  enum Mode { MIRRORED = 0, MAPPED = 1, HYBRID = 2 };
  Arena() : h_(NULL), d_(NULL), cap_(0), used_(0), in_end_(0), mode_(MIRRORED), outputs_(false) {}
  void setMode(Mode m);  // before the first alloc of a call
  Mode mode() const { return mode_; }
  void reserve(size_t bytes);
  void reset() { used_ = 0; in_end_ = 0; outputs_ = false; }
  // n elements of T, 256-byte aligned; *dev receives the device address of the same block
  template <typename T> T* alloc(size_t n, T** dev) {
    size_t off = (used_ + 255) & ~(size_t)255;
    size_t end = off + n * sizeof(T);
    if (end > cap_) grow(end);
    used_ = end;
    *dev = reinterpret_cast<T*>((mode_ == MAPPED || (mode_ == HYBRID && outputs_) ? h_ : d_) + off);
    return reinterpret_cast<T*>(h_ + off);
  }
  void endInputs() { in_end_ = used_; outputs_ = true; }  // everything allocated so far is kernel input
  void upload(void* stream);                      // H2D of [0, in_end)
  void uploadAll(void* stream);                   // H2D of [0, used): also blocks a kernel updates in place,
                                                  // allocated after endInputs() and pre-filled by the host
  void download(void* stream);                    // D2H of [in_end, used)
  void downloadRange(size_t begin, size_t end, void* stream);  // D2H of [begin, end), offsets as used() reports them
  // D2H of one block handed out by alloc() (for arrays a kernel updates in place)
  template <typename T> void fetch(T* host_block, size_t n, void* stream) {
    fetchBytes(reinterpret_cast<uint8_t*>(host_block), n * sizeof(T), stream);
  }
  void release();
  size_t used() const { return used_; }

 private:
  void grow(size_t need);
  void fetchBytes(uint8_t* host_block, size_t bytes, void* stream);
  uint8_t* h_;
  uint8_t* d_;
  size_t cap_, used_, in_end_;
  Mode mode_;
  bool outputs_;  // endInputs() has been called since reset()
};

// Work a call leaves RUNNING on the lane's stream for the lane's next call to pick up.  The reprojector's drop-in
// enqueues pose refinement behind its match kernels (svo_hip_select_matches + svo_hip_pose_optimize) and returns as
// soon as the match results have arrived; pose_optimizer's drop-in takes the result when the frame it is handed is
// the frame that was predicted (same features, same pose, same parameters) and runs the call itself otherwise.
// The blocks live in the lane's arena: beginCall() of any other call drains the stream and drops the prediction.
struct Speculation {
  bool valid;      // a prediction is waiting
  bool in_flight;  // kernels / copies enqueued and not waited for yet ...
  void* stream;    // ... on this stream
  int frame_id;
  std::vector<const void*> point;  // the features predicted for Frame::fts_, in order: Feature::point,
  std::vector<double> px;          //   Feature::px [n][2],
  std::vector<int32_t> level;      //   Feature::level,
  std::vector<int32_t> trial;      //   and the trial each came from (what the device's own selection must say)
  double T_init[12], reproj_thresh;
  int n_iter;
  // host addresses of the results (arena blocks)
  const int32_t* n_sel;
  const int32_t* sel;
  const int32_t* ran;
  const double* T;
  const double* Cov;
  const double* stats;
  const uint8_t* has_point;
  Speculation() : valid(false), in_flight(false), stream(NULL), frame_id(-1), reproj_thresh(0), n_iter(0), n_sel(NULL), sel(NULL), ran(NULL),
                  T(NULL), Cov(NULL), stats(NULL), has_point(NULL) {}
};

struct Lane {
  void* stream;
  void* stream_next;      // the predicted next call runs here, behind ev_results: the lane's own stream -- and the host's
                          // wait on it -- ends with the results the current call returns
  void* ev_results;       // recorded on `stream` behind those results
  Speculation spec;
  // The second half of a call that returned with its kernels still running (DepthFilter::updateSeeds in deferred
  // mode): waits for the stream and writes the results back.  Run by the lane's next beginCall() or by
  // Device::joinDeferred(), whichever comes first; the arena and the pinned frames stay as the call left them until then.
  std::function<void()> deferred;
  // The frame's NEXT steps enqueued behind its sparse alignment (dropin/frame_chain.h): the drop-in of
  // Reprojector::reprojectMap registers a hip_dropin::FrameChain here, SparseImgAlign::run's drop-in calls it.  Opaque in
  // this header (no reference types); owned by the registering Reprojector, which clears it under `mut` when it dies.
  void* chain_hook;
  // The depth filter's update of a frame enqueued EARLY (dropin/depth_filter.cpp): a host that runs the filter without its
  // thread calls updateSeeds a few dozen microseconds after the pose optimizer has returned; the optimizer's drop-in calls
  // `early_hook` (set on the calling thread's MAPPING lane by DepthFilter::updateSeeds' drop-in; argument: const FramePtr*)
  // and the update's kernels run while the host finishes the frame.  `early_drop` is set while such an update is pending:
  // the real updateSeeds takes it (same frame, same pose, same seed list) or drops it; beginCall() of any other call on
  // the lane drops it (the arena is about to be refilled).
  std::function<void(const void*, int)> early_hook;  // (frame, phase: 1 = marshal + upload and hold, 2 = launch what is held -- or all of it)
  std::function<void()> early_drop;
  Arena arena;
  Arena arena_chain;      // the chain's INPUT blocks: filled and uploaded while K1 (whose inputs left with `arena`) is running
  void* d_workspace;      // matcher / depth-filter scratch (svo_hip_match_workspace_bytes)
  size_t workspace_bytes;
  void* d_stage;          // packed level-0 image on its way into the tiled store (svo_hip_pyramid_upload_*)
  int index;              // unique per lane: scratch frames of different lanes never share a slot
  double pyr_upload_us;   // time this lane spent uploading pyramids (StageTimer takes it out of "marshal")
  std::mutex mut;
  std::vector<int> touched;  // frames pinned by the lane's current call
  Lane() : stream(NULL), stream_next(NULL), ev_results(NULL), chain_hook(NULL), d_workspace(NULL), workspace_bytes(0), d_stage(NULL), index(0), pyr_upload_us(0) {}
};

class Device {
 public:
  enum { LANE_TRACKING = 0, LANE_MAPPING = 1, N_LANES = 2 };

  // Context for frames of one image geometry (width x height x pyramid levels), created on
  // first use; all frames of one svo::FrameHandlerMono share a camera, several handlers with
  // different cameras (a heterogeneous rig) get one context each.
  static Device& forGeometry(int width, int height, int n_levels);
  // The context used last (the one and only in a single-camera process).
  static Device& instance();

  // (Re)create the store: width/height of level 0, pyramid levels, slots.  Called lazily by
  // ensureConfigured(); call it explicitly to size the slot pool (default 64 frames).
  void configure(int width, int height, int n_levels, int n_slots = 64, int device = 0);
  void ensureConfigured(int width, int height, int n_levels);
  bool configured() const { return d_store_ != NULL; }
  void shutdown();

  const svo_hip_pyr_layout& layout() const { return layout_; }
  const uint8_t* store() const { return d_store_; }
  int nLevels() const { return layout_.n_levels; }
  int slots() const { return n_slots_; }  // size of the pyramid pool

  // Every entry point brackets its work with beginCall(): slots touched since then are
  // pinned (never evicted) until the lane's next beginCall().
  void beginCall(int which_lane) { beginCall(lane(which_lane)); }
  void beginCall(Lane& lane);
  // The host waits for everything enqueued on the lane's stream (svo_hip_stream_sync).  Round 4 timed the alternative --
  // a stream write-value command storing a sequence number into a pinned word the host polls -- on the GPU box: 5-20 us
  // SLOWER per frame on both map sizes (profiles/r04b_wait_modes_*.txt: the command processor's write lands later than
  // the runtime's own completion signal), so it was removed.
  void finish(Lane& lane);
  // Completes a deferred call of the calling thread's lane of that role, if there is one (no lane is created).
  void joinDeferred(int which_lane);
  static void joinDeferredAll();  // the same for every context of the process
  // Deferred mapping (opt-in: SVO_HIP_MAPPER=deferred or setDeferredMapping(true)).  For hosts that run the depth
  // filter synchronously inside addFrame() (DepthFilter without its thread, depth_filter.cpp:82-95): updateSeeds()
  // returns once its kernels are enqueued on the mapping lane's stream; results are written into the seed list,
  // converged seeds handed to the map, at the next point where the reference's control flow reads them -- the next
  // Reprojector::reprojectMap, the next DepthFilter::updateSeeds, FastDetector::detect (DepthFilter::initializeSeeds)
  // -- so every consumer sees the state the synchronous filter would have left.  The mapping then overlaps the end of
  // frame t and pyramid + sparse alignment of frame t+1 the way the reference's mapping thread does, deterministically.
  // Contract: DepthFilter::getSeeds() readers and the DepthFilter destructor need joinDeferredAll() first.
  static bool deferredMapping();
  static void setDeferredMapping(bool on);
  // Slot of frame `id`; on a miss the level-0 image (8-bit, `stride` bytes per row) is
  // uploaded and the pyramid built on the lane's stream, evicting the least recently used
  // unpinned frame when the pool is full.
  int slotOf(int frame_id, const uint8_t* level0, int stride, int which_lane) { return slotOf(frame_id, level0, stride, lane(which_lane)); }
  int slotOf(int frame_id, const uint8_t* level0, int stride, Lane& lane);
  void forget(int frame_id);
  // A stand-alone image (not level 0 of a svo::Frame) as level *level_out of a scratch slot: the level
  // of the layout whose size is w x h.  Uploaded on every call (the caller's buffer may have changed);
  // the slot is pinned like a frame until the lane's next beginCall().  Scratch slots belong to ONE lane
  // (id = -1 - (lane.index * SVO_HIP_MAX_LEVELS + level)): two threads aligning against their own images
  // never overwrite each other's.  Throws when no level matches.
  int scratchSlotOf(const uint8_t* image, int w, int h, int stride, int* level_out, Lane& lane);

  // The calling thread's lane of the given role (created on first use).
  Lane& lane(int which);
  void* workspace(Lane& lane, int n_trials);  // grows the lane's matcher scratch on demand

  // statistics for the latency read-outs.  Per stage (STAGE_*) the host time of a drop-in call is split
  // into: marshal (walking the reference's pointer graph into the pinned arena), device (H2D copy +
  // kernels + D2H copy + stream sync, as seen from the host) and unmarshal (writing results back into
  // Frame / Feature / Point / Seed objects); pyramid uploads (new frames: H2D + K0) are kept apart.
  enum { STAGE_SPARSE_ALIGN = 0, STAGE_REPROJECT = 1, STAGE_POSE_OPT = 2, STAGE_DEPTH_FILTER = 3, N_STAGES = 4 };
  struct Stats {
    uint64_t uploads, evictions, calls;
    uint64_t spec_hits, spec_misses;  // pose refinements taken from / not taken from the reprojector's prediction
    uint64_t chain_hits, chain_misses;  // reprojections + matches taken from / not taken from the chain enqueued behind K1
    uint64_t early_map_hits, early_map_misses;  // depth-filter updates enqueued by the pose optimizer's drop-in: taken / dropped
    uint64_t early_map_two_phase;               // ... of them launched in two phases (tables uploaded before the pose was known)
    uint64_t chain_miss_why[6];         // ... not taken because: 0 not this frame / drained, 1 pose bits, 2 keyframe ranking,
                                        //     3 the map moved on, 4 a capacity was exceeded on the device, 5 (spare)
    double pyr_upload_us;
    double marshal_us[N_STAGES], device_us[N_STAGES], unmarshal_us[N_STAGES], payload_bytes[N_STAGES];
    uint64_t n[N_STAGES];
    Stats() : uploads(0), evictions(0), calls(0), spec_hits(0), spec_misses(0), chain_hits(0), chain_misses(0), early_map_hits(0), early_map_misses(0), early_map_two_phase(0),
              pyr_upload_us(0) {
      for (int i = 0; i < 6; ++i) chain_miss_why[i] = 0;
      for (int i = 0; i < N_STAGES; ++i) { marshal_us[i] = device_us[i] = unmarshal_us[i] = payload_bytes[i] = 0; n[i] = 0; }
    }
  };
  Stats stats;             // written under stats_mut_ only; read it through statsSnapshot() while calls run
  Stats statsSnapshot();
  // count = false: the second half of a call already counted (a deferred call's join)
  void addStage(int stage, double marshal_us, double device_us, double unmarshal_us, double payload_bytes, bool count = true);
  void countSpeculation(bool hit);
  void countChain(bool hit, int why = 0);
  void countEarlyMapping(bool hit);
  void countEarlyTwoPhase();
  // SVO_HIP_EARLY_MAPPER=0: the depth filter's update is enqueued when the reference calls it, as up to round 5
  static bool earlyMappingEnabled();
  // the calling thread's lane of that role if it exists (no lane is created)
  Lane* findLane(int which);
  // SVO_HIP_CHAIN=0 switches the chain behind the sparse alignment off (reprojectMap then starts its own call, as before)
  static bool chainEnabled();
  // SVO_HIP_SPECULATE=0 switches the reprojector's prediction off (every stage then ends with its own stream sync)
  static bool speculationEnabled();

 private:
  Device();
  ~Device();
  Device(const Device&);
  // owner / settled: a pyramid is published as soon as its upload is ENQUEUED on the owner lane's stream.  The
  // owner's later work is ordered behind it by the stream; another lane that finds the entry first asks the slot's
  // event (recorded behind the upload) and, while that has not fired, makes its own stream wait for it.
  struct Entry { int slot; uint64_t last_use; int pins; int owner; bool settled; };
  std::vector<void*> slot_ready_;  // one event per slot
  svo_hip_pyr_layout layout_;
  uint8_t* d_store_;
  int n_slots_;
  std::vector<int> free_slots_;
  std::map<int, Entry> frames_;
  std::mutex frames_mut_;
  uint64_t clock_;
  std::mutex lanes_mut_;
  std::map<std::pair<std::thread::id, int>, Lane*> lanes_;
  int next_lane_index_;
  std::mutex stats_mut_;
  Lane* makeLane();
  // frames_mut_ held: hit -> pin + slot; miss with level0 != NULL -> evict if needed, upload, publish;
  // miss with level0 == NULL -> -1.  wait_upload: the caller's image buffer goes away right after the call
  int slotOfLocked(int frame_id, const uint8_t* level0, int stride, Lane& lane, bool wait_upload = false);
};

// Splits one drop-in call on the host clock (see Device::Stats).  marshal until device(), device until
// unmarshal(), unmarshal until destruction; time THIS LANE spent uploading pyramids inside the marshal
// phase is taken out of it (another lane's uploads do not enter: the counter is the lane's own).
class StageTimer {
 public:
  StageTimer(Device& dev, Lane& lane, int stage) : dev_(dev), lane_(lane), stage_(stage), phase_(0), up0_(lane.pyr_upload_us), up1_(0), bytes_(0) { t_[0] = now(); }
  void device(size_t payload_bytes) { t_[1] = now(); phase_ = 1; up1_ = lane_.pyr_upload_us; bytes_ = (double)payload_bytes; }
  void unmarshal() { t_[2] = now(); phase_ = 2; }
  ~StageTimer() {
    if (phase_ < 2) return;  // the call left early (nothing to do): not a sample
    const double t3 = now();
    dev_.addStage(stage_, (t_[1] - t_[0]) - (up1_ - up0_), t_[2] - t_[1], t3 - t_[2], bytes_);
  }
  static double now();  // microseconds, steady clock

 private:
  Device& dev_;
  Lane& lane_;
  int stage_, phase_;
  double t_[3], up0_, up1_, bytes_;
};

}  // namespace svo_hip
#endif  // SVO_HIP_DEVICE_H_
