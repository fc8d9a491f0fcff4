// svo_hip_device.cpp -- see svo_hip_device.h.
#include "svo_hip_device.h"

#include <chrono>
#include <cstdlib>

#include <cstring>

namespace svo_hip {

void check(int code, const char* what) {
  if (code >= 0) return;
  throw Error(std::string(what) + ": " + svo_hip_strerror(code) + " (hip error " +
              std::to_string(svo_hip_last_hip_error()) + ")");
}

// ---- Arena ---------------------------------------------------------------------------
void Arena::release() {
  if (h_) svo_hip_host_free(h_);
  if (d_) svo_hip_free(d_);
  h_ = d_ = NULL;
  cap_ = used_ = in_end_ = 0;
}

void Arena::reserve(size_t bytes) {
  if (bytes <= cap_) return;
  if (used_ != 0) throw Error("svo_hip::Arena::reserve on a non-empty arena");
  release();
  size_t cap = (size_t)1 << 20;
  while (cap < bytes) cap <<= 1;
  void* h = NULL;
  void* d = NULL;
  check(svo_hip_host_alloc(&h, cap), "svo_hip_host_alloc");
  check(svo_hip_malloc(&d, cap), "svo_hip_malloc");
  h_ = static_cast<uint8_t*>(h);
  d_ = static_cast<uint8_t*>(d);
  cap_ = cap;
}

void Arena::grow(size_t need) {
  // blocks already handed out would dangle: callers size the arena with reserve() first
  throw Error("svo_hip::Arena overflow (" + std::to_string(need) + " > " + std::to_string(cap_) +
              " bytes): reserve() an upper bound before filling");
}

void Arena::setMode(Mode m) {
  if (used_ != 0) throw Error("svo_hip::Arena::setMode on a non-empty arena");
  mode_ = m;
}

void Arena::upload(void* stream) {
  if (mode_ == MAPPED) return;
  if (in_end_) check(svo_hip_memcpy_h2d(d_, h_, in_end_, stream), "arena upload");
}

void Arena::uploadAll(void* stream) {
  if (mode_ == MAPPED) return;
  if (used_) check(svo_hip_memcpy_h2d(d_, h_, used_, stream), "arena upload");
}

void Arena::download(void* stream) {
  if (mode_ == MAPPED) return;
  if (used_ > in_end_) check(svo_hip_memcpy_d2h(h_ + in_end_, d_ + in_end_, used_ - in_end_, stream), "arena download");
}

void Arena::fetchBytes(uint8_t* host_block, size_t bytes, void* stream) {
  if (host_block < h_ || host_block + bytes > h_ + used_) throw Error("svo_hip::Arena::fetch: not an arena block");
  if (mode_ == MAPPED) return;
  check(svo_hip_memcpy_d2h(host_block, d_ + (host_block - h_), bytes, stream), "arena fetch");
}

// ---- Device --------------------------------------------------------------------------
Device::Device() : d_store_(NULL), n_slots_(0), clock_(0), next_lane_index_(0) { std::memset(&layout_, 0, sizeof(layout_)); }
Device::~Device() { shutdown(); }

namespace {
std::mutex g_registry_mut;
std::vector<Device*> g_registry;  // contexts live until process exit
Device* g_last = NULL;
}  // namespace

Device& Device::forGeometry(int width, int height, int n_levels) {
  std::lock_guard<std::mutex> g(g_registry_mut);
  for (size_t i = 0; i < g_registry.size(); ++i) {
    Device* d = g_registry[i];
    if (!d->configured() || (d->layout_.w[0] == width && d->layout_.h[0] == height && d->layout_.n_levels >= n_levels)) {
      if (!d->configured()) d->configure(width, height, n_levels);
      g_last = d;
      return *d;
    }
  }
  Device* d = new Device();
  g_registry.push_back(d);
  d->configure(width, height, n_levels);
  g_last = d;
  return *d;
}

Device& Device::instance() {
  std::lock_guard<std::mutex> g(g_registry_mut);
  if (g_last == NULL) {
    g_last = new Device();
    g_registry.push_back(g_last);
  }
  return *g_last;
}

Lane* Device::makeLane() {
  Lane* l = new Lane();
  check(svo_hip_stream_create(&l->stream), "svo_hip_stream_create");
  l->index = next_lane_index_++;
  check(svo_hip_malloc(&l->d_stage, (size_t)layout_.w[0] * layout_.h[0]), "svo_hip_malloc(stage)");
  l->arena.reserve((size_t)4 << 20);
  // SVO_HIP_ARENA=mapped|mirrored selects how a call's arguments reach the device (Arena)
  const char* mode = std::getenv("SVO_HIP_ARENA");
  if (mode && std::string(mode) == "mapped") l->arena.setMode(Arena::MAPPED);
  else if (mode && std::string(mode) != "mirrored") throw Error("SVO_HIP_ARENA must be 'mapped' or 'mirrored'");
  return l;
}

Lane& Device::lane(int which) {
  std::lock_guard<std::mutex> g(lanes_mut_);
  const std::pair<std::thread::id, int> key(std::this_thread::get_id(), which);
  std::map<std::pair<std::thread::id, int>, Lane*>::iterator it = lanes_.find(key);
  if (it != lanes_.end()) return *it->second;
  if (!d_store_) throw Error("svo_hip::Device::lane before configure()");
  Lane* l = makeLane();
  lanes_[key] = l;
  return *l;
}

void Device::shutdown() {
  {
    std::lock_guard<std::mutex> g(lanes_mut_);
    for (std::map<std::pair<std::thread::id, int>, Lane*>::iterator it = lanes_.begin(); it != lanes_.end(); ++it) {
      Lane& l = *it->second;
      if (l.stream) { svo_hip_stream_sync(l.stream); svo_hip_stream_destroy(l.stream); l.stream = NULL; }
      l.arena.release();
      if (l.d_workspace) { svo_hip_free(l.d_workspace); l.d_workspace = NULL; l.workspace_bytes = 0; }
      if (l.d_stage) { svo_hip_free(l.d_stage); l.d_stage = NULL; }
      delete it->second;
    }
    lanes_.clear();
  }
  if (d_store_) { svo_hip_free(d_store_); d_store_ = NULL; }
  frames_.clear();
  free_slots_.clear();
  n_slots_ = 0;
}

void Device::configure(int width, int height, int n_levels, int n_slots, int device) {
  std::lock_guard<std::mutex> g(frames_mut_);
  shutdown();
  if (svo_hip_device_count() <= 0) throw Error("svo_hip: no HIP device visible (there is no CPU fallback)");
  check(svo_hip_set_device(device), "svo_hip_set_device");
  check(svo_hip_pyr_layout_init(width, height, n_levels, &layout_), "svo_hip_pyr_layout_init");
  const int64_t bytes = svo_hip_pyr_store_bytes(&layout_, n_slots);
  if (bytes < 0) check((int)bytes, "svo_hip_pyr_store_bytes");
  void* p = NULL;
  check(svo_hip_malloc(&p, (size_t)bytes), "svo_hip_malloc(store)");
  d_store_ = static_cast<uint8_t*>(p);
  check(svo_hip_memset(d_store_, 0, (size_t)bytes, NULL), "svo_hip_memset(store)");
  check(svo_hip_stream_sync(NULL), "svo_hip_stream_sync");
  n_slots_ = n_slots;
  for (int s = n_slots - 1; s >= 0; --s) free_slots_.push_back(s);
}

void Device::ensureConfigured(int width, int height, int n_levels) {
  if (d_store_ && layout_.w[0] == width && layout_.h[0] == height && layout_.n_levels >= n_levels) return;
  configure(width, height, n_levels);
}

void Device::beginCall(Lane& lane) {
  {
    std::lock_guard<std::mutex> g(frames_mut_);
    for (size_t i = 0; i < lane.touched.size(); ++i) {  // the previous call of this lane is over: unpin
      std::map<int, Entry>::iterator it = frames_.find(lane.touched[i]);
      if (it != frames_.end() && it->second.pins > 0) --it->second.pins;
    }
    lane.touched.clear();
  }
  std::lock_guard<std::mutex> g(stats_mut_);
  ++stats.calls;
}

Device::Stats Device::statsSnapshot() {
  std::lock_guard<std::mutex> g(stats_mut_);
  return stats;
}

void Device::addStage(int stage, double marshal_us, double device_us, double unmarshal_us, double payload_bytes) {
  std::lock_guard<std::mutex> g(stats_mut_);
  stats.marshal_us[stage] += marshal_us;
  stats.device_us[stage] += device_us;
  stats.unmarshal_us[stage] += unmarshal_us;
  stats.payload_bytes[stage] += payload_bytes;
  ++stats.n[stage];
}

int Device::slotOf(int frame_id, const uint8_t* level0, int stride, Lane& lane) {
  std::lock_guard<std::mutex> g(frames_mut_);
  const int slot = slotOfLocked(frame_id, level0, stride, lane);
  if (slot < 0) throw Error("svo_hip::Device::slotOf: frame " + std::to_string(frame_id) + " is not resident and no image was given");
  return slot;
}

int Device::slotOfLocked(int frame_id, const uint8_t* level0, int stride, Lane& lane) {
  bool mine = false;
  for (size_t i = 0; i < lane.touched.size(); ++i) mine = mine || lane.touched[i] == frame_id;
  std::map<int, Entry>::iterator it = frames_.find(frame_id);
  if (it != frames_.end()) {
    it->second.last_use = ++clock_;
    if (!mine) { ++it->second.pins; lane.touched.push_back(frame_id); }
    return it->second.slot;
  }
  if (level0 == NULL) return -1;
  bool evicted = false;
  if (free_slots_.empty()) {  // evict the least recently used frame no running call has touched
    std::map<int, Entry>::iterator victim = frames_.end();
    for (std::map<int, Entry>::iterator e = frames_.begin(); e != frames_.end(); ++e)
      if (e->second.pins == 0 && (victim == frames_.end() || e->second.last_use < victim->second.last_use)) victim = e;
    if (victim == frames_.end())
      throw Error("svo_hip::Device: the running calls need more than " + std::to_string(n_slots_) +
                  " frames resident; configure() a larger pool");
    free_slots_.push_back(victim->second.slot);
    frames_.erase(victim);
    evicted = true;
  }
  const int slot = free_slots_.back();
  free_slots_.pop_back();
  const double t_up = StageTimer::now();
  try {
    // H2D into the lane's packed staging buffer, then ONE kernel: tiled level 0 + every further level
    check(svo_hip_pyramid_upload_build(&layout_, d_store_, slot, level0, stride, SVO_HIP_HALFSAMPLE_AUTO, lane.d_stage,
                                       lane.stream), "svo_hip_pyramid_upload_build");
    // another lane may consume this slot next: complete the upload before publishing it
    check(svo_hip_stream_sync(lane.stream), "svo_hip_stream_sync(upload)");
  } catch (...) {
    free_slots_.push_back(slot);  // not published: the slot stays free
    throw;
  }
  Entry en;
  en.slot = slot;
  en.last_use = ++clock_;
  en.pins = 1;
  lane.touched.push_back(frame_id);
  frames_[frame_id] = en;
  const double dt = StageTimer::now() - t_up;
  lane.pyr_upload_us += dt;
  std::lock_guard<std::mutex> gs(stats_mut_);
  ++stats.uploads;
  if (evicted) ++stats.evictions;
  stats.pyr_upload_us += dt;
  return slot;
}

double StageTimer::now() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int Device::scratchSlotOf(const uint8_t* image, int w, int h, int stride, int* level_out, Lane& lane) {
  int level = -1;
  for (int l = 0; l < layout_.n_levels; ++l)
    if (layout_.w[l] == w && layout_.h[l] == h) { level = l; break; }
  if (level < 0)
    throw Error("svo_hip::Device: a " + std::to_string(w) + "x" + std::to_string(h) + " image matches no pyramid level of the device context");
  // scratch frames live under negative ids, one per (lane, level); look-up-or-create is ONE critical section
  // (an entry found and then evicted by another lane before it is pinned would otherwise be re-created
  // from a NULL image)
  const int id = -1 - (lane.index * SVO_HIP_MAX_LEVELS + level);
  int slot;
  {
    std::lock_guard<std::mutex> g(frames_mut_);
    slot = slotOfLocked(id, NULL, 0, lane);  // hit: pins it for this call
    if (slot < 0) {
      std::vector<uint8_t> blank((size_t)layout_.w[0] * layout_.h[0], 0);
      slot = slotOfLocked(id, &blank[0], layout_.w[0], lane);  // synchronises: `blank` may go
    }
  }
  // the lane's stream orders this upload behind the lane's earlier kernels and before its next ones; the
  // staging buffer is the lane's own (the caller holds lane.mut)
  check(svo_hip_pyramid_upload_level(&layout_, d_store_, slot, level, image, stride, lane.d_stage, lane.stream),
        "svo_hip_pyramid_upload_level");
  *level_out = level;
  return slot;
}

void Device::forget(int frame_id) {
  std::lock_guard<std::mutex> g(frames_mut_);
  std::map<int, Entry>::iterator it = frames_.find(frame_id);
  if (it == frames_.end()) return;
  free_slots_.push_back(it->second.slot);
  frames_.erase(it);
}

void* Device::workspace(Lane& lane, int n_trials) {
  const size_t need = svo_hip_match_workspace_bytes(n_trials);
  if (need > lane.workspace_bytes) {
    if (lane.d_workspace) { check(svo_hip_stream_sync(lane.stream), "sync"); svo_hip_free(lane.d_workspace); }
    size_t cap = (size_t)1 << 20;
    while (cap < need) cap <<= 1;
    check(svo_hip_malloc(&lane.d_workspace, cap), "svo_hip_malloc(workspace)");
    lane.workspace_bytes = cap;
  }
  return lane.d_workspace;
}

}  // namespace svo_hip
