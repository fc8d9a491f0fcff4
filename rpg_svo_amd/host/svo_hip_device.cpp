// svo_hip_device.cpp -- see svo_hip_device.h.
#include "svo_hip_device.h"

#include <atomic>
#include <chrono>
#include <cstdlib>

#include <cstring>

namespace svo_hip {

void check(int code, const char* what) {
  if (code >= 0) return;
  throw Error(std::string(what) + ": " + svo_hip_strerror(code) + " (hip error " +
              std::to_string(svo_hip_last_hip_error()) + ")");
}

void spinUntil(const volatile int32_t* flag, int32_t value, void* stream) {
  const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0; *flag != value; ++spins) {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
    if ((spins & 0xfff) == 0xfff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(1)) {
      check(svo_hip_stream_sync(stream), "svo_hip_stream_sync(signal)");
      if (*flag != value) throw Error("svo_hip::spinUntil: the stream drained without the signal being stored");
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);  // the results the signal vouches for are read after it
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
  const size_t n = mode_ == HYBRID && outputs_ ? in_end_ : used_;  // HYBRID: the in/out blocks are host memory
  if (n) check(svo_hip_memcpy_h2d(d_, h_, n, stream), "arena upload");
}

void Arena::download(void* stream) {
  if (mode_ != MIRRORED) return;
  if (used_ > in_end_) check(svo_hip_memcpy_d2h(h_ + in_end_, d_ + in_end_, used_ - in_end_, stream), "arena download");
}

void Arena::downloadRange(size_t begin, size_t end, void* stream) {
  if (begin > end || end > used_) throw Error("svo_hip::Arena::downloadRange: not inside the arena");
  if (mode_ == MAPPED || begin == end) return;
  if (mode_ == HYBRID) {  // only what lies in the mirrored part
    const size_t mirrored_end = outputs_ ? in_end_ : used_;
    if (begin >= mirrored_end) return;
    end = end < mirrored_end ? end : mirrored_end;
  }
  check(svo_hip_memcpy_d2h(h_ + begin, d_ + begin, end - begin, stream), "arena download");
}

void Arena::fetchBytes(uint8_t* host_block, size_t bytes, void* stream) {
  if (host_block < h_ || host_block + bytes > h_ + used_) throw Error("svo_hip::Arena::fetch: not an arena block");
  if (mode_ == MAPPED) return;
  if (mode_ == HYBRID && outputs_ && (size_t)(host_block - h_) >= in_end_) return;  // an output block: written in place
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

namespace {
void destroyLane(Lane* l) {  // everything makeLane() / workspace() may have given it; streams are drained first
  if (l->stream) { svo_hip_stream_sync(l->stream); svo_hip_stream_destroy(l->stream); l->stream = NULL; }
  if (l->stream_next) { svo_hip_stream_sync(l->stream_next); svo_hip_stream_destroy(l->stream_next); l->stream_next = NULL; }
  if (l->ev_results) { svo_hip_event_destroy(l->ev_results); l->ev_results = NULL; }
  l->deferred = nullptr;  // its owner is about to lose the device; nothing is written back
  l->early_drop = nullptr;
  l->early_hook = nullptr;
  l->arena.release();
  l->arena_chain.release();
  if (l->d_workspace) { svo_hip_free(l->d_workspace); l->d_workspace = NULL; l->workspace_bytes = 0; }
  if (l->d_stage) { svo_hip_free(l->d_stage); l->d_stage = NULL; }
  delete l;
}
}  // namespace

Lane* Device::makeLane() {
  // the environment switches first: a bad value must not leave streams and buffers behind
  // SVO_HIP_ARENA=hybrid|mapped|mirrored selects how a call's arguments reach the device (Arena)
  Arena::Mode arena_mode = Arena::HYBRID;  // the default
  if (const char* mode = std::getenv("SVO_HIP_ARENA")) {
    const std::string m(mode);
    if (m == "mapped") arena_mode = Arena::MAPPED;
    else if (m == "mirrored") arena_mode = Arena::MIRRORED;
    else if (m != "hybrid") throw Error("SVO_HIP_ARENA must be 'hybrid', 'mirrored' or 'mapped'");
  }
  // SVO_HIP_PIN_HOST=1: the thread that gets a lane is bound to the CPUs next to the GPU (include/svo_hip.h).  Off by default:
  // binding ONE thread of a process measured no gain (0.274 against 0.265 ms per frame, profiles/r06ab_*) where binding the whole
  // process from outside -- `taskset` / `numactl --cpunodebind`, the runtime's own threads included -- is worth 6 % over an
  // unpinned one (profiles/r06aa_*): a deployment choice, like GPU_MAX_HW_QUEUES.
  {
    const char* pin = std::getenv("SVO_HIP_PIN_HOST");
    if (pin && pin[0] == '1') svo_hip_pin_calling_thread();
  }
  Lane* l = new Lane();
  try {
    check(svo_hip_stream_create(&l->stream), "svo_hip_stream_create");
    check(svo_hip_stream_create(&l->stream_next), "svo_hip_stream_create");
    check(svo_hip_event_create(&l->ev_results), "svo_hip_event_create");
    check(svo_hip_malloc(&l->d_stage, (size_t)layout_.w[0] * layout_.h[0]), "svo_hip_malloc(stage)");
    l->arena.reserve((size_t)4 << 20);
    l->arena.setMode(arena_mode);
    l->arena_chain.reserve((size_t)1 << 20);
    l->arena_chain.setMode(arena_mode);  // (never endInputs(): every block is an input -- device mirror, or host memory when mapped)
  } catch (...) {
    destroyLane(l);  // not in lanes_ yet: shutdown() would never see it
    throw;
  }
  l->index = next_lane_index_++;  // only a lane that exists takes an index
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
    for (std::map<std::pair<std::thread::id, int>, Lane*>::iterator it = lanes_.begin(); it != lanes_.end(); ++it)
      destroyLane(it->second);
    lanes_.clear();
  }
  if (d_store_) { svo_hip_free(d_store_); d_store_ = NULL; }
  for (size_t s = 0; s < slot_ready_.size(); ++s)
    if (slot_ready_[s]) svo_hip_event_destroy(slot_ready_[s]);
  slot_ready_.clear();
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
  slot_ready_.assign((size_t)n_slots, NULL);
  for (int s = 0; s < n_slots; ++s) check(svo_hip_event_create(&slot_ready_[s]), "svo_hip_event_create");
}

void Device::ensureConfigured(int width, int height, int n_levels) {
  if (d_store_ && layout_.w[0] == width && layout_.h[0] == height && layout_.n_levels >= n_levels) return;
  configure(width, height, n_levels);
}

namespace {
std::atomic<int> g_deferred_mapping(-1);  // -1: ask the environment
void runDeferred(Lane& lane) {
  if (!lane.deferred) return;
  std::function<void()> f;
  f.swap(lane.deferred);
  f();
}
}  // namespace

bool Device::deferredMapping() {
  int m = g_deferred_mapping.load();
  if (m < 0) {
    const char* v = std::getenv("SVO_HIP_MAPPER");
    if (v && std::string(v) != "deferred" && std::string(v) != "sync") throw Error("SVO_HIP_MAPPER must be 'deferred' or 'sync'");
    m = v && std::string(v) == "deferred";
    g_deferred_mapping.store(m);
  }
  return m != 0;
}
void Device::setDeferredMapping(bool on) { g_deferred_mapping.store(on ? 1 : 0); }

void Device::finish(Lane& lane) { check(svo_hip_stream_sync(lane.stream), "svo_hip_stream_sync"); }

Lane* Device::findLane(int which) {
  std::lock_guard<std::mutex> g(lanes_mut_);
  std::map<std::pair<std::thread::id, int>, Lane*>::iterator it = lanes_.find(std::make_pair(std::this_thread::get_id(), which));
  return it != lanes_.end() ? it->second : NULL;
}

void Device::joinDeferred(int which_lane) {
  Lane* l = findLane(which_lane);
  if (!l) return;
  std::lock_guard<std::mutex> g(l->mut);
  runDeferred(*l);
}

void Device::joinDeferredAll() {
  std::vector<Device*> all;
  {
    std::lock_guard<std::mutex> g(g_registry_mut);
    all = g_registry;
  }
  for (size_t i = 0; i < all.size(); ++i)
    for (int which = 0; which < N_LANES; ++which) all[i]->joinDeferred(which);
}

void Device::beginCall(Lane& lane) {
  runDeferred(lane);  // the caller holds lane.mut
  if (lane.early_drop) {  // an early update of the depth filter nobody has claimed: its blocks are about to be overwritten
    std::function<void()> drop;
    drop.swap(lane.early_drop);
    drop();
  }
  // work a previous call left running reads and writes the arena this call is about to refill
  if (lane.spec.in_flight) {
    lane.spec.in_flight = false;
    check(svo_hip_stream_sync(lane.spec.stream), "svo_hip_stream_sync(speculation)");
  }
  if (lane.spec.valid) {
    lane.spec.valid = false;
    countSpeculation(false);
  }
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

void Device::countSpeculation(bool hit) {
  std::lock_guard<std::mutex> g(stats_mut_);
  if (hit) ++stats.spec_hits; else ++stats.spec_misses;
}

void Device::countChain(bool hit, int why) {
  std::lock_guard<std::mutex> g(stats_mut_);
  if (hit) ++stats.chain_hits;
  else {
    ++stats.chain_misses;
    ++stats.chain_miss_why[why >= 0 && why < 6 ? why : 5];
  }
}

void Device::countEarlyMapping(bool hit) {
  std::lock_guard<std::mutex> g(stats_mut_);
  if (hit) ++stats.early_map_hits; else ++stats.early_map_misses;
}

void Device::countEarlyTwoPhase() {
  std::lock_guard<std::mutex> g(stats_mut_);
  ++stats.early_map_two_phase;
}

bool Device::earlyMappingEnabled() {
  static const bool on = [] {
    const char* v = std::getenv("SVO_HIP_EARLY_MAPPER");
    return !(v && v[0] == '0');
  }();
  return on;
}

bool Device::chainEnabled() {
  static const bool on = [] {
    const char* v = std::getenv("SVO_HIP_CHAIN");
    return !(v && v[0] == '0');
  }();
  return on;
}

bool Device::speculationEnabled() {
  static const bool on = [] {
    const char* v = std::getenv("SVO_HIP_SPECULATE");
    return !(v && v[0] == '0');
  }();
  return on;
}

void Device::addStage(int stage, double marshal_us, double device_us, double unmarshal_us, double payload_bytes, bool count) {
  std::lock_guard<std::mutex> g(stats_mut_);
  stats.marshal_us[stage] += marshal_us;
  stats.device_us[stage] += device_us;
  stats.unmarshal_us[stage] += unmarshal_us;
  stats.payload_bytes[stage] += payload_bytes;
  if (count) ++stats.n[stage];
}

int Device::slotOf(int frame_id, const uint8_t* level0, int stride, Lane& lane) {
  std::lock_guard<std::mutex> g(frames_mut_);
  const int slot = slotOfLocked(frame_id, level0, stride, lane);
  if (slot < 0) throw Error("svo_hip::Device::slotOf: frame " + std::to_string(frame_id) + " is not resident and no image was given");
  return slot;
}

int Device::slotOfLocked(int frame_id, const uint8_t* level0, int stride, Lane& lane, bool wait_upload) {
  bool mine = false;
  for (size_t i = 0; i < lane.touched.size(); ++i) mine = mine || lane.touched[i] == frame_id;
  std::map<int, Entry>::iterator it = frames_.find(frame_id);
  if (it != frames_.end()) {
    Entry& e = it->second;
    e.last_use = ++clock_;
    if (!mine) { ++e.pins; lane.touched.push_back(frame_id); }
    if (!e.settled && e.owner != lane.index) {  // uploaded on another stream and not known to be complete
      const int done = svo_hip_event_query(slot_ready_[e.slot]);
      check(done, "svo_hip_event_query(upload)");
      if (done) e.settled = true;
      else check(svo_hip_stream_wait_event(lane.stream, slot_ready_[e.slot]), "svo_hip_stream_wait_event(upload)");
    }
    return e.slot;
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
    // no host sync: this lane's kernels follow in stream order, other lanes go through the slot's event
    check(svo_hip_event_record(slot_ready_[slot], lane.stream), "svo_hip_event_record(upload)");
    if (wait_upload) check(svo_hip_stream_sync(lane.stream), "svo_hip_stream_sync(upload)");
  } catch (...) {
    free_slots_.push_back(slot);  // not published: the slot stays free
    throw;
  }
  Entry en;
  en.slot = slot;
  en.last_use = ++clock_;
  en.pins = 1;
  en.owner = lane.index;
  en.settled = wait_upload;
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
      slot = slotOfLocked(id, &blank[0], layout_.w[0], lane, true);  // synchronises: `blank` may go
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
