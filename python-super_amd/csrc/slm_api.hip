// slm_api.hip -- the C ABI of libsuper_lm.so (include/super_lm.h): slot workspaces,
// the on-device LM loop, parity entry points.  Host side only orchestrates launches.
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <algorithm>
#include <vector>
#include <cstdio>
#include <cstring>

#include "slm_common.h"
#include "slm_prep.h"

// launchers defined next to their kernels
void launch_data_grad(const FrameDev*, int, int, int, double, hipStream_t);
void launch_data_grad_pairs(const FrameDev*, int, int, int, double, hipStream_t);
void launch_data_loss(const FrameDev*, int, int, int, double, int, hipStream_t);
void launch_data_resid(const FrameDev*, int, int, int, double, double*, uint8_t*, int32_t*, hipStream_t);
void launch_data_gram(const FrameDev*, int, int, double, int, hipStream_t, const int* reuse = nullptr);
void launch_begin_and_gram(const FrameDev*, int, int, double, hipStream_t, const int* reuse, int dag_cut);
void launch_data_eval(const FrameDev*, int, int, double, int mode, hipStream_t, const int* reuse = nullptr);
void launch_band_assemble(const FrameDev*, int, int, hipStream_t);
void launch_reg_grad(const FrameDev*, int, int, int, double, int, double, hipStream_t);
void launch_front_assemble(const FrameDev*, int, int, hipStream_t);
void launch_pair_reduce(const FrameDev*, int, int, hipStream_t);
void launch_pair_scatter(const FrameDev*, int, int, hipStream_t);
void launch_reg_grad_nd(const FrameDev*, int, int, int, double, int, double, hipStream_t);
void launch_front_load_rhs(const FrameDev*, int, int, hipStream_t);
void launch_iter_begin_nd(const FrameDev*, int, hipStream_t, const int* reuse, int dag_cut);
void launch_front_solve(const FrameDev*, int, const NDLevelSched*, int, double, hipStream_t);
int launch_front_solve_dag(const FrameDev*, int, int, double, hipStream_t, int cut, bool reset, bool check);
int dag_device_setup(int dev, int* xcd8_out);
int dag_last_mode();
void launch_after_solve(const FrameDev*, int, int, int, int, double, int, double, int, hipStream_t);
hipError_t set_dag_timeout_ticks(long long);
void launch_dag_abort_check(const FrameDev*, int, hipStream_t);
void launch_front_levels(const FrameDev*, int, const NDLevelSched*, int, int, int, double, hipStream_t);
void launch_reg_loss(const FrameDev*, int, int, int, double, int, double, int, hipStream_t);
void launch_bandwidth(const slm_frame&, int*, hipStream_t);
void launch_band_solve(const FrameDev*, int, int, int, double, hipStream_t);
void launch_band_to_dense(const FrameDev*, int, double*, hipStream_t);
void launch_dense_to_band(const FrameDev*, const double*, const double*, hipStream_t);
void launch_init_slot(const FrameDev*, int, int, const slm_config&, hipStream_t);
void launch_iter_begin(const FrameDev*, int, hipStream_t);
void launch_pack_nodes(const FrameDev*, int, int, hipStream_t);
void launch_make_trial(const FrameDev*, int, int, hipStream_t);
void launch_pack_target(int, const float*, const float*, float4*, hipStream_t);
void launch_pack_target_px(int, const int*, const uint8_t*, const float*, const float*, float4*, hipStream_t);
void launch_accept(const FrameDev*, int, int, int, int, hipStream_t, int* reuse = nullptr, int eval_pass = 0);
void launch_loss_out(const FrameDev*, int, int, double*, hipStream_t);
void launch_zero_reg_part(const FrameDev*, int, int, hipStream_t);
void launch_update(int, int, int, float*, float*, const int*, const float*, float*, float*, const double*,
                   hipStream_t);
void launch_update64(int, int, int, double*, double*, const int*, const double*, double*, double*, const double*,
                     hipStream_t);
void launch_knn(int, int, int, int, const float*, const float*, int*, float*, hipStream_t);
void launch_knn64(int, int, int, int, const double*, const double*, const int*, const int*, int*, double*, int*,
                  hipStream_t);
void launch_knn_weights64(int, int, int, const int*, const double*, const double*, int, const double*, const double*,
                          double*, uint8_t*, hipStream_t);
void launch_knn_weights(int, int, int, const int*, const float*, const float*, float*, uint8_t*,
                        hipStream_t);

static thread_local std::string g_err;

#define HIPCHK(expr)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) {                                                               \
      g_err = std::string(#expr) + ": " + hipGetErrorString(e_);                          \
      return SLM_ERR_HIP;                                                                 \
    }                                                                                     \
  } while (0)

static int fail(int code, const char* msg) {
  g_err = msg;
  return code;
}

// other translation units (slm_gf.hip) report through the same slm_last_error()
void slm_set_error_text(const char* msg) { g_err = msg; }

namespace {
// Host buffer for per-frame device -> host read-backs, PINNED (hipHostMalloc, grow-only).  A read-back into pageable
// memory (a std::vector) goes through the runtime's staging path, and several bind workers doing that at once were
// seen to block for 5-7 ms together once in ~50 steps (tools/studies/stall_hunt.py: the whole rare stall of a bench
// step sat in this one hipMemcpyAsync + hipStreamSynchronize); a pinned destination is a plain DMA.
template <typename T>
struct PinnedBuf {
  T* p = nullptr;
  size_t cap = 0, n = 0;
  hipError_t resize(size_t need) {
    if (need > cap) {
      if (p) (void)hipHostFree(p);
      p = nullptr;
      cap = 0;
      const size_t want = need + need / 8 + 16;
      const hipError_t e = hipHostMalloc((void**)&p, want * sizeof(T), hipHostMallocDefault);
      if (e != hipSuccess) return e;
      cap = want;
    }
    n = need;
    return hipSuccess;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = n = 0;
  }
  T* data() { return p; }
  const T* data() const { return p; }
  size_t size() const { return n; }
  T* begin() { return p; }
  T* end() { return p + n; }
  T& operator[](size_t i) { return p[i]; }
  const T& operator[](size_t i) const { return p[i]; }
};

constexpr int kLossBlocks = 512;   // data-loss partial sums per slot
constexpr int kRegBlocksMax = 64;

struct Slot {
  FrameDev h{};                 // host mirror of the device descriptor
  size_t cap_beta = 0, cap_vec = 0, cap_band = 0, cap_linv = 0, cap_npk = 0, cap_tpn = 0, cap_tpx = 0, cap_ev = 0;
  V1Plan plan;                  // tuple-sorted assembly buffers (grow-only)
  PairPlan pplan;               // K-generic pair path (num_neighbors != 4): pair keys, per-surfel pair indices, surfel order
  // nested-dissection plan: host copy + device mirrors (grow-only)
  NDPlanHost nd;
  uint64_t nd_hash = 0;         // hash of the coupled-pair list + node KNN of the frame bound last
  uint64_t nd_knn_hash = 0;     // hash of (J, K_ED, node KNN) the cached plan was built for
  bool nd_valid = false;
  // The plan is built for a SUPERSET of the frame's coupled node pairs (the union of the pair lists seen
  // since the node graph last changed): surfels that appear / disappear change the list a little from
  // frame to frame, and a frame whose pairs are all in the plan only needs its own pair -> destination
  // table (a merge of two sorted lists), not a new symbolic analysis (1.7 ms at C2).
  std::vector<uint32_t> plan_pairs;     // sorted keys the plan's block_dest is indexed by
  std::vector<NDDest> cur_dest;         // destinations of the bound frame's pairs
  NDDest* d_cur_dests = nullptr;
  size_t cap_cur_dests = 0;
  int cur_n_blocks = -1;
  PinnedBuf<uint32_t> h_pairs;   // read-backs of the bind (pinned)
  PinnedBuf<int32_t> h_knn;
  PinnedBuf<float> h_pts;
  PinnedBuf<double> h_pts64;
  FrameDev* h_pin = nullptr;       // pinned mirror of `h` for the asynchronous descriptor upload of a bind
  NDFront* d_fronts = nullptr;
  NDTileItem* d_items = nullptr;   // work lists of the pull-form kernels
  size_t cap_items = 0;
  uint8_t* d_tile_kind = nullptr;  // FrameDev::tile_kind / zero_tiles (slm_common.h): per tile 1 = pure fill; the tiles to zero
  long long* d_zero_tiles = nullptr;
  size_t cap_tile_kind = 0, cap_zero_tiles = 0;
  int n_piv_tiles = 0, n_pure_tiles = 0;  // pivot-column tiles of the plan / of them pure fill (slm_get_plan_info)
  std::vector<uint8_t> h_tile_kind;      // host sources of the two uploads (kept: the copies are asynchronous)
  std::vector<long long> h_zero_tiles;
  int32_t* d_ints = nullptr;    // level_start | nodes | eamap | node_front | node_pos | ... | dag_tasks | front_nin
  long long* d_dag_trace = nullptr; // diagnostics
  size_t cap_dag_trace = 0;
  int32_t* d_dag_flags = nullptr;   // persistent task-graph solver: ticket, counters, per-tile / per-column flags
  size_t cap_dag_flags = 0;
  NDDest* d_dests = nullptr;    // block_dest | pair_dest
  double *ftiles = nullptr, *fvec = nullptr, *flinv = nullptr, *fmail = nullptr;
  double* pairbuf = nullptr;   // sharded frames: per-pair sums to exchange
  double *band = nullptr, *linv = nullptr;   // block-banded path (allocated on demand)
  bool band_ready = false;
  size_t cap_pairbuf = 0;
  size_t cap_fronts = 0, cap_ints = 0, cap_dests = 0, cap_ftiles = 0, cap_fvec = 0, cap_flinv = 0, cap_fmail = 0;
  // slm_prepare_model: the model-side half of a bind, done ahead of the frame's target (possibly still running on the
  // solver's worker thread: prep_ticket)
  bool model_ready = false;        // the model part below is complete and valid for prep_model
  slm_frame prep_model{};          // the model fields it was prepared for
  unsigned long long prep_ticket = 0;   // job number on the worker (0: none pending)
  int prep_rc = SLM_OK;
  bool prep_recorded = false;      // prep_done was recorded on the solver's stream and nothing has waited for it yet
  std::string prep_err;
  hipEvent_t prep_fork = nullptr, prep_done = nullptr;
};
}  // namespace

// One persistent host thread running queued jobs in order (slm_prepare_model): created on first use, joined in slm_destroy.
namespace {
class AsyncWorker {
 public:
  ~AsyncWorker() { shutdown(); }
  // returns the job's ticket (> 0); throws only from thread creation / allocation, before the job is queued
  unsigned long long submit(std::function<void()> job) {
    std::lock_guard<std::mutex> lk(m_);
    if (!started_) {
      thread_ = std::thread([this] { loop(); });
      started_ = true;
    }
    jobs_.push_back(std::move(job));
    cv_work_.notify_one();
    return ++submitted_;
  }
  void wait(unsigned long long ticket) {
    std::unique_lock<std::mutex> lk(m_);
    cv_done_.wait(lk, [&] { return completed_ >= ticket; });
  }
  void shutdown() {
    {
      std::lock_guard<std::mutex> lk(m_);
      if (!started_) return;
      stop_ = true;
    }
    cv_work_.notify_all();
    if (thread_.joinable()) thread_.join();
    started_ = false;
  }

 private:
  void loop() {
    for (;;) {
      std::function<void()> job;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_work_.wait(lk, [&] { return stop_ || !jobs_.empty(); });
        if (jobs_.empty()) return;     // (stop: queued jobs are still run first)
        job = std::move(jobs_.front());
        jobs_.erase(jobs_.begin());
      }
      job();
      {
        std::lock_guard<std::mutex> lk(m_);
        ++completed_;
      }
      cv_done_.notify_all();
    }
  }
  std::thread thread_;
  std::mutex m_;
  std::condition_variable cv_work_, cv_done_;
  std::vector<std::function<void()>> jobs_;
  unsigned long long submitted_ = 0, completed_ = 0;
  bool started_ = false, stop_ = false;
};
}  // namespace

// Persistent host workers of slm_bind_frames: created on first use, parked on a condition variable between calls, joined
// in slm_destroy.  (One std::thread per frame and call cost a spawn + join per tracking step inside the timed region
// and could throw through the extern "C" boundary.)
namespace {
class BindPool {
 public:
  ~BindPool() { shutdown(); }
  // runs job(1) .. job(n - 1) on the workers and job(0) on the caller; returns when all are done.  Throws only from
  // ensure() (thread creation), before any job has started.
  void run(int n, const std::function<void(int)>& job) {
    ensure(n - 1);
    {
      std::lock_guard<std::mutex> lk(m_);
      job_ = &job;
      n_active_ = n - 1;
      pending_ = n - 1;
      ++generation_;
    }
    cv_work_.notify_all();
    job(0);
    std::unique_lock<std::mutex> lk(m_);
    cv_done_.wait(lk, [&] { return pending_ == 0; });
    job_ = nullptr;
  }
  void shutdown() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_work_.notify_all();
    for (std::thread& t : threads_)
      if (t.joinable()) t.join();
    threads_.clear();
  }

 private:
  void ensure(int n_workers) {
    while ((int)threads_.size() < n_workers) {
      const int id = (int)threads_.size();      // worker id runs job(id + 1)
      threads_.emplace_back([this, id] { loop(id); });
    }
  }
  void loop(int id) {
    int seen = 0;
    for (;;) {
      const std::function<void(int)>* job = nullptr;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_work_.wait(lk, [&] { return stop_ || generation_ != seen; });
        if (stop_) return;
        seen = generation_;
        if (id < n_active_) job = job_;
      }
      if (job) {
        (*job)(id + 1);
        std::lock_guard<std::mutex> lk(m_);
        if (--pending_ == 0) cv_done_.notify_all();
      }
    }
  }
  std::vector<std::thread> threads_;
  std::mutex m_;
  std::condition_variable cv_work_, cv_done_;
  const std::function<void(int)>* job_ = nullptr;
  int generation_ = 0, n_active_ = 0, pending_ = 0;
  bool stop_ = false;
};
}  // namespace

struct slm_solver {
  PrepBuffers* prep = nullptr;
  BindPool bind_pool;
  // slm_prepare_model: worker thread, its stream and scratch buffers
  AsyncWorker prep_worker;
  PrepBuffers* prep_async = nullptr;
  hipStream_t prep_stream = nullptr;
  // slm_bind_frames: one worker (scratch buffers + stream + events) per frame bound concurrently
  std::vector<PrepBuffers*> bind_prep;
  std::vector<hipStream_t> bind_streams;
  std::vector<hipEvent_t> bind_events;
  std::mutex band_mutex;        // the bandwidth read-back buffer of ensure_band is shared
  hipEvent_t drain_event = nullptr;   // slm_bind_frames' wait for the caller's stream (polled, see there)
  double drain_est_ms = 0.0;          // how long that wait took last time
  int last_solver_form = -1;    // diagnostics: 0 per-level launches, 1 task graph, 2 hybrid (slm_debug_last_solver_form)
  int last_dag_mode = -1;       // diagnostics: mode word of the last task-graph launch (bit 0: XCD-affine), -1 none (slm_debug_last_dag_mode)
  bool no_reuse = false;        // SLM_NO_REUSE=1 (tests): every Jacobian pass recomputes its records, also after a reject
  bool hybrid_batches = true;   // solver_path 0, larger batches: per-level launches + task graph for the top levels
  long dag_max_nodes = 8000;    // solver_path 0: launches of at most this many frames x nodes run as ONE task graph (SLM_DAG_MAX_NODES)
  bool fuse_begin = true;       // SLM_FUSE_BEGIN=0 (A/B, tests): k_iter_begin_nd as a launch of its own in front of the Jacobian pass, as in rounds 1-5
  bool pure_fill = true;        // SLM_PURE_FILL=0 (tests): every pivot-column tile is zeroed and read-modify-written, as in rounds 1-4
  bool profile = false;
  std::vector<hipEvent_t> ev_pool;            // recycled events
  std::vector<std::vector<hipEvent_t>> ev_runs;  // per recorded iteration: SLM_PH_COUNT+1 events
  slm_config cfg{};
  std::vector<Slot> slots;
  FrameDev* frames_dev = nullptr;
  int* bw_dev = nullptr;
  int* bw_host = nullptr;       // pinned
  int* reuse_dev = nullptr;     // per slot: 1 after a rejected iteration -- the next Jacobian pass of the LM loop reuses the
                                // records of the last one (k_accept writes, k_data_gram / k_iter_begin_nd read)
  int rank = 0, world = 1;      // surfel sharding of every frame (slm_set_shard)
  bool shard_mode = false;      // slm_set_shard was called (world == 1 included): slots carry the exchange buffers
};
namespace {
bool solve_is_task_graph(const slm_solver* s, int n, int J);   // (below: which form of the solver a launch of n frames of J nodes takes)
}

// diagnostics (slm_debug_counters): device reallocations and symbolic analyses since the library was loaded
static std::atomic<long long> g_reallocs{0}, g_realloc_bytes{0}, g_plan_builds{0}, g_plan_reuses{0}, g_plan_fill_hits{0};   // (binds may run on worker threads)

template <typename T>
static hipError_t grow(T*& p, size_t& cap, size_t need) {
  if (need <= cap) return hipSuccess;
  ++g_reallocs;
  g_realloc_bytes += (long long)((need + need / 8) * sizeof(T));
  if (p) {
    hipError_t e = hipFree(p);
    if (e != hipSuccess) return e;
    p = nullptr;
  }
  size_t want = need + need / 8;
  hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
  if (e == hipSuccess) cap = want;
  return e;
}
template <typename T>
static hipError_t grow(GP<T>& p, size_t& cap, size_t need) {   // (a descriptor field, slm_common.h GP)
  T* q = p;
  const hipError_t e = grow(q, cap, need);
  p = q;
  return e;
}

static int check_slots(slm_solver* s, int first, int n);
static int check_slots_fwd(slm_solver* s, int first, int n) { return check_slots(s, first, n); }

extern "C" {

const char* slm_last_error(void) { return g_err.c_str(); }

int slm_debug_last_solver_form(slm_solver* s) { return s ? s->last_solver_form : -1; }
int slm_debug_last_dag_mode(slm_solver* s) { return s ? s->last_dag_mode : -1; }

int slm_debug_counters(int64_t out[4]) {
  if (!out) return fail(SLM_ERR_INVALID, "slm_debug_counters: null output");
  out[0] = g_reallocs;
  out[1] = g_realloc_bytes;
  out[2] = g_plan_builds;
  out[3] = g_plan_reuses;
  return SLM_OK;
}

int slm_debug_dag_timeout(int64_t ticks) {
  if (ticks < 0) return fail(SLM_ERR_INVALID, "slm_debug_dag_timeout: negative");
  HIPCHK(set_dag_timeout_ticks(ticks == 0 ? 300000000ll : (long long)ticks));
  return SLM_OK;
}

int slm_debug_dag_abort(slm_solver* s, int32_t n_frames, void* stream) {
  int rc = check_slots_fwd(s, 0, n_frames);
  if (rc) return rc;
  for (int i = 0; i < n_frames; ++i)
    if (!s->slots[i].h.nd_ready) return fail(SLM_ERR_UNSUPPORTED, "slm_debug_dag_abort: every slot needs a nested-dissection plan");
  hipStream_t st = (hipStream_t)stream;
  launch_dag_abort_check(s->frames_dev, n_frames, st);
  launch_accept(s->frames_dev, n_frames, s->cfg.phase_test, 0, std::max(s->cfg.num_iterations, 1), st);
  HIPCHK(hipGetLastError());
  return SLM_OK;
}

int slm_debug_dag_trace(slm_solver* s, int32_t slot, int32_t on, void* stream) {
  if (!s || slot < 0 || slot >= (int)s->slots.size()) return fail(SLM_ERR_INVALID, "slm_debug_dag_trace: bad argument");
  Slot& sl = s->slots[slot];
  if (!sl.h.bound || !sl.h.nd_ready) return fail(SLM_ERR_UNBOUND, "slm_debug_dag_trace: no nested-dissection plan bound");
  hipStream_t st = (hipStream_t)stream;
  if (on) {
    HIPCHK(grow(sl.d_dag_trace, sl.cap_dag_trace, (size_t)24 * sl.h.n_dag_tasks + 8));
    HIPCHK(hipMemsetAsync(sl.d_dag_trace, 0, sizeof(long long) * 24 * (size_t)sl.h.n_dag_tasks, st));
  }
  sl.h.dag_trace = on ? sl.d_dag_trace : nullptr;
  HIPCHK(hipMemcpyAsync(s->frames_dev + slot, &sl.h, sizeof(FrameDev), hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));
  return SLM_OK;
}

int slm_debug_read(slm_solver* s, int32_t slot, int32_t what, double* host_out, int64_t max_doubles,
                   int64_t* n_doubles, void* stream) {
  if (!s || slot < 0 || slot >= (int)s->slots.size() || !n_doubles) return fail(SLM_ERR_INVALID, "slm_debug_read: bad argument");
  const Slot& sl = s->slots[slot];
  if (!sl.h.bound) return fail(SLM_ERR_UNBOUND, "slm_debug_read: slot used before slm_bind_frame");
  const double* src = nullptr;
  int64_t n = 0;
  switch (what) {
    case 0: src = sl.ftiles; n = sl.h.nd_ready ? sl.nd.tile_doubles : 0; break;
    case 1: src = sl.fvec; n = sl.h.nd_ready ? sl.nd.vec_doubles : 0; break;
    case 2: src = sl.flinv; n = sl.h.nd_ready ? sl.nd.linv_doubles : 0; break;
    case 3: src = sl.h.delta; n = sl.h.P; break;
    case 4: src = reinterpret_cast<const double*>(sl.h.dag_trace.get()); n = sl.h.dag_trace ? 24 * (int64_t)sl.h.n_dag_tasks : 0; break;
    case 5: src = reinterpret_cast<const double*>(sl.h.dag_tasks.get()); n = sl.h.nd_ready ? sl.h.n_dag_tasks : 0; break;
    case 6: src = reinterpret_cast<const double*>(sl.h.dag_top_tasks.get()); n = sl.h.nd_ready ? sl.h.n_dag_top_tasks : 0; break;
    default: return fail(SLM_ERR_INVALID, "slm_debug_read: unknown buffer");
  }
  *n_doubles = n;
  const int64_t m = n < max_doubles ? n : max_doubles;
  if (m > 0 && host_out && src) HIPCHK(hipMemcpyAsync(host_out, src, sizeof(double) * m, hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  return SLM_OK;
}

int slm_debug_read_plan(slm_solver* s, int32_t slot, int32_t what, void* host_out, int64_t max_bytes, int64_t* n_bytes, void* stream) {
  if (!s || slot < 0 || slot >= (int)s->slots.size() || !n_bytes) return fail(SLM_ERR_INVALID, "slm_debug_read_plan: bad argument");
  int rc = check_slots_fwd(s, slot, 1);
  if (rc) return rc;
  const Slot& sl = s->slots[slot];
  const FrameDev& h = sl.h;
  if (!h.v1_ready) return fail(SLM_ERR_UNSUPPORTED, "slm_debug_read_plan: the slot has no tuple-sorted plan");
  const size_t esz = h.f.state_f64 ? 8 : 4;
  const size_t n_wg = (size_t)h.n_pos / 256 + ((h.n_pos % 256) ? 1 : 0);
  const void* src = nullptr;
  size_t n = 0;
  switch (what) {
    case 0: src = h.s_idx.get(); n = sizeof(int32_t) * 4 * (size_t)h.n_pos; break;
    case 1: src = h.grp_run.get(); n = sizeof(int32_t) * (size_t)h.n_pos / 4; break;
    case 2: src = h.run_nodes.get(); n = sizeof(int32_t) * 4 * (size_t)h.n_runs; break;
    case 3: src = h.s_w.get(); n = esz * 4 * (size_t)h.n_pos; break;
    case 4: src = h.blk_key.get(); n = sizeof(int32_t) * (size_t)h.n_blocks; break;
    case 5: src = h.blk_start.get(); n = sizeof(int32_t) * ((size_t)h.n_blocks + 1); break;
    case 6: src = h.blk_entry.get(); n = sizeof(int32_t) * 10 * (size_t)h.n_runs; break;
    case 7: src = h.v2_ready ? h.wg_first.get() : nullptr; n = h.v2_ready ? sizeof(int32_t) * n_wg : 0; break;
    case 8: src = h.v2_ready ? h.wg_last.get() : nullptr; n = h.v2_ready ? sizeof(int32_t) * n_wg : 0; break;
    case 9: src = h.v2_ready ? h.run_lidx.get() : nullptr; n = h.v2_ready ? 10 * (size_t)h.n_runs : 0; break;
    case 10: src = h.v2_ready ? h.blk2_start.get() : nullptr; n = h.v2_ready ? sizeof(int32_t) * ((size_t)h.n_blocks + 1) : 0; break;
    case 11: src = h.v2_ready ? h.blk2_entry.get() : nullptr; n = h.v2_ready ? sizeof(int32_t) * (size_t)h.n_wblk : 0; break;
    case 12: src = h.s_pts.get(); n = esz * 3 * (size_t)h.n_pos; break;
    default: return fail(SLM_ERR_INVALID, "slm_debug_read_plan: unknown array");
  }
  *n_bytes = (int64_t)n;
  const size_t m = n < (size_t)(max_bytes > 0 ? max_bytes : 0) ? n : (size_t)(max_bytes > 0 ? max_bytes : 0);
  if (m > 0 && host_out && src) HIPCHK(hipMemcpyAsync(host_out, src, m, hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  return SLM_OK;
}

int slm_abi_version(void) { return SLM_ABI_VERSION; }

int slm_abi_check(int32_t abi_version, int32_t sz_config, int32_t sz_frame, int32_t sz_gf_config, int32_t sz_gf_frame,
                  int32_t sz_iter_record) {
  if (abi_version != SLM_ABI_VERSION || sz_config != (int32_t)sizeof(slm_config) || sz_frame != (int32_t)sizeof(slm_frame) ||
      sz_gf_config != (int32_t)sizeof(slm_gf_config) || sz_gf_frame != (int32_t)sizeof(slm_gf_frame) ||
      sz_iter_record != (int32_t)sizeof(slm_iter_record)) {
    char buf[320];
    snprintf(buf, sizeof(buf),
             "slm_abi_check: caller built for ABI %d (slm_config %d, slm_frame %d, slm_gf_config %d, slm_gf_frame %d, "
             "slm_iter_record %d bytes), library is ABI %d (%zu, %zu, %zu, %zu, %zu)",
             abi_version, sz_config, sz_frame, sz_gf_config, sz_gf_frame, sz_iter_record, SLM_ABI_VERSION, sizeof(slm_config),
             sizeof(slm_frame), sizeof(slm_gf_config), sizeof(slm_gf_frame), sizeof(slm_iter_record));
    return fail(SLM_ERR_INVALID, buf);
  }
  return SLM_OK;
}

int slm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int slm_create(const slm_config* cfg, slm_solver** out) {
  if (!cfg || !out) return fail(SLM_ERR_INVALID, "slm_create: null argument");
  if (cfg->max_frames < 1 || cfg->num_iterations < 0 || !(cfg->v > 0.0))
    return fail(SLM_ERR_INVALID, "slm_create: bad config");
  if (cfg->solver_path < 0 || cfg->solver_path > 4) return fail(SLM_ERR_INVALID, "slm_create: solver_path must be 0..4");
  if (cfg->data_path < 0 || cfg->data_path > 2) return fail(SLM_ERR_INVALID, "slm_create: data_path must be 0..2");
  if (slm_device_count() < 1) return fail(SLM_ERR_NO_DEVICE, "slm_create: no HIP device visible");
  slm_solver* s = new (std::nothrow) slm_solver();
  if (!s) return fail(SLM_ERR_HIP, "slm_create: out of host memory");
  s->cfg = *cfg;
  try {
    s->slots.resize(cfg->max_frames);
  } catch (...) {
    delete s;
    return fail(SLM_ERR_HIP, "slm_create: out of host memory");
  }
  hipError_t e = hipMalloc((void**)&s->frames_dev, sizeof(FrameDev) * cfg->max_frames);
  if (e == hipSuccess) e = hipMemset(s->frames_dev, 0, sizeof(FrameDev) * cfg->max_frames);
  if (e == hipSuccess) e = hipMalloc((void**)&s->bw_dev, sizeof(int));
  if (e == hipSuccess) e = hipMalloc((void**)&s->reuse_dev, sizeof(int) * cfg->max_frames);
  if (e == hipSuccess) e = hipMemset(s->reuse_dev, 0, sizeof(int) * cfg->max_frames);
  if (e == hipSuccess) e = hipHostMalloc((void**)&s->bw_host, sizeof(int), hipHostMallocDefault);
  if (e == hipSuccess) {
    s->prep = prep_create();
    if (!s->prep) e = hipErrorOutOfMemory;
  }
  if (e != hipSuccess) {
    g_err = std::string("slm_create: ") + hipGetErrorString(e);
    slm_destroy(s);
    return SLM_ERR_HIP;
  }
  {   // per-device set-up of the task-graph launch (LDS attribute, workgroup count, XCD probe): here, not inside an LM iteration
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)dag_device_setup(dev, nullptr);
  }
  if (const char* hb = getenv("SLM_HYBRID")) s->hybrid_batches = atoi(hb) != 0;   // experiments
  if (const char* dm = getenv("SLM_DAG_MAX_NODES")) s->dag_max_nodes = atol(dm);   // experiments / tests
  if (const char* nr = getenv("SLM_NO_REUSE")) s->no_reuse = atoi(nr) != 0;       // tests: recompute after a reject
  if (const char* pf = getenv("SLM_PURE_FILL")) s->pure_fill = atoi(pf) != 0;     // tests: differential check of the pure-fill tiles
  if (const char* fb = getenv("SLM_FUSE_BEGIN")) s->fuse_begin = atoi(fb) != 0;   // A/B: the zeroing as a launch of its own
  *out = s;
  return SLM_OK;
}

int slm_destroy(slm_solver* s) {
  if (!s) return SLM_OK;
  s->prep_worker.shutdown();      // (runs what is still queued: the jobs reference the slots freed below)
  s->bind_pool.shutdown();
  for (Slot& sl : s->slots) {
    FrameDev& h = sl.h;
    if (h.beta) (void)hipFree(h.beta);
    if (h.delta) (void)hipFree(h.delta);
    if (h.rhs) (void)hipFree(h.rhs);
    if (sl.band) (void)hipFree(sl.band);
    if (sl.linv) (void)hipFree(sl.linv);
    if (h.loss_part) (void)hipFree(h.loss_part);
    if (h.st) (void)hipFree(h.st);
    if (h.rec) (void)hipFree(h.rec);
    if (h.node_pk) (void)hipFree(h.node_pk);
    if (h.tgt_pn) (void)hipFree(h.tgt_pn);
    if (h.tgt_px) (void)hipFree(h.tgt_px);
    if (h.ev_rc) (void)hipFree(h.ev_rc);
    plan_free(sl.plan);
    pairplan_free(sl.pplan);
    if (sl.h_pin) (void)hipHostFree(sl.h_pin);
    sl.h_pairs.release();
    sl.h_knn.release();
    sl.h_pts.release();
    sl.h_pts64.release();
    if (sl.d_fronts) (void)hipFree(sl.d_fronts);
    if (sl.d_items) (void)hipFree(sl.d_items);
    if (sl.d_tile_kind) (void)hipFree(sl.d_tile_kind);
    if (sl.d_zero_tiles) (void)hipFree(sl.d_zero_tiles);
    if (sl.d_ints) (void)hipFree(sl.d_ints);
    if (sl.d_dests) (void)hipFree(sl.d_dests);
    if (sl.d_cur_dests) (void)hipFree(sl.d_cur_dests);
    if (sl.ftiles) (void)hipFree(sl.ftiles);
    if (sl.fvec) (void)hipFree(sl.fvec);
    if (sl.flinv) (void)hipFree(sl.flinv);
    if (sl.fmail) (void)hipFree(sl.fmail);
    if (sl.pairbuf) (void)hipFree(sl.pairbuf);
    if (sl.d_dag_flags) (void)hipFree(sl.d_dag_flags);
    if (sl.d_dag_trace) (void)hipFree(sl.d_dag_trace);
    if (sl.prep_fork) (void)hipEventDestroy(sl.prep_fork);
    if (sl.prep_done) (void)hipEventDestroy(sl.prep_done);
  }
  prep_destroy(s->prep);
  prep_destroy(s->prep_async);
  if (s->prep_stream) (void)hipStreamDestroy(s->prep_stream);
  for (PrepBuffers* pb : s->bind_prep) prep_destroy(pb);
  for (hipStream_t q : s->bind_streams) (void)hipStreamDestroy(q);
  for (hipEvent_t e : s->bind_events) (void)hipEventDestroy(e);
  for (auto& evs : s->ev_runs)
    for (hipEvent_t e : evs)
      if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : s->ev_pool) (void)hipEventDestroy(e);
  if (s->drain_event) (void)hipEventDestroy(s->drain_event);
  if (s->frames_dev) (void)hipFree(s->frames_dev);
  if (s->bw_dev) (void)hipFree(s->bw_dev);
  if (s->reuse_dev) (void)hipFree(s->reuse_dev);
  if (s->bw_host) (void)hipHostFree(s->bw_host);
  delete s;
  return SLM_OK;
}

// Block-banded path on demand: tile half-bandwidth of the normal matrix from the KNN tables (one
// 4-byte read-back), band + diagonal-inverse storage, refreshed device copy of the slot.
static int ensure_band(slm_solver* s, int slot, hipStream_t st) {
  Slot& sl = s->slots[slot];
  if (sl.band_ready) return SLM_OK;
  FrameDev& h = sl.h;
  launch_bandwidth(h.f, s->bw_dev, st);
  HIPCHK(hipMemcpyAsync(s->bw_host, s->bw_dev, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  int wb = *s->bw_host;
  if (wb > h.nt - 1) wb = h.nt - 1;
  h.wb = wb;
  HIPCHK(grow(sl.band, sl.cap_band, (size_t)h.nt * (wb + 1) * SLM_NB * SLM_NB));
  HIPCHK(grow(sl.linv, sl.cap_linv, (size_t)h.nt * SLM_NB * SLM_NB));
  h.band = sl.band;
  h.linv = sl.linv;
  HIPCHK(hipMemcpyAsync(s->frames_dev + slot, &h, sizeof(FrameDev), hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));
  sl.band_ready = true;
  return SLM_OK;
}

static int bind_frame_impl(slm_solver* s, int32_t slot, const slm_frame* f, hipStream_t st, PrepBuffers* prep);

// Diagnostics (SLM_BIND_TRACE=<ms>): a bind that takes longer than <ms> on the host prints the time stamps of its
// stages -- which stage of which worker carried a rare multi-millisecond stall (tools/studies/stall_hunt.py).
#include <chrono>
namespace {
struct BindTrace {
  double t[8];
  int n;
};
thread_local BindTrace g_bt;
inline double bt_now() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline double bind_trace_threshold() {
  static const double th = [] {
    const char* e = getenv("SLM_BIND_TRACE");
    return e ? atof(e) : -1.0;
  }();
  return th;
}
inline void bt_mark() {
  if (bind_trace_threshold() >= 0.0 && g_bt.n < 8) g_bt.t[g_bt.n++] = bt_now();
}
}  // namespace

int slm_bind_frame(slm_solver* s, int32_t slot, const slm_frame* f, void* stream) {
  if (!s || !f) return fail(SLM_ERR_INVALID, "slm_bind_frame: null argument");
  return bind_frame_impl(s, slot, f, (hipStream_t)stream, s->prep);
}

// The MODEL-side half of a bind (what depends on sf.points / sf.knn_indices / sf.knn_w / ED_nodes only: the tuple-sorted
// copies and indices of the data term, the hashes of the coupling graph, the symbolic plan of the solver, the slot's
// grow-only work buffers).  Leaves the slot UNBOUND; bind_target_part completes it.
static int bind_model_part(slm_solver* s, int32_t slot, const slm_frame* f, hipStream_t st, PrepBuffers* prep) {
  if (slot < 0 || slot >= (int)s->slots.size()) return fail(SLM_ERR_INVALID, "slm_bind_frame: bad slot");
  // num_neighbors (reference options.py:49, README.md:175; super/loss.py:213-220 and super/utils.py:30-36 are K-generic):
  // 4 takes the tuple-sorted MFMA path; any other value in 1..8 the per-entry-atomics data path with the block-banded
  // solve (what data_path = 1 runs), like a frame of >= 65 536 nodes
  if (f->K < 1 || f->K > 8) return fail(SLM_ERR_UNSUPPORTED, "slm_bind_frame: num_neighbors must be in 1..8");
  if (f->K_ED < 1 || f->K_ED > SLM_MAX_KED)
    return fail(SLM_ERR_UNSUPPORTED, "slm_bind_frame: num_ED_neighbors must be in 1..8");
  if (f->N < 0 || f->J < 1) return fail(SLM_ERR_INVALID, "slm_bind_frame: bad sizes");
  if ((f->N > 0 && (!f->sf_points || !f->sf_knn_idx || !f->sf_knn_w)) || !f->ed_points || !f->ed_knn_idx)
    return fail(SLM_ERR_INVALID, "slm_bind_frame: null device pointer");
  Slot& sl = s->slots[slot];
  FrameDev& h = sl.h;
  g_bt.n = 0;
  bt_mark();                                   // [0] entry

  const int P = 7 * f->J;
  const int nt = (P + SLM_NB - 1) / SLM_NB;
  // the tile half-bandwidth and the band storage are only needed by the block-banded path
  // (solver_path 1, a frame without an ND plan, slm_assemble): ensure_band() fills them in on demand
  int wb = 0;
  sl.band_ready = false;
  h.dag_trace = nullptr;   // a trace buffer is sized for the plan it was enabled on
  h.bound = 0;             // a bind that fails half way leaves the slot unbound (slm_run refuses it)
  h.f = *f;
  h.P = P;
  h.nt = nt;
  h.wb = wb;
  h.n_loss_part = s->cfg.use_data ? kLossBlocks : 0;
  size_t cap_dummy;
  HIPCHK(grow(h.beta, sl.cap_beta, (size_t)P));
  {
    size_t c = sl.cap_npk;
    HIPCHK(grow(h.node_pk, c, (size_t)2 * SLM_NPK * f->J));
    sl.cap_npk = c;
    h.node_pk_try = h.node_pk + (size_t)SLM_NPK * f->J;
  }
  {
    size_t need = (size_t)nt * SLM_NB, c1 = sl.cap_vec, c2 = sl.cap_vec;
    HIPCHK(grow(h.delta, c1, need));
    HIPCHK(grow(h.rhs, c2, need));
    sl.cap_vec = c1 < c2 ? c1 : c2;
  }
  h.band = sl.band;   // may be null until ensure_band()
  h.linv = sl.linv;
  if (!h.loss_part) {
    cap_dummy = 0;
    HIPCHK(grow(h.loss_part, cap_dummy, (size_t)2 * (kLossBlocks + kRegBlocksMax)));
    HIPCHK(hipMemsetAsync(h.loss_part, 0, sizeof(double) * 2 * (kLossBlocks + kRegBlocksMax), st));
  }
  if (!h.st) HIPCHK(hipMalloc((void**)&h.st, sizeof(LMState)));
#ifdef SLM_STAMPS
  if (!h.dbg) {
    HIPCHK(hipMalloc((void**)&h.dbg, sizeof(unsigned long long) * 64));
    HIPCHK(hipMemset(h.dbg, 0, sizeof(unsigned long long) * 64));
  }
#endif
  if (!h.rec) {
    int n = s->cfg.num_iterations > 0 ? s->cfg.num_iterations : 1;
    HIPCHK(hipMalloc((void**)&h.rec, sizeof(slm_iter_record) * n));
  }
  // tuple-sorted data-term assembly plan (DataLoss.prepare analogue)
  h.v1_ready = 0;
  h.vk_ready = 0;
  uint64_t dev_knn_hash = 0, dev_graph_hash = 0;   // coupling-graph hashes computed by prep_v1 on the device
  if (s->cfg.use_data && s->cfg.data_path != 1 && f->K == SLM_K && f->J < 65536 && f->N > 0) {
    V1Sizes sz;
    HIPCHK(prep_v1(prep, *f, sl.plan, &sz, st));
    if (sz.bad_knn)
      return fail(SLM_ERR_INVALID, "slm_bind_frame: a surfel KNN index (sf_knn_idx) lies outside [0, J)");
    dev_knn_hash = sz.knn_hash;
    dev_graph_hash = sz.graph_hash;
    bt_mark();                                 // [1] tuple-sorted plan done (its size read-backs included)
    if (sz.n_tuples > 0) {
      h.n_tuples = sz.n_tuples;
      h.n_pos = sz.n_pos;
      h.n_runs = sz.n_runs;
      h.n_blocks = sz.n_blocks;
      h.s_pts = sl.plan.s_pts;
      h.s_idx = sl.plan.s_idx;
      h.s_w = sl.plan.s_w;
      h.grp_run = sl.plan.grp_run;
      h.run_nodes = sl.plan.run_nodes;
      h.slab = sl.plan.slab;
      h.blk_key = sl.plan.blk_key;
      h.blk_start = sl.plan.blk_start;
      h.blk_entry = sl.plan.blk_entry;
      HIPCHK(grow(h.ev_rc, sl.cap_ev, (size_t)4 * sz.n_pos));   // evaluation buffer {r, c} per position
      h.v1_ready = 1;
      // workgroup-merged records (default); data_path 2 keeps the per-run slab
      h.v2_ready = 0;
      if (s->cfg.data_path == 0 && sz.n_wblk > 0 && sz.max_wblk_per_wg <= SLM_LB_MAX) {
        h.n_wblk = sz.n_wblk;
        h.wg_first = sl.plan.wg_first;
        h.wg_last = sl.plan.wg_last;
        h.run_lidx = sl.plan.run_lidx;
        h.wgslab = sl.plan.wgslab;
        h.blk2_start = sl.plan.blk2_start;
        h.blk2_entry = sl.plan.blk2_entry;
        h.v2_ready = 1;
      }
    }
  } else if (s->cfg.use_data && s->cfg.data_path != 1 && s->cfg.solver_path != 1 && f->J < 65536 && f->N > 0) {
    // num_neighbors != 4 (round 6): the K-generic pair path -- the coupled-pair list and every surfel's pair indices from
    // one sort (prep_pairs), the multifrontal solver on that list, the data term through per-pair records (pairbuf)
    PairSizes sz;
    HIPCHK(prep_pairs(prep, *f, sl.pplan, &sz, st));
    if (sz.bad_knn)
      return fail(SLM_ERR_INVALID, "slm_bind_frame: a surfel KNN index (sf_knn_idx) lies outside [0, J)");
    dev_knn_hash = sz.knn_hash;
    dev_graph_hash = sz.graph_hash;
    bt_mark();
    if (sz.n_blocks > 0) {
      h.n_blocks = sz.n_blocks;
      h.blk_key = sl.pplan.blk_key;
      h.sf_pidx = sl.pplan.sf_pidx;
      h.sf_perm = sl.pplan.sf_perm;
      h.vk_ready = 1;
    }
  } else if (s->cfg.use_data && f->N > 0) {
    // the per-entry atomics path (data_path 1, J >= 65536) dereferences the table too: same refusal
    bool bad = false;
    HIPCHK(prep_check_knn(prep, *f, &bad, st));
    if (bad) return fail(SLM_ERR_INVALID, "slm_bind_frame: a surfel KNN index (sf_knn_idx) lies outside [0, J)");
  }
  // share of this rank when the frame is sharded over several GPUs (whole frame otherwise)
  {
    const int n_wg = (h.v1_ready ? h.n_pos + 255 : 0) / 256;
    h.wg_lo = (int32_t)((int64_t)n_wg * s->rank / s->world);
    h.wg_hi = (int32_t)((int64_t)n_wg * (s->rank + 1) / s->world);
    h.sf_lo = (int32_t)((int64_t)f->N * s->rank / s->world);
    h.sf_hi = (int32_t)((int64_t)f->N * (s->rank + 1) / s->world);
    h.pairbuf = nullptr;
    if (s->shard_mode) {
      if (!h.v1_ready && !h.vk_ready)
        return fail(SLM_ERR_UNSUPPORTED, "slm_bind_frame: sharded frames need the multifrontal data paths "
                                         "(data_path 0 or 2, J < 65536)");
      HIPCHK(grow(sl.pairbuf, sl.cap_pairbuf, (size_t)h.n_blocks * SLM_WREC + SLM_VK_TAIL + 2));
      h.pairbuf = sl.pairbuf;
      // records (or per-run Grams) of the other ranks' workgroups stay zero for the whole frame
      if (h.v2_ready) HIPCHK(hipMemsetAsync(h.wgslab, 0, sizeof(double) * SLM_WREC * (size_t)h.n_wblk, st));
      else if (h.v1_ready) HIPCHK(hipMemsetAsync(h.slab, 0, sizeof(double) * SLM_SLAB_STRIDE * (size_t)h.n_runs, st));
    } else if (h.vk_ready) {   // the K-generic pair path always assembles through the pair records
      HIPCHK(grow(sl.pairbuf, sl.cap_pairbuf, (size_t)h.n_blocks * SLM_WREC + SLM_VK_TAIL + 2));
      h.pairbuf = sl.pairbuf;
    }
  }
  // nested-dissection plan (symbolic analysis on the host from the coupled-pair list)
  h.nd_ready = 0;
  if ((h.v1_ready || h.vk_ready) && s->cfg.solver_path != 1) {
    // The symbolic plan depends only on the coupling graph (node KNN table + coupled-pair list): reuse it while the
    // graph is unchanged.  The graph's hash comes from the device with the sizes (prep_v1's one read-back): a frame
    // whose graph is the slot's cached one reads nothing else back -- the lists only travel to the host when the
    // hash says that they changed.
    const uint64_t knn_hash = dev_knn_hash, hash = dev_graph_hash;
    const bool same_graph = sl.nd_valid && sl.nd_hash == hash && sl.nd_knn_hash == knn_hash && sl.cur_n_blocks == h.n_blocks;
    if (!same_graph) {
      HIPCHK(sl.h_pairs.resize(h.n_blocks));
      HIPCHK(sl.h_knn.resize((size_t)f->J * f->K_ED));
      HIPCHK(sl.h_pts.resize((size_t)f->J * 3));
      HIPCHK(hipMemcpyAsync(sl.h_pairs.data(), h.blk_key, sizeof(uint32_t) * h.n_blocks, hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(sl.h_knn.data(), f->ed_knn_idx, sizeof(int32_t) * sl.h_knn.size(), hipMemcpyDeviceToHost, st));
      if (f->state_f64) {
        HIPCHK(sl.h_pts64.resize(sl.h_pts.size()));
        HIPCHK(hipMemcpyAsync(sl.h_pts64.data(), f->ed_points, sizeof(double) * sl.h_pts64.size(), hipMemcpyDeviceToHost, st));
      } else {
        HIPCHK(hipMemcpyAsync(sl.h_pts.data(), f->ed_points, sizeof(float) * sl.h_pts.size(), hipMemcpyDeviceToHost, st));
      }
      HIPCHK(hipStreamSynchronize(st));
      // the node positions only steer the geometric bisection of the symbolic plan: float32 is plenty
      if (f->state_f64)
        for (size_t i = 0; i < sl.h_pts.size(); ++i) sl.h_pts[i] = (float)sl.h_pts64[i];
    }
    bt_mark();                                 // [2] pair list / node table on the host (only when the graph changed)
    // pair -> destination table of this frame from the plan's (sorted) pair list; false when a pair is new
    // A pair the plan was not built from still has a place in it when the later-eliminated node lies in the
    // front of the earlier one (a fill position of the dense front): no new analysis then either.
    auto dests_from_plan = [&]() -> bool {
      sl.cur_dest.resize(sl.h_pairs.size());
      std::vector<size_t> fresh;   // pairs answered by nd_dest_of: remembered in the plan's list afterwards
      size_t j = 0;
      for (size_t i = 0; i < sl.h_pairs.size(); ++i) {
        while (j < sl.plan_pairs.size() && sl.plan_pairs[j] < sl.h_pairs[i]) ++j;
        if (j < sl.plan_pairs.size() && sl.plan_pairs[j] == sl.h_pairs[i]) sl.cur_dest[i] = sl.nd.block_dest[j];
        else if (!nd_dest_of(sl.nd, f->J, sl.h_pairs[i], sl.cur_dest[i])) return false;
        else fresh.push_back(i);
      }
      if (!fresh.empty()) {
        g_plan_fill_hits += (long long)fresh.size();
        std::vector<uint32_t> keys(sl.plan_pairs.size() + fresh.size());
        std::vector<NDDest> dests(keys.size());
        size_t a = 0, b = 0, o = 0;
        while (a < sl.plan_pairs.size() || b < fresh.size()) {
          const bool take_old = b == fresh.size() || (a < sl.plan_pairs.size() && sl.plan_pairs[a] < sl.h_pairs[fresh[b]]);
          if (take_old) { keys[o] = sl.plan_pairs[a]; dests[o++] = sl.nd.block_dest[a++]; }
          else { keys[o] = sl.h_pairs[fresh[b]]; dests[o++] = sl.cur_dest[fresh[b++]]; }
        }
        sl.plan_pairs.swap(keys);
        sl.nd.block_dest.swap(dests);
      }
      return true;
    };
    auto upload_cur_dests = [&]() -> hipError_t {
      hipError_t e = grow(sl.d_cur_dests, sl.cap_cur_dests, sl.cur_dest.size() + 1);
      if (e == hipSuccess && !sl.cur_dest.empty())
        e = hipMemcpyAsync(sl.d_cur_dests, sl.cur_dest.data(), sizeof(NDDest) * sl.cur_dest.size(), hipMemcpyHostToDevice, st);
      h.block_dest = sl.d_cur_dests;
      // (a failed grow / copy leaves d_cur_dests freed or half written: the slot's cached pair list must not be trusted
      //  by a later bind of the old graph -- same_graph compares the hash AND this count)
      sl.cur_n_blocks = e == hipSuccess ? h.n_blocks : -1;
      if (e != hipSuccess) sl.nd_hash = 0;
      return e;
    };
    // Pivot-column tiles by what reaches them (FrameDev::tile_kind, slm_common.h).  From the PLAN's destination list --
    // every pair seen since the node graph last changed, fill-position pairs included: a superset of this frame's --,
    // the ARAP pair blocks and the nodes' diagonal blocks (a 7 x 7 block can straddle a tile boundary: its four
    // corners decide).  A tile that none of them reaches and that some child maps into (a pull item) is pure fill.
    auto upload_tile_kinds = [&]() -> hipError_t {
      const NDPlanHost& nd = sl.nd;
      const size_t n_tiles = (size_t)(nd.tile_doubles / (SLM_NB * SLM_NB));
      {   // (an earlier upload from these vectors may still be in flight)
        const hipError_t e0 = hipStreamSynchronize(st);
        if (e0 != hipSuccess) return e0;
      }
      std::vector<uint8_t>& kind = sl.h_tile_kind;
      std::vector<long long>& zero = sl.h_zero_tiles;
      kind.assign(n_tiles + 1, 0);
      zero.clear();
      sl.n_piv_tiles = sl.n_pure_tiles = 0;
      std::vector<uint8_t> assembled(n_tiles + 1, 0), pulled(n_tiles + 1, 0);
      auto nbase = [](const NDFront& fr, int pos) { return pos < fr.nv ? 7 * pos : fr.n1p + 7 * (pos - fr.nv); };
      auto mark = [&](int front, int prow, int pcol) {
        if (front < 0 || front >= (int)nd.fronts.size()) return;
        const NDFront& fr = nd.fronts[front];
        const int rb = nbase(fr, prow), cb = nbase(fr, pcol);
        for (int x = 0; x < 7; x += 6)
          for (int y = 0; y < 7; y += 6) {
            int i = rb + x, j = cb + y;
            if (i < j) std::swap(i, j);
            const int r = i >> 6, c = j >> 6;
            if (c < fr.npt && r < fr.nt) assembled[(size_t)fr.tile_first + (size_t)c * fr.nt - (size_t)c * (c - 1) / 2 + (size_t)(r - c)] = 1;
          }
      };
      for (const NDDest& d : nd.block_dest) mark(d.front, d.prow, d.pcol);
      for (const NDDest& d : nd.pair_dest) mark(d.front, d.prow, d.pcol);
      for (size_t j = 0; j < nd.node_front.size(); ++j) mark(nd.node_front[j], nd.node_pos[j], nd.node_pos[j]);
      for (size_t l = 0; l + 1 < nd.level_start.size(); ++l)
        for (int k = nd.item_off[2 * l + 1]; k < nd.item_off[2 * l + 2]; ++k) pulled[(size_t)nd.tile_items[k].pad0] = 1;
      for (const NDFront& fr : nd.fronts)
        for (int c = 0; c < fr.npt; ++c)
          for (int r = c; r < fr.nt; ++r) {
            const size_t ix = (size_t)c * fr.nt - (size_t)c * (c - 1) / 2 + (size_t)(r - c), t = (size_t)fr.tile_first + ix;
            kind[t] = (s->pure_fill && !assembled[t] && pulled[t]) ? 1 : 0;
            ++sl.n_piv_tiles;
            sl.n_pure_tiles += kind[t];
            if (!kind[t]) zero.push_back((long long)fr.tile_off + (long long)ix * (SLM_NB * SLM_NB));
          }
      hipError_t e = grow(sl.d_tile_kind, sl.cap_tile_kind, n_tiles + 1);
      if (e == hipSuccess) e = grow(sl.d_zero_tiles, sl.cap_zero_tiles, zero.size() + 1);
      if (e == hipSuccess) e = hipMemcpyAsync(sl.d_tile_kind, kind.data(), n_tiles + 1, hipMemcpyHostToDevice, st);
      if (e == hipSuccess && !zero.empty())
        e = hipMemcpyAsync(sl.d_zero_tiles, zero.data(), sizeof(long long) * zero.size(), hipMemcpyHostToDevice, st);
      h.tile_kind = sl.d_tile_kind;
      h.zero_tiles = sl.d_zero_tiles;
      h.n_zero_tiles = (int32_t)zero.size();
      if (e != hipSuccess) { sl.cur_n_blocks = -1; sl.nd_hash = 0; }
      return e;
    };
    if (sl.nd_valid && sl.nd_knn_hash != knn_hash) {   // another node graph: nothing of the old plan applies
      sl.nd_valid = false;
      sl.plan_pairs.clear();
    }
    std::vector<uint32_t> all_pairs;
    if (same_graph) {
      h.nd_ready = 1;   // device mirrors of the plan and of this pair list are still in place (pointers kept in h)
    } else if (sl.nd_valid && dests_from_plan()) {
      ++g_plan_reuses;
      HIPCHK(upload_cur_dests());
      HIPCHK(upload_tile_kinds());   // (the plan's destination list may have grown by fill-position pairs)
      h.nd_ready = 1;
      sl.nd_hash = hash;
    } else if ([&] {
                 // new pairs: analyse the union of what the plan already covers and this frame's list
                 all_pairs.resize(sl.plan_pairs.size() + sl.h_pairs.size());
                 all_pairs.resize(std::set_union(sl.plan_pairs.begin(), sl.plan_pairs.end(), sl.h_pairs.begin(),
                                                 sl.h_pairs.end(), all_pairs.begin()) - all_pairs.begin());
                 // (a solver that runs its launches as one task graph -- at most two slots, or solver_path 2 -- takes the
                 //  larger leaves: SLM_ND_LEAF_LATENCY, slm_nd.h)
                 return nd_build_plan(f->J, f->K_ED, sl.h_pts.data(), sl.h_knn.data(), all_pairs.data(),
                                      (int)all_pairs.size(), sl.nd,
                                      solve_is_task_graph(s, (int)s->slots.size(), f->J) ? SLM_ND_LEAF_LATENCY : SLM_ND_LEAF);
               }()) {
      sl.nd_valid = false;
      ++g_plan_builds;
      sl.plan_pairs.swap(all_pairs);
      NDPlanHost& nd = sl.nd;
      const size_t n_ints = nd.level_start.size() + nd.nodes.size() + nd.eamap.size() + 2 * (size_t)f->J +
                            nd.in_start.size() + nd.in_edge.size() + nd.item_off.size() +
                            nd.dag_tasks.size() + nd.dag_top_tasks.size() + nd.front_kids.size() + nd.pull_off.size() + nd.pullmap.size() + nd.prng_off.size() +
                            nd.prng.size();
      const size_t n_dests = nd.block_dest.size() + nd.pair_dest.size();
      HIPCHK(grow(sl.d_fronts, sl.cap_fronts, nd.fronts.size()));
      HIPCHK(grow(sl.d_ints, sl.cap_ints, n_ints));
      HIPCHK(grow(sl.d_dests, sl.cap_dests, n_dests));
      HIPCHK(grow(sl.ftiles, sl.cap_ftiles, (size_t)nd.tile_doubles));
      HIPCHK(grow(sl.fvec, sl.cap_fvec, (size_t)nd.vec_doubles));
      HIPCHK(grow(sl.flinv, sl.cap_flinv, (size_t)nd.linv_doubles + 1));
      HIPCHK(grow(sl.fmail, sl.cap_fmail, (size_t)(nd.linv_doubles / (SLM_NB * SLM_NB)) * 2560 + 1));   // SLM_MAIL_DOUBLES per pivot tile column
      HIPCHK(hipMemcpyAsync(sl.d_fronts, nd.fronts.data(), sizeof(NDFront) * nd.fronts.size(), hipMemcpyHostToDevice, st));
      int32_t* p = sl.d_ints;
      auto up = [&](const std::vector<int32_t>& v) -> hipError_t {
        hipError_t e = v.empty() ? hipSuccess : hipMemcpyAsync(p, v.data(), sizeof(int32_t) * v.size(), hipMemcpyHostToDevice, st);
        p += v.size();
        return e;
      };
      h.level_start = p; HIPCHK(up(nd.level_start));
      h.nd_nodes = p;    HIPCHK(up(nd.nodes));
      h.nd_eamap = p;    HIPCHK(up(nd.eamap));
      h.node_front = p;  HIPCHK(up(nd.node_front));
      h.node_pos = p;    HIPCHK(up(nd.node_pos));
      h.in_start = p;    HIPCHK(up(nd.in_start));
      h.in_edge = p;     HIPCHK(up(nd.in_edge));
      h.item_off = p;    HIPCHK(up(nd.item_off));
      HIPCHK(grow(sl.d_items, sl.cap_items, nd.tile_items.size() + 1));
      if (!nd.tile_items.empty())
        HIPCHK(hipMemcpyAsync(sl.d_items, nd.tile_items.data(), sizeof(NDTileItem) * nd.tile_items.size(), hipMemcpyHostToDevice, st));
      h.tile_items = sl.d_items;
      h.dag_tasks = p;   HIPCHK(up(nd.dag_tasks));
      h.dag_top_tasks = p; HIPCHK(up(nd.dag_top_tasks));
      h.n_dag_top_tasks = (int32_t)(nd.dag_top_tasks.size() / 2);
      h.dag_cut_depth = nd.dag_cut_depth;
      h.front_kids = p;  HIPCHK(up(nd.front_kids));
      h.pull_off = p;    HIPCHK(up(nd.pull_off));
      h.pullmap = p;     HIPCHK(up(nd.pullmap));
      h.prng_off = p;    HIPCHK(up(nd.prng_off));
      h.prng = p;        HIPCHK(up(nd.prng));
      h.n_dag_tasks = (int32_t)(nd.dag_tasks.size() / 2);
      h.dag_n_tiles = (int32_t)(nd.tile_doubles / (SLM_NB * SLM_NB));
      h.dag_n_pcols = (int32_t)(nd.linv_doubles / (SLM_NB * SLM_NB));
      h.dag_n_flags = 8 + h.dag_n_tiles + 7 * h.dag_n_pcols;
      HIPCHK(grow(sl.d_dag_flags, sl.cap_dag_flags, (size_t)h.dag_n_flags));
      h.dag_flags = sl.d_dag_flags;
      if (!nd.block_dest.empty())
        HIPCHK(hipMemcpyAsync(sl.d_dests, nd.block_dest.data(), sizeof(NDDest) * nd.block_dest.size(), hipMemcpyHostToDevice, st));
      HIPCHK(hipMemcpyAsync(sl.d_dests + nd.block_dest.size(), nd.pair_dest.data(), sizeof(NDDest) * nd.pair_dest.size(), hipMemcpyHostToDevice, st));
      h.pair_dest = sl.d_dests + nd.block_dest.size();
      if (!dests_from_plan()) return fail(SLM_ERR_INVALID, "slm_bind_frame: internal error (pair missing from its own plan)");
      HIPCHK(upload_cur_dests());
      HIPCHK(upload_tile_kinds());
      h.fronts = sl.d_fronts;
      h.n_fronts = (int)nd.fronts.size();
      h.n_levels = (int)nd.level_start.size() - 1;
      h.ftiles = sl.ftiles;
      h.fvec = sl.fvec;
      h.flinv = sl.flinv;
      h.fmail = sl.fmail;
      h.zero_tile_doubles = nd.tile_zero_doubles;
      h.zero_vec_doubles = nd.vec_doubles;
      h.nd_ready = 1;
      sl.nd_hash = hash;
      sl.nd_knn_hash = knn_hash;
      sl.nd_valid = true;
    } else {
      sl.nd_valid = false;
      sl.plan_pairs.clear();
    }
  }
  bt_mark();                                   // [3] symbolic plan settled
  return SLM_OK;
}

// The TARGET-side half: the frame's target tables, image size and intrinsics; descriptor upload, LM state reset.
static int bind_target_part(slm_solver* s, int32_t slot, const slm_frame* f, hipStream_t st) {
  if (f->H < 2 || f->W < 2 || f->T < 0) return fail(SLM_ERR_INVALID, "slm_bind_frame: bad sizes");
  if ((f->T > 0 && (!f->tgt_points || !f->tgt_norms)) || !f->index_map || !f->tgt_valid)
    return fail(SLM_ERR_INVALID, "slm_bind_frame: null device pointer");
  Slot& sl = s->slots[slot];
  FrameDev& h = sl.h;
  h.f = *f;
  HIPCHK(grow(h.tgt_pn, sl.cap_tpn, (size_t)2 * (f->T > 0 ? f->T : 1)));
  launch_pack_target(f->T, f->tgt_points, f->tgt_norms, h.tgt_pn, st);
  HIPCHK(grow(h.tgt_px, sl.cap_tpx, (size_t)2 * f->H * f->W));
  launch_pack_target_px(f->H * f->W, f->index_map, f->tgt_valid, f->tgt_points, f->tgt_norms, h.tgt_px, st);
  h.bound = 1;
  // descriptor -> device through the slot's pinned mirror: no wait for the copy (the mirror always holds the newest
  // host state, and every change of it is followed by another copy on the stream)
  if (!sl.h_pin) HIPCHK(hipHostMalloc((void**)&sl.h_pin, sizeof(FrameDev), hipHostMallocDefault));
  memcpy(sl.h_pin, &h, sizeof(FrameDev));
  HIPCHK(hipMemcpyAsync(s->frames_dev + slot, sl.h_pin, sizeof(FrameDev), hipMemcpyHostToDevice, st));
  bt_mark();                                   // [4] descriptor on its way
  if (!h.nd_ready) {
    std::lock_guard<std::mutex> lock(s->band_mutex);
    int rc = ensure_band(s, slot, st);
    if (rc) return rc;
  }
  launch_init_slot(s->frames_dev, slot, f->J, s->cfg, st);
  HIPCHK(hipMemsetAsync(s->reuse_dev + slot, 0, sizeof(int), st));   // a new frame: nothing to reuse
  HIPCHK(hipGetLastError());
  bt_mark();                                   // [5] end
  if (bind_trace_threshold() >= 0.0 && g_bt.n >= 2 && g_bt.t[g_bt.n - 1] - g_bt.t[0] > bind_trace_threshold()) {
    char buf[256];
    int o = snprintf(buf, sizeof(buf), "[slm bind trace] slot %d start %.3f stages(ms):", slot, g_bt.t[0]);
    for (int i = 1; i < g_bt.n && o < 220; ++i) o += snprintf(buf + o, sizeof(buf) - o, " %.3f", g_bt.t[i] - g_bt.t[i - 1]);
    fprintf(stderr, "%s\n", buf);
  }
  return SLM_OK;
}

static bool same_model(const slm_frame& a, const slm_frame& b) {
  return a.N == b.N && a.J == b.J && a.K == b.K && a.K_ED == b.K_ED && a.state_f64 == b.state_f64 && a.sf_points == b.sf_points &&
         a.sf_knn_idx == b.sf_knn_idx && a.sf_knn_w == b.sf_knn_w && a.ed_points == b.ed_points && a.ed_knn_idx == b.ed_knn_idx;
}

// waits for the slot's queued model-side preparation, if any (host side: the worker has finished the job)
static void join_prepare(slm_solver* s, int slot) {
  Slot& sl = s->slots[slot];
  if (sl.prep_ticket) {
    s->prep_worker.wait(sl.prep_ticket);
    sl.prep_ticket = 0;
  }
}

static int bind_frame_impl(slm_solver* s, int32_t slot, const slm_frame* f, hipStream_t st, PrepBuffers* prep) {
  if (slot < 0 || slot >= (int)s->slots.size()) return fail(SLM_ERR_INVALID, "slm_bind_frame: bad slot");
  Slot& sl = s->slots[slot];
  join_prepare(s, slot);
  // A model prepared ahead (slm_prepare_model) serves ONE bind, and only the bind of the very arrays it read: the
  // frame's update / fusion rewrites them in place afterwards.
  const bool prepared = sl.model_ready && same_model(sl.prep_model, *f);
  sl.model_ready = false;
  // Whatever becomes of the preparation, its launches ran on the solver's own stream and wrote this slot's plan buffers
  // (and read the caller's arrays): everything the bind enqueues comes after them -- also when the preparation is NOT
  // used (another model, slm_discard_prepared) and the full bind rewrites the same buffers.
  if (sl.prep_recorded) {
    sl.prep_recorded = false;
    HIPCHK(hipStreamWaitEvent(st, sl.prep_done, 0));
  }
  if (prepared) {
    if (sl.prep_rc != SLM_OK) return fail(sl.prep_rc, sl.prep_err.c_str());
    g_bt.n = 0;
    bt_mark();
  } else {
    const int rc = bind_model_part(s, slot, f, st, prep);
    if (rc != SLM_OK) return rc;
  }
  return bind_target_part(s, slot, f, st);
}

int slm_prepare_model(slm_solver* s, int32_t slot, const slm_frame* model, void* stream) {
  if (!s || !model) return fail(SLM_ERR_INVALID, "slm_prepare_model: null argument");
  if (slot < 0 || slot >= (int)s->slots.size()) return fail(SLM_ERR_INVALID, "slm_prepare_model: bad slot");
  Slot& sl = s->slots[slot];
  join_prepare(s, slot);
  sl.model_ready = false;
  sl.h.bound = 0;                   // the slot's plan is being rebuilt: unbound until the next slm_bind_frame
  if (!s->prep_stream) HIPCHK(hipStreamCreateWithFlags(&s->prep_stream, hipStreamNonBlocking));
  if (!s->prep_async) {
    s->prep_async = prep_create();
    if (!s->prep_async) return fail(SLM_ERR_HIP, "slm_prepare_model: out of memory");
  }
  if (!sl.prep_fork) HIPCHK(hipEventCreateWithFlags(&sl.prep_fork, hipEventDisableTiming));
  if (!sl.prep_done) HIPCHK(hipEventCreateWithFlags(&sl.prep_done, hipEventDisableTiming));
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  sl.prep_model = *model;
  sl.prep_model.T = 0;              // (only the model fields are read)
  sl.prep_model.tgt_points = sl.prep_model.tgt_norms = nullptr;
  sl.prep_model.index_map = nullptr;
  sl.prep_model.tgt_valid = nullptr;
  sl.prep_rc = SLM_OK;
  HIPCHK(hipEventRecord(sl.prep_fork, (hipStream_t)stream));   // the preparation sees what the caller has enqueued so far
  try {
    sl.prep_err.clear();
    sl.prep_ticket = s->prep_worker.submit([s, slot, dev] {
      Slot& w = s->slots[slot];
      int rc = SLM_OK;
      try {
        if (hipSetDevice(dev) != hipSuccess || hipStreamWaitEvent(s->prep_stream, w.prep_fork, 0) != hipSuccess) {
          rc = fail(SLM_ERR_HIP, "slm_prepare_model: worker setup failed");
        } else {
          rc = bind_model_part(s, slot, &w.prep_model, s->prep_stream, s->prep_async);
          // (recorded after a failed preparation too: its launches up to the failure still touch the slot)
          if (hipEventRecord(w.prep_done, s->prep_stream) != hipSuccess) {
            if (rc == SLM_OK) rc = fail(SLM_ERR_HIP, "slm_prepare_model: hipEventRecord failed");
          } else {
            w.prep_recorded = true;
          }
        }
        if (rc != SLM_OK) w.prep_err = g_err;          // (thread-local text of the worker)
      } catch (...) {
        rc = SLM_ERR_HIP;
        try { w.prep_err = "slm_prepare_model: out of host memory in the worker"; } catch (...) {}
      }
      w.prep_rc = rc;
      w.model_ready = true;         // (an error is reported by the bind that consumes the preparation)
    });
  } catch (...) {
    sl.prep_ticket = 0;
    return fail(SLM_ERR_HIP, "slm_prepare_model: could not start the worker");
  }
  return SLM_OK;
}

int slm_discard_prepared(slm_solver* s, int32_t slot) {
  if (!s) return fail(SLM_ERR_INVALID, "slm_discard_prepared: null argument");
  if (slot < 0 || slot >= (int)s->slots.size()) return fail(SLM_ERR_INVALID, "slm_discard_prepared: bad slot");
  join_prepare(s, slot);
  s->slots[slot].model_ready = false;   // (prep_recorded stays: the next bind of the slot still orders itself behind the launches)
  return SLM_OK;
}

int slm_bind_frames(slm_solver* s, int32_t first_slot, int32_t n_frames, const slm_frame* frames, void* stream) {
  if (!s || !frames) return fail(SLM_ERR_INVALID, "slm_bind_frames: null argument");
  if (first_slot < 0 || n_frames < 1 || first_slot + n_frames > (int)s->slots.size())
    return fail(SLM_ERR_INVALID, "slm_bind_frames: slot range out of bounds");
  hipStream_t st = (hipStream_t)stream;
  if (n_frames == 1) return bind_frame_impl(s, first_slot, frames, st, s->prep);
  // The preparation of a frame is a chain of ~40 small launches and four size read-backs: latency, not
  // throughput.  The frames of a batch are therefore bound CONCURRENTLY -- one host thread, stream and set of
  // scratch buffers per frame (up to kBindWorkers at a time), forked from and joined into the caller's stream.
  static const int kBindWorkers = [] {
    const char* e = getenv("SLM_BIND_WORKERS");     // experiments
    return e && atoi(e) > 0 ? atoi(e) : 8;
  }();
  const int W = std::min(std::min(n_frames, kBindWorkers), 63);
  // per-worker resources: each one is pushed into its pool as soon as it exists, so a failure half way leaves
  // nothing behind that slm_destroy does not release (the pools are reserved first: push_back cannot throw after)
  try {
    s->bind_prep.reserve(64);
    s->bind_streams.reserve(64);
    s->bind_events.reserve(65);
  } catch (...) {
    return fail(SLM_ERR_HIP, "slm_bind_frames: out of host memory");
  }
  while ((int)s->bind_prep.size() < W) {
    PrepBuffers* pb = prep_create();
    if (!pb) return fail(SLM_ERR_HIP, "slm_bind_frames: out of memory");
    s->bind_prep.push_back(pb);
  }
  while ((int)s->bind_streams.size() < W) {
    hipStream_t q = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
    s->bind_streams.push_back(q);
  }
  while ((int)s->bind_events.size() < W + 1) {
    hipEvent_t e = nullptr;
    HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    s->bind_events.push_back(e);
  }
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  // The binds read sizes back, so they wait for the work already on `st` in any case: wait for it HERE, on one
  // thread, rather than with W workers spinning in their first read-back for as long as the previous LM run takes
  // (8 busy threads for tens of milliseconds per step cost the process its CPU quota on the GPU box).
  // This is the one LONG host wait of a tracking step (the LM run of the previous step, tens of milliseconds).  Every
  // blocking wait of this HIP runtime spins (hipStreamSynchronize, hipEventSynchronize with or without
  // hipEventBlockingSync: one CPU at 100 %, tools/studies/wait_cpu.py), which counts where the ranks of a node share a
  // CPU quota: the wait is therefore a poll of an event with naps while the end is far (estimated from the previous
  // wait on this solver) and a tight poll only over the last stretch.  SLM_SPIN_WAIT=1: hipStreamSynchronize as before.
  const double bt0 = bind_trace_threshold() >= 0.0 ? bt_now() : 0.0;
  static const bool spin_wait = [] {
    const char* e = getenv("SLM_SPIN_WAIT");
    return e && atoi(e) != 0;
  }();
  if (spin_wait) {
    HIPCHK(hipStreamSynchronize(st));
  } else {
    if (!s->drain_event) HIPCHK(hipEventCreateWithFlags(&s->drain_event, hipEventDisableTiming));
    HIPCHK(hipEventRecord(s->drain_event, st));
    const auto w0 = std::chrono::steady_clock::now();
    auto waited_ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count(); };
    const double est = s->drain_est_ms;
    for (;;) {
      const hipError_t q = hipEventQuery(s->drain_event);
      if (q == hipSuccess) break;
      (void)hipGetLastError();   // hipErrorNotReady is not an error here: keep it out of the sticky last-error slot
      if (q != hipErrorNotReady) return fail(SLM_ERR_HIP, hipGetErrorString(q));
      const double w = waited_ms();
      if (w < 0.85 * est - 0.4) {
        std::this_thread::sleep_for(std::chrono::microseconds(300));    // far from the expected end
      } else if (w > est + 1.5) {
        std::this_thread::sleep_for(std::chrono::microseconds(100));    // overdue (the step got longer): nap again
      } else {
        __builtin_ia32_pause();                                          // the last stretch: poll
      }
    }
    s->drain_est_ms = waited_ms();
  }
  const double bt1 = bind_trace_threshold() >= 0.0 ? bt_now() : 0.0;
  HIPCHK(hipEventRecord(s->bind_events[W], st));            // fork: the workers see everything enqueued on `st` so far
  // (no exception may cross the extern "C" boundary: the containers are sized before any worker starts, the pool throws
  //  only from thread creation, and a worker catches whatever its bind throws -- std::bad_alloc from a plan vector)
  int rcs[64];
  std::string errs[64];
  for (int w = 0; w < W; ++w) rcs[w] = SLM_OK;
  const std::function<void(int)> work = [&](int w) {
    try {
      if (hipSetDevice(dev) != hipSuccess || hipStreamWaitEvent(s->bind_streams[w], s->bind_events[W], 0) != hipSuccess) {
        rcs[w] = SLM_ERR_HIP;
        errs[w] = "slm_bind_frames: worker setup failed";
        return;
      }
      for (int i = w; i < n_frames; i += W) {
        const int rc = bind_frame_impl(s, first_slot + i, frames + i, s->bind_streams[w], s->bind_prep[w]);
        if (rc != SLM_OK) {
          rcs[w] = rc;
          errs[w] = g_err;        // thread-local text of this worker
          return;
        }
      }
      if (hipEventRecord(s->bind_events[w], s->bind_streams[w]) != hipSuccess) rcs[w] = SLM_ERR_HIP;
    } catch (...) {
      rcs[w] = SLM_ERR_HIP;
      try { errs[w] = "slm_bind_frames: out of host memory in a bind worker"; } catch (...) {}
    }
  };
  try {
    s->bind_pool.run(W, work);
  } catch (...) {
    return fail(SLM_ERR_HIP, "slm_bind_frames: could not start the bind workers");
  }
  for (int w = 0; w < W; ++w)
    if (rcs[w] != SLM_OK) return fail(rcs[w], errs[w].c_str());
  if (bind_trace_threshold() >= 0.0 && bt_now() - bt1 > bind_trace_threshold())
    fprintf(stderr, "[slm bind trace] batch of %d: drained the stream at %.3f (waited %.3f ms), workers done %.3f ms later\n",
            n_frames, bt1, bt1 - bt0, bt_now() - bt1);
  for (int w = 0; w < W; ++w) HIPCHK(hipStreamWaitEvent(st, s->bind_events[w], 0));   // join
  return SLM_OK;
}

}  // extern "C"

static int check_slots(slm_solver* s, int first, int n) {
  if (!s) return fail(SLM_ERR_INVALID, "null solver");
  if (first < 0 || n < 1 || first + n > (int)s->slots.size())
    return fail(SLM_ERR_INVALID, "slot range out of bounds");
  for (int i = first; i < first + n; ++i) {
    join_prepare(s, i);             // (a queued slm_prepare_model owns the slot until it is done; it leaves it unbound)
    if (!s->slots[i].h.bound) return fail(SLM_ERR_UNBOUND, "slot used before slm_bind_frame");
    // the slots of one launch share their per-surfel kernels, which are instantiated per num_neighbors
    if (s->slots[i].h.f.K != s->slots[first].h.f.K)
      return fail(SLM_ERR_UNSUPPORTED, "the frames of one batch must have the same num_neighbors");
  }
  return SLM_OK;
}

namespace {
struct BatchDims {
  int maxN = 0, maxJKe = 0, nt_max = 0, wb_cap = 0, n_reg_part = 0;
  int max_pos = 0, max_blocks = 0, maxP = 0;
  int K = 0;        // num_neighbors of the batch's slots (-1: they differ -- refused by the callers of the per-surfel kernels)
  bool v1 = true;   // every slot of the batch has a tuple-sorted plan
  bool vk = true;   // every slot of the batch takes the K-generic pair path (num_neighbors != 4 on the multifrontal solver)
  int gram_variants = 0;   // bit0: workgroup-merged records in use, bit1: per-run slab in use
  bool nd = true;   // every slot of the batch has a nested-dissection plan
  int max_tasks = 0;   // tasks of the persistent task-graph solver (maximum over the batch)
  // hybrid solve: every slot has the same number of levels and the same top-of-tree cut (-1: not available)
  int hybrid_cut = -2, hybrid_levels = -1, max_top_tasks = 0;
  std::vector<NDLevelSched> sched;   // per-level launch bounds over the batch
};
BatchDims dims_of(slm_solver* s, int first, int n) {
  BatchDims d;
  for (int i = first; i < first + n; ++i) {
    const FrameDev& h = s->slots[i].h;
    d.maxN = std::max(d.maxN, h.f.N);
    d.K = (d.K == 0 || d.K == h.f.K) ? h.f.K : -1;
    d.maxJKe = std::max(d.maxJKe, h.f.J * h.f.K_ED);
    d.nt_max = std::max(d.nt_max, h.nt);
    d.wb_cap = std::max(d.wb_cap, h.wb);
    d.max_pos = std::max(d.max_pos, h.n_pos);
    d.max_blocks = std::max(d.max_blocks, h.n_blocks);
    d.v1 = d.v1 && h.v1_ready;
    d.vk = d.vk && h.vk_ready;
    if (h.v1_ready) d.gram_variants |= h.v2_ready ? 1 : 2;
    d.nd = d.nd && h.nd_ready;
    d.max_tasks = std::max(d.max_tasks, h.nd_ready ? h.n_dag_tasks : 0);
    if (h.nd_ready) {
      const int cut = h.n_dag_top_tasks > 0 ? h.dag_cut_depth : -1;
      if (d.hybrid_cut == -2) { d.hybrid_cut = cut; d.hybrid_levels = h.n_levels; }
      else if (d.hybrid_cut != cut || d.hybrid_levels != h.n_levels) d.hybrid_cut = -1;
      d.max_top_tasks = std::max(d.max_top_tasks, h.n_dag_top_tasks);
    }
    d.maxP = std::max(d.maxP, h.P);
  }
  if (d.nd) {
    for (int i = first; i < first + n; ++i) {
      const auto& sc = s->slots[i].nd.sched;
      if (sc.size() > d.sched.size()) d.sched.resize(sc.size(), NDLevelSched{0, 0, 0, 0, 0, -2, 0, -2, 0, -2});
      for (size_t l = 0; l < sc.size(); ++l) {
        NDLevelSched& m = d.sched[l];
        // same first front and front count in every slot -> passed to the kernels by value
        if (m.first == -2) m.first = sc[l].first;
        else if (m.first != sc[l].first || m.n_fronts != sc[l].n_fronts) m.first = -1;
        m.n_fronts = std::max(m.n_fronts, sc[l].n_fronts);
        m.max_npt = std::max(m.max_npt, sc[l].max_npt);
        m.max_nt = std::max(m.max_nt, sc[l].max_nt);
        m.max_pairs = std::max(m.max_pairs, sc[l].max_pairs);
        m.max_n2p = std::max(m.max_n2p, sc[l].max_n2p);
        if (m.schur_at == -2) m.schur_at = sc[l].schur_at;
        else if (m.schur_at != sc[l].schur_at || m.n_schur != sc[l].n_schur) m.schur_at = -1;
        m.n_schur = std::max(m.n_schur, sc[l].n_schur);
        if (m.pull_at == -2) m.pull_at = sc[l].pull_at;
        else if (m.pull_at != sc[l].pull_at || m.n_pull != sc[l].n_pull) m.pull_at = -1;
        m.n_pull = std::max(m.n_pull, sc[l].n_pull);
      }
    }
  }
  if (d.nd && n > 1) {
    // slots with fewer levels than the batch maximum: their level tables must be read on the device
    for (int i = first; i < first + n; ++i)
      for (size_t l = s->slots[i].nd.sched.size(); l < d.sched.size(); ++l)
        d.sched[l].first = d.sched[l].schur_at = d.sched[l].pull_at = -1;
  }
  for (auto& m : d.sched) {
    if (m.first == -2) m.first = -1;
    if (m.first < 0) m.schur_at = m.pull_at = -1;
    if (m.schur_at == -2) m.schur_at = -1;
    if (m.pull_at == -2) m.pull_at = -1;
  }
  if (s->cfg.use_arap || s->cfg.use_rot)
    d.n_reg_part = std::min(kRegBlocksMax, (d.maxJKe + 255) / 256);
  return d;
}

// factor + substitutions of the assembled fronts: one persistent task-graph launch (solver_path 2) or the
// per-level launches (solver_path 0)
// solver_path 0 picks by batch size: the task graph is a latency scheduler (one or two frames per launch: the
// drop-in case, one frame at a time); larger batches are throughput-bound and run the per-level launches
// Round 5 (two workgroups per CU in the task graph): "small" is frames x nodes, not frames -- the task graph wins while its
// tasks find workgroups.  C2 (2 000 nodes), ms per LM iteration, task graph / hybrid: 3 frames 1.290 / 1.481, 4 frames
// 1.549 / 1.654, 6 frames 2.124 / 2.063, 8 frames 2.754 / 2.409; C1 (512 nodes) 8 frames 0.605 / 0.755; C4 (4 000 nodes)
// 1 frame 1.396 / 1.753, 4 frames 3.510 / 3.330.
static bool solve_is_task_graph(const slm_solver* s, int n, int J) {
  return s->cfg.solver_path == 2 || (s->cfg.solver_path == 0 && (n <= 2 || (long)n * J <= s->dag_max_nodes));
}
static bool solve_is_hybrid(const slm_solver* s, int n, const BatchDims& d) {
  return !solve_is_task_graph(s, n, d.maxP / 7) && (s->cfg.solver_path == 4 || (s->cfg.solver_path == 0 && s->hybrid_batches)) &&
         d.hybrid_cut >= 0 && d.hybrid_levels == (int)d.sched.size() && d.hybrid_cut + 1 < d.hybrid_levels;
}
// what this batch's solve needs reset before its task-graph launch (k_iter_begin_nd's dag_cut): -1 the whole tree, >= 0 the
// fronts of depth <= the hybrid cut, -2 nothing (per-level launches only)
static int dag_cut_of(const slm_solver* s, int n, const BatchDims& d) {
  if (solve_is_task_graph(s, n, d.maxP / 7)) return -1;
  return solve_is_hybrid(s, n, d) ? d.hybrid_cut : -2;
}
// dag_reset_done: this iteration's k_iter_begin_nd already reset the task graph's flags / mailboxes (launched with
// dag_cut_of(...)); otherwise launch_front_solve_dag resets them with a launch of its own
// dag_check_later: the caller's next launch (k_after_solve) settles a timed-out task-graph launch instead of k_dag_check
// Returns the mode word of the task-graph launch (bit 0: XCD-affine ticket streams -- the caller's completion check must then
// be unconditional, launch_front_solve_dag), -1 when the solve ran no task graph.
int enqueue_front_solve(slm_solver* s, const FrameDev* fr, int n, const BatchDims& d, double u_override, hipStream_t st,
                        bool dag_reset_done = false, bool dag_check_later = false) {
  const bool dag = solve_is_task_graph(s, n, d.maxP / 7);
  // batches: the levels with many fronts as launches (throughput-bound), the top of the tree -- a chain of ~20
  // dependent tile columns with a handful of fronts -- as tasks of ONE persistent launch for all frames
  const bool hybrid = solve_is_hybrid(s, n, d);
  s->last_solver_form = dag ? 1 : (hybrid ? 2 : 0);
  int mode = -1;
  if (dag) {
    mode = launch_front_solve_dag(fr, n, d.max_tasks, u_override, st, -1, !dag_reset_done, !dag_check_later);
  } else if (hybrid) {
    const int n_levels = (int)d.sched.size(), l_cut = n_levels - 1 - d.hybrid_cut;
    launch_front_levels(fr, n, d.sched.data(), n_levels, l_cut, 0, u_override, st);
    // the task graph: the fronts above the cut, then the back substitution of the WHOLE tree (the list dag_top_tasks
    // ends with the BACKB / BACK tasks of the deeper fronts -- 24 small per-level launches, 0.25 ms at C2, otherwise)
    mode = launch_front_solve_dag(fr, n, d.max_top_tasks, u_override, st, d.hybrid_cut, !dag_reset_done, !dag_check_later);
  } else {
    launch_front_solve(fr, n, d.sched.data(), (int)d.sched.size(), u_override, st);
  }
  s->last_dag_mode = mode;
  return mode;
}

// zero the fronts of slots [first, first+n) and assemble JtJ / jtl into them
hipError_t enqueue_assemble_nd(slm_solver* s, int first, int n, const BatchDims& d, hipStream_t st) {
  const FrameDev* fr = s->frames_dev + first;
  launch_iter_begin_nd(fr, n, st, nullptr, -2);   // zeroes the pivot columns of the fronts of all n slots in one launch
  if (s->cfg.use_data) {
    launch_data_eval(fr, n, kLossBlocks, s->cfg.w_data, 2, st);   // {r, c} at the current beta (clobbers the loss partials)
    launch_data_gram(fr, n, d.max_pos, s->cfg.w_data, d.gram_variants, st);
    launch_front_assemble(fr, n, d.max_blocks, st);
  }
  launch_reg_grad_nd(fr, n, d.maxP / 7, s->cfg.use_arap, s->cfg.w_arap, s->cfg.use_rot, s->cfg.w_rot, st);
  if (!s->cfg.use_arap && !s->cfg.use_rot) launch_front_load_rhs(fr, n, d.maxP, st);   // (else k_reg_grad_nd did it)
  return hipSuccess;
}

void enqueue_assemble(slm_solver* s, const FrameDev* fr, int n, const BatchDims& d, hipStream_t st) {
  launch_iter_begin(fr, n, st);
  if (s->cfg.use_data) {
    if (d.v1) {
      launch_data_eval(fr, n, kLossBlocks, s->cfg.w_data, 2, st);
      launch_data_gram(fr, n, d.max_pos, s->cfg.w_data, d.gram_variants, st);
      launch_band_assemble(fr, n, d.max_blocks, st);
    } else {
      launch_data_grad(fr, n, d.maxN, d.K, s->cfg.w_data, st);
    }
  }
  launch_reg_grad(fr, n, d.maxJKe, s->cfg.use_arap, s->cfg.w_arap, s->cfg.use_rot, s->cfg.w_rot, st);
}

void enqueue_loss(slm_solver* s, const FrameDev* fr, int n, const BatchDims& d, int use_delta,
                  hipStream_t st) {
  if (s->cfg.use_data) launch_data_loss(fr, n, kLossBlocks, d.K, s->cfg.w_data, use_delta, st);
  if (d.n_reg_part > 0)
    launch_reg_loss(fr, n, d.n_reg_part, s->cfg.use_arap, s->cfg.w_arap, s->cfg.use_rot,
                    s->cfg.w_rot, use_delta, st);
}
}  // namespace

extern "C" {

static hipEvent_t take_event(slm_solver* s) {
  if (!s->ev_pool.empty()) {
    hipEvent_t e = s->ev_pool.back();
    s->ev_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

// ---- one frame sharded over several GPUs ----------------------------------------------------
int slm_set_shard(slm_solver* s, int32_t rank, int32_t world) {
  if (!s || world < 1 || rank < 0 || rank >= world) return fail(SLM_ERR_INVALID, "slm_set_shard: bad rank/world");
  if (s->cfg.data_path != 0 || s->cfg.solver_path == 1)
    return fail(SLM_ERR_UNSUPPORTED, "slm_set_shard: needs data_path 0 and a nested-dissection solver_path (0, 2, 3 or 4)");
  s->rank = rank;
  s->world = world;
  s->shard_mode = true;
  for (size_t i = 0; i < s->slots.size(); ++i) {   // the shares are fixed at bind time
    join_prepare(s, (int)i);
    s->slots[i].model_ready = false;
    s->slots[i].h.bound = 0;
  }
  HIPCHK(hipMemset(s->frames_dev, 0, sizeof(FrameDev) * s->slots.size()));
  return SLM_OK;
}

static int shard_dims(slm_solver* s, int n_frames, BatchDims& d) {
  int rc = check_slots(s, 0, n_frames);
  if (rc) return rc;
  d = dims_of(s, 0, n_frames);
  if (!d.nd || !(d.v1 || d.vk))
    return fail(SLM_ERR_UNSUPPORTED, "sharded LM step: every slot needs a multifrontal data path (tuple-sorted or K-generic) and the ND solver");
  return SLM_OK;
}

int slm_lm_grad_local(slm_solver* s, int32_t n_frames, void* stream) {
  BatchDims d;
  int rc = shard_dims(s, n_frames, d);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const FrameDev* fr = s->frames_dev;
  const int* reuse = (d.v1 && s->cfg.phase_test && !s->no_reuse) ? s->reuse_dev : nullptr;
  launch_iter_begin_nd(fr, n_frames, st, reuse, -2);
  if (s->cfg.use_data) {
    if (d.v1) {
      launch_data_eval(fr, n_frames, kLossBlocks, s->cfg.w_data, 1, st, reuse);   // (only slots whose buffer is not the current beta's)
      launch_data_gram(fr, n_frames, d.max_pos, s->cfg.w_data, d.gram_variants, st, reuse);
      launch_pair_reduce(fr, n_frames, d.max_blocks, st);
    } else {
      launch_data_grad_pairs(fr, n_frames, d.maxN, d.K, s->cfg.w_data, st);   // K-generic: this rank's share of the surfel list, straight into the pair records
    }
  }
  HIPCHK(hipGetLastError());
  return SLM_OK;
}

int slm_lm_solve(slm_solver* s, int32_t n_frames, void* stream) {
  BatchDims d;
  int rc = shard_dims(s, n_frames, d);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const FrameDev* fr = s->frames_dev;
  const slm_config& c = s->cfg;
  if (c.use_data) launch_pair_scatter(fr, n_frames, d.max_blocks, st);
  launch_reg_grad_nd(fr, n_frames, d.maxP / 7, c.use_arap, c.w_arap, c.use_rot, c.w_rot, st);
  if (!c.use_arap && !c.use_rot) launch_front_load_rhs(fr, n_frames, d.maxP, st);   // (else k_reg_grad_nd did it)
  enqueue_front_solve(s, fr, n_frames, d, -1.0, st);
  HIPCHK(hipGetLastError());
  return SLM_OK;
}

int slm_lm_loss_local(slm_solver* s, int32_t n_frames, void* stream) {
  BatchDims d;
  int rc = shard_dims(s, n_frames, d);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const FrameDev* fr = s->frames_dev;
  const slm_config& c = s->cfg;
  if (c.use_data) {
    launch_make_trial(fr, n_frames, d.maxJKe, st);
    if (d.v1) launch_data_eval(fr, n_frames, kLossBlocks, c.w_data, 0, st);   // this rank's positions: loss partials + the evaluation buffer
    else launch_data_loss(fr, n_frames, kLossBlocks, d.K, c.w_data, 1, st);   // K-generic: this rank's surfels [sf_lo, sf_hi)
  }
  if (d.n_reg_part > 0)
    launch_reg_loss(fr, n_frames, d.n_reg_part, c.use_arap, c.w_arap, c.use_rot, c.w_rot, 1, st);
  HIPCHK(hipGetLastError());
  return SLM_OK;
}

int slm_lm_accept(slm_solver* s, int32_t n_frames, void* stream) {
  BatchDims d;
  int rc = shard_dims(s, n_frames, d);
  if (rc) return rc;
  launch_accept(s->frames_dev, n_frames, s->cfg.phase_test, d.n_reg_part, std::max(s->cfg.num_iterations, 1), (hipStream_t)stream,
                (d.v1 && s->cfg.phase_test) ? s->reuse_dev : nullptr, (s->cfg.use_data && d.v1) ? 1 : 0);   // (slm_lm_loss_local ran k_data_eval)
  HIPCHK(hipGetLastError());
  return SLM_OK;
}

static int exchange_buf(slm_solver* s, int slot, int what, double** p, int64_t* n) {
  int rc = check_slots(s, slot, 1);
  if (rc) return rc;
  const FrameDev& h = s->slots[slot].h;
  switch (what) {
    case SLM_X_PAIR_BLOCKS:
      if (!h.pairbuf) return fail(SLM_ERR_UNBOUND, "slm_lm_exchange: the slot was bound without slm_set_shard");
      *p = h.pairbuf;
      *n = (int64_t)h.n_blocks * SLM_WREC + (h.vk_ready ? SLM_VK_TAIL : 1);   // records + matched count (spread on the K-generic path)
      return SLM_OK;
    case SLM_X_DELTA:
      *p = h.delta;
      *n = h.P;
      return SLM_OK;
    case SLM_X_DATA_LOSS:
      *p = h.loss_part;
      *n = 2 * (int64_t)h.n_loss_part;
      return SLM_OK;
  }
  return fail(SLM_ERR_INVALID, "slm_lm_exchange: unknown buffer");
}

int slm_lm_exchange_size(slm_solver* s, int32_t slot, int32_t what, int64_t* n_doubles) {
  double* p;
  if (!n_doubles) return fail(SLM_ERR_INVALID, "slm_lm_exchange_size: null output");
  return exchange_buf(s, slot, what, &p, n_doubles);
}

int slm_lm_exchange_ptr(slm_solver* s, int32_t slot, int32_t what, double** ptr_out, int64_t* n_doubles) {
  if (!ptr_out || !n_doubles) return fail(SLM_ERR_INVALID, "slm_lm_exchange_ptr: null output");
  return exchange_buf(s, slot, what, ptr_out, n_doubles);
}

int slm_lm_exchange_get(slm_solver* s, int32_t slot, int32_t what, double* out, void* stream) {
  double* p;
  int64_t n;
  int rc = exchange_buf(s, slot, what, &p, &n);
  if (rc) return rc;
  if (!out) return fail(SLM_ERR_INVALID, "slm_lm_exchange_get: null output");
  HIPCHK(hipMemcpyAsync(out, p, sizeof(double) * n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return SLM_OK;
}

int slm_lm_exchange_set(slm_solver* s, int32_t slot, int32_t what, const double* in, void* stream) {
  double* p;
  int64_t n;
  int rc = exchange_buf(s, slot, what, &p, &n);
  if (rc) return rc;
  if (!in) return fail(SLM_ERR_INVALID, "slm_lm_exchange_set: null input");
  HIPCHK(hipMemcpyAsync(p, in, sizeof(double) * n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return SLM_OK;
}

// One LM iteration of slots [first, first + n) on `st`.
static void enqueue_lm_iteration(slm_solver* s, int first, int n, const BatchDims& d, hipStream_t st, bool first_iteration) {
  const FrameDev* fr = s->frames_dev + first;
  const slm_config& c = s->cfg;
  std::vector<hipEvent_t>* evs = nullptr;
  if (s->profile) {
    s->ev_runs.emplace_back();
    evs = &s->ev_runs.back();
  }
  auto mark = [&]() {
    if (!evs) return;
    hipEvent_t e = take_event(s);
    if (e) (void)hipEventRecord(e, st);
    evs->push_back(e);
  };
  mark();
  // (records of the Jacobian pass are reused after a rejected step on the multifrontal path, where the assembly
  //  re-reads them; the banded path adds into the band in place)
  const int* reuse = (d.nd && d.v1 && c.phase_test && !s->no_reuse) ? s->reuse_dev + first : nullptr;
  // (round 6) the zeroing rides on the Jacobian pass's launch when every slot takes the workgroup-merged records
  const bool fused_begin = d.nd && d.v1 && c.use_data && d.gram_variants == 1 && d.max_pos > 0 && s->fuse_begin;
  if (d.nd) {
    if (!fused_begin) launch_iter_begin_nd(fr, n, st, reuse, dag_cut_of(s, n, d));
  } else {
    launch_iter_begin(fr, n, st);
  }
  mark();
  if (c.use_data) {
    if (d.v1) {
      // The Jacobian pass reads {r, c} of every position from the evaluation buffer.  Inside the loop the buffer is
      // what the loss pass of the previous iteration left at the accepted trial point (a rejected step reuses the
      // records and runs no Jacobian pass at all); a pass of its own is only needed at the first iteration of a run
      // (skipped on the device for slots whose buffer is valid) and after a reject when records are not reused.
      if (first_iteration || (c.phase_test && !reuse)) launch_data_eval(fr, n, kLossBlocks, c.w_data, 1, st, reuse);
      if (fused_begin) launch_begin_and_gram(fr, n, d.max_pos, c.w_data, st, reuse, dag_cut_of(s, n, d));
      else launch_data_gram(fr, n, d.max_pos, c.w_data, d.gram_variants, st, reuse);
    } else if (d.nd && d.vk) {
      launch_data_grad_pairs(fr, n, d.maxN, d.K, c.w_data, st);   // K-generic: per-pair records (zeroed, then filled)
    } else {
      launch_data_grad(fr, n, d.maxN, d.K, c.w_data, st);
    }
  }
  mark();
  if (d.nd) {
    if (c.use_data) {
      if (d.v1) launch_front_assemble(fr, n, d.max_blocks, st);
      else launch_pair_scatter(fr, n, d.max_blocks, st);          // K-generic: records -> fronts + jtl
    }
    launch_reg_grad_nd(fr, n, d.maxP / 7, c.use_arap, c.w_arap, c.use_rot, c.w_rot, st);
    if (!s->cfg.use_arap && !s->cfg.use_rot) launch_front_load_rhs(fr, n, d.maxP, st);   // (else k_reg_grad_nd did it)
  } else {
    if (c.use_data && d.v1) launch_band_assemble(fr, n, d.max_blocks, st);
    launch_reg_grad(fr, n, d.maxJKe, c.use_arap, c.w_arap, c.use_rot, c.w_rot, st);
  }
  mark();
  // the trial point beta + delta (node_pk_try), the regularisers' loss there and the task graph's abort check: one launch
  const bool tail = c.use_data || d.n_reg_part > 0;
  int dag_mode = -1;
  if (d.nd) dag_mode = enqueue_front_solve(s, fr, n, d, -1.0, st, true, tail);
  else launch_band_solve(fr, n, d.nt_max, d.wb_cap, -1.0, st);
  if (tail)   // (dag_check 2 behind an XCD-affine task-graph launch: the completion check runs whether or not the abort flag is up)
    launch_after_solve(fr, n, c.use_data ? d.maxP / 7 : 0, d.n_reg_part, c.use_arap, c.w_arap, c.use_rot, c.w_rot,
                       (d.nd && dag_cut_of(s, n, d) >= -1) ? ((dag_mode > 0 && (dag_mode & 1)) ? 2 : 1) : 0, st);
  mark();
  if (c.use_data) {
    if (d.v1) launch_data_eval(fr, n, kLossBlocks, c.w_data, 0, st);   // the loss pass; its {r, c} feed the next Jacobian pass
    else launch_data_loss(fr, n, kLossBlocks, d.K, c.w_data, 1, st);
  }
  mark();
  launch_accept(fr, n, c.phase_test, d.n_reg_part, std::max(c.num_iterations, 1), st,
                (d.v1 && c.phase_test) ? s->reuse_dev + first : nullptr, (c.use_data && d.v1) ? 1 : 0);
  mark();
}

int slm_run(slm_solver* s, int32_t n_frames, void* stream) {
  int rc = check_slots(s, 0, n_frames);
  if (rc) return rc;
  if (s->world > 1)
    return fail(SLM_ERR_UNSUPPORTED,
                "slm_run: the frame is sharded; drive slm_lm_grad_local / slm_lm_solve / slm_lm_loss_local / "
                "slm_lm_accept with the exchanges between them");
  hipStream_t st = (hipStream_t)stream;
  const slm_config& c = s->cfg;
  BatchDims d = dims_of(s, 0, n_frames);
  if (!d.nd) {
    // A batch runs ONE solver form.  When some slot has no nested-dissection plan (a frame without surfels, a graph
    // the analysis refuses) all of them take the block-banded path -- whose storage the slots WITH a plan have not
    // been given at bind time: provide it now (a read-back per such slot; the rare path).
    std::lock_guard<std::mutex> lock(s->band_mutex);
    for (int i = 0; i < n_frames; ++i) {
      rc = ensure_band(s, i, st);
      if (rc) return rc;
    }
    d = dims_of(s, 0, n_frames);
  }
  for (int it = 0; it < c.num_iterations; ++it) enqueue_lm_iteration(s, 0, n_frames, d, st, it == 0);
  // The reuse flag of a slot says "the Gram records in HBM were computed at the slot's CURRENT beta".  k_accept keeps it
  // on every path that writes records (banded path included: a later multifrontal run may then reuse them); a run that
  // wrote none (per-entry atomics) leaves nothing to reuse.
  if (!(d.v1 && c.phase_test) && n_frames > 0) HIPCHK(hipMemsetAsync(s->reuse_dev, 0, sizeof(int) * n_frames, st));
  HIPCHK(hipGetLastError());
  return SLM_OK;
}

int slm_get_plan_info(slm_solver* s, int32_t slot, double* out_caller, int32_t capacity) {
  int rc = check_slots(s, slot, 1);
  if (rc) return rc;
  if (!out_caller || capacity < 0) return fail(SLM_ERR_INVALID, "slm_get_plan_info: bad output");
  const Slot& sl = s->slots[slot];
  const FrameDev& h = sl.h;
  double out[SLM_PLAN_INFO_DOUBLES];
  for (int i = 0; i < SLM_PLAN_INFO_DOUBLES; ++i) out[i] = 0.0;
  if (h.nd_ready) {
    out[0] = 0.0;
    out[1] = (double)sl.nd.fronts.size();
    out[2] = (double)sl.nd.level_start.size() - 1.0;
    out[3] = sl.nd.flops;
    out[10] = sl.nd.flops_exact;
    out[11] = (double)sl.nd.dag_tasks.size() / 2.0;
    out[4] = 8.0 * (double)sl.nd.tile_doubles;
    out[12] = (double)sl.n_piv_tiles;
    out[13] = (double)sl.n_pure_tiles;
  } else {
    out[0] = 1.0;
    const double w = (double)h.wb * SLM_NB;
    out[3] = (double)h.P * w * w;
    out[4] = 8.0 * (double)h.nt * (h.wb + 1) * SLM_NB * SLM_NB;
  }
  out[5] = h.n_tuples;
  out[6] = h.n_runs;
  out[7] = h.n_blocks;
  out[8] = h.v1_ready && h.v2_ready ? h.n_wblk : 0;
  out[9] = h.n_pos;
  for (int i = 0; i < capacity; ++i) out_caller[i] = i < SLM_PLAN_INFO_DOUBLES ? out[i] : 0.0;
  return SLM_OK;
}

int slm_profile_enable(slm_solver* s, int32_t on) {
  if (!s) return fail(SLM_ERR_INVALID, "slm_profile_enable: null solver");
  s->profile = on != 0;
  return SLM_OK;
}

int slm_profile_read(slm_solver* s, double* ms_out, int64_t* count_out) {
  if (!s || !ms_out || !count_out) return fail(SLM_ERR_INVALID, "slm_profile_read: null argument");
  for (int p = 0; p < SLM_PH_COUNT; ++p) {
    ms_out[p] = 0.0;
    count_out[p] = 0;
  }
  for (size_t r = 0; r < s->ev_runs.size(); ++r) {
    auto& evs = s->ev_runs[r];
    if ((int)evs.size() == SLM_PH_COUNT + 1 && evs.back()) {
      HIPCHK(hipEventSynchronize(evs.back()));
      for (int p = 0; p < SLM_PH_COUNT; ++p) {
        if (!evs[p] || !evs[p + 1]) continue;
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, evs[p], evs[p + 1]));
        ms_out[p] += ms;
        count_out[p] += 1;
      }
    }
    for (hipEvent_t e : evs)
      if (e) s->ev_pool.push_back(e);
  }
  s->ev_runs.clear();
  return SLM_OK;
}

int slm_get_beta(slm_solver* s, int32_t slot, double* out, void* stream) {
  int rc = check_slots(s, slot, 1);
  if (rc) return rc;
  if (!out) return fail(SLM_ERR_INVALID, "slm_get_beta: null output");
  const FrameDev& h = s->slots[slot].h;
  HIPCHK(hipMemcpyAsync(out, h.beta, sizeof(double) * h.P, hipMemcpyDeviceToDevice,
                        (hipStream_t)stream));
  return SLM_OK;
}

int slm_set_beta(slm_solver* s, int32_t slot, const double* in, void* stream) {
  int rc = check_slots(s, slot, 1);
  if (rc) return rc;
  if (!in) return fail(SLM_ERR_INVALID, "slm_set_beta: null input");
  const FrameDev& h = s->slots[slot].h;
  hipStream_t st = (hipStream_t)stream;
  // fresh LM state (u0, minimal_loss0, records), then the caller's beta
  launch_init_slot(s->frames_dev, slot, h.f.J, s->cfg, st);
  HIPCHK(hipMemsetAsync(s->reuse_dev + slot, 0, sizeof(int), st));   // another beta: the kept records do not apply
  HIPCHK(hipMemcpyAsync(h.beta, in, sizeof(double) * h.P, hipMemcpyDeviceToDevice, st));
  launch_pack_nodes(s->frames_dev, slot, h.f.J, st);
  return SLM_OK;
}

int slm_get_records(slm_solver* s, int32_t slot, slm_iter_record* host_out, int32_t max_records,
                    void* stream) {
  int rc = check_slots(s, slot, 1);
  if (rc) return rc;
  if (!host_out || max_records < 0) return fail(SLM_ERR_INVALID, "slm_get_records: bad output");
  int n = std::min(max_records, s->cfg.num_iterations);
  hipStream_t st = (hipStream_t)stream;
  if (n > 0)
    HIPCHK(hipMemcpyAsync(host_out, s->slots[slot].h.rec, sizeof(slm_iter_record) * n,
                          hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
#ifdef SLM_STAMPS
  if (s->slots[slot].h.dbg) {
    unsigned long long t[64];
    HIPCHK(hipMemcpy(t, s->slots[slot].h.dbg, sizeof(t), hipMemcpyDeviceToHost));
    fprintf(stderr, "[slm stamps slot %d] deltas (cycles):", slot);
    for (int i = 1; i < 16; ++i) fprintf(stderr, " %d:%lld", i, t[i] ? (long long)(t[i] - t[i - 1]) : -1LL);
    fprintf(stderr, "\n");
  }
#endif
  return SLM_OK;
}

static int clear_flags(slm_solver* s, int slot, hipStream_t st) {
  // stopped / counters live at the tail of LMState: zero {iter..pad}
  LMState* p = s->slots[slot].h.st;
  HIPCHK(hipMemsetAsync(&p->stopped, 0, sizeof(int32_t) * 4, st));
  return SLM_OK;
}

int slm_assemble(slm_solver* s, int32_t slot, double* jtj_dense, double* jtl, void* stream) {
  int rc = check_slots(s, slot, 1);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  rc = clear_flags(s, slot, st);
  if (rc) return rc;
  rc = ensure_band(s, slot, st);
  if (rc) return rc;
  const BatchDims d = dims_of(s, slot, 1);
  const FrameDev& h = s->slots[slot].h;
  enqueue_assemble(s, s->frames_dev + slot, 1, d, st);
  if (jtj_dense) launch_band_to_dense(s->frames_dev, slot, jtj_dense, st);
  if (jtl)
    HIPCHK(hipMemcpyAsync(jtl, h.rhs, sizeof(double) * h.P, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipGetLastError());
  return SLM_OK;
}

int slm_loss(slm_solver* s, int32_t slot, double* out, void* stream) {
  int rc = check_slots(s, slot, 1);
  if (rc) return rc;
  if (!out) return fail(SLM_ERR_INVALID, "slm_loss: null output");
  hipStream_t st = (hipStream_t)stream;
  rc = clear_flags(s, slot, st);
  if (rc) return rc;
  const BatchDims d = dims_of(s, slot, 1);
  enqueue_loss(s, s->frames_dev + slot, 1, d, 0, st);
  launch_loss_out(s->frames_dev, slot, d.n_reg_part, out, st);
  HIPCHK(hipGetLastError());
  return SLM_OK;
}

int slm_solve(slm_solver* s, int32_t slot, double u, double* delta, int32_t* status, void* stream) {
  int rc = check_slots(s, slot, 1);
  if (rc) return rc;
  if (!delta || !(u >= 0.0)) return fail(SLM_ERR_INVALID, "slm_solve: bad argument");
  hipStream_t st = (hipStream_t)stream;
  rc = clear_flags(s, slot, st);
  if (rc) return rc;
  const BatchDims d = dims_of(s, slot, 1);
  const FrameDev& h = s->slots[slot].h;
  if (d.nd) {
    HIPCHK(enqueue_assemble_nd(s, slot, 1, d, st));
    enqueue_front_solve(s, s->frames_dev + slot, 1, d, u, st);
  } else {
    enqueue_assemble(s, s->frames_dev + slot, 1, d, st);
    launch_band_solve(s->frames_dev + slot, 1, d.nt_max, d.wb_cap, u, st);
  }
  HIPCHK(hipMemcpyAsync(delta, h.delta, sizeof(double) * h.P, hipMemcpyDeviceToDevice, st));
  if (status)
    HIPCHK(hipMemcpyAsync(status, &h.st->chol_fail, sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  HIPCHK(hipGetLastError());
  return SLM_OK;
}

int slm_solve_dense(int32_t P, const double* A, const double* b, double* x, int32_t* status,
                    void* stream) {
  if (P < 1 || !A || !b || !x) return fail(SLM_ERR_INVALID, "slm_solve_dense: bad argument");
  hipStream_t st = (hipStream_t)stream;
  FrameDev h{};
  h.P = P;
  h.nt = (P + SLM_NB - 1) / SLM_NB;
  h.wb = h.nt - 1;
  h.bound = 1;
  const size_t nvec = (size_t)h.nt * SLM_NB;
  const size_t nband = (size_t)h.nt * (h.wb + 1) * SLM_NB * SLM_NB;
  const size_t nlinv = (size_t)h.nt * SLM_NB * SLM_NB;
  char* ws = nullptr;
  const size_t bytes = sizeof(double) * (2 * nvec + nband + nlinv) + sizeof(LMState) + sizeof(FrameDev);
  HIPCHK(hipMalloc((void**)&ws, bytes));
  double* p = reinterpret_cast<double*>(ws);
  h.delta = p;
  h.rhs = p + nvec;
  h.band = p + 2 * nvec;
  h.linv = h.band + nband;
  h.st = reinterpret_cast<LMState*>(h.linv + nlinv);
  FrameDev* fdev = reinterpret_cast<FrameDev*>(h.st + 1);
  int rc = SLM_OK;
  hipError_t e = hipMemsetAsync(h.st, 0, sizeof(LMState), st);
  if (e == hipSuccess) e = hipMemcpyAsync(fdev, &h, sizeof(FrameDev), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) {
    launch_dense_to_band(fdev, A, b, st);
    launch_band_solve(fdev, 1, h.nt, h.wb, 0.0, st);
    e = hipMemcpyAsync(x, h.delta, sizeof(double) * P, hipMemcpyDeviceToDevice, st);
  }
  if (e == hipSuccess && status)
    e = hipMemcpyAsync(status, &h.st->chol_fail, sizeof(int32_t), hipMemcpyDeviceToDevice, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess) {
    g_err = std::string("slm_solve_dense: ") + hipGetErrorString(e);
    rc = SLM_ERR_HIP;
  }
  (void)hipFree(ws);
  return rc;
}

int slm_data_residuals(slm_solver* s, int32_t slot, double* r, uint8_t* match, int32_t* taps,
                       void* stream) {
  int rc = check_slots(s, slot, 1);
  if (rc) return rc;
  const FrameDev& h = s->slots[slot].h;
  launch_data_resid(s->frames_dev, slot, h.f.N, h.f.K, s->cfg.w_data, r, match, taps, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return SLM_OK;
}

int slm_apply_update(int32_t N, int32_t J, int32_t K, float* sf_points, float* sf_norms,
                     const int32_t* sf_knn_idx, const float* sf_knn_w, float* ed_points,
                     float* ed_norms, const double* beta, void* stream) {
  if (K < 1 || K > 8) return fail(SLM_ERR_UNSUPPORTED, "slm_apply_update: num_neighbors must be in 1..8");
  if (N < 0 || J < 1 || !ed_points || !ed_norms || !beta || (N > 0 && (!sf_points || !sf_norms ||
      !sf_knn_idx || !sf_knn_w)))
    return fail(SLM_ERR_INVALID, "slm_apply_update: bad argument");
  launch_update(N, J, K, sf_points, sf_norms, sf_knn_idx, sf_knn_w, ed_points, ed_norms, beta,
                (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return SLM_OK;
}

int slm_apply_update_f64(int32_t N, int32_t J, int32_t K, double* sf_points, double* sf_norms,
                         const int32_t* sf_knn_idx, const double* sf_knn_w, double* ed_points,
                         double* ed_norms, const double* beta, void* stream) {
  if (K < 1 || K > 8) return fail(SLM_ERR_UNSUPPORTED, "slm_apply_update_f64: num_neighbors must be in 1..8");
  if (N < 0 || J < 1 || !ed_points || !ed_norms || !beta || (N > 0 && (!sf_points || !sf_norms ||
      !sf_knn_idx || !sf_knn_w)))
    return fail(SLM_ERR_INVALID, "slm_apply_update_f64: bad argument");
  launch_update64(N, J, K, sf_points, sf_norms, sf_knn_idx, sf_knn_w, ed_points, ed_norms, beta,
                  (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return SLM_OK;
}

int slm_knn(int32_t Nq, int32_t Nn, int32_t K, int32_t skip_self, const float* q, const float* nodes,
            int32_t* idx, float* dist, void* stream) {
  if (Nq < 0 || Nn < 1 || K < 1 || K + (skip_self ? 1 : 0) > 9 || !nodes || !idx || !dist ||
      (Nq > 0 && !q))
    return fail(SLM_ERR_INVALID, "slm_knn: bad argument (K + skip_self <= 9)");
  // fewer nodes than neighbours asked for would leave index -1 / distance inf in the tables that
  // slm_bind_frame, slm_knn_weights and slm_apply_update index with (the reference's knn_points fails too)
  if (Nn < K + (skip_self ? 1 : 0))
    return fail(SLM_ERR_INVALID, "slm_knn: fewer nodes than K (+ self)");
  launch_knn(Nq, Nn, K, skip_self, q, nodes, idx, dist, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return SLM_OK;
}

int slm_knn_weights(int32_t Nq, int32_t K, int32_t radius_mode, const int32_t* idx, const float* dist,
                    const float* radii, float* w, uint8_t* stable, void* stream) {
  if (Nq < 0 || K < 1 || K > 9 || !idx || !dist || !radii || !w)
    return fail(SLM_ERR_INVALID, "slm_knn_weights: bad argument");
  launch_knn_weights(Nq, K, radius_mode, idx, dist, radii, w, stable, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return SLM_OK;
}

int slm_knn_f64(int32_t Nq, int32_t Nn, int32_t K, int32_t skip_self, const double* q, const double* nodes,
                const int32_t* q_seg, const int32_t* node_seg, int32_t* idx, double* dist, void* stream) {
  if (Nq < 0 || Nn < 1 || K < 1 || K + (skip_self ? 1 : 0) > 9 || !nodes || !idx || !dist || (Nq > 0 && !q) ||
      ((q_seg == nullptr) != (node_seg == nullptr)))
    return fail(SLM_ERR_INVALID, "slm_knn_f64: bad argument (K + skip_self <= 9; both class arrays or none)");
  if (Nn < K + (skip_self ? 1 : 0)) return fail(SLM_ERR_INVALID, "slm_knn_f64: fewer nodes than K (+ self)");
  hipStream_t st = (hipStream_t)stream;
  int* counter = nullptr;
  if (q_seg) {
    HIPCHK(hipMalloc((void**)&counter, sizeof(int)));
    HIPCHK(hipMemsetAsync(counter, 0, sizeof(int), st));
  }
  launch_knn64(Nq, Nn, K, skip_self, q, nodes, q_seg, node_seg, idx, dist, counter, st);
  HIPCHK(hipGetLastError());
  if (counter) {
    int n_short = 0;
    HIPCHK(hipMemcpyAsync(&n_short, counter, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipFree(counter));
    if (n_short > 0)
      return fail(SLM_ERR_INVALID, "slm_knn_f64: a class has fewer nodes than the neighbours asked for");
  }
  return SLM_OK;
}

int slm_knn_weights_f64(int32_t Nq, int32_t K, int32_t radius_mode, const int32_t* idx, const double* dist,
                        const double* radii, int32_t num_classes, const double* q_seg_conf,
                        const double* node_seg_conf, double* w, uint8_t* stable, void* stream) {
  if (Nq < 0 || K < 1 || K > 9 || !idx || !dist || !radii || !w || num_classes < 0 || num_classes > SLM_MAX_CLASSES ||
      (num_classes > 0 && (!q_seg_conf || !node_seg_conf)))
    return fail(SLM_ERR_INVALID, "slm_knn_weights_f64: bad argument");
  launch_knn_weights64(Nq, K, radius_mode, idx, dist, radii, num_classes, q_seg_conf, node_seg_conf, w, stable,
                       (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return SLM_OK;
}

}  // extern "C"
