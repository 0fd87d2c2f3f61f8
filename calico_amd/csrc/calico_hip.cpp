// calico_hip.cpp — host side of libcalico_hip.so: the C ABI of
// include/calico_hip.h, problem flattening, and the LM driver loop.
//
// What the reference does per Optimize() call (batch_optimizer.cpp:53-81) —
// build a ceres::Problem from the sensors / world model / trajectory, run
// ceres::Solve, re-evaluate the residual blocks — maps here to:
//   add_* calls  -> host-side block / observation tables,
//   finalize()   -> cells, work items, gather lists, device upload,
//   calico_solve -> device-resident LM (kernels in eval_kernels.hip and
//                   solve_kernels.hip; this file only enqueues them and reads
//                   back one small state struct per iteration),
//   calico_get_residuals -> cost-only kernel without the loss function.
// There is no CPU compute path in this library.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>     // types only: the library itself is loaded on first use (RcclApi below)
#include <dlfcn.h>
#include <link.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <array>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/calico_hip.h"
#include "calico_hip_testing.h"
#include "problem_dev.hpp"
#include "shard.hpp"

namespace cal {

// ---- kernels (eval_kernels.hip / solve_kernels.hip) -------------------------
void launch_eval(const EvalArgs& a, bool jac, hipStream_t stream);
void launch_eval_frames(const EvalArgs& a, hipStream_t stream);
void launch_eval_jacobian(const EvalArgs& a, hipStream_t stream);
void launch_expand_cells(const EvalArgs& a, hipStream_t stream);
void launch_residual_heatmap(const double* res, const uint8_t* valid, const uint8_t* active, const double* px, const double* py,
                             int begin, int end, int width, int height, int num_rows, int num_cols, double* rmse, long long* count,
                             hipStream_t s);
void launch_inlier_mask(const double* res, uint8_t* valid_then_mask, const uint8_t* active, int begin, int end, int dim, double threshold,
                        hipStream_t s);
void launch_mark_outliers(const double* res, const uint8_t* valid, uint8_t* active, int begin, int end, int dim,
                          double threshold, int* n_marked, hipStream_t s);
hipError_t configure_eval_kernels(size_t max_lds_bytes);

void launch_gather(double* R, const double* src, const int* out_idx_thin, const int64_t* ptr_thin, const int* idx_thin,
                   int n_thin, int n_thin8, int n_thin4, int thin_per_lane, const int* out_idx_fat, const int64_t* ptr_fat, const int* idx_fat, int n_fat,
                   const double* cost_src, int n_cost, const LmState* st, int need_flag, size_t other_stride, hipStream_t s, const ControlTail* tail = nullptr);
void launch_gather_lists(const GatherStruct& gs, int n_out, int* cnt, int* out_idx, int64_t* ptr, int* idx, int zero_slot, long long* scratch,
                         hipStream_t s);
size_t gather_fixed_entries(int n_thin, int n_thin8, int n_thin4, int thin_per_lane);
void launch_gather_pack_fixed(const int64_t* ptr, const int* idx, int n_thin, int n_thin8, int n_thin4, int thin_per_lane, int zero_slot, int* out,
                              hipStream_t s);
void launch_post_eval(const SolveArgs& a, const double* x, const BlockDev* blocks, int n_blocks, const LmOptionsDev& o,
                      IterLog* log, int log_cap, int first, int jacobi, hipStream_t s);
size_t band_cholesky_lds_bytes(const SolveArgs& a);
size_t reduced_solve_lds_bytes(const SolveArgs& a);
size_t frame_lds_doubles(int Ps, int P1e, int n1);
size_t band_backsolve_lds_bytes(const SolveArgs& a);
hipError_t configure_solve_kernels(size_t band_lds, size_t reduced_lds, size_t back_lds);
void launch_solve(const SolveArgs& a, const LmOptionsDev& o, const double* x, double* x_cand, const BlockDev* blocks,
                  int n_blocks, bool dense_in_lds, hipStream_t s, bool with_post_eval, IterLog* log, int log_cap, int jacobi);
void launch_cost_reduce(const double* item_cost, int n_items, double* R2, const LmState* st, hipStream_t s);
void launch_control(LmState* st, const LmOptionsDev& o, double* R2, double* x, const double* x_cand, int n_amb,
                    IterLog* log, int log_cap, const double* item_cost, int n_items, const double* Rbase, size_t r_stride,
                    hipStream_t s, bool commit_by_copy = false, int* progress = nullptr, int seq = 0);
void launch_init_state(LmState* st, double radius, double x_norm, hipStream_t s, const double* upd_ext = nullptr, int upd_ext_n = 0);
void launch_begin_solve(LmState* st, double radius, double x_norm, const double* upd_ext, int upd_ext_n, const ResultSink& sink, double* x,
                        double* x_cand, const double* h_x, int n_amb, hipStream_t s);
void launch_publish_results(const LmState* st, const IterLog* log, int log_rows, const double* x, int n_amb, LmState* h_state,
                            IterLog* h_log, double* h_x, hipStream_t s);
void launch_seed_x(double* x, const double* h_x, int n_amb, hipStream_t s);
void launch_debug_control_replay(LmState* st, const LmOptionsDev& o, const double* rho, const int* infinite, int n, double* R2,
                                 double* radius_out, int* accepted_out, double* cost_out, IterLog* log, int log_cap, hipStream_t s);
size_t bcr_level_lds_bytes();
size_t bcr_back_lds_bytes(int q_max, int m1p);
hipError_t configure_bcr_kernels(int q_max, int m1p);
void roll_table_row(int k, int lane, unsigned* out);      // (host only: test hook)
hipError_t configure_dense_block_solve();
hipError_t configure_reduced_block_step();
hipError_t configure_reduced_fused();
size_t dense_block_solve_lds_bytes();
void launch_bcr_level(const SolveArgs& a, const BcrArgs& b, int node0, int n_nodes, int level, int keep0, int n_keep, const LmOptionsDev& o,
                      const double* x, const BlockDev* blocks, int n_blocks, int with_post_eval, IterLog* log, int log_cap, int jacobi,
                      hipStream_t s, int schur_ks, int* fan_word, const BcrInlineNodes& inl, int q_max);
bool schur_rides_in_last_level(int n_levels, int n_last_nodes, int root);
void launch_bcr_schur(const SolveArgs& a, const BcrArgs& b, int ks, const LmOptionsDev& o, hipStream_t s);
void launch_bcr_back(const SolveArgs& a, const BcrArgs& b, int node0, int n_nodes, bool top, bool extras, bool border_rows, int q_max,
                     const double* x, double* x_cand, const BlockDev* blocks, int n_blocks, const BcrTopSeps& ts, hipStream_t s);

void launch_reduced_solve(const SolveArgs& a, bool reduced_in_lds, int ks, hipStream_t s, int* fan_words);
bool dense_back_fusable(const SolveArgs& a, int ks, int q_max, bool border_rows);
size_t dense_back_lds_bytes(int q_max, int m1p);
hipError_t configure_dense_back_bytes(size_t lds);
void launch_dense_back(const SolveArgs& a, const BcrArgs& b, int ks, int node0, int n_nodes, int q_max, const double* x, double* x_cand,
                       const BlockDev* blocks, int n_blocks, const BcrTopSeps& ts, int* word, int seq, hipStream_t s);
int reduced_schur_slices(const SolveArgs& a);

}  // namespace cal

using namespace cal;

namespace {

constexpr size_t kMaxLds = 160 * 1024;
constexpr int kLogCap = 4096;
constexpr int kNumPhases = 7;   // 5 = calibration: the same event bracket around a trivial kernel; 6 = the reduced-system launch inside phase 2

// RCCL is loaded when the first communicator is asked for (calico_comm_get_unique_id / calico_comm_init_rccl), not at
// link time: a single-GPU user needs no librccl on the machine. An already loaded librccl (e.g. the one torch ships) is
// found by its soname; otherwise $ROCM_PATH/lib, then the loader's search path.
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
  bool ok() const { return lib != nullptr; }
};
RcclApi& rccl() {
  static RcclApi api = [] {
    RcclApi a;
    std::vector<std::string> names;
    if (const char* one = std::getenv("CALICO_RCCL_LIB")) names.push_back(one);      // this library and no other (a particular RCCL build)
    else {
      names = {"librccl.so.1", "librccl.so"};
      for (const char* env : {"ROCM_PATH", "ROCM_HOME"})
        if (const char* r = std::getenv(env)) { names.push_back(std::string(r) + "/lib/librccl.so.1"); names.push_back(std::string(r) + "/lib/librccl.so"); }
      names.push_back("/opt/rocm/lib/librccl.so.1"); names.push_back("/opt/rocm/lib/librccl.so");
    }
    std::string why;
    // An RCCL the process already holds (PyTorch brings its own librccl.so) is the one to use: two copies in one process
    // end in a double free at exit. And a copy this library loads stays private to it (RTLD_LOCAL: the entry points are
    // taken with dlsym) -- loaded globally before PyTorch, its symbols would interpose on the ones PyTorch's own copy
    // expects to bind.
    for (const std::string& n : names) {
      a.lib = dlopen(n.c_str(), RTLD_NOW | RTLD_NOLOAD);
      if (a.lib) break;
    }
    for (const std::string& n : names) {
      if (a.lib) break;
      a.lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (a.lib) break;
      const char* e = dlerror();         // (dlerror() clears its state: one call per failure)
      if (e) why = e;
    }
    if (!a.lib) { a.error = "librccl not found: " + why; return a; }
    auto sym = [&](const char* n) { void* f = dlsym(a.lib, n); if (!f && a.error.empty()) a.error = std::string("librccl lacks ") + n; return f; };
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
    a.CommCount = reinterpret_cast<decltype(a.CommCount)>(sym("ncclCommCount"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
    if (!a.error.empty()) { dlclose(a.lib); a.lib = nullptr; }
    return a;
  }();
  return api;
}

struct HBlock {
  std::vector<double> v;
  int size = 0, manifold = 0;
  bool constant = false, used = false;
  int amb_off = 0;
  int tan = -1;      // solver tangent index (6·cp for control points, 6·n_cp + c for calibration blocks)
  int eff = -1;      // tangent index in the reduced-problem order exported by calico_evaluate
  int tangent_size() const { return manifold == CALICO_MANIFOLD_EIGEN_QUATERNION ? 3 : size; }
};
struct HBody { int q, t; };
struct HSensor {
  int kind, model, K;
  int intr, q, t, lat, grav;
  double sigma, info;
  int loss; double loss_scale;
  std::vector<double> meas, stamps;
  std::vector<int> body, point, seg;
  std::vector<int64_t> sorted_pos;  // original observation -> position in the sorted device arrays
  std::vector<uint8_t> active;      // 0 = tagged as outlier (outlier_ids_, camera.h:185): left out of the problem
  int64_t n_active = -1;            // cached count of the blocks that are in the problem (-1: recount)
  int64_t sorted_begin = 0, sorted_end = 0;   // this sensor's contiguous range in the sorted arrays
  int dim() const { return kind == CALICO_SENSOR_CAMERA ? 2 : 3; }
  int64_t n() const { return int64_t(stamps.size()); }
};

// Device memory of plans and workspaces comes out of a few large slabs instead of one hipMalloc per buffer: a handle
// has some forty buffers, most of them a few KB, and a buffer of its own sits on pages of its own -- every kernel's
// first touch of each (state, descriptors, index lists, ...) then costs an address translation of its own behind the
// kernel boundary. One slab is one allocation of 64 MB: contiguous, mapped with the largest fragments the driver
// has. First fit over a free list ordered by address, neighbours merged on release; a request no slab can serve opens
// a new slab, and if that fails the request goes to hipMalloc as before. A request larger than a slab is an allocation
// of its own (hipMalloc / hipFree: it has large fragments anyway and must not pin memory for good).
// Slabs go back to the driver: trim() frees every slab that is one free extent -- calico_plan_cache_clear() frees all of
// them, calico_problem_destroy() all but CALICO_ARENA_KEEP_SLABS idle ones per device (default 4) -- so a process that once solved a large
// problem, or that shares the GPU with PyTorch / RCCL, does not keep that memory. (No HIP calls during static
// destruction: what is still held at exit is the driver's to reclaim.) CALICO_ARENA=0: hipMalloc per buffer (rounds 1-4).
class DeviceArena {
 public:
  static DeviceArena& get() { static DeviceArena* a = new DeviceArena; return *a; }
  hipError_t alloc(void** out, size_t bytes) {
    if (!enabled_ || bytes > kSlab) return hipMalloc(out, bytes);
    bytes = (bytes + kAlign - 1) / kAlign * kAlign;
    std::lock_guard<std::mutex> g(mu_);
    int dev = 0; (void)hipGetDevice(&dev);
    for (int pass = 0; pass < 2; ++pass) {
      for (Slab& sl : slabs_) {
        if (sl.device != dev) continue;
        for (auto it = sl.free.begin(); it != sl.free.end(); ++it) {
          if (it->second < bytes) continue;
          const size_t off = it->first, len = it->second;
          sl.free.erase(it);
          if (len > bytes) sl.free.emplace(off + bytes, len - bytes);
          *out = sl.base + off;
          used_[*out] = bytes;
          return hipSuccess;
        }
      }
      if (pass == 1) break;
      void* base = nullptr;
      if (hipMalloc(&base, kSlab) != hipSuccess) { (void)hipGetLastError(); break; }
      Slab sl; sl.base = static_cast<char*>(base); sl.size = kSlab; sl.device = dev; sl.free.emplace(0, kSlab);
      slabs_.push_back(std::move(sl));
    }
    return hipMalloc(out, bytes);      // (not in used_: release() hands it to hipFree)
  }
  void release(void* p) {
    if (!p) return;
    int owner = -1;
    {
      std::lock_guard<std::mutex> g(mu_);
      if (used_.count(p))
        for (const Slab& sl : slabs_)
          if (static_cast<char*>(p) >= sl.base && static_cast<char*>(p) < sl.base + sl.size) { owner = sl.device; break; }
    }
    if (owner < 0) { (void)hipFree(p); return; }
    // hipFree waits for the device; a block that goes back to the free list must do the same (a solve returns while the
    // early-exit kernels of the iterations enqueued ahead are still on its stream) -- for the device that OWNS the slab,
    // and once per batch of releases (Batch below), not once per buffer
    if (batch_device() != owner) sync_device(owner);
    std::lock_guard<std::mutex> g(mu_);
    auto u = used_.find(p);
    if (u == used_.end()) return;
    const size_t bytes = u->second;
    used_.erase(u);
    for (Slab& sl : slabs_) {
      char* c = static_cast<char*>(p);
      if (c < sl.base || c >= sl.base + sl.size) continue;
      size_t off = size_t(c - sl.base), len = bytes;
      auto next = sl.free.lower_bound(off);
      if (next != sl.free.end() && next->first == off + len) { len += next->second; next = sl.free.erase(next); }
      if (next != sl.free.begin()) {
        auto prev = std::prev(next);
        if (prev->first + prev->second == off) { off = prev->first; len += prev->second; sl.free.erase(prev); }
      }
      sl.free.emplace(off, len);
      return;
    }
  }
  // Everything a handle or a plan gives back at once (some forty buffers): ONE wait for the owning device, up front.
  struct Batch {
    explicit Batch(int device) : prev_(batch_device()) { if (DeviceArena::get().enabled_) { sync_device(device); batch_device() = device; } }
    ~Batch() { batch_device() = prev_; }
    Batch(const Batch&) = delete;
    Batch& operator=(const Batch&) = delete;
   private:
    int prev_;
  };
  // Frees the slabs nothing lives in; `keep_per_device` of them stay per device for the next handle. Returns the bytes freed.
  size_t trim(int keep_per_device) {
    std::vector<Slab> drop;
    {
      std::lock_guard<std::mutex> g(mu_);
      std::map<int, int> kept;
      for (size_t i = 0; i < slabs_.size();) {
        Slab& sl = slabs_[i];
        const bool idle = sl.free.size() == 1 && sl.free.begin()->first == 0 && sl.free.begin()->second == sl.size;
        if (idle && kept[sl.device]++ >= keep_per_device) { drop.push_back(std::move(sl)); slabs_.erase(slabs_.begin() + long(i)); }
        else ++i;
      }
    }
    size_t bytes = 0;
    for (Slab& sl : drop) { (void)hipFree(sl.base); bytes += sl.size; }     // (hipFree waits for the device itself)
    return bytes;
  }
  size_t slab_bytes() { std::lock_guard<std::mutex> g(mu_); size_t b = 0; for (const Slab& sl : slabs_) b += sl.size; return b; }
 private:
  static constexpr size_t kAlign = 4096, kSlab = size_t(64) << 20;
  struct Slab { char* base = nullptr; size_t size = 0; int device = 0; std::map<size_t, size_t> free; };
  DeviceArena() { const char* e = std::getenv("CALICO_ARENA"); enabled_ = !e || std::atoi(e) != 0; }
  static int& batch_device() { static thread_local int d = -1; return d; }
  static void sync_device(int device) {
    int cur = device;
    (void)hipGetDevice(&cur);
    if (cur != device) (void)hipSetDevice(device);
    (void)hipDeviceSynchronize();
    if (cur != device) (void)hipSetDevice(cur);
  }
  std::mutex mu_;
  std::vector<Slab> slabs_;
  std::unordered_map<void*, size_t> used_;
  bool enabled_ = true;
};

template <class T> struct DevBuf {
  T* p = nullptr; size_t n = 0;
  bool owner = true;        // false: a view of a buffer the plan cache owns (structure shared between handles)
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() { if (p && owner) DeviceArena::get().release(p); p = nullptr; n = 0; owner = true; }
  void alias(const DevBuf& o) { release(); p = o.p; n = o.n; owner = false; }
  void take(DevBuf& o) { release(); p = o.p; n = o.n; owner = o.owner; o.p = nullptr; o.n = 0; o.owner = true; }
  void swap(DevBuf& o) { std::swap(p, o.p); std::swap(n, o.n); std::swap(owner, o.owner); }
  hipError_t alloc(size_t count) {
    if (count == 0) count = 1;
    if (count == n && p && owner) return hipSuccess;
    release();        // (a view is dropped, never written through)
    hipError_t e = DeviceArena::get().alloc(reinterpret_cast<void**>(&p), count * sizeof(T));
    if (e == hipSuccess) n = count;
    return e;
  }
  hipError_t upload(const std::vector<T>& h, hipStream_t s) {
    hipError_t e = alloc(h.size());
    if (e != hipSuccess || h.empty()) return e;
    return hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s);
  }
};

struct PhaseTimer {
  std::vector<hipEvent_t> pool;
  struct Rec { int phase; hipEvent_t a, b; };
  std::vector<Rec> pending;
  size_t next = 0;
  double ms[kNumPhases] = {0, 0, 0, 0, 0, 0};
  int64_t count[kNumPhases] = {0, 0, 0, 0, 0, 0};
  // the same restricted to "working" launches: kernels of iterations enqueued ahead return at once when the solve has
  // terminated (or the step was rejected), and such brackets (shorter than a quarter of the phase's longest) are left out
  double ms_working[kNumPhases] = {0, 0, 0, 0, 0, 0};
  int64_t count_working[kNumPhases] = {0, 0, 0, 0, 0, 0};
  std::vector<float> samples[kNumPhases];
  hipEvent_t get() {
    if (next == pool.size()) { hipEvent_t e; (void)hipEventCreate(&e); pool.push_back(e); }
    return pool[next++];
  }
  int mask = 0;              // no event brackets unless asked for (calico_set_phase_timing): each pair costs ~6 us of stream time
  int every = 1;            // bracket only every `every`-th launch of a phase (an event pair costs ~6 us of stream time)
  int64_t seen[kNumPhases] = {0, 0, 0, 0, 0, 0};
  bool open_rec = false;
  int nested = 0;           // phase 6 sits inside phase 2: a bracket inside an open bracket is not recorded
  void begin(int phase, hipStream_t s) {
    if (open_rec) { ++nested; return; }
    open_rec = (mask >> phase) & 1;
    if (open_rec && phase != 5 && every > 1) open_rec = (seen[phase]++ % every) == 0;
    if (!open_rec) return;
    Rec r; r.phase = phase; r.a = get(); r.b = nullptr; (void)hipEventRecord(r.a, s); pending.push_back(r);
  }
  void end(hipStream_t s) { if (nested) { --nested; return; } if (!open_rec) return; Rec& r = pending.back(); r.b = get(); (void)hipEventRecord(r.b, s); open_rec = false; }
  void resolve() {  // call after a stream sync
    for (const Rec& r : pending) {
      float t = 0;
      if (r.b && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) { ms[r.phase] += t; count[r.phase]++; samples[r.phase].push_back(t); }
    }
    pending.clear(); next = 0;
    for (int ph = 0; ph < kNumPhases; ++ph) {
      float mx = 0; for (float t : samples[ph]) mx = std::max(mx, t);
      ms_working[ph] = 0; count_working[ph] = 0;
      for (float t : samples[ph]) if (t >= 0.25f * mx) { ms_working[ph] += t; count_working[ph]++; }
    }
  }
  void reset() { for (int i = 0; i < kNumPhases; ++i) { ms[i] = 0; count[i] = 0; seen[i] = 0; ms_working[i] = 0; count_working[i] = 0; samples[i].clear(); } }
  ~PhaseTimer() { for (hipEvent_t e : pool) (void)hipEventDestroy(e); }
};

}  // namespace

// ---- what calico_problem_finalize derives from the STRUCTURE of a problem (not from any value): shared between handles
//      of identical structure through the plan cache -------------------------------------------------------------------
struct BcrLevel { int node0, n_nodes, keep0, n_keep, q_max; };
struct PlanHost {
  bool speculative = true;    // evaluate cost AND Jacobian at the candidate point in one pass (two reduce buffers)
  size_t r_size = 0;
  int sep_s = 0, sep_n = 0;   // separator control points of the nested-dissection split (sep_n = 0: none)
  // tree solver (bcr_kernels.hip): elimination plan, level after level
  bool use_bcr = false, bcr_all_active = false;
  int bcr_N = 0, bcr_m1p = 16, bcr_root = -1, bcr_root_pend = 0, bcr_root_par = 0, bcr_br = 0, bcr_q_max = 1, bcr_slots = 1, bcr_q0 = 1;
  std::vector<BcrLevel> bcr_levels;
  std::vector<BcrNodeDev> h_bcr_nodes;
  std::vector<int> h_bcr_keep, h_cp_block;
  bool bcr_merge_top = true;      // the top level's back-substitution rides in the launch below it
  int border_extra() const { return use_bcr ? bcr_br : 6 * sep_n; }   // rows the band hands to the dense reduced solve
  int n_cp = 0, m = 0, n_amb = 0, n_eff = 0, n_items = 0, n_items_all = 0, lds_cols = 0, row_pad = kRowPad;
  int64_t n_obs = 0;
  size_t partial_doubles = 0, partials_alloc = 0;
  std::vector<int> eff_to_tan;
  std::vector<BlockDev> h_blocks;
  std::vector<ItemDev> h_items, h_items_all, h_jac_items;
  std::vector<FrameItemDev> h_fitems;
  std::vector<CellDev> h_cells;
  int cell_chunk = 1, cell_rec_max = 1, row_cell_chunk = 1;
  int frame_lds_doubles = 0;
  int n_fitems = 0, n_jac_items = 0;
  bool fuse_expand = false;    // cell workgroups (EvalArgs.pair_mode): camera cells expanded inside the Jacobian launch, IMU items form their own blocks
  int pair_wave_lds_doubles = 0;
  int n_thin = 0, n_fat = 0;
  int n_thin8 = 0, n_thin4 = 0;  // thin outputs [0, n_thin8) take eight lanes, [n_thin8, n_thin4) four (<= 24 sources), [n_thin4, n_thin) one (<= 8)
  int thin_per_lane = 6;       // sources per lane of a thin output's eight lanes (6: up to 48 sources, 12: up to 96)
  bool gather_fixed = false;   // the thin lists at a fixed stride (d_idx_fixed) instead of CSR
  bool dense_in_lds = true;
  int gather_owner_block = 0;
  bool gs_lists_on_device = false;   // the band / border / spline right-hand side lists were built by the device (launch_gather_lists)
};
struct PlanDev {      // structure on the device: immutable once uploaded
  DevBuf<double> d_knots, d_basis, d_stamp;
  DevBuf<int> d_ctrl_off, d_point_off, d_out_thin, d_idx_thin, d_idx_fixed, d_out_fat, d_idx_fat, d_prim_tab, d_bkeep, d_cp_block, d_gs_tab;
  DevBuf<int64_t> d_ptr_thin, d_ptr_fat;
  DevBuf<uint8_t> d_cp_active;
  DevBuf<SensorDev> d_sensors;
  DevBuf<LayoutDev> d_layouts;
  DevBuf<ItemDev> d_items, d_items_all, d_jac_items;
  DevBuf<FrameItemDev> d_fitems;
  DevBuf<CellDev> d_cells;
  DevBuf<BlockDev> d_blocks;
  DevBuf<BcrNodeDev> d_bnodes;
#define PLAN_DEV_BUFS(X) X(d_knots) X(d_basis) X(d_stamp) X(d_ctrl_off) X(d_point_off) X(d_out_thin) X(d_idx_thin) X(d_idx_fixed) X(d_out_fat) X(d_idx_fat) \
  X(d_prim_tab) X(d_bkeep) X(d_cp_block) X(d_gs_tab) X(d_ptr_thin) X(d_ptr_fat) X(d_cp_active) X(d_sensors) X(d_layouts) X(d_items) X(d_items_all)     \
  X(d_jac_items) X(d_fitems) X(d_cells) X(d_blocks) X(d_bnodes)
  void take_from(PlanDev& o) {
#define X(n) n.take(o.n);
    PLAN_DEV_BUFS(X)
#undef X
  }
  void alias_from(const PlanDev& o) {
#define X(n) n.alias(o.n);
    PLAN_DEV_BUFS(X)
#undef X
  }
};
// ---- what a handle works in: values, normal equations, solver workspaces, result staging. Recycled between handles of
//      identical structure (a destroyed handle leaves its workspace with the cached plan) ----------------------------------
struct Workspace {
  DevBuf<unsigned long long> d_wave_log;   // CALICO_KERNEL_TIMING=3
  DevBuf<double> d_x, d_xc, d_m0, d_m1, d_m2, d_partials, d_R, d_R2, d_Lb, d_Linv, d_Y, d_S, d_Spart, d_Swork, d_zbuf, d_y, d_dadd, d_scale, d_res;
  DevBuf<double> d_bD, d_bG, d_bF, d_bpD, d_bpF, d_bM, d_bZA, d_bZB, d_bY, d_bysol, d_bzb, d_bupd;
  DevBuf<uint8_t> d_valid, d_active;
  DevBuf<int> d_counter;
  DevBuf<int> d_handoff;         // hand-off word of the fused dense-solve + back-substitution launch
  DevBuf<LmState> d_state;
  DevBuf<IterLog> d_log;
  int handoff_seq = 0;           // number of the last such launch (the word carries it when the solve part is through)
  LmState* h_state = nullptr;  // pinned
  int* h_progress = nullptr;   // pinned, device-visible: [epoch << 20 | iterations the control kernel is through with, epoch of the terminated solve]
  int solve_epoch = 0;         // number of the streaming solve under way (1 .. 2047, wraps)
  int* d_progress = nullptr;
  double* h_xpin = nullptr;    // pinned staging for the parameter vector (upload at the start of a call, download at its end)
  size_t h_xpin_n = 0;
  IterLog* h_log = nullptr;    // pinned
  bool ws_ready = false;       // allocated and initialised for the plan at hand
#define WS_BUFS(X) X(d_wave_log) X(d_x) X(d_xc) X(d_m0) X(d_m1) X(d_m2) X(d_partials) X(d_R) X(d_R2) X(d_Lb) X(d_Linv) X(d_Y) X(d_S) X(d_Spart)      \
  X(d_Swork) X(d_zbuf) X(d_y) X(d_dadd) X(d_scale) X(d_res) X(d_bD) X(d_bG) X(d_bF) X(d_bpD) X(d_bpF) X(d_bM) X(d_bZA) X(d_bZB) X(d_bY) X(d_bysol) \
  X(d_bzb) X(d_bupd) X(d_valid) X(d_active) X(d_counter) X(d_handoff) X(d_state) X(d_log)
  void swap_ws(Workspace& o) {
#define X(n) n.swap(o.n);
    WS_BUFS(X)
#undef X
    std::swap(handoff_seq, o.handoff_seq); std::swap(h_state, o.h_state); std::swap(h_progress, o.h_progress);
    std::swap(solve_epoch, o.solve_epoch); std::swap(d_progress, o.d_progress); std::swap(h_xpin, o.h_xpin);
    std::swap(h_xpin_n, o.h_xpin_n); std::swap(h_log, o.h_log); std::swap(ws_ready, o.ws_ready);
  }
  void free_pinned() {
    if (h_state) (void)hipHostFree(h_state);
    if (h_progress) (void)hipHostFree(h_progress);
    if (h_xpin) (void)hipHostFree(h_xpin);
    if (h_log) (void)hipHostFree(h_log);
    h_state = nullptr; h_progress = nullptr; d_progress = nullptr; h_xpin = nullptr; h_xpin_n = 0; h_log = nullptr;
  }
  Workspace() = default;
  Workspace(const Workspace&) = delete;
  Workspace& operator=(const Workspace&) = delete;
  ~Workspace() { free_pinned(); }
};

struct PlanEntry;      // plan cache entry (below)

struct calico_problem : PlanHost, PlanDev, Workspace {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string error;
  std::vector<HBlock> blocks;
  std::vector<HBody> bodies;
  std::vector<HSensor> sensors;
  int order = 0;
  std::vector<double> knots, valid_knots, basis;
  std::vector<int> ctrl;
  bool dirty = true;
  calico_allreduce_fn allreduce = nullptr;
  void* allreduce_ctx = nullptr;
  ncclComm_t comm = nullptr;      // native exchange: RCCL communicator owned by the handle (calico_comm_init_rccl)
  bool has_exchange() const { return allreduce != nullptr || comm != nullptr; }
  int rank = 0, world = 1;
  std::shared_ptr<PlanEntry> plan;    // the cached plan this handle's structure buffers are views of (null: it owns them)

  std::vector<double> h_x;     // staging of the parameter values (alive until the upload is through)
  double* h_mpin = nullptr;    // pinned staging of the measurements in device order [m0 | m1 | m2], borrowed from the process-wide pool for the duration of finalize
  size_t h_mpin_n = 0;
  bool active_dirty = true;
  bool any_tagged = false;       // some observation is tagged as an outlier: the kernels look at the tags only then
  bool xc_stale = true;       // the candidate buffer must be re-seeded with the constant blocks' values
  // residuals of ALL sensors at the parameter values `x` (calico_get_residuals / calico_project are per sensor, as
  // Sensor::UpdateResiduals is: the second to last sensor of a write-back are served from here)
  struct ResCache { bool valid = false, predict = false; std::vector<double> x, r; std::vector<uint8_t> v; } res_cache;
  std::vector<calico_iteration> iterations;
  PhaseTimer timer;

  int set_error(int code, const std::string& msg) { error = msg; return code; }
  int hip_error(hipError_t e, const char* what) {
    return set_error(CALICO_INTERNAL, std::string(what) + ": " + hipGetErrorString(e));
  }
};

#define HIP_TRY(p, expr)                                    \
  do {                                                      \
    hipError_t _e = (expr);                                 \
    if (_e != hipSuccess) return (p)->hip_error(_e, #expr); \
  } while (0)

namespace {

// bspline.hpp:138-150
int spline_index(const calico_problem* p, double t) {
  const std::vector<double>& vk = p->valid_knots;
  if (t == vk.back()) return int(vk.size()) - 2;
  if (!(t < vk.back())) return -1;
  // upper_bound(vk, t) - 1, found from a guess on the (uniform) knot spacing and corrected by comparisons with the knots
  // themselves, so the result is the binary search's for any knot vector
  const int n = int(vk.size());
  if (t < vk.front()) return -1;
  const double dt = (vk.back() - vk.front()) / double(n - 1);
  int i = dt > 0.0 ? int((t - vk.front()) / dt) : 0;
  i = std::max(0, std::min(n - 2, i));
  while (i > 0 && t < vk[size_t(i)]) --i;
  while (i + 1 < n && !(t < vk[size_t(i) + 1])) ++i;
  return i;
}
int camera_num_params(int model) {
  switch (model) { case 1: return 8; case 2: return 11; case 3: return 7; case 4: return 5; case 5: return 4; case 6: return 4;
    case 7: return 5; default: return -1; }
}
int imu_num_params(int model) { return model == 1 ? 1 : (model == 2 ? 4 : (model == 3 ? 12 : -1)); }

SolveArgs make_solve_args(calico_problem* p) {
  SolveArgs a;
  a.R = p->d_R.p; a.r_stride = p->speculative ? p->r_size : 0; a.Lb = p->d_Lb.p; a.Linv = p->d_Linv.p; a.Y = p->d_Y.p; a.S = p->d_S.p; a.Spart = p->d_Spart.p;
  a.Swork = p->d_Swork.p; a.y = p->d_y.p; a.zbuf = p->d_zbuf.p; a.dadd = p->d_dadd.p;
  a.scale = p->d_scale.p; a.cp_active = p->d_cp_active.p; a.st = p->d_state.p; a.n_cp = p->n_cp; a.k = p->order; a.mc = p->m; a.sep_s = p->sep_s; a.sep_n = p->sep_n; a.m = p->m + p->border_extra();
  static const int dbg = [] {
    const int v = std::getenv("CALICO_KERNEL_TIMING") ? std::atoi(std::getenv("CALICO_KERNEL_TIMING")) : 0;
#ifndef CALICO_DEV_TIMING
    if (v) std::fprintf(stderr, "[calico] CALICO_KERNEL_TIMING is set, but this library was built without the kernels' development "
                                "instrumentation (rebuild with CALICO_DEV_TIMING=1 in the environment of __graft_entry__.build())\n");
#endif
    return v;
  }();
  a.debug = dbg;
  a.progress = nullptr;
  return a;
}

BcrArgs make_bcr_args(calico_problem* p) {
  BcrArgs b;
  b.D = p->d_bD.p; b.G = p->d_bG.p; b.F = p->d_bF.p; b.pendD = p->d_bpD.p; b.pendF = p->d_bpF.p; b.M = p->d_bM.p; b.ZA = p->d_bZA.p;
  b.ZB = p->d_bZB.p; b.Y = p->d_bY.p; b.ysol = p->d_bysol.p; b.zb = p->d_bzb.p; b.upd = p->d_bupd.p; b.nodes = p->d_bnodes.p; b.keep = p->d_bkeep.p;
  b.cp_block = p->d_cp_block.p; b.ctrl_off = p->d_ctrl_off.p; b.all_active = p->bcr_all_active ? 1 : 0; b.pad0 = 0; b.N = p->bcr_N; b.m1p = p->bcr_m1p; b.root = p->bcr_root; b.root_pend = p->bcr_root_pend;
  b.root_par = p->bcr_root_par; b.n_slots = p->bcr_slots;
  return b;
}

// Elimination plan of the tree solver: level 0 eliminates chains of q consecutive superblocks between kept
// separators, every further level every other survivor; the last survivor is the root (joins the dense solve).
// q minimises (levels · launch + chain steps · factorisation) for the trajectory length at hand.
void build_bcr_plan(calico_problem* p) {
  const int N = (p->n_cp + kBcrCps - 1) / kBcrCps;
  p->bcr_N = N;
  auto levels_after = [](int n_sep) { int l = 0; while (n_sep > 1) { n_sep /= 2; ++l; } return l; };
  int q = 1;
  {
    double best = 1e300;
    for (int c = 1; c <= kBcrMaxChain; ++c) {
      const int n_sep = N > c ? N / (c + 1) : 0;
      const int L = 1 + levels_after(n_sep);
      double cost = 6.0 * L + 4.0 * (c + L - 1);
      // One workgroup of a level launch fills a CU and the part has 256: a level 0 whose (node, role) workgroups (laid out by XCD:
      // nodes padded to a multiple of eight), eight separators' workgroups and the bookkeeping one do not fit runs its tail in a
      // second dispatch round (1453 control points, chains of four: 256 + 64 + 1 workgroups, level 0 34.7 us; chains of five: +2.9 % it/s)
      const int per = 1 + p->bcr_m1p / 16, nodes = n_sep + 1;
      if (8 * ((nodes + 7) / 8) * per + 9 > 256) cost += 5.0;
      if (cost < best) { best = cost; q = c; }
    }
    if (const char* e = std::getenv("CALICO_BCR_LEAF")) q = std::max(1, std::min(kBcrMaxChain, std::atoi(e)));
  }
  p->bcr_levels.clear(); p->h_bcr_nodes.clear(); p->h_bcr_keep.clear();
  std::vector<int> alive(static_cast<size_t>(N), 0), mask(static_cast<size_t>(N), 0);
  for (int i = 0; i < N; ++i) alive[size_t(i)] = i;
  int level = 0, q_max_all = 1;
  while (!alive.empty() && (level == 0 || alive.size() > 1)) {
    const int chain = level == 0 ? q : 1;
    BcrLevel L;
    L.node0 = int(p->h_bcr_nodes.size()); L.keep0 = int(p->h_bcr_keep.size() / 2); L.q_max = 1;
    std::vector<int> kept, new_mask(size_t(N), 0);
    const size_t n = alive.size();
    size_t pos = 0;
    // level 0 with N <= q: one chain, no separator. Otherwise: [chain of `chain`] [keep] [chain] [keep] ...
    while (pos < n) {
      BcrNodeDev nd = {};
      nd.left = kept.empty() ? -1 : kept.back();
      nd.q = 0;
      nd.blk0 = alive[pos]; nd.pend = mask[size_t(alive[pos])];     // chains longer than one block only exist at level 0 (consecutive, no pending)
      while (pos < n && nd.q < chain) { ++nd.q; ++pos; }
      nd.right = pos < n ? alive[pos] : -1;
      nd.slot = int(p->h_bcr_nodes.size());
      L.q_max = std::max(L.q_max, nd.q);
      if (nd.left >= 0) new_mask[size_t(nd.left)] |= 2;
      if (nd.right >= 0) new_mask[size_t(nd.right)] |= 1;
      p->h_bcr_nodes.push_back(nd);
      if (pos < n) { kept.push_back(alive[pos]); ++pos; }
    }
    // separators that survive this level: level 0 initialises them from R(x), later levels add last level's pending updates
    for (int kb : kept)
      if (level == 0 || mask[size_t(kb)]) { p->h_bcr_keep.push_back(kb); p->h_bcr_keep.push_back(mask[size_t(kb)]); }
    L.n_nodes = int(p->h_bcr_nodes.size()) - L.node0;
    L.n_keep = int(p->h_bcr_keep.size() / 2) - L.keep0;
    q_max_all = std::max(q_max_all, L.q_max);
    p->bcr_levels.push_back(L);
    alive = kept; mask = new_mask;
    ++level;
  }
  p->bcr_root = alive.empty() ? -1 : alive[0];
  p->bcr_root_pend = alive.empty() ? 0 : mask[size_t(alive[0])];
  p->bcr_root_par = (level - 1) & 1;
  p->bcr_br = alive.empty() ? 0 : 6 * kBcrCps;
  p->bcr_q_max = q_max_all;
  p->bcr_q0 = q;       // level 0's chain length: its node table is arithmetic on the node's number (BcrInlineNodes)
  p->bcr_slots = int(p->h_bcr_nodes.size()) + 1;
  { const char* e = std::getenv("CALICO_BCR_MERGE_TOP"); p->bcr_merge_top = !e || std::atoi(e) != 0; }   // (A/B switch, see enqueue_linear_solve)
}

EvalArgs make_eval_args(calico_problem* p, const double* x, int apply_loss, bool want_res) {
  EvalArgs a;
  static const int dbg = std::getenv("CALICO_KERNEL_TIMING") ? std::atoi(std::getenv("CALICO_KERNEL_TIMING")) : 0;
  a.debug = dbg;
  a.x = x; a.sensors = p->d_sensors.p; a.layouts = p->d_layouts.p; a.items = p->d_items.p;
  a.knots = p->d_knots.p; a.basis = p->d_basis.p; a.ctrl_off = p->d_ctrl_off.p;
  a.m0 = p->d_m0.p; a.m1 = p->d_m1.p; a.m2 = p->d_m2.p; a.stamp = p->d_stamp.p; a.point_off = p->d_point_off.p;
  a.partials = p->d_partials.p; a.item_cost = p->d_partials.p + p->partial_doubles;
  a.res_out = want_res ? p->d_res.p : nullptr; a.valid_out = want_res ? p->d_valid.p : nullptr;
  a.order = p->order; a.n_items = p->n_items; a.lds_cols = p->lds_cols; a.row_pad = p->row_pad; a.n_cells = int(p->h_cells.size()); a.cells = p->d_cells.p; a.prim_tab = p->d_prim_tab.p;
  a.cell_chunk = p->cell_chunk; a.cell_rec_max = p->cell_rec_max; a.project = 0; a.row_cell_chunk = p->row_cell_chunk; a.frame_lds_doubles = p->frame_lds_doubles; a.pad5 = 0; a.wave_log = p->d_wave_log.p; a.active = p->any_tagged ? p->d_active.p : nullptr; a.apply_loss = apply_loss;
  a.st = nullptr; a.need_flag = 0; a.cost_index_base = 0;
  a.fitems = p->d_fitems.p; a.n_fitems = p->n_fitems;
  a.hint_progress = nullptr; a.hint_seq = 0; a.hint_first = 0; a.hint_ftol = a.hint_ptol = 0.0;
  a.pair_mode = 0; a.wave_lds_doubles = 0;
  return a;
}

// Flatten the host tables into cells / work items / gather lists, plan the elimination, upload the STRUCTURE (everything
// here depends on what the problem looks like, nothing on a value: the result is what the plan cache shares).
int build_plan(calico_problem* p) {
  const int k = p->order;
  const int n_cp = int(p->ctrl.size());
  p->n_cp = n_cp;
  // CALICO_SETUP_TIMING=1: wall time of the sections of this function (development aid)
  static const bool setup_timing = std::getenv("CALICO_SETUP_TIMING") != nullptr;
  auto t_sec = std::chrono::steady_clock::now();
  auto section = [&](const char* name) {
    if (!setup_timing) return;
    const auto t = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[calico] finalize %-28s %8.3f ms\n", name, std::chrono::duration<double, std::milli>(t - t_sec).count());
    t_sec = t;
  };
  // ---- ambient offsets, used flags ----
  int off = 0;
  for (HBlock& b : p->blocks) { b.amb_off = off; off += b.size; b.used = false; b.tan = -1; b.eff = -1; }
  p->n_amb = off;
  std::vector<char> is_ctrl(p->blocks.size(), 0);
  for (int id : p->ctrl) is_ctrl[id] = 1;
  std::vector<uint8_t> cp_active(n_cp, 0);
  for (HSensor& s : p->sensors) {
    if (s.n() == 0) continue;
    p->blocks[s.intr].used = p->blocks[s.q].used = p->blocks[s.t].used = p->blocks[s.lat].used = true;
    if (s.kind == CALICO_SENSOR_ACCELEROMETER) p->blocks[s.grav].used = true;
    for (int64_t i = 0; i < s.n(); ++i) {
      for (int j = 0; j < k; ++j) cp_active[s.seg[i] + j] = 1;
      if (s.kind == CALICO_SENSOR_CAMERA) {
        p->blocks[s.point[i]].used = true;
        p->blocks[p->bodies[s.body[i]].q].used = p->blocks[p->bodies[s.body[i]].t].used = true;
      }
    }
  }
  for (int i = 0; i < n_cp; ++i) {
    HBlock& b = p->blocks[p->ctrl[i]];
    b.used = cp_active[i] != 0;
    if (b.constant && b.used) return p->set_error(CALICO_UNIMPLEMENTED, "constant control points are not supported");
    b.tan = 6 * i;
  }
  // ---- tangent order ----
  p->h_blocks.clear(); p->eff_to_tan.clear();
  int eff = 0;
  for (int i = 0; i < n_cp; ++i) {
    if (!cp_active[i]) continue;
    HBlock& b = p->blocks[p->ctrl[i]];
    b.eff = eff; eff += 6;
    for (int c = 0; c < 6; ++c) p->eff_to_tan.push_back(6 * i + c);
    p->h_blocks.push_back({b.amb_off, 6, 0, 6 * i});
  }
  int m = 0;
  for (size_t id = 0; id < p->blocks.size(); ++id) {
    HBlock& b = p->blocks[id];
    if (is_ctrl[id] || b.constant || !b.used) continue;
    b.tan = 6 * n_cp + m; b.eff = eff;
    for (int c = 0; c < b.tangent_size(); ++c) p->eff_to_tan.push_back(b.tan + c);
    p->h_blocks.push_back({b.amb_off, b.size, b.manifold, b.tan});
    m += b.tangent_size(); eff += b.tangent_size();
  }
  p->m = m; p->n_eff = eff;
  // Nested dissection with one separator (k-1 control points in the middle of the trajectory): the two halves of
  // the band are then factored and back-substituted side by side, the separator joins the dense border. Used when
  // the enlarged border still fits the in-LDS reduced solve and every control point is observed.
  p->sep_s = 0; p->sep_n = 0;
  {
    bool all_active = true;
    for (int i = 0; i < n_cp; ++i) all_active = all_active && cp_active[size_t(i)] != 0;
    const char* env = std::getenv("CALICO_BAND_SPLIT");
    const bool allowed = !env || std::atoi(env) != 0;
    if (allowed && all_active && n_cp >= 6 * k && m + 6 * (k - 1) + 1 <= 1024) {
      p->sep_n = k - 1;
      p->sep_s = (n_cp - p->sep_n) / 2;
    }
  }
  // Tree solver for spline orders up to 6 (superblocks of five control points are then block tridiagonal); it takes
  // over the split of the band, so the single-separator variant above is switched off. CALICO_SOLVER=band keeps the
  // sequential banded factorisation (A/B switch, and the path of higher spline orders).
  {
    const char* env = std::getenv("CALICO_SOLVER");
    p->use_bcr = k <= 6 && !(env && std::string(env) == "band");
    if (p->use_bcr) {
      p->sep_s = 0; p->sep_n = 0;
      p->bcr_all_active = true;
      for (int i = 0; i < n_cp; ++i) p->bcr_all_active = p->bcr_all_active && cp_active[size_t(i)] != 0;
      p->bcr_m1p = 16 * ((m + 1 + 15) / 16);
      build_bcr_plan(p);
    }
  }
  const int NS = 6 * n_cp;
  section("blocks / tangent order");
  // ---- layouts ----
  std::vector<SensorDev> sd(p->sensors.size());
  std::vector<LayoutDev> layouts;
  std::vector<std::vector<int>> layout_gmap;             // local calibration column -> solver tangent index
  // (sensor, body, free model point or -1) -> layout id. A free model point is one more calibration block of the
  // residual blocks that observe it, so those blocks get a layout (and cells) of their own per point.
  std::map<std::array<int, 3>, int> layout_of;
  auto layout_key = [&](size_t si, const HSensor& s, int64_t i) -> std::array<int, 3> {
    if (s.kind != CALICO_SENSOR_CAMERA) return {int(si), -1, -1};
    return {int(si), s.body[i], p->blocks[s.point[i]].constant ? -1 : s.point[i]};
  };
  auto is_free = [&](int id) { return id >= 0 && !p->blocks[id].constant; };
  for (size_t si = 0; si < p->sensors.size(); ++si) {
    const HSensor& s = p->sensors[si];
    SensorDev& d = sd[si];
    d.kind = s.kind; d.model = s.model; d.K = s.K; d.loss = s.loss;
    d.intr_off = p->blocks[s.intr].amb_off; d.q_off = p->blocks[s.q].amb_off; d.t_off = p->blocks[s.t].amb_off;
    d.lat_off = p->blocks[s.lat].amb_off; d.grav_off = s.grav >= 0 ? p->blocks[s.grav].amb_off : 0; d.pad0 = 0;
    d.info = s.info; d.loss_scale = s.loss_scale;
    std::array<int, 3> seen = {-2, -2, -2};       // (consecutive observations mostly share their layout: one compare instead of a map look-up)
    for (int64_t i = 0; i < s.n(); ++i) {
      const int body = s.kind == CALICO_SENSOR_CAMERA ? s.body[i] : -1;
      const std::array<int, 3> lkey = layout_key(si, s, i);
      if (lkey == seen) continue;
      seen = lkey;
      if (layout_of.count(lkey)) continue;
      LayoutDev L;
      L.sensor = int(si); L.c_pt = -1;
      std::vector<int> gmap;
      int c = 6 * k;
      auto add = [&](int id, int* slot) {
        if (is_free(id)) { *slot = c; for (int q = 0; q < p->blocks[id].tangent_size(); ++q) gmap.push_back(p->blocks[id].tan + q); c += p->blocks[id].tangent_size(); }
        else *slot = -1;
      };
      add(s.intr, &L.c_intr); add(s.q, &L.c_q);
      if (s.kind == CALICO_SENSOR_GYROSCOPE) L.c_t = -1; else add(s.t, &L.c_t);
      add(s.lat, &L.c_lat);
      L.c_bq = L.c_bt = L.c_grav = -1; L.bq_off = L.bt_off = 0;
      if (s.kind == CALICO_SENSOR_CAMERA) {
        add(p->bodies[body].q, &L.c_bq); add(p->bodies[body].t, &L.c_bt);
        L.bq_off = p->blocks[p->bodies[body].q].amb_off; L.bt_off = p->blocks[p->bodies[body].t].amb_off;
        if (lkey[2] >= 0) add(lkey[2], &L.c_pt);
      } else if (s.kind == CALICO_SENSOR_ACCELEROMETER) {
        add(s.grav, &L.c_grav);
      }
      L.ncols = c;
      layout_of[lkey] = int(layouts.size());
      layouts.push_back(L); layout_gmap.push_back(gmap);
    }
  }
  section("layouts");
  // ---- sort observations by (layout, segment) and cut work items ----
  struct Key { int layout, seg, sensor; int64_t idx; double stamp; };
  std::vector<Key> keys;
  int64_t n_obs = 0;
  for (const HSensor& s : p->sensors) n_obs += s.n();
  keys.reserve(size_t(n_obs));
  for (size_t si = 0; si < p->sensors.size(); ++si) {
    HSensor& s = p->sensors[si];
    s.sorted_pos.assign(size_t(s.n()), 0);
    std::array<int, 3> seen = {-2, -2, -2};
    int seen_layout = -1;
    for (int64_t i = 0; i < s.n(); ++i) {
      const std::array<int, 3> lkey = layout_key(si, s, i);
      if (!(lkey == seen)) { seen = lkey; seen_layout = layout_of[lkey]; }
      keys.push_back({seen_layout, s.seg[i], int(si), i, s.stamps[size_t(i)]});
    }
  }
  {
    // order: (layout, segment, stamp), ties in insertion order. A stable counting sort over the cells (layout, segment)
    // does almost all of it -- measurements arrive in time order, sensor by sensor --; a cell whose stamps are not in
    // order gets a stable comparison sort of its own. (One comparison sort over all keys was 1.5 ms of the set-up.)
    const int nseg_all = std::max(1, int(p->valid_knots.size()) - 1);
    const size_t n_cell_ids = layouts.size() * size_t(nseg_all);
    std::vector<int64_t> cstart(n_cell_ids + 1, 0);
    auto cell_of = [&](const Key& kq) { return size_t(kq.layout) * size_t(nseg_all) + size_t(std::max(0, std::min(nseg_all - 1, kq.seg))); };
    for (const Key& kq : keys) ++cstart[cell_of(kq) + 1];
    for (size_t c = 0; c < n_cell_ids; ++c) cstart[c + 1] += cstart[c];
    std::vector<Key> sorted(keys.size());
    {
      std::vector<int64_t> fill(cstart.begin(), cstart.end() - 1);
      for (const Key& kq : keys) sorted[size_t(fill[cell_of(kq)]++)] = kq;
    }
    for (size_t c = 0; c < n_cell_ids; ++c) {
      const int64_t q0 = cstart[c], q1 = cstart[c + 1];
      bool ordered = true;
      for (int64_t q = q0 + 1; q < q1 && ordered; ++q) ordered = !(sorted[size_t(q)].stamp < sorted[size_t(q - 1)].stamp);
      if (!ordered)
        std::stable_sort(sorted.begin() + q0, sorted.begin() + q1, [](const Key& a, const Key& b) { return a.stamp < b.stamp; });
    }
    keys.swap(sorted);
  }
  p->n_obs = n_obs;
  std::vector<double> st(n_obs);
  std::vector<int> point_off(n_obs, 0);
  p->h_items.clear(); p->h_items_all.clear();
  const int imu_chunk_items = [] { const char* e = std::getenv("CALICO_IMU_CHUNK"); return e ? std::max(1, std::min(21, std::atoi(e))) : 21; }();   // (the Jacobian kernel gives an IMU block three lanes)
  int max_cols = 0;
  for (int64_t q = 0; q < n_obs;) {
    int64_t e = q;
    while (e < n_obs && keys[e].layout == keys[q].layout && keys[e].seg == keys[q].seg) ++e;
    const LayoutDev& L = layouts[keys[q].layout];
    const int dim = p->sensors[L.sensor].dim();
    // cameras fill the 128 staged rows; an IMU block is a long single-lane computation and there are few of them, so
    // they are cut finer: more waves in flight, shorter JᵀJ stage, smaller LDS footprint next to the camera frames
    const int chunk = dim == 2 ? kRowsPerItem / 2 : imu_chunk_items;
    max_cols = std::max(max_cols, L.ncols + 1);
    for (int64_t b = q; b < e; b += chunk) {
      ItemDev it;
      it.layout = keys[q].layout; it.seg = keys[q].seg; it.obs_begin = int(b); it.obs_count = int(std::min<int64_t>(chunk, e - b));
      it.partial_off = 0; it.rows_off = -1;
      p->h_items_all.push_back(it);
    }
    q = e;
  }
  // this rank's shard: a contiguous window of spline segments (shard.hpp)
  const int nseg = int(p->valid_knots.size()) - 1;
  std::vector<int64_t> per_seg(size_t(nseg), 0);
  for (const ItemDev& it : p->h_items_all) per_seg[size_t(it.seg)] += it.obs_count;
  const std::vector<int> win = shard_windows(per_seg, p->world);
  const int seg_lo = win[size_t(p->rank)], seg_hi = win[size_t(p->rank) + 1];
  for (const ItemDev& it : p->h_items_all)
    if (it.seg >= seg_lo && it.seg < seg_hi) p->h_items.push_back(it);
  // Jacobian pass: camera cells are cut into FRAMES (blocks sharing the stamp) for the frame kernel
  // when the spline order is 6 and frames are reasonably full; everything else goes to the generic kernel.
  p->h_fitems.clear(); p->h_jac_items.clear(); p->h_cells.clear();
  size_t poff = 0, comp_off = 0;
  // compact record of a camera frame: M_ext (PE×PE) + expansion coefficients (ncols + 1); see eval_kernels.hip
  auto frame_rec = [&](const LayoutDev& L) -> size_t {
    const HSensor& hs = p->sensors[size_t(L.sensor)];
    const int P1 = 7 + (L.c_intr >= 0 ? hs.K : 0) + 3 * (L.c_q >= 0) + 3 * (L.c_t >= 0) + 3 * (L.c_bq >= 0) + 3 * (L.c_bt >= 0);
    const int PE = P1 + 1;     // prim columns + the latency row / column
    return size_t(PE) * PE + size_t(L.ncols + 1);
  };
  {
    std::vector<char> layout_uses_frames(layouts.size(), 0);
    if (k == 6) {
      std::vector<int64_t> n_obs_l(layouts.size(), 0), n_frames_l(layouts.size(), 0);
      for (int64_t q = 0; q < n_obs;) {
        int64_t e = q;
        while (e < n_obs && keys[e].layout == keys[q].layout && keys[e].seg == keys[q].seg && keys[e].stamp == keys[q].stamp) ++e;
        n_obs_l[size_t(keys[q].layout)] += e - q; n_frames_l[size_t(keys[q].layout)] += 1;
        q = e;
      }
      for (size_t l = 0; l < layouts.size(); ++l)
        layout_uses_frames[l] = p->sensors[size_t(layouts[l].sensor)].kind == CALICO_SENSOR_CAMERA && n_frames_l[l] > 0 &&
                                n_obs_l[l] >= 16 * n_frames_l[l] && layouts[l].ncols + 1 - 36 + 6 <= 30 && layouts[l].c_pt < 0;
    }
    for (int64_t q = 0; q < n_obs;) {
      const Key& kq = keys[q];
      int64_t e = q;
      if (layout_uses_frames[size_t(kq.layout)]) {
        while (e < n_obs && keys[e].layout == kq.layout && keys[e].seg == kq.seg && keys[e].stamp == kq.stamp) ++e;
        if (kq.seg >= seg_lo && kq.seg < seg_hi) {
          FrameItemDev f;
          f.layout = kq.layout; f.seg = kq.seg; f.obs_begin = int(q); f.obs_count = int(e - q); f.stamp = kq.stamp;
          f.partial_off = int64_t(comp_off);                  // compact record, rebased below
          comp_off += frame_rec(layouts[size_t(kq.layout)]);
          // frames arrive sorted by (layout, segment, stamp): consecutive frames of one cell share one expanded block
          if (p->h_cells.empty() || p->h_cells.back().layout != kq.layout || p->h_cells.back().seg != kq.seg) {
            CellDev c;
            c.layout = kq.layout; c.seg = kq.seg; c.frame_begin = int(p->h_fitems.size()); c.frame_count = 0;
            c.partial_off = int64_t(poff); c.prim_off = 0;
            poff += size_t(tri_size(layouts[size_t(kq.layout)].ncols + 1));      // (the block's upper triangle, packed: problem_dev.hpp)
            p->h_cells.push_back(c);
          }
          p->h_cells.back().frame_count += 1;
          f.cell = int(p->h_cells.size()) - 1; f.cell_frames = 0; f.cell_prim_off = 0; f.cell_pad = 0; f.cell_partial_off = 0; f.cell_src_off = 0;   // (filled below)
          p->h_fitems.push_back(f);
        }
      } else {
        while (e < n_obs && keys[e].layout == kq.layout) ++e;
      }
      q = e;
    }
    // IMU work items hand their staged rows to the cell kernel ("row cells": one expanded block per (layout, segment)
    // instead of one per item); everything else forms its own block
    // fuse_expand (CALICO_FUSE_EXPAND=0: off): no launch for the cell expansion -- a camera cell is expanded by the last of its
    // frames inside the Jacobian launch (eval_kernels.hip), and the other work items form their blocks themselves, each
    // registered as a cell of its own so that the gather's device-built lists see it. Needs every such (layout, segment) to
    // be ONE work item (an IMU cell of at most imu_chunk_items blocks: the usual case).
    bool fuse = !p->h_fitems.empty() && [] { const char* e = std::getenv("CALICO_FUSE_EXPAND"); return !e || std::atoi(e) != 0; }();
    for (const CellDev& c : p->h_cells) if (c.frame_count > 2) fuse = false;       // (a workgroup is two waves: one frame each)
    {
      // ... and two waves' staging areas must fit the CU's LDS
      size_t need = 0;
      for (size_t l = 0; l < layouts.size(); ++l) {
        const LayoutDev& L = layouts[l];
        const HSensor& hs = p->sensors[size_t(L.sensor)];
        if (layout_uses_frames[l]) {
          const int P1 = 7 + (L.c_intr >= 0 ? hs.K : 0) + 3 * (L.c_q >= 0) + 3 * (L.c_t >= 0) + 3 * (L.c_bq >= 0) + 3 * (L.c_bt >= 0);
          const int Ps = 7 + (L.c_intr >= 0 ? hs.K : 0) + 3 * (L.c_bq >= 0);
          need = std::max(need, frame_lds_doubles(Ps, P1, L.ncols + 1));
        } else {
          need = std::max(need, size_t((L.ncols + 1 + 15) & ~15) * size_t((((3 * imu_chunk_items + 3) & ~3) + 1) | 1));      // (as lds_cols x row_pad below)
        }
      }
      need = (need + 1) & ~size_t(1);
      if (cells_launch_lds_bytes(need) > kCellsMaxLds) fuse = false;     // (the same bound the kernel's attribute is set to)
      p->pair_wave_lds_doubles = int(need);
    }
    {
      int prev_layout = -1, prev_seg = -1;
      for (const ItemDev& it : p->h_items) {
        if (layout_uses_frames[size_t(it.layout)]) continue;
        if (it.layout == prev_layout && it.seg == prev_seg) fuse = false;
        if (p->sensors[size_t(layouts[size_t(it.layout)].sensor)].kind == CALICO_SENSOR_CAMERA) fuse = false;     // (camera blocks outside the frame path)
        prev_layout = it.layout; prev_seg = it.seg;
      }
    }
    p->fuse_expand = fuse;
    if (fuse) {
      // two frame entries per workgroup, so that wave w of workgroup g finds its frame at 2 g + w without reading a descriptor
      // first: the two frames of a cell (they expand the cell's block together), or two one-frame cells (`cell_pad` = 1, "solo":
      // each wave expands its own cell alone, no barrier), or a solo frame and an empty entry (obs_count = 0)
      std::vector<FrameItemDev> packed;
      packed.reserve(2 * p->h_cells.size());
      std::vector<FrameItemDev> solos;
      for (CellDev& c : p->h_cells) {
        if (c.frame_count > 1) {
          FrameItemDev f0 = p->h_fitems[size_t(c.frame_begin)], f1 = p->h_fitems[size_t(c.frame_begin) + 1];
          f0.cell_pad = f1.cell_pad = 0;
          packed.push_back(f0); packed.push_back(f1);
        } else {
          FrameItemDev f0 = p->h_fitems[size_t(c.frame_begin)];
          f0.cell_pad = 1;
          solos.push_back(f0);
        }
      }
      for (size_t i = 0; i < solos.size(); i += 2) {
        packed.push_back(solos[i]);
        FrameItemDev f1 = solos[i];
        if (i + 1 < solos.size()) f1 = solos[i + 1]; else f1.obs_count = 0;
        packed.push_back(f1);
      }
      p->h_fitems.swap(packed);
      for (CellDev& c : p->h_cells) c.frame_begin = -1;      // (the frames are no longer contiguous by cell: FrameItemDev.cell says whose they are)
    }
    const bool row_cells_ok = !fuse && [] { const char* e = std::getenv("CALICO_ROW_CELLS"); return !e || std::atoi(e) != 0; }();
    for (ItemDev it : p->h_items) {
      if (layout_uses_frames[size_t(it.layout)]) continue;
      const LayoutDev& L = layouts[size_t(it.layout)];
      const HSensor& hs = p->sensors[size_t(L.sensor)];
      const int n1 = L.ncols + 1;
      if (fuse) {
        // a cell of one work item that writes the cell's block itself (prim_off = -2: nothing for expand_cells_kernel to do)
        CellDev c;
        c.layout = it.layout; c.seg = it.seg; c.frame_begin = int(p->h_jac_items.size()); c.frame_count = 1;
        c.partial_off = int64_t(poff); c.src_off = 0; c.n1 = n1; c.PE = 0; c.prim_off = -2; c.pad0 = 0;
        p->h_cells.push_back(c);
        it.rows_off = -2;            // (< 0: the item forms its own block; -2: that block is listed as a cell's)
        it.partial_off = int64_t(poff);
        poff += size_t(tri_size(n1));
      } else if (row_cells_ok && hs.kind != CALICO_SENSOR_CAMERA && n1 <= 112) {
        it.partial_off = 0; it.rows_off = 0;   // row store offset assigned below, once the staging dimensions are known
        if (p->h_cells.empty() || p->h_cells.back().prim_off >= 0 || p->h_cells.back().layout != it.layout ||
            p->h_cells.back().seg != it.seg) {
          CellDev c;
          c.layout = it.layout; c.seg = it.seg; c.frame_begin = int(p->h_jac_items.size()); c.frame_count = 0;
          c.partial_off = int64_t(poff); c.src_off = 0; c.n1 = n1; c.PE = hs.dim() * imu_chunk_items; c.prim_off = -1; c.pad0 = 0;
          poff += size_t(tri_size(n1));
          p->h_cells.push_back(c);
        }
        p->h_cells.back().frame_count += 1;
        p->h_cells.back().pad0 += hs.dim() * it.obs_count;
      } else {
        it.rows_off = -1;
        it.partial_off = int64_t(poff);
        poff += size_t(tri_size(n1));
      }
      p->h_jac_items.push_back(it);
    }
  }
  p->n_fitems = int(p->h_fitems.size());
  p->n_jac_items = int(p->h_jac_items.size());
  // buffer layout: [expanded partial blocks: cells, generic items | item costs (2 per item) | compact frame records]
  const size_t n_cost_slots = 2 * size_t(std::max(std::max(int(p->h_items.size()), int(p->h_items_all.size())), p->n_fitems + p->n_jac_items));
  const size_t comp_base = poff + n_cost_slots;
  p->cell_rec_max = 1;
  p->frame_lds_doubles = 0;
  for (FrameItemDev& f : p->h_fitems) {
    {
      const LayoutDev& L = layouts[size_t(f.layout)];
      const HSensor& hs = p->sensors[size_t(L.sensor)];
      const int P1 = 7 + (L.c_intr >= 0 ? hs.K : 0) + 3 * (L.c_q >= 0) + 3 * (L.c_t >= 0) + 3 * (L.c_bq >= 0) + 3 * (L.c_bt >= 0);
      const int Ps = 7 + (L.c_intr >= 0 ? hs.K : 0) + 3 * (L.c_bq >= 0);     // staged (small) prim columns: eval_kernels.hip
      p->frame_lds_doubles = std::max(p->frame_lds_doubles, int(frame_lds_doubles(Ps, P1, L.ncols + 1)));
    }
    f.partial_off += int64_t(comp_base);
    p->cell_rec_max = std::max(p->cell_rec_max, int(frame_rec(layouts[size_t(f.layout)])));
  }
  p->cell_chunk = std::max(1, int((56 * 1024 / sizeof(double)) / size_t(p->cell_rec_max)));
  {
    // no more LDS than the fullest cell needs: the cell kernel's workgroups should all be resident at once
    int most = 1;
    for (const CellDev& c : p->h_cells) if (c.prim_off >= 0) most = std::max(most, c.frame_count);
    p->cell_chunk = std::min(p->cell_chunk, most);
  }
  // per-layout prim-column table of the cell kernel (mirror of prim_map / the frame kernel's column order)
  std::vector<int> prim_tab;
  {
    std::vector<int> tab_off(layouts.size(), -1);
    for (CellDev& c : p->h_cells) {
      if (c.prim_off < 0) continue;   // row cell
      const LayoutDev& L = layouts[size_t(c.layout)];
      const HSensor& hs = p->sensors[size_t(L.sensor)];
      int pc = 6;
      const int p_intr = pc; if (L.c_intr >= 0) pc += hs.K;
      const int p_q = pc; if (L.c_q >= 0) pc += 3;
      const int p_t = pc; if (L.c_t >= 0) pc += 3;
      const int p_bq = pc; if (L.c_bq >= 0) pc += 3;
      const int p_bt = pc; if (L.c_bt >= 0) pc += 3;
      const int p_r = pc, PT = pc + 1;     // latency row / column right behind the prim columns
      if (tab_off[size_t(c.layout)] < 0) {
        tab_off[size_t(c.layout)] = int(prim_tab.size());
        std::vector<int> prim;
        for (int lc = 0; lc <= L.ncols; ++lc) {
          int pr;
          if (lc < 36) pr = lc % 6;
          else if (lc == L.c_lat) pr = PT;
          else if (lc == L.ncols) pr = p_r;
          else if (L.c_intr >= 0 && lc >= L.c_intr && lc < L.c_intr + hs.K) pr = p_intr + (lc - L.c_intr);
          else if (L.c_q >= 0 && lc >= L.c_q && lc < L.c_q + 3) pr = p_q + (lc - L.c_q);
          else if (L.c_t >= 0 && lc >= L.c_t && lc < L.c_t + 3) pr = p_t + (lc - L.c_t);
          else if (L.c_bq >= 0 && lc >= L.c_bq && lc < L.c_bq + 3) pr = p_bq + (lc - L.c_bq);
          else pr = p_bt + (lc - L.c_bt);
          prim.push_back(pr);
        }
        // pair table of the cell kernel: row-major upper triangle of the (c+1)×(c+1) block
        const int n1 = L.ncols + 1, PEc = PT + 1;
        for (int i = 0; i < n1; ++i)
          for (int j = i; j < n1; ++j) prim_tab.push_back(i | (j << 8) | ((prim[size_t(i)] * PEc + prim[size_t(j)]) << 16));
      }
      c.prim_off = tab_off[size_t(c.layout)]; c.pad0 = 0;
      c.n1 = L.ncols + 1; c.PE = PT + 1;
      c.src_off = c.frame_begin >= 0 ? p->h_fitems[size_t(c.frame_begin)].partial_off : 0;      // (no compact records with cell workgroups)
    }
    for (FrameItemDev& fi : p->h_fitems) {      // (copies of the cell's fields for the cell's workgroup: fuse_expand)
      const CellDev& c = p->h_cells[size_t(fi.cell)];
      fi.cell_frames = c.frame_count; fi.cell_prim_off = c.prim_off; fi.cell_partial_off = c.partial_off; fi.cell_src_off = c.src_off;
    }
  }
  for (HSensor& s : p->sensors) { s.sorted_begin = n_obs; s.sorted_end = 0; }
  for (int64_t q = 0; q < n_obs; ++q) {
    HSensor& s = p->sensors[keys[q].sensor];
    const int64_t i = keys[q].idx;
    s.sorted_pos[size_t(i)] = q;
    s.sorted_begin = std::min(s.sorted_begin, q); s.sorted_end = std::max(s.sorted_end, q + 1);   // layouts are per sensor: contiguous
    st[q] = s.stamps[i];
    if (s.kind == CALICO_SENSOR_CAMERA) point_off[q] = p->blocks[s.point[i]].amb_off;
  }
  p->n_items = int(p->h_items.size());
  p->n_items_all = int(p->h_items_all.size());
  p->partial_doubles = poff;
  if (poff + 2 * size_t(std::max(p->n_items, p->n_fitems + p->n_jac_items)) >= size_t(0x7fffffff))
    return p->set_error(CALICO_UNIMPLEMENTED, "problem too large for 32-bit gather indices");
  {
    // LDS staging of the generic Jacobian kernel: sized by the items that actually go through it
    int jc = 4, jr = 2;
    for (const ItemDev& it : p->h_jac_items) {
      const LayoutDev& L = layouts[size_t(it.layout)];
      jc = std::max(jc, L.ncols + 1);
      jr = std::max(jr, p->sensors[size_t(L.sensor)].dim() * it.obs_count);
    }
    (void)max_cols;
    // whole groups of sixteen columns and of four rows: stage B reads them without masks (eval_kernels.hip, stage_b_mfma;
    // the padding is cleared by the work item)
    p->lds_cols = (jc + 15) & ~15;
    p->row_pad = (((jr + 3) & ~3) + 1) | 1;
  }
  if (size_t(p->lds_cols) * p->row_pad * sizeof(double) > kMaxLds)
    return p->set_error(CALICO_UNIMPLEMENTED, "too many Jacobian columns per residual block for the LDS staging area");
  // row store of the items that leave [J r]ᵀ[J r] to the cell kernel: behind the compact frame records
  size_t row_store = 0;
  {
    const size_t stride = (size_t(p->lds_cols) * p->row_pad + 1) & ~size_t(1);   // even: the rows travel as 16-byte words
    row_store = (comp_base + comp_off) & 1;                                      // ... from an even offset
    for (size_t i = 0; i < p->h_jac_items.size(); ++i) {
      ItemDev& it = p->h_jac_items[i];
      if (it.rows_off < 0) continue;
      it.rows_off = int64_t(comp_base + comp_off + row_store);
      row_store += stride;
    }
    for (CellDev& c : p->h_cells)
      if (c.prim_off == -1) c.src_off = p->h_jac_items[size_t(c.frame_begin)].rows_off;
    p->row_cell_chunk = std::max(1, int((56 * 1024 / sizeof(double)) / std::max<size_t>(1, stride)));
    int most = 1;
    for (const CellDev& c : p->h_cells) if (c.prim_off == -1) most = std::max(most, c.frame_count);
    p->row_cell_chunk = std::min(p->row_cell_chunk, most);
  }
  section("sort + work items");
  // ---- gather lists ----
  SolveArgs sa; sa.n_cp = n_cp; sa.k = k; sa.mc = m; sa.sep_s = p->sep_s; sa.sep_n = p->sep_n; sa.m = m + p->border_extra(); sa.debug = 0; sa.progress = nullptr;
  const size_t r_size = sa.r_size();
  if (r_size >= size_t(0x7fffffff)) return p->set_error(CALICO_UNIMPLEMENTED, "normal-equation buffer too large");
  struct Pair { int dst, src; };
  std::vector<Pair> pairs;
  const int n_cells = int(p->h_cells.size());
  const int n_part = n_cells + p->n_jac_items;     // producers of expanded partial blocks
  // Lists built on the device: when every producer is a cell (camera frames' cells, IMU row cells) and the layouts are
  // few, the sources of the band, the border and the spline part of the right-hand side follow from the outputs' indices
  // (the band is uniform in time): the host uploads three small tables and the device builds those lists itself
  // (launch_gather_lists: count, scan, fill -- the same CSR form the per-iteration gather reads). Only the corner and the
  // calibration part of the right-hand side (2 % of the outputs, sources in every segment) are listed here.
  // CALICO_GATHER_STRUCT=0: everything listed by the host (A/B switch, and the path of problems with free model points or
  // other spline orders' generic items).
  bool gs_ok = [] { const char* e = std::getenv("CALICO_GATHER_STRUCT"); return !e || std::atoi(e) != 0; }();
  gs_ok = gs_ok && int(layouts.size()) * k <= 96 && int(layouts.size()) >= 1 && m >= 1 && n_cells > 0;
  for (int itn = n_cells; gs_ok && itn < n_part; ++itn) {   // no block of its own, or one that is listed as a cell's (fuse_expand)
    const int64_t ro = p->h_jac_items[size_t(itn - n_cells)].rows_off;
    gs_ok = ro >= 0 || ro == -2;
  }
  const int64_t gs_n_out = int64_t(NS) * m + int64_t(n_cp) * k * 36 + NS;
  gs_ok = gs_ok && gs_n_out * 96 < int64_t(0x7fffffff);
  p->gs_lists_on_device = gs_ok;
  std::vector<int> gs_tab;
  GatherStruct gsd = {};
  if (gs_ok) {
    const int n_lay = int(layouts.size()), nsg = int(p->valid_knots.size()) - 1;
    gs_tab.assign(size_t(n_lay) * nsg + size_t(n_lay) * m + size_t(n_lay), -1);
    for (const CellDev& c : p->h_cells) gs_tab[size_t(c.layout) * nsg + size_t(c.seg)] = int(c.partial_off);
    for (int l = 0; l < n_lay; ++l) {
      const std::vector<int>& gmap = layout_gmap[size_t(l)];
      for (size_t q = 0; q < gmap.size(); ++q) gs_tab[size_t(n_lay) * nsg + size_t(l) * m + size_t(gmap[q] - NS)] = 6 * k + int(q);
      gs_tab[size_t(n_lay) * nsg + size_t(n_lay) * m + size_t(l)] = layouts[size_t(l)].ncols + 1;
    }
    gsd.n_lay = n_lay; gsd.nseg = nsg; gsd.n_cp = n_cp; gsd.k = k; gsd.m = m;
    {   // band blocks at distance d from the diagonal have (k - d) segments per layout: four lanes in the gather where that is <= 24 sources
      int d4 = k;
      while (d4 > 0 && (k - (d4 - 1)) * n_lay <= 24) --d4;
      gsd.d_split = d4;
    }
    gsd.off_g = sa.off_g(); gsd.off_B = sa.off_B(); gsd.off_E = sa.off_E();
  }
  pairs.reserve(gs_ok ? size_t(n_part) * 256 : poff / 2 + 4 * size_t(p->n_items));
  for (int itn = 0; itn < n_part; ++itn) {
    const bool is_cell = itn < n_cells;
    if (!is_cell && (p->h_jac_items[size_t(itn - n_cells)].rows_off >= 0 || p->h_jac_items[size_t(itn - n_cells)].rows_off == -2)) continue;   // its block is a cell's
    const int it_layout = is_cell ? p->h_cells[size_t(itn)].layout : p->h_jac_items[size_t(itn - n_cells)].layout;
    const int it_seg = is_cell ? p->h_cells[size_t(itn)].seg : p->h_jac_items[size_t(itn - n_cells)].seg;
    const int64_t it_poff = is_cell ? p->h_cells[size_t(itn)].partial_off : p->h_jac_items[size_t(itn - n_cells)].partial_off;
    const LayoutDev& L = layouts[size_t(it_layout)];
    const std::vector<int>& gmap = layout_gmap[size_t(it_layout)];
    const int nc = L.ncols, n1 = nc + 1;
    auto tan_of = [&](int c) { return c < 6 * k ? 6 * (it_seg + c / 6) + c % 6 : gmap[size_t(c - 6 * k)]; };
    for (int i = gs_ok ? 6 * k : 0; i < nc; ++i) {      // (structured gather: the spline rows have no lists)
      const int ti = tan_of(i);
      pairs.push_back({int(sa.off_g()) + ti, int(it_poff) + tri_off(i, nc, n1)});
      for (int j = i; j < nc; ++j) {
        const int tj = tan_of(j);
        const int src = int(it_poff) + tri_off(i, j, n1);
        if (ti < NS && tj < NS) {
          const int a = ti / 6, b = tj / 6;  // a <= b
          pairs.push_back({int(sa.off_B()) + (a * k + (b - a)) * 36 + (ti % 6) * 6 + (tj % 6), src});
          if (a == b && ti != tj) pairs.push_back({int(sa.off_B()) + (a * k) * 36 + (tj % 6) * 6 + (ti % 6), src});
        } else if (ti < NS) {
          pairs.push_back({int(sa.off_E() + size_t(ti) * m + (tj - NS)), src});
        } else {
          const int a = ti - NS, b = tj - NS;
          pairs.push_back({int(sa.off_C() + size_t(a) * m + b), src});
          if (a != b) pairs.push_back({int(sa.off_C() + size_t(b) * m + a), src});
        }
      }
    }
  }
  // (outputs 0 and 1 -- cost and invalid count -- are summed by the gather's first workgroup straight from the slot pairs
  //  of the frames and work items behind the partial blocks: no index list)
  // Group the pairs by output, keeping the order in which they were generated inside every group (the summation
  // order of the device's gather, hence its rounding): a counting sort over the outputs -- linear, where a comparison
  // sort of the ~10^6 pairs took most of the set-up time.
  std::vector<int> out_thin, idx_thin, out_fat, idx_fat;
  std::vector<int64_t> ptr_thin(1, 0), ptr_fat(1, 0);
  {
    std::vector<int64_t> start(r_size + 1, 0);
    for (const Pair& pr : pairs) ++start[size_t(pr.dst) + 1];
    for (size_t d = 0; d < r_size; ++d) start[d + 1] += start[d];
    std::vector<int> sorted_src(pairs.size());
    {
      std::vector<int64_t> fill(start.begin(), start.end() - 1);
      for (const Pair& pr : pairs) sorted_src[size_t(fill[size_t(pr.dst)]++)] = pr.src;
    }
    // thin outputs: eight lanes, 6 sources per lane -- or 12 when the problem has outputs of 49..96 sources (many
    // layouts: their band and right-hand-side entries would each take a whole wave otherwise)
    int thin_cap = 48;
    if (gs_ok) thin_cap = int(layouts.size()) * k <= 48 ? 48 : 96;
    else {
      size_t n_mid = 0;
      for (size_t d = 0; d < r_size; ++d) { const int64_t c = start[d + 1] - start[d]; if (c > 48 && c <= 96) ++n_mid; }
      if (n_mid > 0) thin_cap = 96;
    }
    p->thin_per_lane = thin_cap / 8;
    size_t n_thin_src = 0, n_fat_src = 0;
    for (size_t d = 0; d < r_size; ++d) {
      const int64_t c = start[d + 1] - start[d];
      if (gs_ok || c > thin_cap) n_fat_src += size_t(c); else n_thin_src += size_t(c);
    }
    idx_thin.reserve(n_thin_src); idx_fat.reserve(n_fat_src);
    for (size_t d = 0; d < r_size; ++d) {
      const int64_t q0 = start[d], q1 = start[d + 1];
      if (q1 == q0) continue;
      const bool fat = gs_ok || (q1 - q0) > thin_cap;        // (the device's lists are the thin ones: what the host lists goes to the waves)
      std::vector<int>& out = fat ? out_fat : out_thin;
      std::vector<int>& idx = fat ? idx_fat : idx_thin;
      std::vector<int64_t>& ptr = fat ? ptr_fat : ptr_thin;
      out.push_back(int(d));
      idx.insert(idx.end(), sorted_src.begin() + q0, sorted_src.begin() + q1);
      ptr.push_back(int64_t(idx.size()));
    }
  }
  p->n_thin = int(out_thin.size()); p->n_fat = int(out_fat.size());
  p->n_thin8 = p->n_thin; p->n_thin4 = p->n_thin;
  p->gather_owner_block = 0;
  section("gather lists");
  // ---- upload of the structure ----
  hipStream_t s = p->stream;
  std::vector<int> ctrl_off(n_cp);
  for (int i = 0; i < n_cp; ++i) ctrl_off[i] = p->blocks[p->ctrl[i]].amb_off;
  {
    // every work item / frame carries copies of its layout, its sensor and the offsets of its control points
    auto fill = [&](auto& it) {
      it.L = layouts[size_t(it.layout)];
      it.S = sd[size_t(it.L.sensor)];
      for (int i = 0; i < 8; ++i) it.ctrl_off[i] = (i < k && it.seg + i < n_cp) ? ctrl_off[size_t(it.seg + i)] : 0;
    };
    for (ItemDev& it : p->h_items) fill(it);
    for (ItemDev& it : p->h_items_all) fill(it);
    for (ItemDev& it : p->h_jac_items) fill(it);
    for (FrameItemDev& it : p->h_fitems) fill(it);
  }
  HIP_TRY(p, p->d_knots.upload(p->knots, s)); HIP_TRY(p, p->d_basis.upload(p->basis, s));
  HIP_TRY(p, p->d_ctrl_off.upload(ctrl_off, s));
  HIP_TRY(p, p->d_stamp.upload(st, s)); HIP_TRY(p, p->d_point_off.upload(point_off, s));
  HIP_TRY(p, p->d_sensors.upload(sd, s)); HIP_TRY(p, p->d_layouts.upload(layouts, s));
  HIP_TRY(p, p->d_items.upload(p->h_items, s)); HIP_TRY(p, p->d_items_all.upload(p->h_items_all, s));
  HIP_TRY(p, p->d_jac_items.upload(p->h_jac_items, s)); HIP_TRY(p, p->d_fitems.upload(p->h_fitems, s));
  HIP_TRY(p, p->d_blocks.upload(p->h_blocks, s));
  HIP_TRY(p, p->d_cp_active.upload(cp_active, s));
  DevBuf<int> d_cnt;                    // (scratch of the device's list build; freed behind the synchronisation below)
  DevBuf<long long> d_scan;
  // a word of the partials nobody writes (allocated and cleared with them): what padded list entries point to. The lists
  // hold 32-bit positions, so the whole partials buffer must be addressable by one -- checked for every kind of list
  if (comp_base + comp_off + row_store + 2 >= size_t(0x7fffffff)) return p->set_error(CALICO_UNIMPLEMENTED, "problem too large for 32-bit gather indices");
  const int zero_slot = int(comp_base + comp_off + row_store);
  if (gs_ok) {
    HIP_TRY(p, p->d_gs_tab.upload(gs_tab, s));
    gsd.tab = p->d_gs_tab.p;
    const int per_out = int(layouts.size()) * k;      // (<= 96)
    HIP_TRY(p, p->d_out_thin.alloc(size_t(gs_n_out))); HIP_TRY(p, p->d_ptr_thin.alloc(size_t(gs_n_out) + 1));
    HIP_TRY(p, p->d_idx_thin.alloc(size_t(gs_n_out) * per_out)); HIP_TRY(p, d_cnt.alloc(size_t(gs_n_out)));
    HIP_TRY(p, d_scan.alloc(size_t(gs_n_out) / 1024 + 2));      // block sums of the lists' prefix scan
    launch_gather_lists(gsd, int(gs_n_out), d_cnt.p, p->d_out_thin.p, p->d_ptr_thin.p, p->d_idx_thin.p, zero_slot, d_scan.p, s);
    p->n_thin = int(gs_n_out);
    // the border's outputs (behind the right-hand side and the band) have one source per segment and layout that holds
    // their calibration column: at most k where every column belongs to ONE layout -- one lane each in the gather
    bool one_layout = k <= 8;
    for (int tc = 0; one_layout && tc < m; ++tc) {
      int holders = 0;
      for (int l = 0; l < gsd.n_lay; ++l) holders += gs_tab[size_t(gsd.n_lay) * gsd.nseg + size_t(l) * m + size_t(tc)] >= 0 ? 1 : 0;
      one_layout = holders <= 1;
    }
    const bool tiny_env = [] { const char* e = std::getenv("CALICO_GATHER_TINY"); return !e || std::atoi(e) != 0; }();      // (per plan, like the key that hashes it)
    const int n_border0 = NS + n_cp * k * 36;          // first border output
    p->n_thin4 = (one_layout && tiny_env) ? n_border0 : p->n_thin;
    // band blocks at distance d from the diagonal have (k - d) segments per layout: four lanes where that is <= 24 sources
    const int d4 = gsd.d_split;
    p->n_thin8 = (one_layout && tiny_env) ? std::min(n_border0, NS + d4 * n_cp * 36) : p->n_thin;     // (the classes are ranges: [8 | 4 | 1])
  } else {
    HIP_TRY(p, p->d_out_thin.upload(out_thin, s)); HIP_TRY(p, p->d_idx_thin.upload(idx_thin, s));
    HIP_TRY(p, p->d_ptr_thin.upload(ptr_thin, s));
  }
  HIP_TRY(p, p->d_out_fat.upload(out_fat, s)); HIP_TRY(p, p->d_idx_fat.upload(idx_fat, s));
  HIP_TRY(p, p->d_ptr_fat.upload(ptr_fat, s));
  // the thin outputs' lists at a fixed stride per lane class: the gather then needs no pointer load in front of its index
  // loads (CALICO_GATHER_FIXED=0: the CSR form; read per plan and part of its key)
  // Only for the device-built lists: their lengths are bounded by the structure (layouts x k <= thin_per_lane x 8 per output),
  // which is what the fixed stride relies on; host-built lists (plans the table cannot describe) keep the CSR form -- padding
  // each of their short lists to 48 / 96 slots would multiply the index memory, and nothing bounds their length.
  p->gather_fixed = [] { const char* e = std::getenv("CALICO_GATHER_FIXED"); return !e || std::atoi(e) != 0; }() && p->n_thin > 0 && gs_ok;
  if (p->gather_fixed) {
    HIP_TRY(p, p->d_idx_fixed.alloc(gather_fixed_entries(p->n_thin, p->n_thin8, p->n_thin4, p->thin_per_lane) + 8));
    launch_gather_pack_fixed(p->d_ptr_thin.p, p->d_idx_thin.p, p->n_thin, p->n_thin8, p->n_thin4, p->thin_per_lane, zero_slot, p->d_idx_fixed.p, s);
  }
  HIP_TRY(p, p->d_cells.upload(p->h_cells, s)); HIP_TRY(p, p->d_prim_tab.upload(prim_tab, s));
  p->partials_alloc = comp_base + comp_off + row_store + 2;      // (+ the word that is always zero, see zero_slot)
  p->r_size = r_size;
  {
    const char* env = std::getenv("CALICO_SPECULATIVE");
    p->speculative = !env || std::atoi(env) != 0;
  }
  {
    sa = make_solve_args(p);
    if (band_cholesky_lds_bytes(sa) > kMaxLds) return p->set_error(CALICO_UNIMPLEMENTED, "spline order too high for the banded factorisation window");
    p->dense_in_lds = reduced_solve_lds_bytes(sa) <= kMaxLds - 1024;
    if (band_backsolve_lds_bytes(sa) > kMaxLds) return p->set_error(CALICO_UNIMPLEMENTED, "trajectory too long for the back-substitution window");
  }
  if (p->use_bcr) {
    HIP_TRY(p, p->d_bnodes.upload(p->h_bcr_nodes, s)); HIP_TRY(p, p->d_bkeep.upload(p->h_bcr_keep, s));
    // (a member, not a local: the asynchronous upload reads it until the synchronisation below)
    p->h_cp_block.assign(size_t(n_cp), -1);
    for (size_t bi = 0; bi < p->h_blocks.size(); ++bi)
      if (p->h_blocks[bi].tan_off < NS) p->h_cp_block[size_t(p->h_blocks[bi].tan_off / 6)] = int(bi);
    HIP_TRY(p, p->d_cp_block.upload(p->h_cp_block, s));
    if (bcr_level_lds_bytes() > kMaxLds || bcr_back_lds_bytes(p->bcr_q_max, p->bcr_m1p) > kMaxLds)
      return p->set_error(CALICO_UNIMPLEMENTED, "tree solver workspace exceeds the LDS");
  }
  HIP_TRY(p, hipStreamSynchronize(s));      // the uploads read locals of this function
  section("structure uploads");
  return CALICO_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Plan cache. The reference rebuilds its ceres::Problem on every Optimize() (batch_optimizer.cpp:57-70); rebuilt here,
// the flattening would cost more than the solve it feeds. What finalize derives depends on the STRUCTURE of the problem
// only -- block sizes / manifolds / constancy, the spline's knots and basis, the sensors' models, blocks, noise and loss
// settings, and per observation its stamp, rigid body and model point -- so it is keyed on a 128-bit hash of exactly
// that and shared: a handle whose structure has been seen before adopts the cached plan (views of its device buffers)
// and a pooled workspace, and only uploads its values. A changed structure hashes differently and is planned afresh.
// CALICO_PLAN_CACHE=0 switches the cache off; calico_plan_cache_clear() empties it.
// ---------------------------------------------------------------------------------------------------------------------
struct PlanKey {
  uint64_t h1 = 0, h2 = 0; size_t n_blocks = 0, n_obs = 0; int device = 0;
  bool operator==(const PlanKey& o) const { return h1 == o.h1 && h2 == o.h2 && n_blocks == o.n_blocks && n_obs == o.n_obs && device == o.device; }
};
}  // namespace
struct PlanEntry {
  PlanKey key;
  PlanHost host;
  PlanDev dev;
  // per parameter block / per sensor: what finalize writes into the handle's own tables
  struct BlockMeta { int amb_off, tan, eff; bool used; };
  std::vector<BlockMeta> block_meta;
  struct SensorMeta { std::vector<int64_t> sorted_pos; int64_t sorted_begin, sorted_end; };
  std::vector<SensorMeta> sensor_meta;
  std::vector<std::unique_ptr<Workspace>> pool;     // workspaces of destroyed handles, ready for the next one
  uint64_t last_use = 0;
};
namespace {
struct PlanCache {
  std::mutex mu;
  std::vector<std::shared_ptr<PlanEntry>> entries;
  uint64_t tick = 0;
  int64_t hits = 0, misses = 0;
  static constexpr size_t kMaxEntries = 8, kMaxPool = 2;
};
// (never destroyed, like the stream and pinned pools: its buffers would be freed after the HIP runtime is gone)
PlanCache& plan_cache() { static PlanCache* c = new PlanCache; return *c; }
bool plan_cache_enabled() {
  static const bool on = [] { const char* e = std::getenv("CALICO_PLAN_CACHE"); return !e || std::atoi(e) != 0; }();
  return on;
}
struct Hasher {
  uint64_t a = 0x9E3779B97F4A7C15ull, b = 0xC2B2AE3D27D4EB4Full;
  void word(uint64_t w) {
    a = (a ^ w) * 0xff51afd7ed558ccdull; a ^= a >> 32;
    b = (b + w) * 0xc4ceb9fe1a85ec53ull; b ^= b >> 29;
  }
  void bytes(const void* p, size_t n) {
    const unsigned char* c = static_cast<const unsigned char*>(p);
    size_t i = 0;
    if (n >= 256) {
      // long arrays (stamps, ids): four independent lanes of 64-bit words, folded into the two running mixes -- the
      // single multiply chain above hashes at the latency of its multiplications, not at memory speed
      uint64_t l0 = 0x243F6A8885A308D3ull, l1 = 0x13198A2E03707344ull, l2 = 0xA4093822299F31D0ull, l3 = 0x082EFA98EC4E6C89ull;
      for (; i + 32 <= n; i += 32) {
        uint64_t w[4];
        std::memcpy(w, c + i, 32);
        l0 = (l0 ^ w[0]) * 0xff51afd7ed558ccdull; l0 ^= l0 >> 32;
        l1 = (l1 ^ w[1]) * 0xc4ceb9fe1a85ec53ull; l1 ^= l1 >> 29;
        l2 = (l2 ^ w[2]) * 0x9E3779B97F4A7C15ull; l2 ^= l2 >> 31;
        l3 = (l3 ^ w[3]) * 0xD6E8FEB86659FD93ull; l3 ^= l3 >> 30;
      }
      word(l0); word(l1); word(l2); word(l3);
    }
    for (; i + 8 <= n; i += 8) { uint64_t w; std::memcpy(&w, c + i, 8); word(w); }
    if (i < n) { uint64_t w = 0; std::memcpy(&w, c + i, n - i); word(w ^ (uint64_t(n - i) << 56)); }
  }
  template <class T> void vec(const std::vector<T>& v) { word(v.size()); bytes(v.data(), v.size() * sizeof(T)); }
  void dbl(double d) { uint64_t w; std::memcpy(&w, &d, 8); word(w); }
};
PlanKey structure_key(const calico_problem* p) {
  Hasher h;
  h.word(uint64_t(p->order)); h.vec(p->knots); h.vec(p->basis); h.vec(p->ctrl);
  h.word(uint64_t(p->rank)); h.word(uint64_t(p->world));
  h.word(p->blocks.size());
  for (const HBlock& b : p->blocks) h.word(uint64_t(b.size) | (uint64_t(b.manifold) << 32) | (uint64_t(b.constant) << 40));
  h.word(p->bodies.size());
  for (const HBody& b : p->bodies) h.word(uint64_t(uint32_t(b.q)) | (uint64_t(uint32_t(b.t)) << 32));
  h.word(p->sensors.size());
  size_t n_obs = 0;
  for (const HSensor& s : p->sensors) {
    h.word(uint64_t(s.kind) | (uint64_t(s.model) << 8) | (uint64_t(s.K) << 16) | (uint64_t(s.loss) << 32));
    h.word(uint64_t(uint32_t(s.intr)) | (uint64_t(uint32_t(s.q)) << 32)); h.word(uint64_t(uint32_t(s.t)) | (uint64_t(uint32_t(s.lat)) << 32));
    h.word(uint64_t(uint32_t(s.grav)));
    h.dbl(s.sigma); h.dbl(s.info); h.dbl(s.loss_scale);
    h.vec(s.stamps); h.vec(s.body); h.vec(s.point);
    n_obs += s.stamps.size();
  }
  // the switches finalize reads from the environment
  for (const char* name : {"CALICO_SOLVER", "CALICO_SPECULATIVE", "CALICO_BAND_SPLIT", "CALICO_BCR_LEAF", "CALICO_BCR_MERGE_TOP", "CALICO_IMU_CHUNK",
                           "CALICO_ROW_CELLS", "CALICO_FUSE_EXPAND", "CALICO_GATHER_STRUCT", "CALICO_GATHER_TINY", "CALICO_GATHER_FIXED"}) {
    const char* e = std::getenv(name);
    h.word(e ? 1 : 0);
    if (e) h.bytes(e, std::strlen(e));
  }
  PlanKey k; k.h1 = h.a; k.h2 = h.b; k.n_blocks = p->blocks.size(); k.n_obs = n_obs; k.device = p->device;
  return k;
}

// A handle adopts a cached plan: copies of the host-side plan, views of the device-side structure.
void adopt_plan(calico_problem* p, const std::shared_ptr<PlanEntry>& e) {
  static_cast<PlanHost&>(*p) = e->host;
  static_cast<PlanDev&>(*p).alias_from(e->dev);
  for (size_t i = 0; i < p->blocks.size(); ++i) {
    const PlanEntry::BlockMeta& bm = e->block_meta[i];
    p->blocks[i].amb_off = bm.amb_off; p->blocks[i].tan = bm.tan; p->blocks[i].eff = bm.eff; p->blocks[i].used = bm.used;
  }
  for (size_t i = 0; i < p->sensors.size(); ++i) {
    p->sensors[i].sorted_pos = e->sensor_meta[i].sorted_pos;
    p->sensors[i].sorted_begin = e->sensor_meta[i].sorted_begin; p->sensors[i].sorted_end = e->sensor_meta[i].sorted_end;
  }
  p->plan = e;
}

// Everything a handle works in, sized by the plan: a pooled workspace of the same plan if there is one, else allocated
// (and the parts the kernels expect zero-filled cleared) here.
int prepare_workspace(calico_problem* p) {
  hipStream_t s = p->stream;
  if (p->plan) {
    std::unique_ptr<Workspace> w;
    {
      std::lock_guard<std::mutex> lock(plan_cache().mu);
      if (!p->plan->pool.empty()) { w = std::move(p->plan->pool.back()); p->plan->pool.pop_back(); }
    }
    if (w) { static_cast<Workspace&>(*p).swap_ws(*w); p->active_dirty = true; p->xc_stale = true; return CALICO_OK; }   // (w takes the handle's old one along)
  }
  const int n_cp = p->n_cp, m = p->m, k = p->order, NS = 6 * n_cp;
  const int64_t n_obs = p->n_obs;
  const size_t r_size = p->r_size;
  HIP_TRY(p, p->d_x.alloc(size_t(p->n_amb))); HIP_TRY(p, p->d_xc.alloc(size_t(p->n_amb)));
  HIP_TRY(p, p->d_m0.alloc(size_t(n_obs))); HIP_TRY(p, p->d_m1.alloc(size_t(n_obs))); HIP_TRY(p, p->d_m2.alloc(size_t(n_obs)));
  HIP_TRY(p, p->d_partials.alloc(p->partials_alloc));
  HIP_TRY(p, hipMemsetAsync(p->d_partials.p + (p->partials_alloc - 2), 0, 2 * sizeof(double), s));      // the word the lists point to for "nothing"
  if (std::getenv("CALICO_KERNEL_TIMING") && std::atoi(std::getenv("CALICO_KERNEL_TIMING")) >= 3) HIP_TRY(p, p->d_wave_log.alloc(2 * size_t(p->n_jac_items + p->n_fitems) + 8));
  HIP_TRY(p, p->d_R.alloc(2 * r_size)); HIP_TRY(p, hipMemsetAsync(p->d_R.p, 0, 2 * r_size * sizeof(double), s));
  HIP_TRY(p, p->d_R2.alloc(2));
  const int NT = 6 * n_cp + m;
  const int mw = m + p->border_extra();     // border width the solver kernels work with
  HIP_TRY(p, p->d_Lb.alloc(size_t(NS) * 6 * k)); HIP_TRY(p, p->d_Linv.alloc(size_t(n_cp) * 36));
  HIP_TRY(p, p->d_Y.alloc(size_t(NS) * (mw + 1)));
  HIP_TRY(p, p->d_S.alloc(size_t(mw + 1) * (mw + 1)));
  HIP_TRY(p, p->d_y.alloc(size_t(NT) + p->border_extra() + 64)); HIP_TRY(p, p->d_zbuf.alloc(size_t(NS) + 64)); HIP_TRY(p, p->d_dadd.alloc(NT)); HIP_TRY(p, p->d_scale.alloc(2 * size_t(NT)));      // [Jacobi scale s | 1 / s^2]
  HIP_TRY(p, p->d_res.alloc(size_t(n_obs) * 3)); HIP_TRY(p, p->d_valid.alloc(size_t(n_obs)));
  HIP_TRY(p, p->d_active.alloc(size_t(n_obs))); HIP_TRY(p, p->d_counter.alloc(1));
  p->active_dirty = true; p->xc_stale = true;
  HIP_TRY(p, p->d_state.alloc(1)); HIP_TRY(p, p->d_log.alloc(kLogCap));
  // fine-grained (coherent): the terminating stage of a solve writes its results here and the host reads them while
  // later kernels are still on the stream
  if (!p->h_state) {
    HIP_TRY(p, hipHostMalloc(reinterpret_cast<void**>(&p->h_state), sizeof(LmState), hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(p->h_state, 0, sizeof(LmState));
  }
  if (!p->h_log) HIP_TRY(p, hipHostMalloc(reinterpret_cast<void**>(&p->h_log), size_t(kLogCap) * sizeof(IterLog), hipHostMallocMapped | hipHostMallocCoherent));
  if (p->h_xpin_n < size_t(p->n_amb)) {
    if (p->h_xpin) (void)hipHostFree(p->h_xpin);
    p->h_xpin = nullptr; p->h_xpin_n = 0;
    HIP_TRY(p, hipHostMalloc(reinterpret_cast<void**>(&p->h_xpin), std::max<size_t>(1, size_t(p->n_amb)) * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
    p->h_xpin_n = size_t(p->n_amb);
  }
  if (!p->h_progress) {
    HIP_TRY(p, hipHostMalloc(reinterpret_cast<void**>(&p->h_progress), 64, hipHostMallocMapped | hipHostMallocCoherent));   // fine-grained: the host polls it while kernels run
    HIP_TRY(p, hipHostGetDevicePointer(reinterpret_cast<void**>(&p->d_progress), p->h_progress, 0));
    std::memset(p->h_progress, 0, 64);     // epoch 0 is never used: words of "no solve yet"
  }
  const SolveArgs sa = make_solve_args(p);
  const size_t reduced_lds = reduced_solve_lds_bytes(sa);
  HIP_TRY(p, p->d_Spart.alloc(size_t(mw + 1 <= 128 ? 8 : 4) * size_t(mw + 1) * (mw + 1) + 64));   // up to eight K-slices of the Schur complement (long trajectories; four for the blocked path) (+ slack: the blocked factorisation reads whole 32-column panels)
  HIP_TRY(p, p->d_Swork.alloc(std::max<size_t>(reduced_lds / sizeof(double) + 8, size_t(mw + 1) * 16 * 13 + 8)));
  if (p->use_bcr) {
    const size_t N = size_t(p->bcr_N), bb = size_t(kBcrBP) * kBcrBP, fb = size_t(kBcrBP) * p->bcr_m1p;
    HIP_TRY(p, p->d_bD.alloc(N * bb)); HIP_TRY(p, p->d_bG.alloc(2 * N * bb)); HIP_TRY(p, p->d_bF.alloc(N * fb));
    HIP_TRY(p, p->d_bpD.alloc(4 * N * bb)); HIP_TRY(p, p->d_bpF.alloc(4 * N * fb));
    HIP_TRY(p, p->d_bM.alloc(N * bb)); HIP_TRY(p, p->d_bZA.alloc(N * bb)); HIP_TRY(p, p->d_bZB.alloc(N * bb));
    HIP_TRY(p, p->d_bY.alloc(N * fb)); HIP_TRY(p, p->d_bysol.alloc(N * kBcrBP)); HIP_TRY(p, p->d_bzb.alloc(N * kBcrBP)); HIP_TRY(p, p->d_bupd.alloc(size_t(p->bcr_slots) * 4));
    HIP_TRY(p, hipMemsetAsync(p->d_bY.p, 0, N * fb * sizeof(double), s));       // rows of the root are never written
    HIP_TRY(p, hipMemsetAsync(p->d_bG.p, 0, 2 * N * bb * sizeof(double), s));
    HIP_TRY(p, hipMemsetAsync(p->d_bpD.p, 0, 4 * N * bb * sizeof(double), s)); HIP_TRY(p, hipMemsetAsync(p->d_bpF.p, 0, 4 * N * fb * sizeof(double), s));
    HIP_TRY(p, hipMemsetAsync(p->d_bupd.p, 0, size_t(p->bcr_slots) * 4 * sizeof(double), s));
    HIP_TRY(p, p->d_handoff.alloc(16)); HIP_TRY(p, hipMemsetAsync(p->d_handoff.p, 0, 16 * sizeof(int), s));
    p->handoff_seq = 0;
  }
  p->ws_ready = true;
  return CALICO_OK;
}

// Pinned staging buffers for the measurement upload, shared by all handles of the process: a handle holds one from
// upload_values to the synchronisation at the end of finalize (hipHostMalloc costs more than the upload it speeds up).
struct PinnedPool {
  std::mutex mu;
  std::vector<std::pair<double*, size_t>> idle;
  double* acquire(size_t n, size_t* cap) {
    {
      std::lock_guard<std::mutex> lock(mu);
      for (size_t i = 0; i < idle.size(); ++i)
        if (idle[i].second >= n) { double* q = idle[i].first; *cap = idle[i].second; idle.erase(idle.begin() + long(i)); return q; }
    }
    double* q = nullptr;
    const size_t c = n + n / 4;       // (some slack: the next structure is often a little larger)
    if (hipHostMalloc(reinterpret_cast<void**>(&q), c * sizeof(double), hipHostMallocDefault) != hipSuccess) return nullptr;
    *cap = c;
    return q;
  }
  void release(double* q, size_t cap) {
    if (!q) return;
    std::lock_guard<std::mutex> lock(mu);
    if (idle.size() < 2) { idle.emplace_back(q, cap); return; }
    size_t small = 0;
    for (size_t i = 1; i < idle.size(); ++i) if (idle[i].second < idle[small].second) small = i;
    if (idle[small].second < cap) { (void)hipHostFree(idle[small].first); idle[small] = {q, cap}; }
    else (void)hipHostFree(q);
  }
};
PinnedPool& pinned_pool() { static PinnedPool* pp = new PinnedPool(); return *pp; }

// The values: measurements in the device's (sorted) order, parameter vector.
int upload_values(calico_problem* p) {
  hipStream_t s = p->stream;
  const size_t n = size_t(std::max<int64_t>(p->n_obs, 1));
  // pinned staging (part of the workspace, so a pooled one brings it along): the three copies below are DMA transfers
  // that return at once, where pageable vectors went through the runtime's bounce buffers synchronously
  if (p->h_mpin && p->h_mpin_n < 3 * n) { pinned_pool().release(p->h_mpin, p->h_mpin_n); p->h_mpin = nullptr; p->h_mpin_n = 0; }
  if (!p->h_mpin) {
    p->h_mpin = pinned_pool().acquire(3 * n, &p->h_mpin_n);
    if (!p->h_mpin) return p->set_error(CALICO_INTERNAL, "hipHostMalloc (measurement staging) failed");
  }
  double* m0 = p->h_mpin; double* m1 = m0 + n; double* m2 = m1 + n;
  for (const HSensor& sn : p->sensors) {
    const int dim = sn.dim();
    const int64_t ns = sn.n();
    const double* me = sn.meas.data();
    const int64_t* sp = sn.sorted_pos.data();
    if (dim == 2) for (int64_t i = 0; i < ns; ++i) { const size_t q = size_t(sp[i]); m0[q] = me[2 * i]; m1[q] = me[2 * i + 1]; m2[q] = 0.0; }
    else for (int64_t i = 0; i < ns; ++i) { const size_t q = size_t(sp[i]); m0[q] = me[3 * i]; m1[q] = me[3 * i + 1]; m2[q] = me[3 * i + 2]; }
  }
  if (p->n_obs > 0) {
    HIP_TRY(p, hipMemcpyAsync(p->d_m0.p, m0, size_t(p->n_obs) * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(p, hipMemcpyAsync(p->d_m1.p, m1, size_t(p->n_obs) * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(p, hipMemcpyAsync(p->d_m2.p, m2, size_t(p->n_obs) * sizeof(double), hipMemcpyHostToDevice, s));
  }
  p->h_x.assign(size_t(p->n_amb), 0.0);
  for (const HBlock& b : p->blocks) std::copy(b.v.begin(), b.v.end(), p->h_x.begin() + b.amb_off);
  if (p->n_amb > 0) {
    HIP_TRY(p, hipMemcpyAsync(p->d_x.p, p->h_x.data(), p->h_x.size() * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(p, hipMemcpyAsync(p->d_xc.p, p->h_x.data(), p->h_x.size() * sizeof(double), hipMemcpyHostToDevice, s));
  }
  return CALICO_OK;
}

int configure_kernels(calico_problem* p) {
  const SolveArgs sa = make_solve_args(p);
  const size_t reduced_lds = reduced_solve_lds_bytes(sa);
  // The kernels' dynamic-LDS limits depend on a handful of sizes. They are only ever RAISED on a device (two live handles
  // of different shapes must not lower each other's limits), and the forty-odd hipFuncSetAttribute calls (0.3 ms) are
  // skipped when the device already allows what this handle needs -- every handle of a known structure, and most others.
  const bool db_fits = p->use_bcr && std::max(dense_block_solve_lds_bytes(), bcr_back_lds_bytes(std::min(p->bcr_q_max, 4), p->bcr_m1p)) + 1024 <= kMaxLds;
  const std::array<size_t, 8> want = {size_t(p->lds_cols) * p->row_pad * sizeof(double), band_cholesky_lds_bytes(sa),
                                      p->dense_in_lds ? reduced_lds : 0, band_backsolve_lds_bytes(sa), size_t(p->use_bcr ? 1 : 0),
                                      size_t(p->use_bcr ? p->bcr_q_max : 0), size_t(p->use_bcr ? p->bcr_m1p : 0),
                                      db_fits ? dense_back_lds_bytes(std::min(p->bcr_q_max, 4), p->bcr_m1p) : 0};
  static std::mutex mu;
  static std::map<int, std::array<size_t, 8>> allowed;
  std::lock_guard<std::mutex> lock(mu);
  std::array<size_t, 8>& cur = allowed[p->device];       // (zeros for a device seen for the first time)
  std::array<size_t, 8> nw;
  for (size_t i = 0; i < nw.size(); ++i) nw[i] = std::max(cur[i], want[i]);
  if (nw == cur) return CALICO_OK;
  HIP_TRY(p, configure_eval_kernels(nw[0]));
  HIP_TRY(p, configure_solve_kernels(nw[1], nw[2], nw[3]));
  HIP_TRY(p, configure_dense_block_solve());
  HIP_TRY(p, configure_reduced_block_step());
  HIP_TRY(p, configure_reduced_fused());
  if (nw[4]) HIP_TRY(p, configure_bcr_kernels(int(nw[5]), int(nw[6])));
  if (nw[7]) HIP_TRY(p, configure_dense_back_bytes(nw[7]));
  cur = nw;
  return CALICO_OK;
}

// Plan (cached or built), workspace (pooled or allocated), values.
int finalize(calico_problem* p) {
  if (!p->dirty) return CALICO_OK;
  p->res_cache.valid = false;
  if (p->order <= 0) return p->set_error(CALICO_FAILED_PRECONDITION, "spline not set");
  if (p->order > 8) return p->set_error(CALICO_UNIMPLEMENTED, "spline order > 8 is not supported by the HIP kernels");
  HIP_TRY(p, hipSetDevice(p->device));
  static const bool setup_timing = std::getenv("CALICO_SETUP_TIMING") != nullptr;
  auto t_sec = std::chrono::steady_clock::now();
  auto section = [&](const char* name) {
    if (!setup_timing) return;
    const auto t = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[calico] finalize %-28s %8.3f ms\n", name, std::chrono::duration<double, std::milli>(t - t_sec).count());
    t_sec = t;
  };
  // a workspace that belongs to the plan the handle is leaving goes back to that plan's pool
  const bool use_cache = plan_cache_enabled();
  PlanKey key;
  std::shared_ptr<PlanEntry> hit;
  if (use_cache) {
    key = structure_key(p);
    PlanCache& c = plan_cache();
    std::lock_guard<std::mutex> lock(c.mu);
    for (const std::shared_ptr<PlanEntry>& e : c.entries) if (e->key == key) { hit = e; break; }
    if (hit) { hit->last_use = ++c.tick; ++c.hits; } else ++c.misses;
  }
  section("structure key + look-up");
  if (hit && p->plan == hit && p->ws_ready) {
    // same structure as before on the same handle (values re-registered): nothing to rebuild
  } else {
    if (p->plan && p->ws_ready) {        // leaving another plan: its workspace stays with it
      HIP_TRY(p, hipStreamSynchronize(p->stream));
      auto w = std::make_unique<Workspace>();
      w->swap_ws(static_cast<Workspace&>(*p));
      std::lock_guard<std::mutex> lock(plan_cache().mu);
      if (p->plan->pool.size() < PlanCache::kMaxPool) p->plan->pool.push_back(std::move(w));
    }
    p->plan.reset();
    p->ws_ready = false;
    if (hit) adopt_plan(p, hit);
    else {
      const int rc = build_plan(p);
      if (rc != CALICO_OK) return rc;
      if (use_cache) {
        auto e = std::make_shared<PlanEntry>();
        e->key = key;
        e->host = static_cast<const PlanHost&>(*p);
        e->block_meta.resize(p->blocks.size());
        for (size_t i = 0; i < p->blocks.size(); ++i) e->block_meta[i] = {p->blocks[i].amb_off, p->blocks[i].tan, p->blocks[i].eff, p->blocks[i].used};
        e->sensor_meta.resize(p->sensors.size());
        for (size_t i = 0; i < p->sensors.size(); ++i)
          e->sensor_meta[i] = {p->sensors[i].sorted_pos, p->sensors[i].sorted_begin, p->sensors[i].sorted_end};
        e->dev.take_from(static_cast<PlanDev&>(*p));
        static_cast<PlanDev&>(*p).alias_from(e->dev);
        p->plan = e;
        PlanCache& c = plan_cache();
        std::shared_ptr<PlanEntry> evicted;
        {
          std::lock_guard<std::mutex> lock(c.mu);
          e->last_use = ++c.tick;
          if (c.entries.size() >= PlanCache::kMaxEntries) {
            size_t old = 0;
            for (size_t i = 1; i < c.entries.size(); ++i) if (c.entries[i]->last_use < c.entries[old]->last_use) old = i;
            evicted = std::move(c.entries[old]);
            c.entries.erase(c.entries.begin() + long(old));      // (handles that still use it keep it alive)
          }
          c.entries.push_back(e);
        }
        if (evicted && evicted.use_count() == 1) {       // its buffers go back now: one wait for ITS device, outside the cache's lock
          DeviceArena::Batch batch(evicted->key.device);
          evicted.reset();
        }
      }
    }
    section(hit ? "plan adopted" : "plan built");
    const int rc = prepare_workspace(p);
    if (rc != CALICO_OK) return rc;
    section("workspace");
  }
  int rc = upload_values(p);
  if (rc != CALICO_OK) return rc;
  rc = configure_kernels(p);
  if (rc != CALICO_OK) return rc;
  HIP_TRY(p, hipStreamSynchronize(p->stream));
  pinned_pool().release(p->h_mpin, p->h_mpin_n); p->h_mpin = nullptr; p->h_mpin_n = 0;     // (the uploads are through)
  section("values + kernel attributes");
  p->dirty = false;
  return CALICO_OK;
}

int upload_x(calico_problem* p, bool seed = true) {
  if (p->active_dirty) {   // outlier tags, in the sorted order of the device arrays
    std::vector<uint8_t> act(size_t(std::max<int64_t>(p->n_obs, 1)), 1);
    bool tagged = false;
    for (const HSensor& s : p->sensors)
      for (int64_t i = 0; i < s.n(); ++i) { act[size_t(s.sorted_pos[size_t(i)])] = s.active[size_t(i)]; tagged = tagged || !s.active[size_t(i)]; }
    p->any_tagged = tagged;
    HIP_TRY(p, hipMemcpyAsync(p->d_active.p, act.data(), size_t(p->n_obs), hipMemcpyHostToDevice, p->stream));
    HIP_TRY(p, hipStreamSynchronize(p->stream));   // `act` is a local
    p->active_dirty = false;
  }
  for (const HBlock& b : p->blocks) std::copy(b.v.begin(), b.v.end(), p->h_x.begin() + b.amb_off);
  // through the pinned staging buffer: a true asynchronous DMA (every API call ends with a stream synchronisation, so
  // the buffer is never rewritten while a transfer is pending)
  std::copy(p->h_x.begin(), p->h_x.end(), p->h_xpin);
  if (!seed) return CALICO_OK;   // calico_solve: the kernel that resets the LM state reads the staging buffer
  launch_seed_x(p->d_x.p, p->h_xpin, int(p->h_x.size()), p->stream);      // the kernel reads the pinned buffer: no DMA copy (~13 us) on the stream
  // d_xc needs no upload: every parameter block, constant ones included, is rewritten by the update kernel... except
  // the constant blocks, which the update never touches -- so it is seeded once per finalisation (below) and whenever
  // a constant block may have changed
  if (p->xc_stale) {
    HIP_TRY(p, hipMemcpyAsync(p->d_xc.p, p->h_xpin, p->h_x.size() * sizeof(double), hipMemcpyHostToDevice, p->stream));
    p->xc_stale = false;
  }
  return CALICO_OK;
}

int do_allreduce(calico_problem* p, double* buf, int64_t n) {
  if (p->comm) {     // native: one in-place RCCL all-reduce on the handle's stream, no host code in between
    const ncclResult_t r = rccl().AllReduce(buf, buf, size_t(n), ncclDouble, ncclSum, p->comm, p->stream);
    if (r != ncclSuccess) return p->set_error(CALICO_INTERNAL, std::string("ncclAllReduce: ") + rccl().GetErrorString(r));
    return CALICO_OK;
  }
  if (!p->allreduce) return CALICO_OK;
  const int st = p->allreduce(p->allreduce_ctx, buf, n, p->stream);
  if (st != 0) return p->set_error(CALICO_INTERNAL, "all-reduce callback failed");
  return CALICO_OK;
}

// residual + Jacobian evaluation at d_x into the reduce buffer R. With st != nullptr the
// kernels skip themselves on the device when the solve has terminated or (need_flag) when
// the last step was rejected, so whole iterations can be enqueued without a host round trip.
// `spec`: evaluation at the candidate point x_at = x_cand into the reduce buffer that does NOT hold R(x) (chosen on
// the device from LmState.rcur); otherwise evaluation at x into buffer 0.
// `end_hint` (streaming solve loop, fused Jacobian launch only: end_hint_available): the launch tells the host whether the
// control stage behind it is about to end the solve (eval_kernels.hip, end_hint_body).
static bool end_hint_available(const calico_problem* p) { return p->order == 6 && p->n_fitems > 0; }
int enqueue_jacobian_eval(calico_problem* p, const LmState* st, int need_flag, const double* x_at = nullptr, bool spec = false,
                          const ControlTail* tail = nullptr, bool end_hint = false) {
  p->timer.begin(0, p->stream);
  EvalArgs ea = make_eval_args(p, x_at ? x_at : p->d_x.p, 1, false);
  ea.st = st; ea.need_flag = need_flag;
  if (end_hint && tail && st && end_hint_available(p)) {
    ea.hint_progress = tail->progress; ea.hint_seq = tail->seq;
    // (the hint's workgroup first in the grid: it is dispatched with the launch, not behind the first workgroups that end
    //  -- the host has the whole evaluation to enqueue the next iteration. CALICO_HINT_FIRST=0: last, A/B switch read per solve)
    ea.hint_first = [] { const char* e = std::getenv("CALICO_HINT_FIRST"); return !e || std::atoi(e) != 0; }() ? 1 : 0;
    ea.hint_ftol = tail->o.function_tolerance; ea.hint_ptol = tail->o.parameter_tolerance;
  }
  ea.items = p->d_jac_items.p; ea.n_items = p->n_jac_items; ea.cost_index_base = p->n_fitems;
  if (p->fuse_expand) { ea.pair_mode = 1; ea.wave_lds_doubles = p->pair_wave_lds_doubles; }      // (the plan has frames and order 6 then)
  // a launch the runtime refuses (too much LDS for the kernel's attribute, a bad grid) would leave last iteration's blocks
  // in place and the solve would go wrong silently on stale partials: ask behind EVERY launch (a thread-local read, no
  // synchronisation). The thread's error word is cleared first -- a benign error some other code on this thread left behind
  // (a PyTorch probe, a hipMalloc fallback) is not this solve's --, and asked per launch: hipGetLastError() reports the last
  // call only on some runtimes, so a refused first launch must not hide behind a second one that went through.
  (void)hipGetLastError();
  if (p->order == 6 && p->n_fitems > 0) {
    launch_eval_jacobian(ea, p->stream);                  // camera frames (item-cost slots [0, n_fitems)) + everything else
    HIP_TRY(p, hipGetLastError());
  } else {
    launch_eval_frames(ea, p->stream);
    HIP_TRY(p, hipGetLastError());
    launch_eval(ea, true, p->stream);
    HIP_TRY(p, hipGetLastError());
  }
  p->timer.end(p->stream);
  p->timer.begin(1, p->stream);
  // the host knows which buffer is filled: multi-rank runs either read the state back every iteration or (batched)
  // always evaluate the candidate into buffer 1
  double* target = p->d_R.p + ((spec && p->h_state && !p->h_state->rcur) ? p->r_size : 0);
  if (p->has_exchange() && p->world > 1) {
    // a rank's gather only writes the entries its own residual blocks contribute to; the others must enter the sum
    // as zeros, not as what the previous reduction left there
    HIP_TRY(p, hipMemsetAsync(target, 0, p->r_size * sizeof(double), p->stream));
  }
  if (!p->fuse_expand) launch_expand_cells(ea, p->stream);   // compact frame records -> one expanded block per cell
  launch_gather(p->d_R.p, p->d_partials.p, p->d_out_thin.p, p->gather_fixed ? nullptr : p->d_ptr_thin.p, p->gather_fixed ? p->d_idx_fixed.p : p->d_idx_thin.p, p->n_thin, p->n_thin8, p->n_thin4, p->thin_per_lane, p->d_out_fat.p,
                p->d_ptr_fat.p, p->d_idx_fat.p, p->n_fat, p->d_partials.p + p->partial_doubles, p->n_fitems + p->n_jac_items, st, need_flag,
                spec ? p->r_size : 0, p->stream, tail);
  p->timer.end(p->stream);
  if (!p->has_exchange()) return CALICO_OK;  // single rank: no exchange
  return do_allreduce(p, target, int64_t(p->r_size));
}

// One linear solve + update of the candidate point: tree solver or sequential banded factorisation.
// with_post_eval: 0 none, 1 the bookkeeping of the step just accepted rides in the first launch, 2 the bookkeeping of the
// solve's FIRST evaluation does (tree solver only: level 0 then forms the Jacobi scale of its diagonal entries itself)
void enqueue_linear_solve(calico_problem* p, const SolveArgs& sa, const LmOptionsDev& o, int with_post_eval, int jacobi) {
  hipStream_t s = p->stream;
  const int n_blocks = int(p->h_blocks.size());
  if (!p->use_bcr) {
    launch_solve(sa, o, p->d_x.p, p->d_xc.p, p->d_blocks.p, n_blocks, p->dense_in_lds, s, with_post_eval == 1, p->d_log.p, kLogCap, jacobi);
    return;
  }
  const BcrArgs b = make_bcr_args(p);
  const int L = int(p->bcr_levels.size());
  const int ks = reduced_schur_slices(sa);
  // The Schur complement rides in the last level's launch (its tiles over the rows eliminated below that level run beside
  // the level's chains; the level's own rows and the root's rows follow an in-launch fan-in): one launch less.
  bool schur_rides = schur_rides_in_last_level(L, p->bcr_levels[size_t(L - 1)].n_nodes, p->bcr_root);
  for (int i = 0; schur_rides && i < p->bcr_levels[size_t(L - 1)].n_nodes; ++i)
    schur_rides = p->h_bcr_nodes[size_t(p->bcr_levels[size_t(L - 1)].node0 + i)].q == 1;
  int* const fan_word = p->d_handoff.p + 4;
  // (A/B switch, read per solve: 0 = every level reads its node descriptors from the table)
  const bool inline_nodes = [] { const char* e = std::getenv("CALICO_INLINE_NODES"); return !e || std::atoi(e) != 0; }();
  for (int l = 0; l < L; ++l) {
    const BcrLevel& lv = p->bcr_levels[size_t(l)];
    BcrInlineNodes inl = {};
    if (inline_nodes) {
      if (l == 0) inl.q_regular = p->bcr_q0;
      else if (lv.n_nodes <= 4) { inl.n = lv.n_nodes; for (int i = 0; i < lv.n_nodes; ++i) inl.nd[i] = p->h_bcr_nodes[size_t(lv.node0 + i)]; }
    }
    launch_bcr_level(sa, b, lv.node0, lv.n_nodes, l, lv.keep0, lv.n_keep, o, p->d_x.p, p->d_blocks.p, n_blocks, l == 0 ? with_post_eval : 0,
                     p->d_log.p, kLogCap, jacobi, s, schur_rides && l == L - 1 ? ks : 0, schur_rides ? fan_word : nullptr, inl, lv.q_max);
  }
  if (!schur_rides) launch_bcr_schur(sa, b, ks, o, s);
  // The top level of the tree is one or two single superblocks next to the root: their back-substitution rides in the
  // launch of the level below (every node there solves the top separators beside it itself -- a few more loads next to
  // the ones it waits for anyway) instead of costing a launch of its own.
  BcrTopSeps ts = {};
  if (p->bcr_merge_top && L >= 2) {
    const BcrLevel& tl = p->bcr_levels[size_t(L - 1)];
    bool ok = tl.n_nodes <= 2 && p->bcr_levels[size_t(L - 2)].q_max <= 4;
    for (int i = 0; ok && i < tl.n_nodes; ++i) {
      const BcrNodeDev& nd = p->h_bcr_nodes[size_t(tl.node0 + i)];
      ok = nd.q == 1 && (nd.left < 0 || nd.left == p->bcr_root) && (nd.right < 0 || nd.right == p->bcr_root);
      ts.blk[i] = nd.blk0; ts.left[i] = nd.left; ts.right[i] = nd.right;
    }
    ts.n = ok ? tl.n_nodes : 0;
  }
  // The first back-substitution launch rides in the launch of the dense reduced solve where the shapes allow it (the
  // nodes fetch what they need while the solve runs and take its solution over a hand-off word: dense_back_kernel).
  const int l_first = ts.n > 0 ? L - 2 : L - 1;
  const BcrLevel& lf = p->bcr_levels[size_t(l_first)];
  const bool fused = l_first == 0 && dense_back_fusable(sa, ks, lf.q_max, /*border_rows=*/l_first > 0) &&
                     std::max(dense_block_solve_lds_bytes(), bcr_back_lds_bytes(lf.q_max, p->bcr_m1p)) + 1024 <= kMaxLds;
  p->timer.begin(6, s);       // the launch that solves the reduced system: the longest kernel of an iteration at configs[3]
  if (fused) {
    p->handoff_seq = p->handoff_seq % 0x3fffffff + 1;
    launch_dense_back(sa, b, ks, lf.node0, lf.n_nodes, lf.q_max, p->d_x.p, p->d_xc.p, p->d_blocks.p, n_blocks, ts, p->d_handoff.p, p->handoff_seq, s);
  } else {
    launch_reduced_solve(sa, p->dense_in_lds, ks, s, p->d_handoff.p + 8);      // (words 8..15: the fan-ins of a blocked factorisation in one launch)
  }
  p->timer.end(s);
  for (int l = L - 1; l >= 0; --l) {
    const BcrLevel& lv = p->bcr_levels[size_t(l)];
    if (ts.n > 0 && l == L - 1) continue;
    const bool first = l == L - 1 || (ts.n > 0 && l == L - 2);     // the first launch behind the reduced solve
    if (first && fused) continue;
    const BcrTopSeps none = {};
    launch_bcr_back(sa, b, lv.node0, lv.n_nodes, first, first, /*border_rows=*/l > 0, lv.q_max, p->d_x.p, p->d_xc.p, p->d_blocks.p, n_blocks,
                    first ? ts : none, s);
  }
  // development aid (CALICO_CHECK_FINITE=1): where does the first non-finite value of a solve sit?
  static const bool check = std::getenv("CALICO_CHECK_FINITE") != nullptr;
  if (check) {
    (void)hipStreamSynchronize(s);
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) std::fprintf(stderr, "[calico] launch error after the tree solve: %s\n", hipGetErrorString(le));
    auto scan = [&](const char* name, const double* d, size_t n) {
      std::vector<double> h(n);
      (void)hipMemcpy(h.data(), d, n * sizeof(double), hipMemcpyDeviceToHost);
      size_t bad = 0, first = 0;
      for (size_t i = 0; i < n; ++i) if (!std::isfinite(h[i])) { if (!bad) first = i; ++bad; }
      if (bad) std::fprintf(stderr, "[calico] %s: %zu of %zu non-finite, first at %zu\n", name, bad, n, first);
    };
    const size_t N = size_t(p->bcr_N), bb = size_t(kBcrBP) * kBcrBP, fb = size_t(kBcrBP) * p->bcr_m1p, m1 = size_t(sa.m) + 1;
    scan("R", p->d_R.p, 2 * p->r_size); scan("D", b.D, N * bb); scan("F", b.F, N * fb); scan("M", b.M, N * bb); scan("ZA", b.ZA, N * bb);
    scan("ZB", b.ZB, N * bb); scan("Y", b.Y, N * fb); scan("Spart", sa.Spart, size_t(ks) * m1 * m1); scan("y", sa.y, size_t(sa.NT()) + p->border_extra());
    scan("dadd", sa.dadd, size_t(sa.NT())); scan("scale", sa.scale, size_t(sa.NT()));
    scan("zb", b.zb, N * kBcrBP); scan("ysol", b.ysol, N * kBcrBP); scan("x", p->d_x.p, size_t(p->n_amb)); scan("x_cand", p->d_xc.p, size_t(p->n_amb));
    {
      std::vector<double> h(N * kBcrBP);
      (void)hipMemcpy(h.data(), b.ysol, h.size() * sizeof(double), hipMemcpyDeviceToHost);
      std::string okb;
      for (size_t I = 0; I < N; ++I) { bool ok = true; for (int r = 0; r < kBcrBP; ++r) ok = ok && std::isfinite(h[I * kBcrBP + r]); okb += ok ? '.' : 'X'; }
      std::fprintf(stderr, "[calico] ysol by superblock (X = non-finite): %s  root %d levels %zu\n", okb.c_str(), p->bcr_root, p->bcr_levels.size());
    }
  }
}

int read_state(calico_problem* p) {
  HIP_TRY(p, hipMemcpyAsync(p->h_state, p->d_state.p, sizeof(LmState), hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(p, hipStreamSynchronize(p->stream));
  p->timer.resolve();
  return CALICO_OK;
}

void fill_counts(calico_problem* p, calico_summary* sm) {
  int nrb = 0, nr = 0;
  for (HSensor& s : p->sensors) {   // tagged outliers are not part of the problem (camera.cpp:121-124)
    if (s.n_active < 0) { int64_t c = 0; for (uint8_t a : s.active) c += a ? 1 : 0; s.n_active = c; }   // per solve otherwise: 100k bytes
    const int64_t na = s.n_active;
    nrb += int(na); nr += int(na) * s.dim();
  }
  sm->num_residual_blocks = nrb; sm->num_residuals = nr;
  sm->num_residual_blocks_reduced = nrb; sm->num_residuals_reduced = nr;
  sm->num_parameter_blocks = int(p->blocks.size());
  int np = 0, ne = 0, npr = 0;
  for (const HBlock& b : p->blocks) { np += b.size; ne += b.tangent_size(); }
  sm->num_parameters = np; sm->num_effective_parameters = ne;
  sm->num_parameter_blocks_reduced = int(p->h_blocks.size());
  for (const BlockDev& b : p->h_blocks) npr += b.size;
  sm->num_parameters_reduced = npr;
  sm->num_effective_parameters_reduced = p->n_eff;
}

const char* reason_message(int reason) {
  switch (reason) {
    case 1: return "Maximum number of iterations reached.";
    case 2: return "Gradient tolerance reached.";
    case 3: return "Minimum trust region radius reached.";
    case 4: return "Parameter tolerance reached.";
    case 5: return "Function tolerance reached.";
    case 10: return "Initial residual and Jacobian evaluation failed.";
    case 11: return "Residual and Jacobian evaluation failed.";
    case 12: return "Number of consecutive invalid steps more than Solver::Options::max_num_consecutive_invalid_steps.";
    default: return "";
  }
}

}  // namespace

namespace {
struct StreamPool {
  std::mutex mu;
  std::map<int, std::vector<hipStream_t>> idle;     // per device: streams of destroyed handles
  static constexpr size_t kMaxIdle = 4;
};
StreamPool& stream_pool() { static StreamPool* sp = new StreamPool(); return *sp; }     // (never destroyed: the streams outlive static destruction)
}  // namespace

extern "C" {

int32_t calico_problem_create(calico_problem** out, int32_t device) {
  if (!out) return CALICO_INVALID_ARGUMENT;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return CALICO_INTERNAL;
  if (hipSetDevice(device) != hipSuccess) return CALICO_INTERNAL;
  calico_problem* p = new calico_problem();
  p->device = device;
  // (a stream costs a few hundred microseconds to create: the handles of a create-solve-destroy loop pass theirs on)
  {
    StreamPool& sp = stream_pool();
    std::lock_guard<std::mutex> lock(sp.mu);
    std::vector<hipStream_t>& v = sp.idle[device];
    if (!v.empty()) { p->stream = v.back(); v.pop_back(); }
  }
  if (!p->stream && hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) != hipSuccess) { delete p; return CALICO_INTERNAL; }
  p->own_stream = true;
  *out = p;
  return CALICO_OK;
}

void calico_problem_destroy(calico_problem* p) {
  if (!p) return;
  (void)hipSetDevice(p->device);
  // first drain the stream -- the iterations enqueued ahead of a terminated multi-rank solve each still carry an
  // all-reduce --, then give the communicator back
  if (p->stream) (void)hipStreamSynchronize(p->stream);
  if (p->comm) { (void)rccl().CommDestroy(p->comm); p->comm = nullptr; }
  // the workspace stays with the cached plan: the next handle of this structure takes it over instead of allocating
  if (p->plan && p->ws_ready) {
    auto w = std::make_unique<Workspace>();
    w->swap_ws(static_cast<Workspace&>(*p));
    std::lock_guard<std::mutex> lock(plan_cache().mu);
    if (p->plan->pool.size() < PlanCache::kMaxPool) p->plan->pool.push_back(std::move(w));
  }
  pinned_pool().release(p->h_mpin, p->h_mpin_n); p->h_mpin = nullptr;      // (only set if a finalize failed half-way)
  if (p->own_stream && p->stream) {      // (drained above)
    StreamPool& sp = stream_pool();
    std::lock_guard<std::mutex> lock(sp.mu);
    std::vector<hipStream_t>& v = sp.idle[p->device];
    if (v.size() < StreamPool::kMaxIdle) v.push_back(p->stream); else (void)hipStreamDestroy(p->stream);
  }
  {
    DeviceArena::Batch batch(p->device);     // what the handle owns goes back to the arena behind ONE wait for its device
    delete p;
  }
  // Idle slabs beyond a few go back to the driver -- lazily: hipFree waits for the whole device (other handles' streams,
  // PyTorch's), and a create / solve / destroy loop over a problem of several slabs would hipMalloc them again every cycle.
  // CALICO_ARENA_KEEP_SLABS (default 4 = 256 MB per device) idle slabs stay; calico_plan_cache_clear() frees them all.
  static const int keep_slabs = [] { const char* e = std::getenv("CALICO_ARENA_KEEP_SLABS"); return e ? std::max(0, std::atoi(e)) : 4; }();
  DeviceArena::get().trim(keep_slabs);
}

int32_t calico_plan_cache_stats(int64_t* hits, int64_t* misses, int64_t* entries) {
  PlanCache& c = plan_cache();
  std::lock_guard<std::mutex> lock(c.mu);
  if (hits) *hits = c.hits;
  if (misses) *misses = c.misses;
  if (entries) *entries = int64_t(c.entries.size());
  return CALICO_OK;
}

int32_t calico_plan_cache_clear(void) {
  PlanCache& c = plan_cache();
  std::vector<std::shared_ptr<PlanEntry>> drop;
  {
    std::lock_guard<std::mutex> lock(c.mu);
    drop.swap(c.entries);      // (entries that live handles still refer to are freed with the last of them)
  }
  for (std::shared_ptr<PlanEntry>& e : drop) {
    DeviceArena::Batch batch(e->key.device);
    {
      std::lock_guard<std::mutex> lock(c.mu);      // (a live handle of this plan may be taking a workspace from the pool)
      e->pool.clear();
    }
    e.reset();
  }
  DeviceArena::get().trim(0);   // every slab no live handle has a buffer in goes back to the driver
  return CALICO_OK;
}

const char* calico_last_error(const calico_problem* p) { return p ? p->error.c_str() : "null problem"; }

void calico_default_solver_options(calico_solver_options* o) {
  // DefaultSolverOptions() (batch_optimizer.cpp:10-17) over Ceres' Solver::Options defaults.
  o->max_num_iterations = 50; o->num_threads = 1; o->minimizer_progress_to_stdout = 1; o->jacobi_scaling = 1;
  o->max_num_consecutive_invalid_steps = 5; o->sync_every = 1;
  o->function_tolerance = 1e-8; o->gradient_tolerance = 1e-10; o->parameter_tolerance = 1e-10;
  o->initial_trust_region_radius = 1e4; o->max_trust_region_radius = 1e16; o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3; o->min_lm_diagonal = 1e-6; o->max_lm_diagonal = 1e32;
}

int32_t calico_problem_add_param_block(calico_problem* p, const double* values, int32_t size, int32_t manifold,
                                       int32_t is_constant, int32_t* block_id_out) {
  if (!p) return CALICO_INVALID_ARGUMENT;
  if (size <= 0 || !values) return p->set_error(CALICO_INVALID_ARGUMENT, "bad parameter block");
  if (manifold == CALICO_MANIFOLD_EIGEN_QUATERNION && size != 4)
    return p->set_error(CALICO_INVALID_ARGUMENT, "quaternion manifold needs size 4");
  if (manifold != CALICO_MANIFOLD_EUCLIDEAN && manifold != CALICO_MANIFOLD_EIGEN_QUATERNION)
    return p->set_error(CALICO_INVALID_ARGUMENT, "unknown manifold");
  HBlock b; b.v.assign(values, values + size); b.size = size; b.manifold = manifold; b.constant = is_constant != 0;
  p->blocks.push_back(b);
  p->dirty = true;
  if (block_id_out) *block_id_out = int32_t(p->blocks.size()) - 1;
  return CALICO_OK;
}

int32_t calico_problem_add_param_blocks(calico_problem* p, int32_t n, int32_t size, int32_t manifold, const uint8_t* is_constant,
                                        const double* values, int32_t* block_ids_out) {
  if (!p) return CALICO_INVALID_ARGUMENT;
  if (n < 0 || size <= 0 || (n > 0 && !values)) return p->set_error(CALICO_INVALID_ARGUMENT, "bad parameter blocks");
  if (manifold == CALICO_MANIFOLD_EIGEN_QUATERNION && size != 4)
    return p->set_error(CALICO_INVALID_ARGUMENT, "quaternion manifold needs size 4");
  if (manifold != CALICO_MANIFOLD_EUCLIDEAN && manifold != CALICO_MANIFOLD_EIGEN_QUATERNION)
    return p->set_error(CALICO_INVALID_ARGUMENT, "unknown manifold");
  p->blocks.reserve(p->blocks.size() + size_t(n));
  for (int i = 0; i < n; ++i) {
    HBlock b; b.v.assign(values + size_t(i) * size, values + size_t(i + 1) * size); b.size = size; b.manifold = manifold;
    b.constant = is_constant && is_constant[i] != 0;
    p->blocks.push_back(std::move(b));
    if (block_ids_out) block_ids_out[i] = int32_t(p->blocks.size()) - 1;
  }
  p->dirty = true;
  return CALICO_OK;
}

int32_t calico_get_param_block(calico_problem* p, int32_t id, double* out) {
  if (!p || id < 0 || id >= int(p->blocks.size()) || !out) return p ? p->set_error(CALICO_INVALID_ARGUMENT, "bad block id") : CALICO_INVALID_ARGUMENT;
  std::copy(p->blocks[id].v.begin(), p->blocks[id].v.end(), out);
  return CALICO_OK;
}

int32_t calico_set_param_block(calico_problem* p, int32_t id, const double* v) {
  if (!p || id < 0 || id >= int(p->blocks.size()) || !v) return p ? p->set_error(CALICO_INVALID_ARGUMENT, "bad block id") : CALICO_INVALID_ARGUMENT;
  std::copy(v, v + p->blocks[id].size, p->blocks[id].v.begin());
  if (p->blocks[id].constant || !p->blocks[id].used) p->xc_stale = true;   // the update kernel never rewrites these
  return CALICO_OK;
}

int32_t calico_get_param_blocks(calico_problem* p, int32_t n, const int32_t* ids, double* out) {
  if (!p) return CALICO_INVALID_ARGUMENT;
  if (n < 0 || (n > 0 && (!ids || !out))) return p->set_error(CALICO_INVALID_ARGUMENT, "bad block list");
  for (int32_t i = 0; i < n; ++i)
    if (ids[i] < 0 || ids[i] >= int(p->blocks.size())) return p->set_error(CALICO_INVALID_ARGUMENT, "bad block id");
  for (int32_t i = 0; i < n; ++i) {
    const HBlock& b = p->blocks[size_t(ids[i])];
    std::copy(b.v.begin(), b.v.end(), out);
    out += b.size;
  }
  return CALICO_OK;
}

int32_t calico_set_param_blocks(calico_problem* p, int32_t n, const int32_t* ids, const double* v) {
  if (!p || n < 0 || (n > 0 && (!ids || !v))) return p ? p->set_error(CALICO_INVALID_ARGUMENT, "bad arguments") : CALICO_INVALID_ARGUMENT;
  for (int i = 0; i < n; ++i)
    if (ids[i] < 0 || ids[i] >= int(p->blocks.size())) return p->set_error(CALICO_INVALID_ARGUMENT, "bad block id");
  for (int i = 0; i < n; ++i) {
    HBlock& b = p->blocks[ids[i]];
    std::copy(v, v + b.size, b.v.begin());
    v += b.size;
    if (b.constant || !b.used) p->xc_stale = true;
  }
  return CALICO_OK;
}

int32_t calico_problem_set_spline(calico_problem* p, int32_t order, int32_t n_knots, const double* knots,
                                  const double* basis, const int32_t* ctrl) {
  if (!p) return CALICO_INVALID_ARGUMENT;
  if (order < 2 || order > 16 || n_knots < 2 * order || !knots || !basis || !ctrl)
    return p->set_error(CALICO_INVALID_ARGUMENT, "bad spline");
  const int deg = order - 1;
  for (int i = 0; i < n_knots - order; ++i)
    if (ctrl[i] < 0 || ctrl[i] >= int(p->blocks.size()) || p->blocks[ctrl[i]].size != 6)
      return p->set_error(CALICO_INVALID_ARGUMENT, "control point blocks must be 6-vectors");
  p->order = order;
  p->knots.assign(knots, knots + n_knots);
  p->valid_knots.assign(knots + deg, knots + n_knots - deg);
  const int nseg = int(p->valid_knots.size()) - 1;
  p->basis.assign(basis, basis + size_t(nseg) * order * order);
  p->ctrl.assign(ctrl, ctrl + (n_knots - order));
  p->dirty = true;
  return CALICO_OK;
}

int32_t calico_problem_add_rigid_body(calico_problem* p, int32_t q, int32_t t, int32_t* id_out) {
  if (!p) return CALICO_INVALID_ARGUMENT;
  const int nb = int(p->blocks.size());
  if (q < 0 || q >= nb || t < 0 || t >= nb || p->blocks[q].size != 4 || p->blocks[t].size != 3)
    return p->set_error(CALICO_INVALID_ARGUMENT, "rigid body pose blocks must be a quaternion and a 3-vector");
  p->bodies.push_back({q, t});
  p->dirty = true;
  if (id_out) *id_out = int32_t(p->bodies.size()) - 1;
  return CALICO_OK;
}

int32_t calico_problem_add_sensor(calico_problem* p, int32_t kind, int32_t model, int32_t intr, int32_t q, int32_t t,
                                  int32_t lat, int32_t grav, double sigma, int32_t loss, double loss_scale,
                                  int32_t* id_out) {
  if (!p) return CALICO_INVALID_ARGUMENT;
  if (kind < 0 || kind > 2) return p->set_error(CALICO_INVALID_ARGUMENT, "unknown sensor kind");
  const int K = kind == CALICO_SENSOR_CAMERA ? camera_num_params(model) : imu_num_params(model);
  // camera.cpp:94-97 / gyroscope.cpp:12-15: model not set -> FailedPrecondition
  if (K < 0) return p->set_error(CALICO_FAILED_PRECONDITION, "Cannot add sensor parameters. Sensor model is not yet defined.");
  const int nb = int(p->blocks.size());
  auto ok = [&](int id, int size) { return id >= 0 && id < nb && p->blocks[id].size == size; };
  if (!ok(intr, K)) return p->set_error(CALICO_INVALID_ARGUMENT, "intrinsics block size does not match the model");
  if (!ok(q, 4) || !ok(t, 3) || !ok(lat, 1)) return p->set_error(CALICO_INVALID_ARGUMENT, "bad extrinsics / latency blocks");
  if (kind == CALICO_SENSOR_ACCELEROMETER && !ok(grav, 3)) return p->set_error(CALICO_INVALID_ARGUMENT, "bad gravity block");
  if (loss < 0 || loss > 2) return p->set_error(CALICO_INVALID_ARGUMENT, "unknown loss function");
  HSensor s; s.kind = kind; s.model = model; s.K = K; s.intr = intr; s.q = q; s.t = t; s.lat = lat;
  s.grav = kind == CALICO_SENSOR_ACCELEROMETER ? grav : -1;
  s.sigma = sigma; s.info = sigma > 0.0 ? 1.0 / sigma : 1.0;  // camera_cost_functor.cpp:15 (Q9)
  s.loss = loss; s.loss_scale = loss_scale;
  p->sensors.push_back(s);
  p->dirty = true;
  if (id_out) *id_out = int32_t(p->sensors.size()) - 1;
  return CALICO_OK;
}

static int32_t add_obs(calico_problem* p, int32_t sid, int64_t n, const double* meas, const double* stamps,
                       const int32_t* body, const int32_t* point) {
  if (sid < 0 || sid >= int(p->sensors.size())) return p->set_error(CALICO_INVALID_ARGUMENT, "bad sensor id");
  if (p->order <= 0) return p->set_error(CALICO_FAILED_PRECONDITION, "spline must be set before residuals");
  if (n < 0 || (n > 0 && (!meas || !stamps))) return p->set_error(CALICO_INVALID_ARGUMENT, "bad observation arrays");
  HSensor& s = p->sensors[sid];
  const int dim = s.dim();
  // validate first so a failing call adds nothing
  std::vector<int> segs(static_cast<size_t>(n));
  double last_t = 0.0;
  int last_sg = -2;       // (the blocks of a camera frame share their stamp: one knot search per frame)
  for (int64_t i = 0; i < n; ++i) {
    const int sg = (last_sg != -2 && stamps[i] == last_t) ? last_sg : spline_index(p, stamps[i]);
    last_t = stamps[i]; last_sg = sg;
    if (sg < 0)
      return p->set_error(CALICO_INVALID_ARGUMENT, "measurement stamp is outside the spline's valid knots");
    segs[size_t(i)] = sg;
    if (body) {
      // camera.cpp:126-131
      if (body[i] < 0 || body[i] >= int(p->bodies.size()))
        return p->set_error(CALICO_FAILED_PRECONDITION,
                            "Attempted to create cost function from an observation for a rigidbody that does not exist in the world model.");
      if (point[i] < 0 || point[i] >= int(p->blocks.size()) || p->blocks[point[i]].size != 3)
        return p->set_error(CALICO_INVALID_ARGUMENT, "model point block must be a 3-vector");
    }
  }
  s.seg.insert(s.seg.end(), segs.begin(), segs.end());
  s.stamps.insert(s.stamps.end(), stamps, stamps + n);
  if (body) { s.body.insert(s.body.end(), body, body + n); s.point.insert(s.point.end(), point, point + n); }
  s.meas.insert(s.meas.end(), meas, meas + n * dim);
  s.active.insert(s.active.end(), size_t(n), uint8_t(1));
  s.n_active = -1;
  p->dirty = true;
  return CALICO_OK;
}

int32_t calico_problem_add_camera_residuals(calico_problem* p, int32_t sid, int64_t n, const double* px, const double* st,
                                            const int32_t* body, const int32_t* point) {
  if (!p) return CALICO_INVALID_ARGUMENT;
  if (sid >= 0 && sid < int(p->sensors.size()) && p->sensors[sid].kind != CALICO_SENSOR_CAMERA)
    return p->set_error(CALICO_INVALID_ARGUMENT, "sensor is not a camera");
  if (n > 0 && (!body || !point)) return p->set_error(CALICO_INVALID_ARGUMENT, "bad observation arrays");
  return add_obs(p, sid, n, px, st, body, point);
}

int32_t calico_problem_add_imu_residuals(calico_problem* p, int32_t sid, int64_t n, const double* m, const double* st) {
  if (!p) return CALICO_INVALID_ARGUMENT;
  if (sid >= 0 && sid < int(p->sensors.size()) && p->sensors[sid].kind == CALICO_SENSOR_CAMERA)
    return p->set_error(CALICO_INVALID_ARGUMENT, "sensor is not an IMU sensor");
  return add_obs(p, sid, n, m, st, nullptr, nullptr);
}

int32_t calico_solve(calico_problem* p, const calico_solver_options* opt, calico_summary* sm) {
  if (!p || !opt || !sm) return CALICO_INVALID_ARGUMENT;
  const auto t_start = std::chrono::steady_clock::now();
  // CALICO_SOLVE_TIMING=1: host time of the sections of this call and since the previous call returned (development aid)
  static const bool solve_timing = std::getenv("CALICO_SOLVE_TIMING") != nullptr;
  static std::chrono::steady_clock::time_point t_last_return = t_start;
  double t_mark[6] = {0, 0, 0, 0, 0, 0};
  auto mark = [&](int i) { if (solve_timing) t_mark[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_start).count(); };
  std::memset(sm, 0, sizeof(*sm));
  if (p->world > 1 && !p->has_exchange())
    return p->set_error(CALICO_FAILED_PRECONDITION, "calico_problem_set_shard(world > 1) needs an exchange: calico_comm_init_rccl or calico_problem_set_allreduce");
  int rc = finalize(p);
  if (rc != CALICO_OK) return rc;
  HIP_TRY(p, hipSetDevice(p->device));
  rc = upload_x(p, /*seed=*/false);
  if (rc != CALICO_OK) return rc;
  fill_counts(p, sm);
  p->iterations.clear();
  // event brackets nobody has asked about yet: resolved here once they pile up (a solve returns without draining them)
  if (p->timer.pending.size() > 8192) { HIP_TRY(p, hipStreamSynchronize(p->stream)); p->timer.resolve(); }
  LmOptionsDev o;
  o.max_num_iterations = opt->max_num_iterations; o.max_num_consecutive_invalid_steps = opt->max_num_consecutive_invalid_steps;
  o.function_tolerance = opt->function_tolerance; o.gradient_tolerance = opt->gradient_tolerance;
  o.parameter_tolerance = opt->parameter_tolerance; o.max_radius = opt->max_trust_region_radius;
  o.min_radius = opt->min_trust_region_radius; o.min_relative_decrease = opt->min_relative_decrease;
  o.min_lm_diagonal = opt->min_lm_diagonal; o.max_lm_diagonal = opt->max_lm_diagonal;
  double xn = 0.0;
  for (const BlockDev& b : p->h_blocks) for (int i = 0; i < b.size; ++i) xn += p->h_x[b.amb_off + i] * p->h_x[b.amb_off + i];
  hipStream_t s = p->stream;
  const auto t_loop = std::chrono::steady_clock::now();
  const double* upd_ext = p->use_bcr ? p->d_bupd.p : nullptr;
  const int upd_ext_n = p->use_bcr ? p->bcr_slots : 0;
  // Single rank, speculative evaluation: the host never blocks inside the solve. The control kernel publishes the
  // number of the iteration it has finished with (and post_eval / control the termination flag) in host-mapped
  // memory; the host keeps `depth` iterations enqueued ahead of that and stops when the flag goes up. Compared with
  // batches of `sync_every` iterations and a blocking read-back per batch this takes the read-back gaps out of the
  // stream and leaves at most `depth` iterations of early-exit kernels behind a terminated solve. The stage that
  // terminates the solve writes the results (state, log, parameters) into pinned host memory itself, so the call
  // returns as soon as the flag is up: the early-exit kernels drain while the caller prepares its next call.
  const int stream_depth = [] { const char* e = std::getenv("CALICO_STREAM_DEPTH"); return e ? std::atoi(e) : 2; }();
  // (the progress word carries the iteration count in 20 bits: budgets beyond that take the batched loop)
  const bool streaming = p->speculative && !p->has_exchange() && stream_depth > 0 && p->h_progress != nullptr &&
                         opt->max_num_iterations <= 0xfffff;
  const int log_rows = std::min(kLogCap, std::max(0, opt->max_num_iterations) + 2);
  ResultSink sink = {};
  if (streaming) {
    p->solve_epoch = p->solve_epoch % 2047 + 1;
    sink.state = p->h_state; sink.log = p->h_log; sink.x = p->h_xpin; sink.src_log = p->d_log.p; sink.src_x = p->d_x.p;
    sink.rows = log_rows; sink.n_amb = int(p->h_x.size()); sink.epoch = p->solve_epoch;
  }
  // what a hipEventRecord pair costs around a ~2 us kernel on this stream: lets the caller take the bracket
  // overhead out of the per-launch phase times (phase 5)
  if ((p->timer.mask >> 5) & 1) {
    for (int r = 0; r < 4; ++r) {
      p->timer.begin(5, s);
      launch_init_state(p->d_state.p, opt->initial_trust_region_radius, std::sqrt(xn), s, upd_ext, upd_ext_n);
      p->timer.end(s);
    }
  }
  mark(0);
  launch_begin_solve(p->d_state.p, opt->initial_trust_region_radius, std::sqrt(xn), upd_ext, upd_ext_n, sink, p->d_x.p,
                     p->xc_stale ? p->d_xc.p : nullptr, p->h_xpin, int(p->h_x.size()), s);
  p->xc_stale = false;
  mark(1);
  SolveArgs sa = make_solve_args(p);
  const int n_blocks = int(p->h_blocks.size());
  if (streaming) sa.progress = p->d_progress;
  const int epoch = p->solve_epoch;
  // iteration 0
  rc = enqueue_jacobian_eval(p, nullptr, 0);
  if (rc != CALICO_OK) return rc;
  // The bookkeeping of the first evaluation (initial cost, gradient norms, Jacobi scaling, log row 0) rides in the first
  // linear solve's level-0 launch where the streaming loop and the tree solver run (CALICO_FOLD_FIRST=0: its own launch)
  const bool fold_first = streaming && p->use_bcr && !p->has_exchange() && opt->max_num_iterations > 0 &&
                          [] { const char* e = std::getenv("CALICO_FOLD_FIRST"); return !e || std::atoi(e) != 0; }();
  if (!fold_first) {
    p->timer.begin(4, s);
    launch_post_eval(sa, p->d_x.p, p->d_blocks.p, int(p->h_blocks.size()), o, p->d_log.p, kLogCap, 1, opt->jacobi_scaling, s);
    p->timer.end(s);
  }
  const bool fused_control = [] { const char* e = std::getenv("CALICO_FUSED_CONTROL"); return !e || std::atoi(e) != 0; }();
  // The iteration enqueued ahead of the device is wasted when the one in front of it ends the solve (six early-exit kernels,
  // 40 us at configs[3], in front of the caller's next solve). With the end hint the Jacobian launch of iteration i says, from
  // what the linear solve left, whether iteration i's control stage will end the solve; iteration i + 1 is enqueued on its
  // "go" (progress word 2) -- the evaluation chain is still running then, so the device does not wait -- or, without one,
  // once iteration i has ended without terminating (CALICO_PREDICT_END=0: always one iteration ahead, rounds 2-3).
  const bool predict_end = streaming && fused_control && end_hint_available(p) &&
                           [] { const char* e = std::getenv("CALICO_PREDICT_END"); return !e || std::atoi(e) != 0; }();
  mark(2);
  int dbg_enq = 0, dbg_go = 0, dbg_wait = 0;      // CALICO_SOLVE_TIMING: iterations enqueued, on a go word, behind a finished iteration
  if (streaming) {
    __atomic_store_n(p->h_progress + 2, 0, __ATOMIC_RELEASE);     // (a go word of the same epoch, 2047 solves ago)
    int enq = 0;
    auto t_progress = std::chrono::steady_clock::now();     // when the device last reported a finished iteration
    int last_seen = 0;
    int64_t spins = 0;
    bool budget_spent = false;
    for (;;) {
      bool done = false;
      for (;;) {
        if (__atomic_load_n(p->h_progress + 1, __ATOMIC_ACQUIRE) == epoch) { done = true; break; }
        const int word = __atomic_load_n(p->h_progress, __ATOMIC_ACQUIRE);
        const int seen = (word >> 20) == epoch ? (word & 0xfffff) : 0;    // words of another epoch: early-exit kernels of the previous solve
        if (seen != last_seen) { last_seen = seen; t_progress = std::chrono::steady_clock::now(); spins = 0; }
        const bool room = predict_end
                              ? (seen >= enq || __atomic_load_n(p->h_progress + 2, __ATOMIC_ACQUIRE) == ((epoch << 20) | enq))
                              : enq - seen < stream_depth;
        if (!budget_spent && room) {
          if (solve_timing) { ++dbg_enq; if (seen >= enq) ++dbg_wait; else ++dbg_go; }
          // the device raises the termination word BEFORE the iteration count: having seen the count move, look at the
          // flag once more, or one solve in two enqueues a whole iteration of early-exit kernels for nothing
          if (__atomic_load_n(p->h_progress + 1, __ATOMIC_ACQUIRE) == epoch) done = true;
          break;
        }
        __builtin_ia32_pause();
        if ((++spins & 0xfffff) == 0) {   // a device fault must not leave the host spinning
          const hipError_t qe = hipStreamQuery(s);
          if (qe != hipSuccess && qe != hipErrorNotReady) return p->set_error(CALICO_INTERNAL, hipGetErrorString(qe));
          if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_progress).count() > 600.0) {
            (void)hipStreamSynchronize(s);     // nothing of this solve stays behind on the stream
            return p->set_error(CALICO_INTERNAL, "solve loop: no progress from the device");
          }
        }
      }
      if (done) break;
      if (enq >= std::max(0, opt->max_num_iterations)) {
        // the iteration budget is enqueued: all that can still be due is the bookkeeping of the last step, should it be
        // accepted (it ends the solve by the iteration count) -- one kernel instead of an iteration of early exits
        launch_post_eval(sa, p->d_x.p, p->d_blocks.p, n_blocks, o, p->d_log.p, kLogCap, 0, opt->jacobi_scaling, s);
        budget_spent = true;
        continue;
      }
      p->timer.begin(2, s);
      enqueue_linear_solve(p, sa, o, /*with_post_eval=*/enq > 0 ? 1 : (fold_first ? 2 : 0), opt->jacobi_scaling);
      p->timer.end(s);
      // the control stage rides in the last workgroup of the gather kernel
      ControlTail tail;
      tail.enabled = 1; tail.n_amb = p->n_amb; tail.log_cap = kLogCap; tail.seq = ++enq; tail.o = o; tail.x = p->d_x.p;
      tail.x_cand = p->d_xc.p; tail.log = p->d_log.p; tail.Rbase = p->d_R.p; tail.r_stride = p->r_size;
      tail.progress = p->d_progress; tail.owner_block = p->gather_owner_block;
      rc = enqueue_jacobian_eval(p, p->d_state.p, 0, p->d_xc.p, true, fused_control ? &tail : nullptr, predict_end);
      if (rc != CALICO_OK) return rc;
      if (!fused_control) {
        p->timer.begin(4, s);
        launch_control(p->d_state.p, o, p->d_R2.p, p->d_x.p, p->d_xc.p, p->n_amb, p->d_log.p, kLogCap, nullptr, 0, p->d_R.p,
                       p->r_size, s, false, p->d_progress, enq);
        p->timer.end(s);
      }
    }
  } else {
    rc = read_state(p);
    if (rc != CALICO_OK) return rc;
  }
  // One LM iteration = linear solve + candidate cost + control (+ Jacobian evaluation if the
  // step was accepted). `sync_every` complete iterations are enqueued per host round trip, every kernel deciding on
  // the device whether it still has work. With several ranks this needs the speculative evaluation: the candidate is
  // then always evaluated into reduce buffer 1, so the collective gets a fixed address and runs in every enqueued
  // iteration on every rank (re-reducing a stale buffer 1 behind a terminated solve is harmless), and an accepted
  // candidate is committed by a copy (commit_kernel) instead of the pointer swap. Without the speculative evaluation
  // a multi-rank run needs the host between the phases (the all-reduce must not run when the evaluation was skipped).
  const bool spec = p->speculative;
  const bool multi = p->has_exchange();
  const bool multi_async_ok = [] { const char* e = std::getenv("CALICO_MULTIRANK_ASYNC"); return !e || std::atoi(e) != 0; }();
  const bool async = !multi || (spec && multi_async_ok);
  const int batch = async ? std::max(1, opt->sync_every) : 1;
  int batch_now = batch;
  while (!streaming && !p->h_state->terminated) {
    // The iterations enqueued behind a terminated solve are wasted (with several ranks each still carries a real
    // all-reduce), so the batch shrinks when the cost changes of the last two successful steps predict convergence
    // by the function tolerance within fewer iterations: linear convergence, ratio r -> log(tol / change) / log(r).
    batch_now = batch;
    {
      const LmState& hs = *p->h_state;
      const double tol = opt->function_tolerance * hs.x_cost;
      if (batch > 1 && hs.last_cost_change > 0.0 && hs.prev_cost_change > hs.last_cost_change && tol > 0.0) {
        const double r = hs.last_cost_change / hs.prev_cost_change;
        const double left = hs.last_cost_change <= tol ? 0.0 : std::ceil(std::log(tol / hs.last_cost_change) / std::log(r));
        batch_now = int(std::max(1.0, std::min(double(batch), left + 1.0)));
      }
    }
    for (int b = 0; b < batch_now; ++b) {
      // (speculative, single rank) the bookkeeping of the step accepted in the previous iteration of this batch rides
      // in the prepare kernel of this one; the last iteration of a batch gets a stand-alone post_eval below
      const bool ride = spec && async && b > 0;
      p->timer.begin(2, s);
      enqueue_linear_solve(p, sa, o, ride, opt->jacobi_scaling);
      p->timer.end(s);
      if (spec) {
        // Speculative evaluation: cost AND Jacobian at the candidate point in one pass, into the reduce buffer that
        // does not hold R(x). Its first two entries are the candidate's [cost, invalid]; when the step is accepted the
        // control kernel swaps the buffers and the next linear solve starts at once -- no separate cost-only pass,
        // and with several ranks a single all-reduce per iteration. A rejected step wastes the Jacobian work.
        rc = enqueue_jacobian_eval(p, p->d_state.p, 0, p->d_xc.p, true);
        if (rc != CALICO_OK) return rc;
        p->timer.begin(4, s);
        launch_control(p->d_state.p, o, p->d_R2.p, p->d_x.p, p->d_xc.p, p->n_amb, p->d_log.p, kLogCap, nullptr, 0, p->d_R.p,
                       p->r_size, s, /*commit_by_copy=*/multi && async);
        if (!async || b == batch_now - 1)
          launch_post_eval(sa, p->d_x.p, p->d_blocks.p, n_blocks, o, p->d_log.p, kLogCap, 0, opt->jacobi_scaling, s);
        p->timer.end(s);
        if (!async) {
          rc = read_state(p);
          if (rc != CALICO_OK) return rc;
          if (p->h_state->terminated) break;
        }
        continue;
      }
      p->timer.begin(3, s);
      {
        EvalArgs ea = make_eval_args(p, p->d_xc.p, 1, false);
        ea.st = p->d_state.p;
        launch_eval(ea, false, s);
      }
      const bool fuse_cost = !p->has_exchange();    // single rank: the cost sum rides in the control kernel
      if (!fuse_cost) launch_cost_reduce(p->d_partials.p + p->partial_doubles, p->n_items, p->d_R2.p, p->d_state.p, s);
      p->timer.end(s);
      rc = do_allreduce(p, p->d_R2.p, 2);
      if (rc != CALICO_OK) return rc;
      p->timer.begin(4, s);
      launch_control(p->d_state.p, o, p->d_R2.p, p->d_x.p, p->d_xc.p, p->n_amb, p->d_log.p, kLogCap,
                     fuse_cost ? p->d_partials.p + p->partial_doubles : nullptr, p->n_items, nullptr, 0, s);
      p->timer.end(s);
      if (!async) {
        rc = read_state(p);
        if (rc != CALICO_OK) return rc;
        if (p->h_state->terminated || !p->h_state->need_jacobian) continue;
      }
      rc = enqueue_jacobian_eval(p, p->d_state.p, async ? 1 : 0);
      if (rc != CALICO_OK) return rc;
      p->timer.begin(4, s);
      launch_post_eval(sa, p->d_x.p, p->d_blocks.p, n_blocks, o, p->d_log.p, kLogCap, 0, opt->jacobi_scaling, s);
      p->timer.end(s);
    }
    rc = read_state(p);
    if (rc != CALICO_OK) return rc;
  }
  // results: final state, iteration log and parameters come back in one go (pinned buffers, one synchronisation)
  // one small kernel writes them into the pinned host buffers (three DMA copies cost ~13 us of stream time each). R(x)
  // may sit in either reduce buffer afterwards: nobody reads it (every entry point that needs it evaluates first).
  // (streaming loop: the terminating stage has written them already, and nothing is waited for; event brackets of the
  //  phase timer are resolved when somebody asks for the times)
  mark(3);
  if (!streaming) launch_publish_results(p->d_state.p, p->d_log.p, log_rows, p->d_x.p, int(p->h_x.size()), p->h_state, p->h_log, p->h_xpin, s);
  if (!streaming) HIP_TRY(p, hipStreamSynchronize(s));
  sm->num_jacobian_evaluations = p->h_state->n_jac_evals;
  sm->num_cost_evaluations = p->h_state->n_cost_evals;
  const double t_solve = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_loop).count();
  const LmState st = *p->h_state;
  std::copy(p->h_xpin, p->h_xpin + p->h_x.size(), p->h_x.begin());
  for (HBlock& b : p->blocks) std::copy(p->h_x.begin() + b.amb_off, p->h_x.begin() + b.amb_off + b.size, b.v.begin());
  const std::vector<IterLog> log(p->h_log, p->h_log + std::max(0, std::min(st.n_log, log_rows)));
  for (const IterLog& r : log) {
    calico_iteration it;
    it.iteration = r.iteration; it.step_is_valid = r.step_is_valid; it.step_is_successful = r.step_is_successful; it.reserved = 0;
    it.cost = r.cost; it.cost_change = r.cost_change; it.gradient_max_norm = r.gradient_max_norm; it.step_norm = r.step_norm;
    it.relative_decrease = r.relative_decrease; it.trust_region_radius = r.trust_region_radius;
    p->iterations.push_back(it);
    if (opt->minimizer_progress_to_stdout) {
      if (r.iteration == 0) std::printf("iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius\n");
      std::printf("%4d % 8e   % 3.2e   % 3.2e  % 3.2e  % 3.2e % 3.2e\n", r.iteration, r.cost, r.cost_change, r.gradient_max_norm,
                  r.step_norm, r.relative_decrease, r.trust_region_radius);
    }
  }
  if (p->d_wave_log.p && p->d_wave_log.n > 1) {   // development aid: the last Jacobian launch of the solve, workgroup by workgroup
    std::vector<unsigned long long> wl(p->d_wave_log.n);
    HIP_TRY(p, hipMemcpy(wl.data(), p->d_wave_log.p, wl.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (size_t i = 0; i + 1 < wl.size(); i += 2)
      std::fprintf(stderr, "WAVE %zu %s t0 %llu t1 %llu\n", i / 2, int(i / 2) < ((p->n_jac_items + 1) & ~1) ? "item" : "frame", wl[i], wl[i + 1]);
  }
  sm->termination_type = st.termination_type;
  sm->num_successful_steps = st.num_successful; sm->num_unsuccessful_steps = st.num_unsuccessful;
  sm->num_iterations = st.last_logged_iteration;      // Summary::iterations.size() - 1; not read from the log buffer, which is capped at kLogCap rows
  sm->initial_cost = st.initial_cost;
  sm->final_cost = st.termination_type == CALICO_FAILURE ? 0.0 : std::min(st.initial_cost, st.min_cost);
  std::snprintf(sm->message, sizeof(sm->message), "%s", reason_message(st.termination_reason));
  sm->solve_time_in_seconds = t_solve;
  sm->total_time_in_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
  if (solve_timing) {
    mark(4);
    std::fprintf(stderr, "solve host us: since last return %.1f | prep %.1f | begin launch %.1f | first evaluation enqueued %.1f | loop %.1f | results %.1f"
                 " | iterations: device %d, enqueued %d (ahead of the device %d, behind a finished iteration %d), reason %d\n",
                 std::chrono::duration<double, std::micro>(t_start - t_last_return).count(), t_mark[0], t_mark[1] - t_mark[0],
                 t_mark[2] - t_mark[1], t_mark[3] - t_mark[2], t_mark[4] - t_mark[3], st.iteration, dbg_enq, dbg_go, dbg_wait,
                 st.termination_reason);
    t_last_return = std::chrono::steady_clock::now();
  }
  return CALICO_OK;
}

int32_t calico_debug_lm_control_replay(int32_t device, int32_t n, const double* rho, const int32_t* infinite,
                                       const calico_solver_options* opt, double* radius_out, int32_t* accepted_out,
                                       double* cost_column_out) {
  if (n <= 0 || n > kLogCap - 2 || !rho || !infinite || !opt || !radius_out || !accepted_out || !cost_column_out)
    return CALICO_INVALID_ARGUMENT;
  if (hipSetDevice(device) != hipSuccess) return CALICO_INTERNAL;
  DevBuf<double> d_rho, d_R2, d_rad, d_cost;
  DevBuf<int> d_inf, d_acc;
  DevBuf<LmState> d_st;
  DevBuf<IterLog> d_log;
  std::vector<double> h_rho(rho, rho + n);
  std::vector<int> h_inf(infinite, infinite + n);
  if (d_rho.upload(h_rho, nullptr) != hipSuccess || d_inf.upload(h_inf, nullptr) != hipSuccess || d_R2.alloc(2) != hipSuccess ||
      d_rad.alloc(size_t(n)) != hipSuccess || d_cost.alloc(size_t(n)) != hipSuccess || d_acc.alloc(size_t(n)) != hipSuccess ||
      d_st.alloc(1) != hipSuccess || d_log.alloc(kLogCap) != hipSuccess)
    return CALICO_INTERNAL;
  LmOptionsDev o;
  o.max_num_iterations = 1 << 30; o.max_num_consecutive_invalid_steps = opt->max_num_consecutive_invalid_steps;
  o.function_tolerance = 0.0; o.gradient_tolerance = 0.0; o.parameter_tolerance = 0.0;     // the replay never converges
  o.max_radius = opt->max_trust_region_radius; o.min_radius = opt->min_trust_region_radius;
  o.min_relative_decrease = opt->min_relative_decrease; o.min_lm_diagonal = opt->min_lm_diagonal; o.max_lm_diagonal = opt->max_lm_diagonal;
  launch_init_state(d_st.p, opt->initial_trust_region_radius, 1.0, nullptr);
  launch_debug_control_replay(d_st.p, o, d_rho.p, d_inf.p, n, d_R2.p, d_rad.p, d_acc.p, d_cost.p, d_log.p, kLogCap, nullptr);
  if (hipMemcpy(radius_out, d_rad.p, size_t(n) * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(accepted_out, d_acc.p, size_t(n) * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(cost_column_out, d_cost.p, size_t(n) * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
    return CALICO_INTERNAL;
  return CALICO_OK;
}

int32_t calico_debug_roll_table(int32_t spline_order, int32_t lane, uint32_t* out48) {
  if (!out48 || spline_order < 1 || spline_order > 6 || lane < 0 || lane > 63) return CALICO_INVALID_ARGUMENT;
  cal::roll_table_row(spline_order, lane, out48);
  return CALICO_OK;
}

int32_t calico_debug_plan_info(calico_problem* p, int32_t* out, int32_t n) {
  if (!p || !out || n < 0 || n > 9) return CALICO_INVALID_ARGUMENT;
  int rc = finalize(p);
  if (rc != CALICO_OK) return rc;
  int max_frames = 0, max_items = 0, run = 0, prev_layout = -1, prev_seg = -1;
  for (const CellDev& c : p->h_cells) if (c.prim_off >= 0) max_frames = std::max(max_frames, c.frame_count);   // (camera cells; < 0: IMU cells)
  for (const ItemDev& it : p->h_jac_items) {
    run = (it.layout == prev_layout && it.seg == prev_seg) ? run + 1 : 1;
    prev_layout = it.layout; prev_seg = it.seg;
    max_items = std::max(max_items, run);
  }
  const int v[9] = {p->fuse_expand ? 1 : 0, p->n_fitems, p->n_jac_items, int(p->h_cells.size()), max_frames, max_items,
                    p->use_bcr ? 1 : 0, p->m, p->bcr_all_active ? 1 : 0};
  for (int i = 0; i < n; ++i) out[i] = v[i];
  return CALICO_OK;
}

int32_t calico_get_iterations(calico_problem* p, calico_iteration* out, int32_t max_rows, int32_t* n_out) {
  if (!p || !out || !n_out) return CALICO_INVALID_ARGUMENT;
  const int n = std::min<int>(max_rows, int(p->iterations.size()));
  for (int i = 0; i < n; ++i) out[i] = p->iterations[size_t(i)];
  *n_out = n;
  return CALICO_OK;
}

static int32_t residuals_or_prediction(calico_problem* p, int32_t sid, double* out, uint8_t* valid, bool predict) {
  if (!p) return CALICO_INVALID_ARGUMENT;
  if (sid < 0 || sid >= int(p->sensors.size())) return p->set_error(CALICO_INVALID_ARGUMENT, "bad sensor id");
  if (p->sensors[size_t(sid)].n() == 0) return CALICO_OK;   // a sensor without measurements has nothing to report
  if (!out) return p->set_error(CALICO_INVALID_ARGUMENT, "null output buffer");
  int rc = finalize(p);
  if (rc != CALICO_OK) return rc;
  HIP_TRY(p, hipSetDevice(p->device));
  // one evaluation and one download serve every sensor as long as no parameter value, measurement or tag has changed
  calico_problem::ResCache& rc_ = p->res_cache;
  std::vector<double> xnow(size_t(p->n_amb), 0.0);
  for (const HBlock& b : p->blocks) std::copy(b.v.begin(), b.v.end(), xnow.begin() + b.amb_off);
  if (!(rc_.valid && rc_.predict == predict && !p->active_dirty && rc_.x == xnow)) {
    rc_.valid = false;
    rc = upload_x(p);
    if (rc != CALICO_OK) return rc;
    {
      EvalArgs ea = make_eval_args(p, p->d_x.p, 0, true);
      ea.items = p->d_items_all.p; ea.n_items = p->n_items_all;  // every rank re-evaluates all blocks here
      ea.project = predict ? 1 : 0;
      launch_eval(ea, false, p->stream);
    }
    rc_.r.resize(size_t(p->n_obs) * 3);
    rc_.v.resize(size_t(p->n_obs));
    HIP_TRY(p, hipMemcpyAsync(rc_.r.data(), p->d_res.p, rc_.r.size() * sizeof(double), hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(p, hipMemcpyAsync(rc_.v.data(), p->d_valid.p, rc_.v.size(), hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(p, hipStreamSynchronize(p->stream));
    rc_.x.swap(xnow); rc_.predict = predict; rc_.valid = true;
  }
  const std::vector<double>& r = rc_.r;
  const std::vector<uint8_t>& v = rc_.v;
  const HSensor& s = p->sensors[sid];
  const int dim = s.dim();
  bool all = true;
  for (int64_t i = 0; i < s.n(); ++i) {
    const int64_t q = s.sorted_pos[size_t(i)];
    for (int c = 0; c < dim; ++c) out[i * dim + c] = v[size_t(q)] ? r[size_t(q) * 3 + c] : 0.0;
    if (valid) valid[i] = v[size_t(q)];
    if (!v[size_t(q)] && s.active[size_t(i)]) all = false;     // tagged outliers have no residual and are no failure
  }
  // camera.cpp:73-76: a failing block makes UpdateResiduals return kInternal; Project just skips such points
  // (camera.cpp:172-174), here they come back with valid = 0
  if (predict) return CALICO_OK;
  return all ? CALICO_OK : p->set_error(CALICO_INTERNAL, "Failed to update residual");
}

int32_t calico_get_residuals(calico_problem* p, int32_t sid, double* out, uint8_t* valid) {
  return residuals_or_prediction(p, sid, out, valid, false);
}

int32_t calico_project(calico_problem* p, int32_t sid, double* out, uint8_t* valid) {
  return residuals_or_prediction(p, sid, out, valid, true);
}

int32_t calico_get_inlier_mask(calico_problem* p, int32_t sid, double threshold, uint8_t* mask) {
  if (!p) return CALICO_INVALID_ARGUMENT;
  if (sid < 0 || sid >= int(p->sensors.size()) || !mask) return p->set_error(CALICO_INVALID_ARGUMENT, "bad sensor id");
  int rc = finalize(p);
  if (rc != CALICO_OK) return rc;
  HIP_TRY(p, hipSetDevice(p->device));
  rc = upload_x(p);
  if (rc != CALICO_OK) return rc;
  const HSensor& s = p->sensors[sid];
  if (s.n() == 0) return CALICO_OK;
  {
    EvalArgs ea = make_eval_args(p, p->d_x.p, 0, true);   // residuals without the loss function (camera.cpp:70-80)
    ea.items = p->d_items_all.p; ea.n_items = p->n_items_all;
    launch_eval(ea, false, p->stream);
  }
  // the test runs on the device; one byte per observation comes back (a block that failed to evaluate, or one tagged
  // as an outlier, is no inlier)
  launch_inlier_mask(p->d_res.p, p->d_valid.p, p->d_active.p, int(s.sorted_begin), int(s.sorted_end), s.dim(), threshold, p->stream);
  const int64_t nrange = std::max<int64_t>(0, s.sorted_end - s.sorted_begin);
  std::vector<uint8_t> m(size_t(std::max<int64_t>(nrange, 1)));
  if (nrange > 0) HIP_TRY(p, hipMemcpyAsync(m.data(), p->d_valid.p + s.sorted_begin, size_t(nrange), hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(p, hipStreamSynchronize(p->stream));
  for (int64_t i = 0; i < s.n(); ++i) mask[i] = m[size_t(s.sorted_pos[size_t(i)] - s.sorted_begin)];
  return CALICO_OK;
}

int32_t calico_problem_set_outlier_mask(calico_problem* p, int32_t sid, const uint8_t* is_outlier) {
  if (!p) return CALICO_INVALID_ARGUMENT;
  if (sid < 0 || sid >= int(p->sensors.size())) return p->set_error(CALICO_INVALID_ARGUMENT, "bad sensor id");
  HSensor& s = p->sensors[sid];
  for (int64_t i = 0; i < s.n(); ++i) s.active[size_t(i)] = (is_outlier && is_outlier[i]) ? 0 : 1;
  s.n_active = -1;
  p->active_dirty = true;
  p->res_cache.valid = false;
  return CALICO_OK;
}

int32_t calico_mark_outliers(calico_problem* p, int32_t sid, double threshold, int64_t* n_marked) {
  if (!p) return CALICO_INVALID_ARGUMENT;
  if (sid < 0 || sid >= int(p->sensors.size())) return p->set_error(CALICO_INVALID_ARGUMENT, "bad sensor id");
  int rc = finalize(p);
  if (rc != CALICO_OK) return rc;
  HIP_TRY(p, hipSetDevice(p->device));
  rc = upload_x(p);
  if (rc != CALICO_OK) return rc;
  HSensor& s = p->sensors[sid];
  {
    EvalArgs ea = make_eval_args(p, p->d_x.p, 0, true);   // residuals without the loss function (camera.cpp:70-80)
    ea.items = p->d_items_all.p; ea.n_items = p->n_items_all;
    launch_eval(ea, false, p->stream);
  }
  HIP_TRY(p, hipMemsetAsync(p->d_counter.p, 0, sizeof(int), p->stream));
  launch_mark_outliers(p->d_res.p, p->d_valid.p, p->d_active.p, int(s.sorted_begin), int(s.sorted_end), s.dim(), threshold,
                       p->d_counter.p, p->stream);
  // mirror the tags on the host (they decide counts and survive a re-finalisation)
  const int64_t nrange = std::max<int64_t>(0, s.sorted_end - s.sorted_begin);
  std::vector<uint8_t> act(size_t(std::max<int64_t>(nrange, 1)));
  int marked = 0;
  if (nrange > 0) HIP_TRY(p, hipMemcpyAsync(act.data(), p->d_active.p + s.sorted_begin, size_t(nrange), hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(p, hipMemcpyAsync(&marked, p->d_counter.p, sizeof(int), hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(p, hipStreamSynchronize(p->stream));
  for (int64_t i = 0; i < s.n(); ++i) s.active[size_t(i)] = act[size_t(s.sorted_pos[size_t(i)] - s.sorted_begin)];
  s.n_active = -1;
  p->res_cache.valid = false;
  if (marked > 0) p->any_tagged = true;
  if (n_marked) *n_marked = marked;
  return CALICO_OK;
}

int32_t calico_residual_heatmap(calico_problem* p, int32_t sid, int32_t image_width, int32_t image_height, int32_t num_rows,
                                int32_t num_cols, double* rmse_out, int64_t* count_out) {
  if (!p) return CALICO_INVALID_ARGUMENT;
  if (sid < 0 || sid >= int(p->sensors.size())) return p->set_error(CALICO_INVALID_ARGUMENT, "bad sensor id");
  if (p->sensors[size_t(sid)].kind != CALICO_SENSOR_CAMERA) return p->set_error(CALICO_INVALID_ARGUMENT, "not a camera");
  if (image_width <= 0 || image_height <= 0 || num_rows <= 0 || num_cols <= 0 || !rmse_out || !count_out)
    return p->set_error(CALICO_INVALID_ARGUMENT, "bad heat-map dimensions");
  int rc = finalize(p);
  if (rc != CALICO_OK) return rc;
  HIP_TRY(p, hipSetDevice(p->device));
  rc = upload_x(p);
  if (rc != CALICO_OK) return rc;
  const HSensor& s = p->sensors[size_t(sid)];
  {
    EvalArgs ea = make_eval_args(p, p->d_x.p, 0, true);   // residuals without the loss function (camera.cpp:70-80)
    ea.items = p->d_items_all.p; ea.n_items = p->n_items_all;
    launch_eval(ea, false, p->stream);
  }
  const size_t nb = size_t(num_rows) * num_cols;
  DevBuf<double> d_rmse; DevBuf<long long> d_cnt;
  HIP_TRY(p, d_rmse.alloc(nb)); HIP_TRY(p, d_cnt.alloc(nb));
  launch_residual_heatmap(p->d_res.p, p->d_valid.p, p->d_active.p, p->d_m0.p, p->d_m1.p, int(std::min(s.sorted_begin, s.sorted_end)),
                          int(s.sorted_end), image_width, image_height, num_rows, num_cols, d_rmse.p, d_cnt.p, p->stream);
  std::vector<long long> cnt(nb);
  HIP_TRY(p, hipMemcpyAsync(rmse_out, d_rmse.p, nb * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(p, hipMemcpyAsync(cnt.data(), d_cnt.p, nb * sizeof(long long), hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(p, hipStreamSynchronize(p->stream));
  for (size_t i = 0; i < nb; ++i) count_out[i] = int64_t(cnt[i]);
  return CALICO_OK;
}

int32_t calico_num_effective_parameters(calico_problem* p, int32_t* n_out) {
  if (!p || !n_out) return CALICO_INVALID_ARGUMENT;
  const int rc = finalize(p);
  if (rc != CALICO_OK) return rc;
  *n_out = p->n_eff;
  return CALICO_OK;
}

int32_t calico_evaluate(calico_problem* p, double* cost, double* gradient, double* jtj) {
  if (!p) return CALICO_INVALID_ARGUMENT;
  if (p->world > 1 && !p->has_exchange())
    return p->set_error(CALICO_FAILED_PRECONDITION, "calico_problem_set_shard(world > 1) needs an exchange: calico_comm_init_rccl or calico_problem_set_allreduce");
  int rc = finalize(p);
  if (rc != CALICO_OK) return rc;
  HIP_TRY(p, hipSetDevice(p->device));
  rc = upload_x(p);
  if (rc != CALICO_OK) return rc;
  rc = enqueue_jacobian_eval(p, nullptr, 0);
  if (rc != CALICO_OK) return rc;
  SolveArgs sa = make_solve_args(p);
  std::vector<double> R(sa.r_size());
  HIP_TRY(p, hipMemcpyAsync(R.data(), p->d_R.p, R.size() * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(p, hipStreamSynchronize(p->stream));
  p->timer.resolve();
  if (R[1] > 0.0) return p->set_error(CALICO_INTERNAL, "residual evaluation failed");
  if (cost) *cost = R[0];
  const int n = p->n_eff, NS = 6 * p->n_cp, m = p->m, k = p->order;
  auto H = [&](int ta, int tb) -> double {  // solver tangent indices
    if (ta > tb) std::swap(ta, tb);
    if (tb < NS) {
      const int a = ta / 6, b = tb / 6;
      if (b - a >= k) return 0.0;
      return R[sa.off_B() + (size_t(a) * k + (b - a)) * 36 + (ta % 6) * 6 + (tb % 6)];
    }
    if (ta < NS) return R[sa.off_E() + size_t(ta) * m + (tb - NS)];
    return R[sa.off_C() + size_t(ta - NS) * m + (tb - NS)];
  };
  if (gradient) for (int i = 0; i < n; ++i) gradient[i] = R[sa.off_g() + p->eff_to_tan[size_t(i)]];
  if (jtj)
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) jtj[size_t(i) * n + j] = H(p->eff_to_tan[size_t(i)], p->eff_to_tan[size_t(j)]);
  return CALICO_OK;
}

int32_t calico_problem_set_allreduce(calico_problem* p, calico_allreduce_fn fn, void* ctx) {
  if (!p) return CALICO_INVALID_ARGUMENT;
  p->allreduce = fn; p->allreduce_ctx = ctx;
  return CALICO_OK;
}

int32_t calico_problem_finalize(calico_problem* p) {
  if (!p) return CALICO_INVALID_ARGUMENT;
  const int rc = finalize(p);
  if (rc != CALICO_OK) return rc;
  return upload_x(p) == CALICO_OK && hipStreamSynchronize(p->stream) == hipSuccess ? CALICO_OK : CALICO_INTERNAL;
}

int32_t calico_comm_get_unique_id(uint8_t* id_out) {
  if (!id_out) return CALICO_INVALID_ARGUMENT;
  static_assert(sizeof(ncclUniqueId) == CALICO_COMM_ID_BYTES, "RCCL unique id size");
  ncclUniqueId id;
  if (!rccl().ok() || rccl().GetUniqueId(&id) != ncclSuccess) return CALICO_INTERNAL;
  std::memcpy(id_out, &id, sizeof(id));
  return CALICO_OK;
}

// How many distinct librccl images the process holds. More than one means this library loaded its private copy BEFORE the
// application brought its own (PyTorch imported after the first communicator call): both copies work, but two RCCLs in one
// process have ended in a double free at exit. The load order that avoids it -- the application's RCCL first -- is documented
// in include/calico_hip.h; here the situation is detected and said out loud, once.
static int rccl_images_loaded() {
  std::vector<std::string> seen;
  dl_iterate_phdr([](struct dl_phdr_info* info, size_t, void* data) {
    auto* v = static_cast<std::vector<std::string>*>(data);
    const std::string n = info->dlpi_name ? info->dlpi_name : "";
    const size_t slash = n.rfind('/');
    if (n.compare(slash == std::string::npos ? 0 : slash + 1, 7, "librccl") == 0 && std::find(v->begin(), v->end(), n) == v->end()) v->push_back(n);
    return 0;
  }, &seen);
  return int(seen.size());
}

int32_t calico_comm_init_rccl(calico_problem* p, const uint8_t* id, int32_t rank, int32_t world_size) {
  if (!p || !id) return CALICO_INVALID_ARGUMENT;
  if (world_size < 1 || rank < 0 || rank >= world_size) return p->set_error(CALICO_INVALID_ARGUMENT, "bad rank / world size");
  if (!rccl().ok()) return p->set_error(CALICO_INTERNAL, rccl().error);
  if (rccl_images_loaded() > 1) {
    static std::atomic<bool> said{false};
    if (!said.exchange(true))
      std::fprintf(stderr, "[calico] warning: two librccl images are loaded in this process (this library loaded its own before the "
                           "application's -- e.g. torch was imported after the first calico_comm_* call). Load the application's RCCL "
                           "first, or point CALICO_RCCL_LIB at the same file.\n");
  }
  HIP_TRY(p, hipSetDevice(p->device));
  if (p->comm) { if (p->stream) (void)hipStreamSynchronize(p->stream); (void)rccl().CommDestroy(p->comm); p->comm = nullptr; }
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  const ncclResult_t r = rccl().CommInitRank(&p->comm, world_size, uid, rank);
  if (r != ncclSuccess) { p->comm = nullptr; return p->set_error(CALICO_INTERNAL, std::string("ncclCommInitRank: ") + rccl().GetErrorString(r)); }
  p->rank = rank; p->world = world_size; p->dirty = true;
  return CALICO_OK;
}

int32_t calico_comm_info(calico_problem* p, int32_t* rank_out, int32_t* world_out, int64_t* local_blocks_out, int64_t* total_blocks_out) {
  if (!p) return CALICO_INVALID_ARGUMENT;
  int world = p->world;
  if (p->comm) {      // what the communicator itself says, not what the caller asked for
    if (rccl().CommCount(p->comm, &world) != ncclSuccess) return p->set_error(CALICO_INTERNAL, "ncclCommCount failed");
  }
  const int rc = finalize(p);
  if (rc != CALICO_OK) return rc;
  int64_t local = 0, total = 0;
  for (const ItemDev& it : p->h_items) local += it.obs_count;
  for (const ItemDev& it : p->h_items_all) total += it.obs_count;
  if (rank_out) *rank_out = p->rank;
  if (world_out) *world_out = world;
  if (local_blocks_out) *local_blocks_out = local;
  if (total_blocks_out) *total_blocks_out = total;
  return CALICO_OK;
}

int32_t calico_problem_set_shard(calico_problem* p, int32_t rank, int32_t world_size) {
  if (!p) return CALICO_INVALID_ARGUMENT;
  if (world_size < 1 || rank < 0 || rank >= world_size) return p->set_error(CALICO_INVALID_ARGUMENT, "bad rank / world size");
  p->rank = rank; p->world = world_size; p->dirty = true;
  return CALICO_OK;
}

int32_t calico_problem_set_stream(calico_problem* p, void* stream) {
  if (!p) return CALICO_INVALID_ARGUMENT;
  if (p->stream) (void)hipStreamSynchronize(p->stream);
  if (p->own_stream && p->stream) (void)hipStreamDestroy(p->stream);
  p->stream = reinterpret_cast<hipStream_t>(stream);
  p->own_stream = false;
  return CALICO_OK;
}

int32_t calico_set_phase_timing(calico_problem* p, int32_t mask) {
  if (!p) return CALICO_INVALID_ARGUMENT;
  // the phase times accumulate from this call on, over as many solves as follow
  if (!p->timer.pending.empty()) { (void)hipSetDevice(p->device); (void)hipStreamSynchronize(p->stream); p->timer.resolve(); }
  p->timer.reset();
  p->timer.mask = mask & 0xff;
  p->timer.every = std::max(1, (mask >> 8) & 0xff);
  return CALICO_OK;
}

int32_t calico_get_phase_time(calico_problem* p, int32_t phase, double* ms, int64_t* launches) {
  const bool working = (phase & 0x100) != 0;
  phase &= 0xff;
  if (!p || phase < 0 || phase >= kNumPhases) return CALICO_INVALID_ARGUMENT;
  if (!p->timer.pending.empty()) {   // brackets still on the stream (a solve returns without waiting for it to drain)
    HIP_TRY(p, hipSetDevice(p->device));
    HIP_TRY(p, hipStreamSynchronize(p->stream));
    p->timer.resolve();
  }
  if (ms) *ms = working ? p->timer.ms_working[phase] : p->timer.ms[phase];
  if (launches) *launches = working ? p->timer.count_working[phase] : p->timer.count[phase];
  return CALICO_OK;
}

}  // extern "C"
