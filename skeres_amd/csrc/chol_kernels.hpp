// Dense fp64 Cholesky on gfx950 matrix cores — host entry points (chol_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "dev_knobs.hpp"

namespace sk {

// HIP-event timing of named kernel launches on the stream they run on.
// Off by default; bench.py switches it on to measure the dominant kernel's
// average launch duration live (roofline.achieved).
class KernelTimer {
 public:
  struct Stat { double seconds = 0.0; int launches = 0; };
  void enable(bool on) { enabled_ = on; }
  // Time only launches of this name (empty: all).  Events are barrier packets: timing every
  // small kernel of the panel chain would slow the chain it measures.
  void only(const std::string& name) { only_ = name; }
  bool enabled() const { return enabled_; }
  bool times_all() const { return enabled_ && only_.empty(); }
  void begin(const char* name, hipStream_t s) {
    if (!enabled_) return;
    if (!only_.empty() && only_ != name) { skip_ = true; return; }
    skip_ = false;
    hipEvent_t e = get();
    hipEventRecord(e, s);
    pending_.push_back({name, e, nullptr});
  }
  // The two events of one launch, to ride on its own dispatch (hipExtLaunchKernelGGL start / stop): the kernel's begin
  // and end timestamps with no record packets around it in the queue (which delay whatever follows by 3-6 us each —
  // on the panel chain that is the measurement changing the thing it measures).  Null when this name is not timed.
  void pair(const char* name, hipEvent_t* start, hipEvent_t* stop) {
    *start = *stop = nullptr;
    if (!enabled_ || (!only_.empty() && only_ != name)) return;
    *start = get(); *stop = get();
    pending_.push_back({name, *start, *stop});
  }
  void end(const char* name, hipStream_t s) {
    if (!enabled_ || skip_ || pending_.empty()) return;
    hipEvent_t e = get();
    hipEventRecord(e, s);
    pending_.back().stop = e;
    (void)name;
  }
  // Resolve everything recorded so far (blocks until the events have completed).
  void collect() {
    for (auto& p : pending_) {
      if (!p.stop) { free_.push_back(p.start); continue; }
      hipEventSynchronize(p.stop);
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
        Stat& st = stats_[p.name];
        st.seconds += 1e-3 * ms;
        st.launches += 1;
      }
      free_.push_back(p.start);
      free_.push_back(p.stop);
    }
    pending_.clear();
  }
  Stat get_stat(const std::string& name) {
    collect();
    auto it = stats_.find(name);
    return it == stats_.end() ? Stat() : it->second;
  }
  void reset() { collect(); stats_.clear(); }
  ~KernelTimer() {
    collect();
    for (hipEvent_t e : free_) hipEventDestroy(e);
  }

 private:
  struct Pending { std::string name; hipEvent_t start, stop; };
  hipEvent_t get() {
    if (!free_.empty()) { hipEvent_t e = free_.back(); free_.pop_back(); return e; }
    hipEvent_t e; hipEventCreate(&e); return e;
  }
  bool enabled_ = false, skip_ = false;
  std::string only_;
  std::vector<Pending> pending_;
  std::vector<hipEvent_t> free_;
  std::map<std::string, Stat> stats_;
};

// One helper thread that runs host-side jobs (enqueueing a factorisation's launches) next to the caller's thread.
// A launch-by-launch factorisation costs the host about as much time to enqueue as it costs the GPU to run (four launches
// and as many event operations per block column): two of them side by side on the GPU need two enqueueing threads.
class AsyncRunner {
 public:
  explicit AsyncRunner(int device) : device_(device) {}
  ~AsyncRunner() {
    { std::lock_guard<std::mutex> l(m_); stop_ = true; }
    cv_.notify_all();
    if (th_.joinable()) th_.join();
  }
  void run(std::function<void()> job) {
    wait();
    { std::lock_guard<std::mutex> l(m_); job_ = std::move(job); busy_ = true; }
    if (!th_.joinable()) th_ = std::thread([this] { loop(); });
    cv_.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> l(m_);
    cv_.wait(l, [this] { return !busy_; });
  }

 private:
  void loop() {
    (void)hipSetDevice(device_);
    for (;;) {
      std::function<void()> job;
      {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [this] { return stop_ || (busy_ && job_); });
        if (stop_) return;
        job = std::move(job_); job_ = nullptr;
      }
      job();
      { std::lock_guard<std::mutex> l(m_); busy_ = false; }
      cv_.notify_all();
    }
  }
  int device_;
  std::thread th_;
  std::mutex m_;
  std::condition_variable cv_;
  std::function<void()> job_;
  bool busy_ = false, stop_ = false;
};

// Streams + events of the look-ahead factorisation: the panel chain of the next block-column group
// runs on `panel` while the trailing SYRK of the current group runs on `bulk`, whose CU mask leaves
// a few CUs per XCD free so that the panel kernels are never queued behind a 6000-workgroup grid.
struct DeviceQueues;  // chol_kernels.hip: the queues of one device, shared by its contexts, alive until process exit
struct CholeskyContext {
  DeviceQueues* dq = nullptr;        // the device's queue set this context uses (init(): the device current at that time)
  bool prepared = false;             // the device's queue choice has been made and adopted (cholesky_prepare)
  bool in_trial = false;             // this context is the one running its device's queue trial
  bool resident = true;              // false: this context's factorisations never use the resident panel chain (sk_options_set_resident_kernels): the same plan, launch by launch
  int device = -1;
  void use(DeviceQueues* q);
  hipStream_t panel = nullptr;
  hipStream_t bulk = nullptr;        // SYRK of the late groups (short SYRK, the panel chain decides: more CUs kept free)
  hipStream_t bulk_early = nullptr;  // SYRK of the early groups (long SYRK hides a slow chain: fewer CUs kept free)
  int early_tiles = 0;               // groups with at least this many trailing tile rows use bulk_early
  int reserved_cus = 0;              // CUs kept free of the SYRK on `bulk` (0: ordinary stream)
  hipStream_t server = nullptr;      // the resident potrf workgroup of the chain-server factorisation (own hardware queue)
  int* sync = nullptr;               // its counters (device memory), sized for sync_blk block columns
  int sync_blk = 0;
  double* xs = nullptr;              // 128 x 128 scratch: X(j+1,j) of the column in flight
  int* sync_for(int nblk);
  std::vector<hipEvent_t> events;
  hipError_t init();
  // A second set of streams on the same device, for a factorisation that runs NEXT TO the primary context's (the tail
  // front of a dissected system): the panel / bulk queues of the device's candidates that `primary` does not use, no
  // resident server, and `fork`, a plain stream that stands in for the caller's stream.
  hipError_t init_secondary(const CholeskyContext& primary);
  hipStream_t fork = nullptr;
  hipEvent_t fork_ev = nullptr, join_ev = nullptr;
  std::unique_ptr<AsyncRunner> runner;   // enqueues this context's factorisation from a thread of its own
  hipEvent_t event(size_t i);
  ~CholeskyContext();
};

hipError_t cholesky_init();
size_t potrf128_lds_bytes();
// Factor the lower triangle of S (npad x ld, npad % 128 == 0) in place.
// Linv: (npad/128) blocks of 128x128, zero-initialised once by the caller.
// ctx == nullptr: everything on `s`; otherwise the panel chain overlaps the trailing SYRK.
// allow_chain: the grouping is the library's to choose (cholesky_plan): resident panel chain for the chain-bound columns.
// partner: a SECOND leaf front whose block columns — all under the resident chain — ride in the launches of the first one's
// chain-bound columns (one server workgroup each, the column launches and thin SYRKs of both fronts as ONE launch each, blockIdx.y
// picking the front: DESIGN.md section 8, item 0).  ctx lends its counters and scratch only.  taken: whether it was factored.
struct CholeskyPartner {
  double* S; long ld; int nblk, ncols, tail_rows; const int* last; double* Linv; CholeskyContext* ctx;
  const int* tail = nullptr;  // the partner's tail profile (as cholesky_factor's `tail`)
  mutable bool taken = false;
};
// tail (optional, nblk entries; round 4): a PROFILE of the trailing block rows instead of the uniform `tail_rows` — block rows
// [tail[c], nblk) are active in block column c on top of its contiguous run (non-increasing in c, tail[c] <= nblk - 1: the
// right-hand-side row is active everywhere).  With it a FULL factorisation may have a border as well: the trailing block
// columns are then the border's own (dense among themselves, last[c] == nblk - 1 there) — the cameras of loop closures, ordered
// behind the band of the camera sequence ("Bordered envelope").
void cholesky_factor(double* S, long ld, int npad, double* Linv, int* info, int group, hipStream_t s, CholeskyContext* ctx,
                     KernelTimer* kt, const int* last = nullptr, bool allow_chain = false, int ncols = -1, int tail_rows = 1,
                     const CholeskyPartner* partner = nullptr, const int* tail = nullptr);
void cholesky_prepare(CholeskyContext* ctx, hipStream_t s);
bool cholesky_note_info(CholeskyContext* ctx, int info);
bool cholesky_chain_enabled(const CholeskyContext* ctx);
// the right to run two resident servers per factorisation on this device (a partner front): at most four solvers hold it
bool cholesky_claim_pair_servers(CholeskyContext* ctx);
void cholesky_release_pair_servers(CholeskyContext* ctx);
void cholesky_disable_chain(CholeskyContext* ctx);
struct CholeskyPlan {
  std::vector<int> bounds;  // group start columns + nblk
  std::vector<char> resident;  // per block column: under the resident panel chain
  std::vector<char> paired;    // ... as the first (1) / second (2) column of a resident pair (one K = 256 SYRK for both); else 0
};
CholeskyPlan cholesky_plan(int nblk, int group, const int* last, bool chain, int ncols = -1, int tail_rows = 1, const int* tail = nullptr);
int cholesky_plan_max_group(const CholeskyPlan& plan);
// info != nullptr: one resident launch (bs_resident_kernel; a time-out of its polls raises *info to 2); nullptr: one launch per block step
// zero_after (the resident launch only): every block below the diagonal that is read is overwritten with zeros once it is in
// registers — the envelope is ready for the next assembly but for its diagonal blocks (BalSolver's reduced zero_envelope pass).
// resident: -1 the process-wide switch decides now (g_bs_resident: a time-out elsewhere may clear it at any time); 0 / 1 the CALLER's decision,
// taken once for all the back-substitutions of one linear solve (cholesky_backsolve_resident) — the fronts of a dissected system must
// all be solved the same way, and what zero_after promises the next assembly depends on it
void cholesky_backsolve(double* S, long ld, int n, int npad, int rhs_row, const double* Linv, double* w, double* y,
                        hipStream_t s, KernelTimer* kt, const int* last = nullptr, int* info = nullptr, const int* tail = nullptr, bool zero_after = false, int resident = -1,
                        bool prefilled = false);  // prefilled (resident launch): y already holds the "not there yet" pattern (all bits set) — the caller filled it off the critical path
bool cholesky_backsolve_resident(int nblk);  // would cholesky_backsolve(..., info != nullptr) of a system of nblk block rows be the resident launch?
// --- dissected factorisation (chol_kernels.hip, "Two-way dissection") ---
struct FrontView {
  double* S = nullptr; long ld = 0;   // row-major, lower triangle
  int nblk = 0, ncols = 0;            // block rows; block columns that are factored (leaf: the interior; root: all)
  const int* last = nullptr;          // block envelope (nblk entries) or nullptr
  double* Linv = nullptr;             // ncols inverted diagonal blocks
  int rhs_row = 0;                    // row that carries the right-hand side
  int tail_rows = 1;                  // block rows at the end that are active in every column (cholesky_plan)
  bool spike = false;                 // the border has rows that couple with the first interior columns (SegmentLayout)
  const int* tail = nullptr;          // a PROFILE of trailing block rows instead of the uniform tail_rows (cholesky_factor): the leaf's border ends
                                      // with the rows of retained points, which are active from the first camera that sees the point
};
struct DissectedSystem {
  FrontView A, B, R;                  // head, tail (B.ncols == 0: none), root
  int border_blocks = 0;              // block rows of the leaves' borders == R.nblk
  const int* mapB = nullptr;          // device: B's border index -> root index (< 0: padding)
  bool mapB_involution = false;       // mapB[mapB[i]] == i wherever mapB[i] >= 0 (a reversal is): the two fronts' borders are then added to the root in ONE launch
};
// Factor A and B side by side (B on ctxB's streams, launch by launch), add their Schur complements to the root, factor it.
// ktB: the timer of the tail's launches (they are enqueued by ctxB's own thread: a KernelTimer belongs to one thread).
void cholesky_dissected_factor(const DissectedSystem& d, int* info, int group, hipStream_t s, CholeskyContext* ctxA, CholeskyContext* ctxB,
                               KernelTimer* kt, KernelTimer* ktB, bool allow_chain);
// Solve: root, then the two interiors side by side.  yR / yA / yB: solutions in each front's own order; w*: scratch of the
// fronts' sizes; ybB: scratch of border size.  The right-hand sides are the fronts' rhs rows after the factorisation.
void cholesky_dissected_backsolve(const DissectedSystem& d, int n_root, double* wR, double* yR, double* wA, double* yA, double* wB, double* yB, double* ybB,
                                  hipStream_t s, CholeskyContext* ctxB, KernelTimer* kt, int* info = nullptr, bool zero_after = false, int resident = -1, bool prefilled = false);
// root += border x border block of a leaf front (front: ncols interior block columns, then border_blocks block rows);
// map: root index of each border index (nullptr: identity; < 0: skip)
void cholesky_border_add(double* root, long ld_r, const double* front, long ld_f, int ncols, int border_blocks, const int* map, hipStream_t s);
// info != nullptr: one resident launch (the kernel of cholesky_backsolve; tail_rows as in cholesky_factor); nullptr: one launch per block step
void cholesky_backsolve_front(double* S, long ld, int nblk, int ncols, int rhs_row, const double* Linv, const double* yb, double* w, double* y,
                              hipStream_t s, const int* last, bool spike = false, int tail_rows = 1, int* info = nullptr, bool zero_after = false, const int* tail = nullptr,
                              const int* yb_map = nullptr,  // yb_map (resident launch only): border index -> index into yb (< 0: zero) instead of a gathered copy
                              int resident = -1, bool prefilled = false);
void cholesky_gather_map(const double* src, const int* map, double* dst, int m, hipStream_t s);
// --- multi-way dissection: R segments of a block-banded system with R - 1 separators between them ---
// Leaf front of one segment, in scalar rows.  The interior is followed by a border:
//   first segment    [right separator, forward | rhs]                         tail_rows 1 — the head of the two-way case
//   last segment     [left separator, REVERSED | rhs], interior reversed too  tail_rows 1 — the tail of the two-way case
//   between two      [right separator, forward, padded to whole blocks | left separator, forward | rhs]
//                    eliminated front to back: its last columns reach the right separator as part of their contiguous run;
//                    the left separator couples with the FIRST columns and fills in along the whole interior (the spike):
//                    its block rows are the tail rows of the partial factorisation (cholesky_plan).
struct SegmentLayout {
  int ncols = 0, nblk = 0, tail_rows = 1;
  int rhs_row = 0;                    // absolute row of the right-hand side in the front
  int right_off = -1, left_off = -1;  // first border row (relative to the border) of the right / left separator; -1: none
  bool reversed = false;
  bool spike = false;                 // a segment between two separators: the left one's rows reach every interior column
};
inline SegmentLayout segment_layout(int interior_n, int left_n, int right_n) {
  SegmentLayout L;
  L.ncols = (interior_n + 127) / 128;
  if (left_n <= 0) {          // first segment (or the only one)
    L.right_off = 0;
    L.nblk = L.ncols + (right_n + 1 + 127) / 128;
    L.rhs_row = L.ncols * 128 + right_n;
  } else if (right_n <= 0) {  // last segment
    L.left_off = 0; L.reversed = true;
    L.nblk = L.ncols + (left_n + 1 + 127) / 128;
    L.rhs_row = L.ncols * 128 + left_n;
  } else {
    const int rb = (right_n + 127) / 128;
    L.right_off = 0; L.left_off = rb * 128; L.spike = true;
    L.tail_rows = (left_n + 1 + 127) / 128;
    L.nblk = L.ncols + rb + L.tail_rows;
    L.rhs_row = L.ncols * 128 + L.left_off + left_n;
  }
  return L;
}
// Block envelope of the root (every separator in sequence order, then the right-hand side): separator k couples with
// separator k - 1 through the Schur complement of the segment between them.  sep_off: R entries, scalar offset of each
// separator in the root and, last, their total.  Empty result: dense (one separator).
// members_n > 0: that many scalar rows behind the last separator couple with EVERY separator (the members of a border: pseudo-cameras of
// retained points) — a border of the root in the sense of cholesky_envelope_bordered, its profile in *tail_out.
std::vector<int> root_envelope(const std::vector<int>& sep_off, int members_n = 0, std::vector<int>* tail_out = nullptr);

double cholesky_syrk_flops(int npad, int group, const int* last = nullptr, bool chain = false, double* c_tiles = nullptr, int ncols = -1, int tail_rows = 1,
                           const int* tail = nullptr);
double cholesky_plan_flops(int nblk, const int* last, int ncols = -1, int tail_rows = 1, const int* tail = nullptr);
std::vector<int> cholesky_envelope_last(const std::vector<int>& first_col, int tail_rows = 1);
// Bordered envelope of a whole system: block rows [border_begin, nblk) are the border (the last one carries the right-hand side).
// From the block rows' first non-zero block columns: last[c] over the rows before the border (nblk - 1 for the border's own
// columns), and the profile tail[c] = first border row active in column c — a border row, once reached, stays active, and so
// does every border row behind it (the caller orders the border so that the rows reached first come last).
void cholesky_envelope_bordered(const std::vector<int>& first_col, int border_begin, std::vector<int>* last, std::vector<int>* tail);
// first block column in which block row i is active, for every i (from the envelope: the run `last`, the tail profile or uniform tail)
std::vector<int> cholesky_row_first_cols(int nblk, const int* last, const int* tail, int tail_rows = 1);
std::vector<int> cholesky_group_bounds(int nblk, int group);
void launch_syrk_gram(double* H, long ldh, const double* A, long lda, int Kc, int nslabs, double* slabs, int tiles, hipStream_t s,
                      KernelTimer* kt);

}  // namespace sk
