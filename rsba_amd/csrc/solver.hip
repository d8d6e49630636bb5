// Host side of rsba_solve: the symbolic phase (once per problem) and the trust-region loop.
//
// The loop is a rule-for-rule restatement of Ceres-Solver 1.9's TrustRegionMinimizer +
// LevenbergMarquardtStrategy with an exact Schur-complement linear solve — what
// ceres::Solve(SPARSE_SCHUR) runs for /root/reference/src/rsba/CeresHandler.h:394-426 (SURVEY
// Appendix C.5).  Every array lives on the device; per iteration the host reads back a few scalars
// (costs, model decrease, step / parameter norms, gradient max-norm, failure flags) and decides.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <limits>
#include <string>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

#include "handle.hpp"
#include "devmem.hpp"
#include "solver_state.hpp"
#include "tile_order.hpp"
#include "plan_device.hpp"
#include "test_hooks.hpp"

namespace rsba {


// HIP-event timing of the phases of an LM iteration (rsba_solver_options::profile_phases): events are recorded on the
// solver's stream around each group of launches and read after the iteration's own synchronisation point, so the
// timed run has the same launch sequence and no extra waits.
struct PhaseTimer {
  bool on = false;
  struct Rec { int phase; hipEvent_t a, b; };
  std::vector<hipEvent_t> pool; size_t next = 0;
  std::vector<Rec> pending;
  double ms[RSBA_NUM_PHASES] = {}; int32_t calls[RSBA_NUM_PHASES] = {};
  hipEvent_t get() { if (next == pool.size()) { hipEvent_t e = nullptr; (void)hipEventCreate(&e); pool.push_back(e); } return pool[next++]; }
  void reset() { for (int p = 0; p < RSBA_NUM_PHASES; ++p) { ms[p] = 0.0; calls[p] = 0; } pending.clear(); next = 0; }
  void resolve() {   // the stream is idle
    for (const Rec& r : pending) { float t = 0.f; if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) { ms[r.phase] += t; ++calls[r.phase]; } }
    pending.clear(); next = 0;
  }
  ~PhaseTimer() { for (hipEvent_t e : pool) (void)hipEventDestroy(e); }
};

struct Solver {
  PhaseTimer timer;
  PhaseTimer xtimer;   // the same for the collectives of a sharded solve, by kind (RSBA_EXCHANGE_*)
  rsba_plan_stats stats{};
  SolverDev sv{};
  std::vector<void*> allocs;
  // Cholesky plan over the packed tile slots, level-scheduled on the elimination structure (see build_solver)
  int nlev = 0;
  std::vector<int32_t> lev_diag_ptr, lev_sub_ptr, lev_upd_ptr, upd;   // [nlev+1] ranges of diag / sub / update items per level
  int32_t* d_upd = nullptr;
  std::vector<int32_t> diag_info, diag_ptr, diag_list;                // per column: {slot_jj, old tile}; contributors {slot_jk, old tile k}
  std::vector<int32_t> diag_own, sub_own;                             // first contributor the owner multiplies itself (the ones before arrive as partial tiles)
  std::vector<int32_t> diag_fuse;                                     // per column j: SUB item of tile (j, k*), k* = its last contributor, which the DIAG task forms itself (-1: none)
  int32_t* d_diag_fuse = nullptr;
  std::vector<int32_t> sub_pub;                                       // per SUB item: DIAG item that wants its X = S_ij - updates published (sv.Xpub slot), -1: nobody
  int32_t* d_sub_pub = nullptr;
  std::vector<int32_t> sub_col;                                       // per sub tile: tile index of its column (W_j, z_j)
  std::vector<int32_t> sub_info, sub_ptr, sub_list;                   // per tile (i,j): {slot_ij, slot_jj}; contributors {slot_ik, slot_jk}
  std::vector<int32_t> back_info, back_ptr, back_list;                // per column: {slot_jj, old tile}; tiles {slot_ij, old tile i}
  int32_t *d_diag_info = nullptr, *d_diag_ptr = nullptr, *d_diag_list = nullptr, *d_sub_info = nullptr, *d_sub_ptr = nullptr,
          *d_sub_list = nullptr, *d_sub_col = nullptr, *d_diag_own = nullptr, *d_sub_own = nullptr, *d_back_info = nullptr, *d_back_ptr = nullptr, *d_back_list = nullptr;
  std::vector<int32_t> tasks;                                         // {kind, item} in level order, backward solve last
  DagArgs* d_dag_args = nullptr;
  int32_t* d_slot_tiles = nullptr;                                    // [nslots][2] {row tile, column tile} of every packed tile
  double* d_verify = nullptr;                                         // [2 * npad] residual and yardstick of the DAG verification
  // The verification runs on a stream of its own, beside the back-substitution / candidate / trial evaluation of the iteration: it
  // only has to be done when the step scalars are packed.  verify_b = the right-hand side of the solve (sv.rhs is overwritten by the step).
  hipStream_t vstream = nullptr; hipEvent_t ev_solved = nullptr, ev_verified = nullptr; double* verify_b = nullptr; bool verify_pending = false;
  bool verify_dag = true;                                             // RSBA_CHOL_VERIFY=0 switches the check off
  bool test_corrupt_once = false;                                     // RSBA_CHOL_TEST_CORRUPT=1 (tests): the first DAG solve loses one entry of y
  int dag_fallbacks = 0;                                              // solves repeated on the level schedule after a failed check                                      // device copy of {sv, plan} for the persistent kernel
  int32_t* d_tasks = nullptr;
  unsigned int* d_dag_sync = nullptr;                                 // [ticket, pad x3]
  long long* d_trace = nullptr;                                       // RSBA_CHOL_TRACE=<file>: task time stamps of the last factorisation
  CholPlan plan{};
  int dag_workgroups = 0;
  bool dag_one_per_cu = true;                                         // LDS request above half a CU's: two persistent workgroups never share a CU (RSBA_CHOL_WGS above the CU count lifts it)
  bool use_levels = false;                                            // RSBA_CHOL_LEVELS=1: one launch per (level, kind)
  std::vector<int32_t> fwd_full, fwd_a, fwd_b, diag_toprow; int32_t *d_fwd_full = nullptr, *d_fwd_a = nullptr, *d_fwd_b = nullptr, *d_diag_toprow = nullptr;   // second right-hand side: contributor ranges per DIAG item and plan
  uint8_t* d_row_sep = nullptr;                                       // [nt] tile columns (old index) in the separators
  bool two_rhs = false;                                               // the plan carries a second right-hand side through the factorisation (free interFrameRatio: FWD2 / ETA tasks)
  int last_diag_slot = 0;
  int32_t* d_obs_slot = nullptr;
  double *d_gpose = nullptr, *d_gpoint = nullptr;
  int64_t num_pairs = 0;
  // The write-once cells of the DAG Cholesky (Lf | chol_part | Winv | zv | yv | Xpub) exist TWICE: while one set is in use the other
  // is re-armed (one memset) on a stream of its own, off the iteration's critical path; consecutive solves alternate.
  double* cells[2] = {nullptr, nullptr}; size_t ncells = 0, cell_off[5] = {0, 0, 0, 0, 0};
  DagArgs* d_dag_args2[2] = {nullptr, nullptr};
  int cur_cells = 0;
  hipStream_t mstream = nullptr; hipEvent_t ev_armed[2] = {nullptr, nullptr}, ev_released = nullptr; bool arm_pending[2] = {false, false};
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;   // the virtual-record sweep of a large shared-intrinsics problem runs on mstream beside the projection (reduce_system)
  int64_t schur_launches = 0;                                         // launches of the Schur kernel since the plan was built (statistics)
  double* d_ctl = nullptr;                                            // trust-region state on the device (device_state.hpp: LmCtlSlot; all zero while the host decides)
  rsba_iteration* d_trace_it = nullptr; int trace_it_cap = 0;         // the iteration records the deciding kernels write
  static constexpr int kCtlRing = 4;                                  // snapshots of d_ctl in flight (one per enqueued iteration): pinned host memory + the event behind each copy
  bool clamp_with_factor = false; double clamp_lo_hi[2] = {0.0, 0.0};   // device-side trust region: the diagonal's clamp rides in the point factor's launch
  double* h_ctl = nullptr; double* h_ctl_dev = nullptr;   // (h_ctl_dev: the same memory as the deciding kernel addresses it)
  bool gradmax_done = false;                              // the last linearisation's camera exchange carried max |g_i| (several ranks): gradient_max() has nothing left to do
  double ctl_seq = 0.0;                                   // stamp of the last snapshot asked for (never repeats within a handle: a stale slot cannot be mistaken for a new one)
  // Sharded factorisation (several ranks whose points respect the cut of tile_order.hpp): this rank factors the columns
  // of ITS part of the elimination tree from its own partial S (launch A), the ranks all-reduce the separators' tiles less what
  // their parts subtract from them, every rank factors the separators and solves them backward, then its own part (launch B).
  bool sharded = false, sharded_off = false;                          // the plan has that form; a suspect solve switched it off for this handle
  std::vector<int32_t> tasks_a, tasks_b; int32_t *d_tasks_a = nullptr, *d_tasks_b = nullptr;
  std::vector<int32_t> diag_info_sh, sub_info_sh; int32_t *d_diag_info_sh = nullptr, *d_sub_info_sh = nullptr;   // {.., part0, nparts} of the separators' items without the parts' partials
  CholPlan plan_a{}, plan_b{}; DagArgs* d_dag_args_a[2] = {nullptr, nullptr}; DagArgs* d_dag_args_b[2] = {nullptr, nullptr};
  int ntop_slots = 0, ntop_tiles = 0;
  int32_t *d_top_slots = nullptr, *d_top_info = nullptr, *d_asm_ptr = nullptr, *d_asm_list = nullptr, *d_top_tiles = nullptr;
  double* topx_buf = nullptr;                                         // exchange (2) of the sharded form: the separators' tiles | their rows of the right-hand side
  uint8_t* d_row_mine = nullptr;                                      // [nt] tiles (old index) whose rows of y this rank contributes to the gather (its part; rank 0: the separators)
  uint8_t* d_row_check = nullptr;                                     // [nt] ... and whose residual it can check: its part (every tile of those rows is complete here)
  double* ybuf = nullptr;                                             // [npad] y of this rank's tiles, zero elsewhere: summed over the ranks
  int32_t* d_top_fill = nullptr; int ntop_fill = 0;                   // separator tiles that exist through fill only (zero in S; the sharded solve leaves its reduced values there)
  int num_reduced_blocks = 0, num_reduced_params = 0, num_priors_reduced = 0;
  int32_t* exch_slots = nullptr; double* exch_buf = nullptr; int exch_tiles = 0;   // exchange (2) of a sharded solve: the plan's tile pairs, packed
  double* zy2 = nullptr;                                              // [2][npad] z | y of one more right-hand side through the last factorisation (solve_again)
  double* border = nullptr, *ratio4 = nullptr;                       // free interFrameRatio: its column of S [npad]; its scalars on the device (solver_state.hpp: RatioSlot)
  PosePriorDev pp{};                                                  // per-pose priors: linearisation of the priorPoses coordinates
  double* merge_buf = nullptr;                                        // sharded solve: [4 M] owned point values | owner flags
  double* ucross = nullptr;                                           // [F][CD][CD] motion-prior blocks (f, f-1), behind sv.U's J^T J blocks
};

}  // namespace rsba

using namespace rsba;

#define HIP_TRY(expr)                                                                                 \
  do {                                                                                                \
    hipError_t e_ = (expr);                                                                           \
    if (e_ != hipSuccess)                                                                             \
      return rsba_set_error(e_ == hipErrorOutOfMemory ? RSBA_ERR_OUT_OF_MEMORY : RSBA_ERR_HIP,        \
                            (std::string(#expr) + ": " + hipGetErrorString(e_)).c_str());             \
  } while (0)

namespace {

template <class T>
int32_t s_alloc(Solver* s, T** p, size_t count) {
  void* q = nullptr;
  HIP_TRY(dev_malloc(&q, std::max<size_t>(count, 1) * sizeof(T)));
  s->allocs.push_back(q);
  *p = static_cast<T*>(q);
  return RSBA_OK;
}
template <class T>
int32_t s_upload(Solver* s, T** p, const std::vector<T>& v) {
  int32_t rc = s_alloc(s, p, v.size());
  if (rc) return rc;
  if (!v.empty()) HIP_TRY(hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return RSBA_OK;
}
template <class T>
int32_t s_upload_const(Solver* s, const T** p, const std::vector<T>& v) {
  T* q = nullptr;
  int32_t rc = s_upload(s, &q, v);
  *p = q;
  return rc;
}

// The symbolic phase hands its finished arrays to ONE background thread that allocates and copies them while the host goes on
// with the next pass (a fresh handle's plan is the per-call cost of windowedBA, VideoSfMHandler.cc:185-214: at 1k cameras 60 MB of
// index arrays, 6 ms of copies from pageable memory that used to sit behind the passes instead of under them).  The vectors must
// stay untouched until finish(); the allocations end up in Solver::allocs like everyone else's.
struct Uploader {
  Solver* s; int device;
  std::thread th; std::mutex m; std::condition_variable cv; std::deque<std::function<hipError_t()>> q;
  bool closing = false, joined = false; hipError_t err = hipSuccess; std::string what;
  std::vector<void*> allocs;
  Uploader(Solver* s_, int dev) : s(s_), device(dev) {
    th = std::thread([this]() {
      (void)hipSetDevice(device);
      for (;;) {
        std::function<hipError_t()> job;
        { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return closing || !q.empty(); }); if (q.empty()) return; job = std::move(q.front()); q.pop_front(); }
        if (err == hipSuccess) err = job();
      }
    });
  }
  void push(std::function<hipError_t()> job) { { std::lock_guard<std::mutex> lk(m); q.push_back(std::move(job)); } cv.notify_one(); }
  // *_ref: the vector outlives this object and is not touched before finish() (the plan scratch, members of the Solver); the plain
  // forms take a copy (small tables that are locals of build_solver: an early return destroys them before this object)
  template <class T>
  void upload_ref(T** dst, const std::vector<T>& v) {
    push([this, dst, &v]() -> hipError_t {
      void* d = nullptr;
      hipError_t e = dev_malloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T));
      if (e != hipSuccess) return e;
      allocs.push_back(d); *dst = static_cast<T*>(d);
      return v.empty() ? hipSuccess : hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    });
  }
  template <class T>
  void upload(T** dst, const std::vector<T>& v) {
    auto own = std::make_shared<std::vector<T>>(v);
    push([this, dst, own]() -> hipError_t {
      void* d = nullptr;
      hipError_t e = dev_malloc(&d, std::max<size_t>(own->size(), 1) * sizeof(T));
      if (e != hipSuccess) return e;
      allocs.push_back(d); *dst = static_cast<T*>(d);
      return own->empty() ? hipSuccess : hipMemcpy(d, own->data(), own->size() * sizeof(T), hipMemcpyHostToDevice);
    });
  }
  template <class T>
  void upload_const(const T** dst, const std::vector<T>& v) { upload(const_cast<T**>(dst), v); }
  template <class T>
  void upload_const_ref(const T** dst, const std::vector<T>& v) { upload_ref(const_cast<T**>(dst), v); }
  hipError_t finish() {
    if (!joined) {
      { std::lock_guard<std::mutex> lk(m); closing = true; } cv.notify_one();
      th.join(); joined = true;
      s->allocs.insert(s->allocs.end(), allocs.begin(), allocs.end()); allocs.clear();
    }
    return err;
  }
  ~Uploader() { (void)finish(); }
};

// Host scratch of the symbolic phase — everything sized by the observations or the entries (85 MB at 1k cameras).  It lives across
// calls: as plain locals these vectors cost more than the passes that fill them — every fresh handle page-faulted them in and
// unmapped them on return (on a 256-core host, after 16 threads had touched them, the unmap alone was 25 ms of a 51 ms plan;
// measured with glibc told to keep its memory: 14.5 ms).  One build at a time uses the shared set (a second concurrent one gets
// its own, freed on return); rsba_release_host_scratch() gives the memory back.
struct PlanScratch {
  std::vector<int64_t> point_ptr, fill, vgroup_ptr, pt_group;
  std::vector<int32_t> obs_slot, real_frame, slot_frame, slot_point, g_tile, g_rows, ent_pt, vgroup_point, vgroup_intr;
  std::vector<uint32_t> slot_gpos, ent_groups, g_off;
  std::vector<int64_t> pt_goff;
  std::vector<uint8_t> group_mask, group_present;
  std::vector<uint16_t> ent_mask;
  std::vector<std::vector<int32_t>> thread_cnt;
  std::vector<double> inprog_point;
};
std::mutex g_plan_scratch_mutex;
std::unique_ptr<PlanScratch> g_plan_scratch;

// fn(a, b, t) over nthr contiguous ranges of [0, n)
template <class F>
void parallel_ranges(int nthr, int64_t n, F&& fn) {
  if (nthr <= 1) { fn((int64_t)0, n, 0); return; }
  std::vector<std::thread> pool;
  for (int t = 0; t < nthr; ++t) pool.emplace_back([&, t]() { fn(n * t / nthr, n * (t + 1) / nthr, t); });
  for (auto& th : pool) th.join();
}

int32_t exchange(rsba_handle* h, double* buf, int64_t count, int op, int kind);   // (below)

struct PhaseScope {
  PhaseTimer* t = nullptr; int phase; hipStream_t st; hipEvent_t a = nullptr;
  PhaseScope(rsba_handle* h, int ph) : phase(ph), st(h->stream) {
    if (h->solver && h->solver->timer.on) { t = &h->solver->timer; start(); }
  }
  // a phase that another one interrupts (the exchange between the two launches of a sharded factorisation): stop() ... start()
  void start() { if (t && !a) { a = t->get(); (void)hipEventRecord(a, st); } }
  void stop() { if (t && a) { hipEvent_t b = t->get(); (void)hipEventRecord(b, st); t->pending.push_back({phase, a, b}); a = nullptr; } }
  ~PhaseScope() { stop(); }
};

// Symbolic phase: frame / point adjacency, the per-block pair lists of the reduced camera system and
// the tile-level fill pattern of its Cholesky factor.  Ceres does the equivalent in its preprocessor
// (block structure detection, Schur ordering, CHOLMOD analyse) — SURVEY Appendix C.4.
int32_t build_solver_impl(rsba_handle* h) {
  const DeviceProblem& dp = h->dp;
  Solver* s = new Solver();
  h->solver = s;   // owned by the handle from here on (freed by rsba_destroy_solver)
  const bool dbg_plan = std::getenv("RSBA_DEBUG_PLAN") != nullptr;
  std::string phases; double t_phase = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
  auto tick = [&](const char* name) {
    if (!dbg_plan) return;
    const double t = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    char b[64]; std::snprintf(b, sizeof b, " %s %.1f ms;", name, (t - t_phase) * 1e3); phases += b; t_phase = t;
  };
  SolverDev& sv = s->sv;
  const int FR = dp.F, M = dp.M, CD = 6 * dp.P;          // FR real frames
  const int NIB = dp.calibrated ? 0 : dp.NI;               // intrinsics parameter blocks: sess.cam and / or per-frame f.cam (CeresHandler.h:260,277)
  const int NPF = dp.calibrated ? 0 : (9 + CD - 1) / CD;   // pseudo frames per intrinsics block (solver_state.hpp)
  const int F = FR + NIB * NPF;                            // camera-side blocks of the reduced system
  const int64_t N = dp.N;
  sv.F = FR; sv.Fx = F; sv.NPF = NPF; sv.NIB = NIB;
  const std::vector<int32_t>& fi = h->frame_intr;          // frame -> intrinsics block
  auto intr_of = [&](int f) { return NIB > 1 ? fi[f] : 0; };
  sv.CD = CD; sv.n = (int64_t)F * CD;
  const int FT = kTile / CD;
  sv.nt = (F + FT - 1) / FT; sv.npad = (int64_t)sv.nt * kTile;
  const std::vector<int32_t>& of = h->obs_frame; const std::vector<int32_t>& op = h->obs_point;

  std::unique_lock<std::mutex> scratch_lock(g_plan_scratch_mutex, std::try_to_lock);
  std::unique_ptr<PlanScratch> own_scratch;
  if (scratch_lock.owns_lock()) { if (!g_plan_scratch) g_plan_scratch.reset(new PlanScratch()); } else own_scratch.reset(new PlanScratch());
  PlanScratch& scr = scratch_lock.owns_lock() ? *g_plan_scratch : *own_scratch;
  // host threads of the passes over observations / points / entries: sixteen are worth their start-up (~1 ms on a busy 256-thread host)
  // from a few hundred thousand observations on; a 100-camera window (187 k) plans fastest on four — symbolic phase 4.1 / 3.9 / 2.8 /
  // 3.4 ms on 1 / 2 / 4 / 8 threads (RSBA_PLAN_THREADS overrides: A/B)
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  int plan_threads = N >= 400000 ? (int)std::min(16u, hw) : N >= 50000 ? (int)std::min(4u, hw) : 1;
  if (const char* e = std::getenv("RSBA_PLAN_THREADS")) plan_threads = std::max(1, std::min(64, std::atoi(e)));
  const int nt = sv.nt;
  // Which frame tiles store their groups FACTORED (solver_state.hpp: kGroupFactored): two-pose frames of a problem whose point-side passes
  // recompute the records — the 12 camera-side rows of a frame are (1 - tau) q | tau q, so 6 rows and tau say it all —
  // except a tile that holds an intrinsics pseudo frame (its virtual records have no such structure).  RSBA_FACTORED=0: none (A/B).
  bool recompute = dp.calibrated != 0 || NIB == 1;
  if (const char* e = std::getenv("RSBA_RECORDS")) recompute = recompute && e[0] != '1';
  bool factored = recompute && dp.P == 2;
  if (const char* e = std::getenv("RSBA_FACTORED")) factored = factored && e[0] != '0';
  std::vector<uint8_t> tile_factored((size_t)nt, 0);
  for (int t = 0; t < nt && factored; ++t) tile_factored[t] = !((int64_t)(t + 1) * FT > FR && (int64_t)t * FT < F && F > FR);   // (no pseudo frame in [t FT, (t + 1) FT))
  // The passes over observations, points and entries run ON THE DEVICE (plan_device.hip: stable sorts and prefix sums — the lists come
  // out as from the host passes below, which stay as the path for what the device form leaves out: several intrinsics blocks (their
  // per-point block lists), more than 8 192 tile columns (a dense pair map), and RSBA_PLAN_DEVICE=0 for A/B runs and the test that
  // compares the two).  The host keeps the O(tiles) part: ordering, symbolic factorisation, task lists, chunk numbering.
  bool dev_plan = N > 0 && NIB <= 1 && (int64_t)nt * nt <= ((int64_t)1 << 26) && N < ((int64_t)1 << 31);
  if (const char* e = std::getenv("RSBA_PLAN_DEVICE")) dev_plan = dev_plan && e[0] != '0';
  DevicePlanOut dpo;
  std::vector<int64_t> frame_ptr(FR + 1, 0);
  std::vector<int64_t>& point_ptr = scr.point_ptr;
  if (!dev_plan) point_ptr.assign((size_t)M + 1, 0);
  else for (int f = 0; f <= FR; ++f) frame_ptr[f] = (int64_t)(std::lower_bound(of.begin(), of.end(), (int32_t)f) - of.begin());   // (the list is frame-major)
  std::vector<int32_t>& obs_slot = scr.obs_slot; std::vector<int32_t>& real_frame = scr.real_frame;
  std::vector<int64_t>& vgroup_ptr = scr.vgroup_ptr; std::vector<int32_t>& vgroup_point = scr.vgroup_point; std::vector<int32_t>& vgroup_intr = scr.vgroup_intr;
  std::vector<int32_t>& slot_frame_host = scr.slot_frame; std::vector<int32_t>& slot_point = scr.slot_point;
  int64_t NVG = 0, NS = N;
  if (!dev_plan) {
  // slots: stable counting sort of the frame-major list by point -> ascending frame inside a point.  On several threads: every
  // thread counts the points of ITS range of observations, the counts of the threads before it are where its share of a point's
  // slots starts — the slots come out exactly as from one thread.
  obs_slot.resize((size_t)N);
  real_frame.resize((size_t)N);
  const int nthr_obs = (N >= 200000 && (int64_t)plan_threads * M <= ((int64_t)1 << 26)) ? plan_threads : 1;
  if (nthr_obs > 1) {
    std::vector<std::vector<int32_t>> cnt((size_t)nthr_obs);
    std::vector<std::vector<int64_t>> fcnt((size_t)nthr_obs);
    parallel_ranges(nthr_obs, N, [&](int64_t a, int64_t b, int t) {
      std::vector<int32_t>& c = cnt[(size_t)t]; c.assign((size_t)M, 0);
      std::vector<int64_t>& fc = fcnt[(size_t)t]; fc.assign((size_t)FR, 0);
      for (int64_t i = a; i < b; ++i) { ++c[op[i]]; ++fc[of[i]]; }
    });
    for (int f = 0; f < FR; ++f) { int64_t sum = 0; for (int t = 0; t < nthr_obs; ++t) sum += fcnt[(size_t)t][f]; frame_ptr[f + 1] = frame_ptr[f] + sum; }
    // point_ptr, and per thread the first slot of its share of every point (in place of its count)
    parallel_ranges(nthr_obs, M, [&](int64_t a, int64_t b, int) {
      for (int64_t j = a; j < b; ++j) { int64_t sum = 0; for (int t = 0; t < nthr_obs; ++t) sum += cnt[(size_t)t][j]; point_ptr[j + 1] = sum; }
    });
    for (int j = 0; j < M; ++j) point_ptr[j + 1] += point_ptr[j];
    std::vector<std::vector<int64_t>> first((size_t)nthr_obs);
    for (auto& v : first) v.resize((size_t)M);
    parallel_ranges(nthr_obs, M, [&](int64_t a, int64_t b, int) {
      for (int64_t j = a; j < b; ++j) { int64_t at = point_ptr[j]; for (int t = 0; t < nthr_obs; ++t) { first[(size_t)t][j] = at; at += cnt[(size_t)t][j]; } }
    });
    parallel_ranges(nthr_obs, N, [&](int64_t a, int64_t b, int t) {
      std::vector<int64_t>& fill = first[(size_t)t];
      for (int64_t i = a; i < b; ++i) { const int64_t sl = fill[op[i]]++; obs_slot[i] = (int32_t)sl; real_frame[sl] = of[i]; }
    });
  } else {
    for (int64_t i = 0; i < N; ++i) { frame_ptr[of[i] + 1]++; point_ptr[op[i] + 1]++; }
    for (int f = 0; f < FR; ++f) frame_ptr[f + 1] += frame_ptr[f];
    for (int j = 0; j < M; ++j) point_ptr[j + 1] += point_ptr[j];
    std::vector<int64_t>& fill = scr.fill; fill.assign(point_ptr.begin(), point_ptr.end() - 1);
    for (int64_t i = 0; i < N; ++i) { const int64_t sl = fill[op[i]]++; obs_slot[i] = (int32_t)sl; real_frame[sl] = of[i]; }
  }
  // virtual groups: one per (observed point, intrinsics block it is seen through), blocks ascending; each owns NPF virtual
  // slots behind the real ones
  vgroup_ptr.assign((size_t)M + 1, 0);
  vgroup_point.clear(); vgroup_intr.clear();
  if (NIB == 1) {   // one shared block (the usual uncalibrated session): every observed point is seen through it — no lists to sort
    vgroup_point.reserve((size_t)M); vgroup_intr.reserve((size_t)M);
    for (int j = 0; j < M; ++j) {
      if (point_ptr[j + 1] > point_ptr[j]) { vgroup_point.push_back(j); vgroup_intr.push_back(0); }
      vgroup_ptr[j + 1] = (int64_t)vgroup_point.size();
    }
  } else if (NIB > 0) {
    std::vector<int32_t> seen;
    for (int j = 0; j < M; ++j) {
      seen.clear();
      for (int64_t x = point_ptr[j]; x < point_ptr[j + 1]; ++x) seen.push_back(intr_of(real_frame[x]));
      std::sort(seen.begin(), seen.end()); seen.erase(std::unique(seen.begin(), seen.end()), seen.end());
      for (int32_t c : seen) { vgroup_point.push_back(j); vgroup_intr.push_back(c); }
      vgroup_ptr[j + 1] = (int64_t)vgroup_point.size();
    }
  }
  NVG = (int64_t)vgroup_point.size();
  NS = N + NVG * NPF;
  slot_frame_host.resize((size_t)NS);
  slot_point.resize((size_t)NS);
  parallel_ranges(nthr_obs, N, [&](int64_t a, int64_t b, int) { for (int64_t x = a; x < b; ++x) slot_frame_host[x] = real_frame[x]; });
  parallel_ranges(nthr_obs, M, [&](int64_t a, int64_t b, int) { for (int64_t j = a; j < b; ++j) for (int64_t x = point_ptr[j]; x < point_ptr[j + 1]; ++x) slot_point[x] = (int32_t)j; });
  for (int64_t g = 0; g < NVG; ++g) for (int v = 0; v < NPF; ++v) { slot_frame_host[N + g * NPF + v] = FR + vgroup_intr[g] * NPF + v; slot_point[N + g * NPF + v] = vgroup_point[g]; }
  }   // (!dev_plan)
  std::vector<int32_t>& slot_frame = dev_plan ? dpo.slot_frame_h : slot_frame_host;   // (device plan: the real slots only, and only when a later pass asks for them)
  // the slots of point j in ascending frame order (virtual ones last, by intrinsics block; only for points that are observed)
  auto slots_of = [&](int j, std::vector<int64_t>& out) {
    out.clear();
    for (int64_t x = point_ptr[j]; x < point_ptr[j + 1]; ++x) out.push_back(x);
    for (int64_t g = vgroup_ptr[j]; g < vgroup_ptr[j + 1]; ++g) for (int v = 0; v < NPF; ++v) out.push_back(N + g * NPF + v);
  };
  Uploader up(s, h->device);
  up.upload_const(&sv.frame_ptr, frame_ptr);
  if (!dev_plan) {
    up.upload_const_ref(&sv.point_ptr, point_ptr);
    up.upload_const_ref(&sv.slot_frame, slot_frame_host);
    up.upload_const_ref(&sv.slot_point, slot_point);
    up.upload_ref(&s->d_obs_slot, obs_slot);
  }
  tick("slots");
  // ---- work list of the point elimination: one ENTRY per (point, pair of frame tiles I >= J) ----
  // An entry lists the point's observation slot in each of the FT frames of tile I (sa) and of tile J (sb),
  // -1 where it is not observed.  One wave turns an entry into up to FT x FT block products P_a P_b^T with
  // every P record loaded once (frame-pair-major accumulation, no atomics).  A point seen
  // twice in one frame gets a second "layer" of slots and the cross-layer entries.
  // Per point, the (tile, layer) groups of its slots — computed once, flat (no per-point allocations: this pass used
  // to be 85 % of the symbolic phase): group g of point j covers one tile and one layer and owns FT slot entries.
  std::vector<int64_t>& pt_group = scr.pt_group;     // groups of point j: [pt_group[j], pt_group[j+1])
  std::vector<int32_t>& g_tile = scr.g_tile; std::vector<int32_t>& g_rows = scr.g_rows;   // tile of each group; FT slots per group (NS = not observed)
  std::vector<uint32_t>& slot_gpos = scr.slot_gpos;         // where every slot's P record goes: group offset | position << 1 | kind (solver_state.hpp)
  if (!dev_plan) { pt_group.assign((size_t)M + 1, 0); slot_gpos.resize((size_t)NS); }
  std::vector<uint8_t>& group_mask = scr.group_mask;        // which of the three 16-row blocks of a group's records can be non-zero
  std::vector<uint8_t>& group_present = scr.group_present;  // frames of the group's tile that see the point (plan statistics)
  std::vector<uint32_t>& g_off = scr.g_off;                 // element offset of every group in Pm
  std::vector<int64_t>& pt_goff = scr.pt_goff;   // doubles of the groups of the points before j
  const int nthr_pts = M >= 4096 ? plan_threads : 1;
  int64_t pt_total = 0;                          // doubles of all groups
  if (!dev_plan) {
    pt_goff.assign((size_t)M + 1, 0);
    // one walk over a point's slots: on_group(g, tile) for every new (tile, layer) group g = 0, 1, .. of the point, on_slot(g, pos, slot)
    auto walk = [&](int j, std::vector<int64_t>& pslots, auto&& on_group, auto&& on_slot) -> int64_t {
      slots_of(j, pslots);
      int prev_frame = -1, layer = 0, cur_tile = -1;
      int64_t ng = 0, tile_first = 0;            // first group (layer 0) of the current tile
      for (int64_t sl : pslots) {
        const int f = slot_frame[sl], tile = f / FT, pos = f % FT;
        layer = (f == prev_frame) ? layer + 1 : 0; prev_frame = f;
        if (tile != cur_tile) { cur_tile = tile; tile_first = ng; }
        while (ng - tile_first <= layer) { on_group(ng, tile); ++ng; }   // a new layer of this tile
        on_slot(tile_first + layer, pos, sl);
      }
      return ng;
    };
    // count, prefix, fill — over contiguous point ranges on a few threads (the lists come out as from one thread)
    parallel_ranges(nthr_pts, M, [&](int64_t a, int64_t b, int) {
      std::vector<int64_t> ps;
      for (int64_t j = a; j < b; ++j) {
        int64_t doubles = 0;
        pt_group[j + 1] = walk((int)j, ps, [&](int64_t, int tile) { doubles += tile_factored[tile] ? kGroupFactored : kGroupFull; }, [](int64_t, int, int64_t) {});
        pt_goff[j + 1] = doubles;
      }
    });
    for (int j = 0; j < M; ++j) { pt_group[j + 1] += pt_group[j]; pt_goff[j + 1] += pt_goff[j]; }
    const int64_t NG = pt_group[M];
    // (more than 2^32 doubles of groups: the offsets below wrap; the plan is given up right behind this pass — on several ranks through the vote)
    g_tile.resize((size_t)NG); g_rows.assign((size_t)NG * FT, (int32_t)NS);   // NS = the all-zero record behind the last slot: "not observed"
    g_off.resize((size_t)NG + 1);
    group_mask.assign((size_t)NG + 1, 0); group_present.assign((size_t)NG + 1, 0);
    parallel_ranges(nthr_pts, M, [&](int64_t a, int64_t b, int) {
      std::vector<int64_t> ps;
      for (int64_t j = a; j < b; ++j) {
        const int64_t base = pt_group[j];
        int64_t at = pt_goff[j];
        walk((int)j, ps, [&](int64_t g, int tile) { g_tile[(size_t)(base + g)] = tile; g_off[(size_t)(base + g)] = (uint32_t)at; at += tile_factored[tile] ? kGroupFactored : kGroupFull; },
             [&](int64_t g, int pos, int64_t sl) {
               const size_t gg = (size_t)(base + g);
               const bool fac = tile_factored[g_tile[gg]] != 0;
               g_rows[gg * FT + pos] = (int32_t)sl;
               slot_gpos[sl] = g_off[gg] | ((uint32_t)pos << 1) | (fac ? 1u : 0u);
               ++group_present[gg];
               if (fac) group_mask[gg] |= (uint8_t)(pos < 2 ? 0b011 : pos == 2 ? 0b111 : 0b100);   // (factored: blocks 0 / 1 hold sources 0..15 = frames 0, 1 and two thirds of 2; block 2 the rest)
               else for (int row = pos * CD; row < (pos + 1) * CD; row += 4) group_mask[gg] |= (uint8_t)(1u << (row / 16));
             });
      }
    });
    g_off[(size_t)NG] = (uint32_t)pt_goff[M];
    pt_total = pt_goff[M];
    sv.ngroups = (int64_t)g_tile.size();
  }
  // A plan that cannot be built on THIS rank must not leave the other ranks waiting in the vote further down (one all-reduce in the
  // middle of the plan): a rank-local failure is carried into that vote and every rank fails together; a single rank returns here.
  const bool plan_votes = h->allreduce && h->world > 1 && !h->union_mask.empty();
  int32_t local_fail = RSBA_OK; const char* local_why = ""; std::string dev_why;
  if (test_hook("RSBA_TEST_FAIL_PLAN")) { local_fail = RSBA_ERR_UNSUPPORTED; local_why = "RSBA_TEST_FAIL_PLAN: the plan was made to fail (test hook)"; }   // after the uploader has started
  else if (pt_total + kGroupFull >= ((int64_t)1 << 32)) { local_fail = RSBA_ERR_UNSUPPORTED; local_why = "more than 2^32 doubles of P records: the Schur kernel indexes them with 32 bits"; }
  if (local_fail && !plan_votes) return rsba_set_error(local_fail, local_why);
  if (!dev_plan) up.upload_const_ref(&sv.slot_gpos, slot_gpos);
  const bool dense_keys = (int64_t)nt * nt <= (int64_t)1 << 26;
  std::vector<int64_t> dense_cnt; std::unordered_map<int64_t, int64_t> sparse_cnt;
  if (dense_keys && !dev_plan) dense_cnt.assign((size_t)nt * nt, -1);
  std::vector<uint32_t> struct_keys;   // (device plan: the keys of the pairs that exist whatever the points say)
  auto bump = [&](int I, int J, int64_t by) {
    const int64_t key = (int64_t)I * nt + J;
    if (dev_plan) { struct_keys.push_back((uint32_t)key); return; }
    if (dense_keys) { int64_t& c = dense_cnt[key]; c = (c < 0 ? 0 : c) + by; }
    else sparse_cnt[key] += by;
  };
  for (int I = 0; I < nt; ++I) bump(I, I, 0);                // every diagonal tile exists (U + D^2, rhs)
  auto bump_blocks = [&](int a, int b) { const int I = std::max(a, b) / FT, J = std::min(a, b) / FT; bump(I, J, 0); };
  if (!h->union_mask.empty())                                  // multi-GPU: tiles other ranks fill, so all ranks share one layout
    for (int a = 0; a < FR; ++a) for (int b = 0; b <= a; ++b) if (h->union_mask[(size_t)a * FR + b]) {
      bump(a / FT, b / FT, 0);
      for (int v = 0; v < NPF; ++v) {                          // ... and the rows of the two frames' intrinsics blocks
        bump_blocks(FR + intr_of(a) * NPF + v, b); bump_blocks(FR + intr_of(b) * NPF + v, a);
        for (int w = 0; w < NPF; ++w) bump_blocks(FR + intr_of(a) * NPF + v, FR + intr_of(b) * NPF + w);
      }
    }
  // J^T J blocks that do not come from a point: (intrinsics block of a frame) x (that frame), and an intrinsics block with itself
  for (int f = 0; f < FR && NIB > 0; ++f) for (int v = 0; v < NPF; ++v) bump_blocks(FR + intr_of(f) * NPF + v, f);
  for (int c = 0; c < NIB; ++c) for (int v = 0; v < NPF; ++v) for (int w = 0; w <= v; ++w) bump_blocks(FR + c * NPF + v, FR + c * NPF + w);
  for (int32_t f : h->prior_frames) bump(f / FT, (f - 1) / FT, 0);   // motion priors couple frame f with f - 1 (every rank: one layout)
  // entries of point j: every pair of its tiles (X >= Y) times every combination of their layers — for X == Y both
  // orders of two different layers (the diagonal tile pair is stored in full)
  auto for_each_entry = [&](int j, auto&& fn) {
    for (int64_t xa = pt_group[j]; xa < pt_group[j + 1];) {
      int64_t xb = xa; while (xb < pt_group[j + 1] && g_tile[xb] == g_tile[xa]) ++xb;
      for (int64_t ya = pt_group[j]; ya < xb;) {
        int64_t yb = ya; while (yb < pt_group[j + 1] && g_tile[yb] == g_tile[ya]) ++yb;
        for (int64_t gx = xa; gx < xb; ++gx) for (int64_t gy = ya; gy < yb; ++gy) fn(gx, gy);
        ya = yb;
      }
      xa = xb;
    }
  };
  // The two passes over all (point, tile pair) entries — count, then fill — are most of the symbolic phase (2 M entries at 1k cameras):
  // with dense keys they run on a few host threads over contiguous point ranges, each with its own counters per tile pair, and
  // the fill starts every thread where the threads before it end: the entry lists come out exactly as from one thread.
  // Beyond 2 048 tile columns (8 k cameras) a counter per tile pair and thread would be 4 nt^2 bytes each: the threads first mark which
  // pairs exist (one shared byte map), the pairs are numbered, and the per-thread counters are as long as that list (~9 nt).
  const bool small_keys = dense_keys && (int64_t)nt * nt <= ((int64_t)1 << 22) && !std::getenv("RSBA_PLAN_LISTED_KEYS");   // (the variable: the large-problem path at any size — its plan must be the same)
  const bool listed_keys = dense_keys && !small_keys && plan_threads > 1;
  const int nthreads = (small_keys || listed_keys) && M >= 4096 ? plan_threads : 1;   // (small_keys: per-thread counters of 4 nt^2 bytes)
  std::vector<std::vector<int32_t>>& thread_cnt = scr.thread_cnt; thread_cnt.resize(nthreads > 1 ? nthreads : 0);
  auto point_range = [&](int t) { return std::pair<int, int>((int)((int64_t)M * t / nthreads), (int)((int64_t)M * (t + 1) / nthreads)); };
  std::vector<int32_t> tp_I, tp_J; std::vector<int64_t> tp_ptr(1, 0);
  int64_t nent = 0;
  std::vector<int32_t>& ent_pt = scr.ent_pt;   // (host passes only; the device plan hands the chunk numbering the entry list's segments instead)
  if (dev_plan) ent_pt.clear();
  std::vector<int64_t> products_part(1, 0);   // (plan statistics: block products that are not structurally zero)
  if (dev_plan) {
    // ---- the device form of the passes above and below (plan_device.hip) ----
    // (a failure here — the arena allocations are the likeliest out-of-memory of the symbolic phase — is carried into the vote below
    // when several ranks plan together, like the two checks above: a rank that returned early would leave the others waiting in it)
    uint8_t* d_tf = nullptr;
    if (factored) {
      if (int32_t rc_ = s_upload(s, &d_tf, tile_factored)) {
        if (!plan_votes) return rc_;
        if (!local_fail) { local_fail = rc_; dev_why = std::string("device plan: ") + rsba_last_error(); local_why = dev_why.c_str(); }
      }
    }
    bool any_const_point = false;
    for (int j = 0; j < M && !any_const_point; ++j) any_const_point = h->mask_point[(size_t)j * 3] == 0.0;
    int64_t chunk_block = nt > 500 ? 2048 : 0;   // (the chunk numbering by blocks of points, below: it gets the entry list's segments instead of the list)
    if (const char* e = std::getenv("RSBA_SCHUR_BLOCK")) chunk_block = std::atoi(e) > 0 ? std::max(16, std::atoi(e)) : 0;
    if (chunk_block >= M) chunk_block = 0;
    DevicePlanIn din{dp.obs_frame, dp.obs_point, N, M, FR, NPF, NIB, FT, CD, nt, d_tf, struct_keys.data(), (int64_t)struct_keys.size(), chunk_block, any_const_point, h->stream};
    hipError_t pe = local_fail ? hipSuccess : device_plan_lists(din, &dpo);
    if (pe == hipSuccess && !local_fail && test_hook("RSBA_TEST_FAIL_DEVICE_PLAN")) pe = hipErrorOutOfMemory;   // (test hook: the lists' allocations fail on this rank)
    s->allocs.insert(s->allocs.end(), dpo.owned.begin(), dpo.owned.end());
    if (pe != hipSuccess) {
      (void)hipGetLastError();
      const int32_t code = pe == hipErrorOutOfMemory ? RSBA_ERR_OUT_OF_MEMORY : pe == hipErrorInvalidValue ? RSBA_ERR_UNSUPPORTED : RSBA_ERR_HIP;
      dev_why = pe == hipErrorInvalidValue ? std::string("device plan: 2^32 or more (point, tile pair) entries — the lists are indexed with 32 bits") : std::string("device plan: ") + hipGetErrorString(pe);
      if (!plan_votes) return rsba_set_error(code, dev_why.c_str());
      local_fail = code; local_why = dev_why.c_str();
    }
    if (local_fail) {
      // nothing of the lists can be used: an empty plan (no points' entries) walks through the host-side passes up to the vote, where every rank gives up together
      dpo = DevicePlanOut{};
      dpo.point_ptr_h.assign((size_t)M + 1, 0); dpo.tp_ptr.assign(1, 0);
    }
    point_ptr.swap(dpo.point_ptr_h);
    sv.point_ptr = dpo.point_ptr; sv.slot_frame = dpo.slot_frame; sv.slot_point = dpo.slot_point; s->d_obs_slot = dpo.obs_slot; sv.slot_gpos = dpo.slot_gpos;
    sv.ent_groups = dpo.ent_groups; sv.ent_pt = dpo.ent_pt; sv.ent_mask = dpo.ent_mask;
    NVG = dpo.nvgroups; NS = N + NVG * NPF;
    sv.ngroups = dpo.ngroups; pt_total = dpo.group_doubles;
    tp_I.swap(dpo.tp_I); tp_J.swap(dpo.tp_J); tp_ptr.swap(dpo.tp_ptr);
    nent = tp_ptr.back();
    products_part[0] = dpo.products;
    if (!local_fail && pt_total + kGroupFull >= ((int64_t)1 << 32)) {
      local_fail = RSBA_ERR_UNSUPPORTED; local_why = "more than 2^32 doubles of P records: the Schur kernel indexes them with 32 bits";
      if (!plan_votes) return rsba_set_error(local_fail, local_why);
    }
  } else {
  std::vector<int32_t> pair_no;   // listed_keys: key -> number of the pair in (I, J) order (what dense_index will hold further down)
  if (nthreads > 1 && listed_keys) {
    std::vector<uint8_t> seen((size_t)nt * nt, 0);
    {
      std::vector<std::thread> pool;
      for (int t = 0; t < nthreads; ++t)
        pool.emplace_back([&, t]() {
          const auto r = point_range(t);
          for (int j = r.first; j < r.second; ++j) for_each_entry(j, [&](int64_t gx, int64_t gy) { __atomic_store_n(&seen[(size_t)g_tile[gx] * nt + g_tile[gy]], (uint8_t)1, __ATOMIC_RELAXED); });
        });
      for (auto& th : pool) th.join();
    }
    pair_no.assign((size_t)nt * nt, -1);
    int32_t np = 0;
    for (int I = 0; I < nt; ++I) for (int J = 0; J <= I; ++J) {
      const size_t key = (size_t)I * nt + J;
      if (seen[key] && dense_cnt[key] < 0) dense_cnt[key] = 0;   // the pair exists (its entries are counted below)
      if (dense_cnt[key] >= 0) pair_no[key] = np++;
    }
    std::vector<uint8_t>().swap(seen);
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; ++t)
      pool.emplace_back([&, t]() {
        std::vector<int32_t>& c = thread_cnt[t];
        c.assign((size_t)np, 0);
        const auto r = point_range(t);
        for (int j = r.first; j < r.second; ++j) for_each_entry(j, [&](int64_t gx, int64_t gy) { ++c[pair_no[(size_t)g_tile[gx] * nt + g_tile[gy]]]; });
      });
    for (auto& th : pool) th.join();
    for (int I = 0; I < nt; ++I) for (int J = 0; J <= I; ++J) {
      const size_t key = (size_t)I * nt + J;
      if (pair_no[key] < 0) continue;
      int64_t sum = 0;
      for (int t = 0; t < nthreads; ++t) sum += thread_cnt[t][pair_no[key]];
      dense_cnt[key] += sum;
    }
  } else if (nthreads > 1) {
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; ++t)
      pool.emplace_back([&, t]() {
        std::vector<int32_t>& c = thread_cnt[t];
        c.assign((size_t)nt * nt, 0);
        const auto r = point_range(t);
        for (int j = r.first; j < r.second; ++j) for_each_entry(j, [&](int64_t gx, int64_t gy) { ++c[(size_t)g_tile[gx] * nt + g_tile[gy]]; });
      });
    for (auto& th : pool) th.join();
    for (size_t key = 0; key < (size_t)nt * nt; ++key) {
      int64_t sum = 0;
      for (int t = 0; t < nthreads; ++t) sum += thread_cnt[t][key];
      if (sum > 0) { int64_t& c = dense_cnt[key]; c = (c < 0 ? 0 : c) + sum; }
    }
  } else
  for (int j = 0; j < M; ++j) for_each_entry(j, [&](int64_t gx, int64_t gy) { bump(g_tile[gx], g_tile[gy], 1); });
  std::unordered_map<int64_t, int32_t> tp_index; std::vector<int32_t> dense_index;
  if (dense_keys) {
    dense_index.assign((size_t)nt * nt, -1);
    for (int I = 0; I < nt; ++I) for (int J = 0; J <= I; ++J) {
      const int64_t c = dense_cnt[(size_t)I * nt + J];
      if (c >= 0) { dense_index[(size_t)I * nt + J] = (int32_t)tp_I.size(); tp_I.push_back(I); tp_J.push_back(J); tp_ptr.push_back(tp_ptr.back() + c); }
    }
    std::vector<int64_t>().swap(dense_cnt);
  } else {
    std::vector<int64_t> keys; keys.reserve(sparse_cnt.size());
    for (auto& kv : sparse_cnt) keys.push_back(kv.first);
    std::sort(keys.begin(), keys.end());
    for (int64_t key : keys) { tp_index[key] = (int32_t)tp_I.size(); tp_I.push_back((int32_t)(key / nt)); tp_J.push_back((int32_t)(key % nt)); tp_ptr.push_back(tp_ptr.back() + sparse_cnt[key]); }
  }
  auto index_of = [&](int I, int J) -> int32_t { return dense_keys ? dense_index[(size_t)I * nt + J] : tp_index[(int64_t)I * nt + J]; };
  nent = tp_ptr.back();
  // an entry is the pair of groups (of tile I, of tile J) plus its point: the kernel looks the slots up in g_rows
  std::vector<uint32_t>& ent_groups = scr.ent_groups; ent_groups.resize((size_t)nent * 2);   // (where the two groups start in Pm | kind: solver_state.hpp)
  ent_pt.resize((size_t)nent);
  // ... and, per entry, which of the 3 x 3 block products of its two groups can be non-zero
  std::vector<uint16_t>& ent_mask = scr.ent_mask; ent_mask.resize((size_t)nent);
  products_part.assign((size_t)std::max(nthreads, 1), 0);
  auto put_entry = [&](int64_t w, int64_t gx, int64_t gy, int j, int64_t& prod) {
    ent_groups[2 * (size_t)w] = g_off[(size_t)gx] | (tile_factored[g_tile[(size_t)gx]] ? 1u : 0u);
    ent_groups[2 * (size_t)w + 1] = g_off[(size_t)gy] | (tile_factored[g_tile[(size_t)gy]] ? 1u : 0u);
    ent_pt[w] = j | (gx == gy ? (int32_t)0x80000000 : 0);   // top bit: the entry carries the rhs term P z
    const unsigned ma = group_mask[(size_t)gx], mb = group_mask[(size_t)gy];
    unsigned pm = 0;
    for (int I = 0; I < 3; ++I) if ((ma >> I) & 1u) pm |= mb << (3 * I);
    ent_mask[(size_t)w] = (uint16_t)pm;
    prod += (int64_t)group_present[(size_t)gx] * group_present[(size_t)gy];
  };
  {
    std::vector<int64_t> fill(tp_ptr.begin(), tp_ptr.end() - 1);
    if (nthreads > 1) {
      // per thread and tile pair: where its entries start (the counters become cursors)
      std::vector<std::vector<int64_t>> cursor(nthreads, std::vector<int64_t>(tp_I.size(), 0));
      for (size_t t_ = 0; t_ < tp_I.size(); ++t_) {
        const size_t key = listed_keys ? t_ : (size_t)tp_I[t_] * nt + tp_J[t_];   // (listed_keys: the counters are indexed by the pair's number, which is t_)
        int64_t at = fill[t_];
        for (int t = 0; t < nthreads; ++t) { cursor[t][t_] = at; at += thread_cnt[t][key]; }
      }
      std::vector<std::thread> pool;
      for (int t = 0; t < nthreads; ++t)
        pool.emplace_back([&, t]() {
          std::vector<int64_t>& cur = cursor[t];
          const auto r = point_range(t);
          int64_t prod = 0;
          for (int j = r.first; j < r.second; ++j)
            for_each_entry(j, [&](int64_t gx, int64_t gy) { put_entry(cur[dense_index[(size_t)g_tile[gx] * nt + g_tile[gy]]]++, gx, gy, j, prod); });
          products_part[t] = prod;
        });
      for (auto& th : pool) th.join();
    } else {
      int64_t prod = 0;
      for (int j = 0; j < M; ++j) for_each_entry(j, [&](int64_t gx, int64_t gy) { put_entry(fill[index_of(g_tile[gx], g_tile[gy])]++, gx, gy, j, prod); });
      products_part[0] = prod;
    }
  }
  std::vector<int32_t>().swap(dense_index);
  up.upload_const_ref(&sv.ent_groups, ent_groups);
  up.upload_const_ref(&sv.ent_pt, ent_pt);
  up.upload_const_ref(&sv.ent_mask, ent_mask);
  }   // (!dev_plan)
  sv.nvgroups = NVG;
  const int ntp = (int)tp_I.size();
  s->num_pairs = nent;

  // ---- tile graph of S, fill-reducing / parallelism-exposing ordering, symbolic factorisation ----
  std::vector<std::vector<int32_t>> adj(nt);
  for (int t = 0; t < ntp; ++t) if (tp_I[t] != tp_J[t]) { adj[tp_I[t]].push_back(tp_J[t]); adj[tp_J[t]].push_back(tp_I[t]); }
  tick("entries");
  // Nested dissection by BFS level structures (tile_order.hpp).  A sharded solve (several ranks, one tile layout: the co-visibility
  // structure of all ranks is installed) asks for the top of the tree to be cut into one part per rank; whether THIS rank's points
  // respect the cut — rsba_partition_points places them so — is checked below (sharded plan).
  const bool want_parts = h->allreduce && h->world > 1 && !h->union_mask.empty();
  std::vector<double> tile_weight(nt, 0.0);
  for (int f = 0; f < FR; ++f) tile_weight[f / FT] += (double)(h->frame_obs_total.empty() ? frame_ptr[f + 1] - frame_ptr[f] : h->frame_obs_total[f]);
  TileOrder tord = nested_dissection(nt, adj, plan_leaf_size(want_parts ? h->world : 1, nt), want_parts ? h->world : 1, &tile_weight);
  const std::vector<int32_t>& perm = tord.perm;          // perm[new] = old tile
  std::vector<int32_t> iperm(nt);
  for (int k = 0; k < nt; ++k) iperm[perm[k]] = k;
  tick("ordering");
  // symbolic factorisation in the new order: col[k] = rows i > k of column k (after fill), row[j] = columns k < j of row j
  std::vector<std::vector<int32_t>> col(nt), row(nt);
  {
    std::vector<std::vector<uint8_t>> mark(nt);
    for (int k = 0; k < nt; ++k) mark[k].assign(nt - k, 0);       // mark[k][i-k] for i >= k
    for (int t = 0; t < nt; ++t) for (int u : adj[t]) { const int i = std::max(iperm[t], iperm[u]), k = std::min(iperm[t], iperm[u]); mark[k][i - k] = 1; }
    for (int k = 0; k < nt; ++k) {
      for (int i = k + 1; i < nt; ++i) if (mark[k][i - k]) col[k].push_back(i);
      for (size_t u = 0; u < col[k].size(); ++u) for (size_t v = u; v < col[k].size(); ++v) mark[col[k][u]][col[k][v] - col[k][u]] = 1;
      for (int32_t i : col[k]) row[i].push_back(k);
    }
  }
  // packed tile slots, column by column: (k,k) first, then the sub-diagonal tiles of column k
  std::vector<int32_t> slot_base(nt + 1, 0);
  for (int k = 0; k < nt; ++k) slot_base[k + 1] = slot_base[k] + 1 + (int32_t)col[k].size();
  sv.nslots = slot_base[nt];
  auto slot_of = [&](int i, int k) -> int32_t {   // new indices, i >= k; -1 if the tile is structurally zero
    if (i == k) return slot_base[k];
    auto it = std::lower_bound(col[k].begin(), col[k].end(), i);
    return (it != col[k].end() && *it == i) ? slot_base[k] + 1 + (int32_t)(it - col[k].begin()) : -1;
  };
  s->last_diag_slot = slot_base[iperm[nt - 1]];      // the (possibly padded) last tile of the natural order
  tick("symbolic");
  // ---- sharded factorisation: does every rank's share of the points respect the cut? ----
  // (part of a column = the rank whose subtree it belongs to, -1 = a separator the ranks share.)  Every kind of block the solver takes
  // is in (rounds 4 - 6): motion priors are shared out like the frames (below), GoodPosePrior / SphericalPrior terms go to the rank whose
  // part holds the pose, a free interFrameRatio has its column's forward solve run part by part — and SEVERAL intrinsics blocks (a 9-block
  // per frame, CeresHandler.h:256-264,273-280; round 6) need nothing of their own: a block's pseudo frames sit in a tile that is adjacent to the
  // tiles of exactly the frames seen through it, so the dissection puts it in those frames' part or in a separator, every point seen through
  // the block is owned by that part's rank (rsba_partition_points builds the same graph), and the tile's replicated terms — damping, identity
  // padding, the gradient after exchange (1) — follow frame_lead like any frame tile's.
  std::vector<int32_t> cpart(nt, -1);
  bool sharded = want_parts && tord.parts_ok;
  if (const char* e = std::getenv("RSBA_SHARDED")) sharded = sharded && e[0] != '0';   // A/B switch
  if (want_parts) {
    double bad = sharded ? 0.0 : 1.0;
    for (int64_t i = 0; i < N && bad == 0.0; ++i) { const int p = tord.part_of[of[i] / FT]; if (p >= 0 && p != h->rank) bad = 1.0; }
    if (local_fail) bad = 2.0;   // this rank cannot build its plan at all: every rank gives up together
    // every rank must take the same form: one all-reduce (max) of the verdicts — through the handle's cost slot (allocated with the
    // handle, rewritten by every evaluation): no allocation here that could fail on one rank and leave the others waiting
    double* d_bad = h->d_cost2;
    hipError_t e = hipMemcpyAsync(d_bad, &bad, sizeof bad, hipMemcpyHostToDevice, h->stream);
    const int32_t rcx = exchange(h, d_bad, 1, 1, RSBA_EXCHANGE_SETUP);
    if (e == hipSuccess && rcx == RSBA_OK) e = hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess && rcx == RSBA_OK) e = hipStreamSynchronize(h->stream);
    if (rcx) return rcx;
    if (e != hipSuccess) return rsba_set_error(e == hipErrorOutOfMemory ? RSBA_ERR_OUT_OF_MEMORY : RSBA_ERR_HIP, hipGetErrorString(e));
    if (local_fail) return rsba_set_error(local_fail, local_why);
    if (bad >= 2.0) return rsba_set_error(RSBA_ERR_UNSUPPORTED, "another rank could not build its plan (its own call says why)");
    sharded = bad == 0.0;
  }
  s->sharded = sharded;
  if (sharded) for (int j = 0; j < nt; ++j) cpart[j] = tord.part_of[perm[j]];
  if (sharded && !h->prior_frames.empty() && !h->prior_split) {
    // The prior between frames f and f - 1 (CeresHandler.h:147-185) adds to U_f, U_f-1, the (f, f-1) block and both gradients: it
    // belongs to the rank that owns the part either frame is in (two adjacent frames are never in two different parts: their tiles
    // are the same or neighbours), rank 0 when both sit in separators — so every tile of a part's columns stays complete on its rank.
    auto part_of_frame = [&](int f) { return tord.part_of[f / FT]; };
    std::vector<int32_t> own((size_t)FR + 1, 0);
    int mine = 0;
    for (int32_t f : h->prior_frames) {
      int r = part_of_frame(f);
      if (r < 0) r = part_of_frame(f - 1);
      if (r < 0) r = 0;
      if (r == h->rank) { own[f] = 1; ++mine; }
    }
    int32_t* d_own = nullptr;
    if (int32_t rc_ = s_upload(s, &d_own, own)) return rc_;
    h->prior_of_all = h->dp.prior_of; h->prior_invalid_all = h->prior_invalid;   // (d_own belongs to this plan: rsba_destroy_solver restores the handle's own table)
    h->dp.prior_of = d_own;
    if (h->prior_invalid > 0) h->prior_invalid = mine;
    h->prior_split = true;
  }
  // level schedule: column j is ready once every column of row[j] is done
  std::vector<int32_t> level(nt, 0);
  int nlev = 0;
  for (int j = 0; j < nt; ++j) { int l = 0; for (int32_t k : row[j]) l = std::max(l, level[k] + 1); level[j] = l; nlev = std::max(nlev, l + 1); }
  s->nlev = nlev;
  std::vector<std::vector<int32_t>> lev_cols(nlev);
  for (int j = 0; j < nt; ++j) lev_cols[level[j]].push_back(j);
  // flattened work lists (device copies below)
  //   diag item d: column j -> {slot_jj, old tile of j}, contributors k in row[j]: {slot_jk, old tile of k}
  //   sub  item t: tile (i,j) -> {slot_ij, slot_jj}, contributors k in row[j] & row[i]: {slot_ik, slot_jk}
  //   back item  : column j -> {slot_jj, old tile of j}, tiles i in col[j]: {slot_ij, old tile of i}
  // A tile with many contributors (separator rows are dense across their segments) would serialise dozens
  // of 48^3 products in one workgroup; its contributor list is cut into chunks of kChunk that separate
  // workgroups reduce to partial tiles (fixed split, fixed order: still deterministic), summed by the
  // factor kernels.  upd items: {kind 0 diag / 1 sub, list begin, list end, scratch slot}.
  // The owner of a tile keeps the last kTail contributors — the ones from the latest levels, the list being sorted by
  // level — for itself: on the critical path a freshly finished tile is then multiplied by its consumer directly
  // instead of passing through a partial tile in HBM (two memory round trips less per level).
  // An UPDATE task becomes runnable one level after its last contributor, which is where it enters the ticket order.
  // Look-ahead on the critical path: the LAST contributor k* of column j finishes one level before j and would reach the
  // DIAG task through the SUB task of tile (j, k*) — W_k* out, L_jk* = X W_k*^T back, two hand-offs through memory.  The DIAG
  // task takes X = S_jk* - (older updates) — which the SUB task publishes as soon as it has it, long before W_k* exists — and
  // multiplies by W_k* itself the moment it lands (one hand-off).  The SUB task still writes L_jk* for everyone else; both
  // follow the same arithmetic, bit for bit.
  bool fuse_last = true;
  if (const char* e = std::getenv("RSBA_CHOL_FUSE")) fuse_last = e[0] != '0';
  std::vector<int32_t> sub_base(nt, 0);   // first SUB item of each column (items of a column follow col[] order)
  int kChunk = 12, kTail = 2;   // (swept on C4 / C5 after the look-ahead: fewer, longer UPDATE tasks and a short own share — 2.45 -> 2.36 ms per C4 iteration)
  if (const char* e = std::getenv("RSBA_CHOL_TAIL")) kTail = std::max(1, std::atoi(e));       // tuning aids
  if (const char* e = std::getenv("RSBA_CHOL_CHUNK")) kChunk = std::max(kTail, std::atoi(e));
  s->lev_diag_ptr.assign(1, 0); s->lev_sub_ptr.assign(1, 0); s->lev_upd_ptr.assign(1, 0);
  s->diag_ptr.assign(1, 0); s->sub_ptr.assign(1, 0); s->back_ptr.assign(1, 0);
  int parts = 0;
  std::vector<std::vector<int32_t>> upd_by_level(nlev);
  std::vector<int32_t> klev;   // level of each contributor of the list being built
  std::vector<int32_t> upd_owner, diag_colnew, sub_colnew; int cur_part = -1;   // rank that runs each UPDATE item (-1: every rank); column (new order) of each DIAG / SUB item
  std::vector<std::vector<int32_t>> asm_of_slot((size_t)sv.nslots);   // per separator tile of a sharded plan: {part, partial tile} of the parts' UPDATE items that subtract from it
  for (int l = 0; l < nlev; ++l) {
    const size_t diag0 = s->diag_info.size() / 4, sub0 = s->sub_info.size() / 4;
    // contributors [p0, p1) of the list being built (levels in klev, klev[0] belonging to list position lbase); the first nb of them
    // — sharded plans, separator items only — are the columns of the ranks' parts, grouped by part (pk = their parts): each part's
    // share is cut into UPDATE items that ITS rank runs, whatever their number; {part0, nparts} covers every partial tile of the item
    // — what the replicated factorisation subtracts —, info_sh the ones of the separators' own columns only — what is left to
    // subtract after the exchange has summed the rest in (assemble_top)
    auto chunk_it = [&](int kind, int32_t p0, int32_t p1, std::vector<int32_t>& info, std::vector<int32_t>& own, int32_t lbase, int nb, const std::vector<int32_t>& pk,
                        std::vector<int32_t>* info_sh, int32_t tile_slot) {
      const int32_t first_part = parts;
      int cnt = 0;
      for (int32_t q = p0; q < p0 + nb;) {   // the parts' shares
        int32_t qe = q; while (qe < p0 + nb && pk[qe - p0] == pk[q - p0]) ++qe;
        for (int32_t c = q; c < qe; c += kChunk) {
          const int32_t c1 = std::min(c + kChunk, qe);
          upd_by_level[klev[c1 - 1 - lbase] + 1].push_back((int32_t)(s->upd.size() / 4));
          upd_owner.push_back(pk[q - p0]); asm_of_slot[tile_slot].push_back(pk[q - p0]); asm_of_slot[tile_slot].push_back(parts);
          s->upd.push_back(kind); s->upd.push_back(c); s->upd.push_back(c1); s->upd.push_back(parts++); ++cnt;
        }
        q = qe;
      }
      const int nbparts = cnt;
      p0 += nb;
      if (p1 - p0 <= kChunk) own.push_back(p0);
      else {
        const int32_t own0 = p1 - kTail;
        for (int32_t q = p0; q < own0; q += kChunk) {
          const int32_t q1 = std::min(q + kChunk, own0);
          upd_by_level[klev[q1 - 1 - lbase] + 1].push_back((int32_t)(s->upd.size() / 4));
          upd_owner.push_back(cur_part);
          s->upd.push_back(kind); s->upd.push_back(q); s->upd.push_back(q1); s->upd.push_back(parts++); ++cnt;
        }
        own.push_back(own0);
      }
      info.push_back(cnt ? first_part : 0); info.push_back(cnt);
      if (info_sh) { info_sh->push_back(cnt - nbparts ? first_part + nbparts : 0); info_sh->push_back(cnt - nbparts); }
    };
    // contributors of a separator item of a sharded plan: the parts' columns first, part by part, then the separators' own — each
    // group in the order its columns finish
    auto split_parts = [&](std::vector<int32_t>& ks, std::vector<int32_t>& pk) {
      std::stable_sort(ks.begin(), ks.end(), [&](int32_t x, int32_t y) { const unsigned px = (unsigned)cpart[x], py = (unsigned)cpart[y]; return px < py; });   // (-1 = separators: last)
      pk.clear();
      for (int32_t k : ks) if (cpart[k] >= 0) pk.push_back(cpart[k]);
      return (int)pk.size();
    };
    for (int32_t j : lev_cols[l]) {
      std::vector<int32_t> rj(row[j]);   // contributors in the order they finish
      std::stable_sort(rj.begin(), rj.end(), [&](int32_t x, int32_t y) { return level[x] < level[y]; });
      const bool topcol = sharded && cpart[j] < 0;
      cur_part = cpart[j];
      std::vector<int32_t> pk;
      const int nbj = topcol ? split_parts(rj, pk) : 0;
      s->diag_info.push_back(slot_base[j]); s->diag_info.push_back(perm[j]);
      if (sharded) { s->diag_info_sh.push_back(slot_base[j]); s->diag_info_sh.push_back(perm[j]); }
      diag_colnew.push_back(j);
      const int32_t dp0 = (int32_t)(s->diag_list.size() / 2);
      klev.clear();
      for (int32_t k : rj) { s->diag_list.push_back(slot_of(j, k)); s->diag_list.push_back(perm[k]); klev.push_back(level[k]); }
      s->diag_ptr.push_back((int32_t)(s->diag_list.size() / 2));
      {
        // the contributors a second right-hand side's forward task sums over (FWD2 / FWD2P, cholesky.hip): all of them; in a sharded plan a
        // separator column takes THIS rank's part in launch A (its share travels) and the separators' own columns in launch B
        const int32_t dp1 = (int32_t)(s->diag_list.size() / 2);
        s->fwd_full.push_back(dp0); s->fwd_full.push_back(dp1);
        if (sharded) {
          int32_t lo = dp0, hi = dp0;
          if (topcol) { int q = 0; while (q < nbj && pk[q] != h->rank) ++q; lo = dp0 + q; while (q < nbj && pk[q] == h->rank) ++q; hi = dp0 + q; }
          else { lo = dp0; hi = dp1; }
          s->fwd_a.push_back(lo); s->fwd_a.push_back(hi);
          s->fwd_b.push_back(topcol ? dp0 + nbj : dp0); s->fwd_b.push_back(topcol ? dp1 : dp0);
        }
      }
      chunk_it(0, dp0, (int32_t)(s->diag_list.size() / 2), s->diag_info, s->diag_own, dp0, nbj, pk, sharded ? &s->diag_info_sh : nullptr, slot_base[j]);
      if (fuse_last && !rj.empty() && (!topcol || cpart[rj.back()] < 0)) {   // (a separator column of a sharded plan never takes a part's column by the hand: it lives on another rank)
        const int32_t ks = rj.back();
        const int32_t fs = sub_base[ks] + (int32_t)(std::lower_bound(col[ks].begin(), col[ks].end(), j) - col[ks].begin());
        s->diag_fuse.push_back(fs);
        s->sub_pub[fs] = (int32_t)(s->diag_info.size() / 4) - 1;
      } else s->diag_fuse.push_back(-1);
      sub_base[j] = (int32_t)(s->sub_info.size() / 4);
      s->back_info.push_back(slot_base[j]); s->back_info.push_back(perm[j]);
      for (auto it = col[j].rbegin(); it != col[j].rend(); ++it) { s->back_list.push_back(slot_of(*it, j)); s->back_list.push_back(perm[*it]); }   // bottom-up: the order the y_i arrive in
      s->back_ptr.push_back((int32_t)(s->back_list.size() / 2));
      for (int32_t i : col[j]) {
        s->sub_info.push_back(slot_of(i, j)); s->sub_info.push_back(slot_base[j]); s->sub_col.push_back(perm[j]); s->sub_pub.push_back(-1);
        if (sharded) { s->sub_info_sh.push_back(slot_of(i, j)); s->sub_info_sh.push_back(slot_base[j]); }
        sub_colnew.push_back(j);
        const int32_t sp0 = (int32_t)(s->sub_list.size() / 2);
        // k in row[j] with tile (i,k) present (rj: for a separator column of a sharded plan the parts' columns first, part by part)
        klev.clear(); pk.clear();
        for (int32_t k : rj) { const int32_t sik = slot_of(i, k); if (sik >= 0) { s->sub_list.push_back(sik); s->sub_list.push_back(slot_of(j, k)); klev.push_back(level[k]); if (topcol && cpart[k] >= 0) pk.push_back(cpart[k]); } }
        s->sub_ptr.push_back((int32_t)(s->sub_list.size() / 2));
        chunk_it(1, sp0, (int32_t)(s->sub_list.size() / 2), s->sub_info, s->sub_own, sp0, (int)pk.size(), pk, sharded ? &s->sub_info_sh : nullptr, slot_of(i, j));
      }
    }
    s->lev_diag_ptr.push_back((int32_t)(s->diag_info.size() / 4));
    s->lev_sub_ptr.push_back((int32_t)(s->sub_info.size() / 4));
    s->lev_upd_ptr.push_back((int32_t)(s->upd.size() / 4));
    (void)diag0; (void)sub0;
  }
  // A free interFrameRatio brings a second right-hand side (its column of the normal equations): z2 = L^-1 b is formed by FWD2 tasks
  // right behind the DIAG tasks of their columns — light tasks for workgroups the factorisation leaves idle — and ONE ETA task between
  // the forward and the backward phase turns both forward solves into the ratio's step (solver_state.hpp; cholesky.hip)
  const bool two_rhs = h->prior_free && !h->prior_frames.empty();
  s->two_rhs = two_rhs;
  for (int l = 0; l < nlev; ++l) {
    for (int32_t u : upd_by_level[l]) { s->tasks.push_back(kTaskUpdate); s->tasks.push_back(u); }
    for (int d = s->lev_diag_ptr[l]; d < s->lev_diag_ptr[l + 1]; ++d) { s->tasks.push_back(kTaskDiag); s->tasks.push_back(d); }
    if (two_rhs) for (int d = s->lev_diag_ptr[l]; d < s->lev_diag_ptr[l + 1]; ++d) { s->tasks.push_back(kTaskFwd2); s->tasks.push_back(d); }
    for (int t = s->lev_sub_ptr[l]; t < s->lev_sub_ptr[l + 1]; ++t) { s->tasks.push_back(kTaskSub); s->tasks.push_back(t); }
  }
  if (two_rhs) { s->tasks.push_back(kTaskEta); s->tasks.push_back(0); }
  for (int l = nlev - 1; l >= 0; --l)
    for (int d = s->lev_diag_ptr[l]; d < s->lev_diag_ptr[l + 1]; ++d) { s->tasks.push_back(kTaskBack); s->tasks.push_back(d); }
  if (sharded) {
    // launch A: this rank's part, forward (the parts' UPDATE items of the separators' tiles included); launch B: the separators,
    // forward and backward, then this rank's part backward
    const int me = h->rank;
    for (int l = 0; l < nlev; ++l) {
      for (int32_t u : upd_by_level[l]) { std::vector<int32_t>& t = upd_owner[u] == me ? s->tasks_a : s->tasks_b; if (upd_owner[u] == me || upd_owner[u] < 0) { t.push_back(kTaskUpdate); t.push_back(u); } }
      for (int d = s->lev_diag_ptr[l]; d < s->lev_diag_ptr[l + 1]; ++d) { const int p = cpart[diag_colnew[d]]; if (p == me) { s->tasks_a.push_back(kTaskDiag); s->tasks_a.push_back(d); } else if (p < 0) { s->tasks_b.push_back(kTaskDiag); s->tasks_b.push_back(d); } }
      if (two_rhs) for (int d = s->lev_diag_ptr[l]; d < s->lev_diag_ptr[l + 1]; ++d) { const int p = cpart[diag_colnew[d]]; if (p == me) { s->tasks_a.push_back(kTaskFwd2); s->tasks_a.push_back(d); } else if (p < 0) { s->tasks_b.push_back(kTaskFwd2); s->tasks_b.push_back(d); } }
      for (int t = s->lev_sub_ptr[l]; t < s->lev_sub_ptr[l + 1]; ++t) { const int p = cpart[sub_colnew[t]]; if (p == me) { s->tasks_a.push_back(kTaskSub); s->tasks_a.push_back(t); } else if (p < 0) { s->tasks_b.push_back(kTaskSub); s->tasks_b.push_back(t); } }
    }
    if (two_rhs) {
      // launch A ends with what travels: this rank's part's share of every separator column's second right-hand side, and of the two dots;
      // launch B's ETA task sits between its forward and its backward phase
      for (int d = 0; d < (int)diag_colnew.size(); ++d) if (cpart[diag_colnew[d]] < 0) { s->tasks_a.push_back(kTaskFwd2P); s->tasks_a.push_back(d); }
      s->tasks_a.push_back(kTaskEta); s->tasks_a.push_back(0);
      s->tasks_b.push_back(kTaskEta); s->tasks_b.push_back(0);
    }
    for (int l = nlev - 1; l >= 0; --l)
      for (int d = s->lev_diag_ptr[l]; d < s->lev_diag_ptr[l + 1]; ++d) { const int p = cpart[diag_colnew[d]]; if (p == me || p < 0) { s->tasks_b.push_back(kTaskBack); s->tasks_b.push_back(d); } }
  }
  tick("tasks");
  // Chunks of the Schur kernel (one workgroup each): at most kSchurChunk consecutive entries of one tile pair, numbered tile
  // pair by tile pair in (I, J) order — the pairs of one tile row, which read the same A_j(I) groups, next to each other; the
  // kernel's blockIdx -> chunk map keeps consecutive chunks on one XCD.  What was measured around this choice (C4, round 3):
  //   * this numbering: 40 % L2 hits, 2.5 GB from the fabric per launch, 4 975 chunks, kernel 0.53 ms — its MFMA loops run at
  //     88 % of the matrix pipe (two waves per SIMD), the rest is tables / epilogue (14 %) and the ramp-down of the launch;
  //   * point-block-major (RSBA_SCHUR_BLOCK=<points>: every tile pair cut at the same blocks of consecutive points, all pairs of
  //     a block next to each other — rsba numbers tracks in the order the video first sees them, so a block spans a few tiles
  //     and an XCD's L2 holds its records): 80 % L2 hits, 0.76 GB from the fabric, but 8 700 shorter chunks: 0.60 ms;
  //   * equal parts of up to 1024 entries launched longest first, wherever their records are: 3 100 chunks, 0.67 ms — the
  //     loops then wait for memory (2.2 us per group of four entries instead of 1.5).
  // Per tile pair the chunk ids are listed in entry order for the merge kernel.
  std::vector<int32_t> chunk_tp, chunk_n; std::vector<int64_t> chunk_e0;
  std::vector<std::vector<int32_t>> pair_chunks(ntp);
  //   * round 4, 4k cameras (1 001 tile columns, 29 678 chunks): there the kernel pulls 20.5 GB from the fabric in 3.6 ms — the
  //     point-block-major numbering with blocks of 2 048 points is worth 3 % (3.59 -> 3.47 ms, 35 502 chunks), so it is the default
  //     above 500 tile columns — as long as it does not multiply the chunks (point numbers that do not follow the video would).
  auto number_chunks = [&](int64_t kBlock) {
    chunk_tp.clear(); chunk_n.clear(); chunk_e0.clear();
    for (auto& pc : pair_chunks) pc.clear();
    if (dev_plan && kBlock < M) {
      // the device plan's segments — maximal runs of one tile pair's entries inside one block of points, in entry order — in the order of
      // the walk below: block by block, inside a block the pairs in (I, J) order, every segment cut into chunks
      const size_t nseg = dpo.seg_pair.size();
      std::vector<int32_t> order(nseg);
      for (size_t q = 0; q < nseg; ++q) order[q] = (int32_t)q;
      std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return dpo.seg_block[a] != dpo.seg_block[b] ? dpo.seg_block[a] < dpo.seg_block[b] : dpo.seg_pair[a] < dpo.seg_pair[b]; });
      for (int32_t q : order) {
        const int tp_ = dpo.seg_pair[q];
        const int64_t q0 = dpo.seg_start[q], q1 = ((size_t)q + 1 < nseg && dpo.seg_pair[q + 1] == tp_) ? dpo.seg_start[q + 1] : tp_ptr[tp_ + 1];
        for (int64_t a = q0; a < q1; a += kSchurChunk) {
          pair_chunks[tp_].push_back((int32_t)chunk_tp.size());
          chunk_tp.push_back(tp_); chunk_e0.push_back(a); chunk_n.push_back((int32_t)std::min<int64_t>(kSchurChunk, q1 - a));
        }
      }
      return;
    }
    // per tile pair the cursor into its entry list (entries are in point order); pairs that still have entries, in (I, J) order
    std::vector<int64_t> cursor(tp_ptr.begin(), tp_ptr.end() - 1);
    std::vector<int32_t> live; live.reserve(64);
    int next_pair = 0;           // pairs enter `live` when the block reaches their first point
    std::vector<int32_t> by_first(ntp);
    for (int t = 0; t < ntp; ++t) by_first[t] = t;
    // (the device plan brings the entries' points to the host only for a numbering by point blocks: without them every pair is live from the first — and only — block)
    auto first_point = [&](int t) { return tp_ptr[t] < tp_ptr[t + 1] ? (ent_pt.empty() ? 0 : (ent_pt[tp_ptr[t]] & 0x7fffffff)) : std::numeric_limits<int32_t>::max(); };
    std::stable_sort(by_first.begin(), by_first.end(), [&](int a, int b) { return first_point(a) < first_point(b); });
    for (int64_t p0 = 0; p0 < M; p0 += kBlock) {
      const int64_t p1 = std::min<int64_t>(p0 + kBlock, M);
      while (next_pair < ntp && first_point(by_first[next_pair]) < p1) live.push_back(by_first[next_pair++]);
      std::sort(live.begin(), live.end());   // (I, J) order inside the block: the pairs of one tile row next to each other
      size_t keep = 0;
      for (size_t x = 0; x < live.size(); ++x) {
        const int tp_ = live[x];
        int64_t q = cursor[tp_];
        const int64_t qend = tp_ptr[tp_ + 1];
        if (p1 >= M) q = qend;
        else q = std::lower_bound(ent_pt.begin() + q, ent_pt.begin() + qend, (int32_t)p1, [](int32_t e, int32_t p) { return (e & 0x7fffffff) < p; }) - ent_pt.begin();   // (a pair's entries are in point order)
        for (int64_t q0 = cursor[tp_]; q0 < q; q0 += kSchurChunk) {
          pair_chunks[tp_].push_back((int32_t)chunk_tp.size());
          chunk_tp.push_back(tp_); chunk_e0.push_back(q0); chunk_n.push_back((int32_t)std::min<int64_t>(kSchurChunk, q - q0));
        }
        cursor[tp_] = q;
        if (q < qend) live[keep++] = tp_;
      }
      live.resize(keep);
    }
  };
  {
    int64_t pair_major = 0;
    for (int t = 0; t < ntp; ++t) pair_major += (tp_ptr[t + 1] - tp_ptr[t] + kSchurChunk - 1) / kSchurChunk;
    int64_t kBlock = nt > 500 ? 2048 : std::max<int64_t>(M, 1);
    if (const char* e = std::getenv("RSBA_SCHUR_BLOCK")) kBlock = std::atoi(e) > 0 ? std::max(16, std::atoi(e)) : std::max<int64_t>(M, 1);   // tuning aid (0: tile pair by tile pair)
    number_chunks(kBlock);
    if (kBlock < M && !std::getenv("RSBA_SCHUR_BLOCK") && (int64_t)chunk_tp.size() > pair_major + pair_major / 3) number_chunks(std::max<int64_t>(M, 1));
  }
  // A pair with very many chunks (the diagonal pair of the intrinsics pseudo tile has one per 512 points of the whole
  // problem) would be summed by a single workgroup of the merge kernel: its chunk list is pre-reduced in groups of
  // kMergeGroup, one workgroup each, into the partial tile of the group's first chunk, and only those heads go to the merge.
  const int kMergeGroup = 32;
  std::vector<int32_t> tp_chunk0(ntp + 1, 0), tp_chunk_list, pm_ptr(1, 0), pm_list;
  for (int t = 0; t < ntp; ++t) {
    tp_chunk0[t] = (int32_t)tp_chunk_list.size();
    const std::vector<int32_t>& pc = pair_chunks[t];
    if ((int)pc.size() <= kMergeGroup) { tp_chunk_list.insert(tp_chunk_list.end(), pc.begin(), pc.end()); continue; }
    for (size_t g = 0; g < pc.size(); g += kMergeGroup) {
      const size_t g1 = std::min(pc.size(), g + kMergeGroup);
      tp_chunk_list.push_back(pc[g]);
      if (g1 - g > 1) { pm_list.insert(pm_list.end(), pc.begin() + g, pc.begin() + g1); pm_ptr.push_back((int32_t)pm_list.size()); }
    }
  }
  tp_chunk0[ntp] = (int32_t)tp_chunk_list.size();
  sv.npremerge = (int)pm_ptr.size() - 1;
  sv.nchunk = (int)chunk_tp.size(); sv.ntp = ntp; sv.FT = FT;
  { const char* e = std::getenv("RSBA_SCHUR_LINEAR"); sv.schur_linear = e && e[0] == '1'; }
  { const char* e = std::getenv("RSBA_SCHUR_VARIANT"); sv.schur_variant = e ? std::atoi(e) : 0; if (!kTestHooks && sv.schur_variant >= 4) sv.schur_variant = 0; }   // (4 / 5: ablations, instrumented build only)
  sv.schur_trace = nullptr;
  if (std::getenv("RSBA_SCHUR_TRACE")) { if (int32_t rc_ = s_alloc(s, &sv.schur_trace, 8 * (size_t)std::max(sv.nchunk, 1))) return rc_; }
  std::vector<uint8_t> has_prior((size_t)FR + 1, 0);
  for (int32_t f : h->prior_frames) has_prior[f] = 1;
  const int64_t ucross_base = ((int64_t)FR + (int64_t)NPF * FR + (int64_t)NIB * NPF * NPF) * CD * CD;   // behind the J^T J blocks in sv.U
  std::vector<int32_t> tp_dst(ntp); std::vector<uint8_t> tp_trans(ntp, 0);
  std::vector<int64_t> tp_add((size_t)ntp * FT * FT, -1);
  for (int t = 0; t < ntp; ++t) {
    const int I = tp_I[t], J = tp_J[t], pI = iperm[I], pJ = iperm[J];
    // tile of the pair in the permuted order; if the ordering swapped the two tiles it is stored transposed
    if (pI >= pJ) tp_dst[t] = slot_of(pI, pJ); else { tp_dst[t] = slot_of(pJ, pI); tp_trans[t] = 1; }
    // which J^T J block enters block (a,b) of this tile: U layout [frames][pseudo x frames][pseudo x pseudo]
    for (int x = 0; x < FT; ++x) for (int y = 0; y < FT; ++y) {
      const int a = I * FT + x, b = J * FT + y;
      if (a >= F || b >= F || a < b) continue;
      int64_t add = -1;
      if (a < FR) {
        if (a == b) add = (int64_t)a * CD * CD;
        else if (b == a - 1 && has_prior[a]) add = ucross_base + (int64_t)a * CD * CD;   // motion prior block (a, a-1)
      }
      else {
        const int ca = (a - FR) / NPF, va = (a - FR) % NPF;     // pseudo frame va of intrinsics block ca
        if (b < FR) { if (intr_of(b) == ca) add = ((int64_t)FR + (int64_t)va * FR + b) * CD * CD; }   // only with the frames that use the block
        else if ((b - FR) / NPF == ca) add = ((int64_t)FR + (int64_t)NPF * FR + ((int64_t)ca * NPF + va) * NPF + (b - FR) % NPF) * CD * CD;
      }
      tp_add[((size_t)t * FT + x) * FT + y] = add;
    }
  }

  tick("chunks");
  // which coordinates belong to the reduced program (for |x| and |step|): blocks that are not constant
  // and are touched by at least one residual block
  std::vector<double> inprog_pose((size_t)FR * CD, 0.0), inprog_intr((size_t)std::max(NIB * NPF, 1) * CD, 0.0);
  std::vector<double>& inprog_point = scr.inprog_point; inprog_point.assign((size_t)M * 3, 0.0);
  int nfree = 0;
  const bool lead = h->rank == 0;
  sv.lead = lead;
  std::vector<uint8_t> has_pose_prior((size_t)FR, 0);
  for (int32_t b : h->pp_blocks) has_pose_prior[b / dp.P] = 1;
  if (dp.pp_spherical >= 0) has_pose_prior[dp.pp_spherical / dp.P] = 1;
  auto frame_has_obs = [&](int f) {
    if (has_prior[f] || has_prior[f + 1] || has_pose_prior[f]) return true;   // touched by a motion prior / pose prior block
    return h->frame_obs_total.empty() ? frame_ptr[f + 1] > frame_ptr[f] : h->frame_obs_total[f] > 0;
  };
  {
    // an intrinsics block is part of the program when it is not constant and a residual block touches it
    std::vector<uint8_t> touched((size_t)std::max(NIB, 1), 0);
    for (int f = 0; f < FR && NIB > 0; ++f) if (h->frame_obs_total.empty() ? frame_ptr[f + 1] > frame_ptr[f] : h->frame_obs_total[f] > 0) touched[intr_of(f)] = 1;
    for (int c = 0; c < NIB; ++c) if (lead && touched[c] && h->mask_intr[(size_t)c * 9] != 0.0)
      for (int k = 0; k < 9; ++k) { inprog_intr[((size_t)c * NPF + k / CD) * CD + k % CD] = 1.0; ++nfree; }
  }
  for (int f = 0; f < FR; ++f) for (int q = 0; q < dp.P; ++q) {
    bool any_free = false;
    for (int k = 0; k < 6; ++k) any_free = any_free || h->mask_pose[((size_t)f * dp.P + q) * 6 + k] != 0.0;
    if (lead && any_free && frame_has_obs(f)) for (int k = 0; k < 6; ++k) { inprog_pose[((size_t)f * dp.P + q) * 6 + k] = 1.0; nfree += h->mask_pose[((size_t)f * dp.P + q) * 6 + k] != 0.0; }
  }
  for (int j = 0; j < M; ++j) if (h->mask_point[(size_t)j * 3] != 0.0 && point_ptr[j + 1] > point_ptr[j]) { for (int k = 0; k < 3; ++k) inprog_point[(size_t)j * 3 + k] = 1.0; nfree += 3; }
  s->num_reduced_params = nfree;
  {
    // residual blocks whose parameter blocks are all constant leave the program: every observation of a free point stays; those
    // of a constant point stay where the frame's poses or its intrinsics block are free (per point, not per observation)
    std::vector<uint8_t> frame_const((size_t)FR, 1);
    for (int f = 0; f < FR; ++f) {
      bool c = NIB == 0 || h->mask_intr[(size_t)intr_of(f) * 9] == 0.0;
      for (int k = 0; k < CD && c; ++k) c = h->mask_pose[(size_t)f * CD + k] == 0.0;
      frame_const[f] = c;
    }
    int64_t nred = 0;
    for (int j = 0; j < M; ++j) {
      if (h->mask_point[(size_t)j * 3] != 0.0) { nred += point_ptr[j + 1] - point_ptr[j]; continue; }
      for (int64_t x = point_ptr[j]; x < point_ptr[j + 1]; ++x) nred += !frame_const[slot_frame[x]];
    }
    s->num_priors_reduced = 0;
    if (lead) for (int32_t f : h->prior_frames) {
      bool all_const = !h->prior_free;
      for (int k = 0; k < 24 && all_const; ++k) all_const = h->mask_pose[(size_t)(f - 1) * CD + k] == 0.0;
      s->num_priors_reduced += !all_const;
    }
    if (lead) {   // per-pose priors: a GoodPosePrior always keeps its free priorPoses block; a SphericalPrior on a constant pose is dropped
      s->num_priors_reduced += (int)h->pp_blocks.size();
      nfree += 6 * (int)h->pp_blocks.size();
      s->num_reduced_params = nfree;
      if (dp.pp_spherical >= 0) { bool all_const = true; for (int k = 0; k < 6; ++k) all_const = all_const && h->mask_pose[(size_t)dp.pp_spherical * 6 + k] == 0.0; s->num_priors_reduced += !all_const; }
    }
    s->num_reduced_blocks = (int)nred;
  }

  int32_t rc;
  sv.tile_factored = nullptr;
  if (factored) up.upload_const(&sv.tile_factored, tile_factored);
  sv.all_real_factored = factored ? 1 : 0;
  for (int t = 0; t < nt && (int64_t)t * FT < FR; ++t) if (!tile_factored[t]) sv.all_real_factored = 0;
  sv.fused_sweep = 0;   // (set once the plan knows its virtual groups, below)
  up.upload_const(&sv.tp_I, tp_I);
  up.upload_const(&sv.tp_J, tp_J);
  up.upload_const(&sv.tp_ptr, tp_ptr);
  up.upload_const(&sv.inprog_pose, inprog_pose);
  up.upload_const_ref(&sv.inprog_point, inprog_point);
  up.upload_const(&sv.inprog_intr, inprog_intr);
  std::vector<int32_t> ifp((size_t)NIB + 1, 0), ifl;       // (alive until the uploads have finished)
  std::vector<int64_t> point_vgroup((NIB == 1 && !dev_plan) ? (size_t)M : 0, -1);
  {
    for (int f = 0; f < FR && NIB > 0; ++f) ifp[intr_of(f) + 1]++;
    for (int c = 0; c < NIB; ++c) ifp[c + 1] += ifp[c];
    ifl.resize(NIB > 0 ? FR : 0);
    { std::vector<int32_t> fill(ifp.begin(), ifp.end() - 1); for (int f = 0; f < FR && NIB > 0; ++f) ifl[fill[intr_of(f)]++] = f; }
    up.upload_const(&sv.intr_frame_ptr, ifp);
    up.upload_const(&sv.intr_frame_list, ifl);
    if (dev_plan) { sv.vgroup_point = dpo.vgroup_point; sv.vgroup_intr = dpo.vgroup_intr; sv.point_vgroup = dpo.point_vgroup; }
    else {
      up.upload_const_ref(&sv.vgroup_point, vgroup_point);
      up.upload_const_ref(&sv.vgroup_intr, vgroup_intr);
      for (int j = 0; j < M && NIB == 1; ++j) if (vgroup_ptr[j + 1] > vgroup_ptr[j]) point_vgroup[j] = vgroup_ptr[j];
      up.upload_const(&sv.point_vgroup, point_vgroup);
    }
  }
  up.upload_const(&sv.chunk_tp, chunk_tp);
  up.upload_const(&sv.chunk_e0, chunk_e0);
  up.upload_const(&sv.tp_chunk0, tp_chunk0);
  up.upload_const(&sv.tp_chunk_list, tp_chunk_list);
  up.upload_const(&sv.chunk_n, chunk_n);
  std::vector<int4> chunk_info(chunk_tp.size());
  for (size_t c = 0; c < chunk_tp.size(); ++c) {
    const int I_ = tp_I[chunk_tp[c]], J_ = tp_J[chunk_tp[c]];
    chunk_info[c] = int4{(int)(uint32_t)(chunk_e0[c] & 0xffffffff), (int)(chunk_e0[c] >> 32), chunk_n[c], (I_ == J_ ? 1 : 0) | (tile_factored[I_] ? 2 : 0) | (tile_factored[J_] ? 4 : 0)};
  }
  up.upload_const(&sv.chunk_info, chunk_info);
  up.upload_const(&sv.pm_ptr, pm_ptr);
  up.upload_const(&sv.pm_list, pm_list);
  up.upload_const(&sv.tp_dst, tp_dst);
  std::vector<int32_t> exch_slots(tp_dst);   // (a tile pair has a packed tile of its own: distinct slots; ascending = the order they sit in memory)
  std::sort(exch_slots.begin(), exch_slots.end());
  s->exch_tiles = (int)exch_slots.size();
  if (h->allreduce) {
    up.upload(&s->exch_slots, exch_slots);
    if ((rc = s_alloc(s, &s->exch_buf, (size_t)exch_slots.size() * kTile * kTile + (size_t)sv.npad))) return rc;
  }
  up.upload_const(&sv.tp_trans, tp_trans);
  up.upload_const(&sv.tp_add, tp_add);
  {
    // one 64-byte line per pair for the merge kernel (kernels_normal.hip, PairDesc): what it used to collect from five arrays in two dependent rounds
    std::vector<int32_t> tp_desc((size_t)ntp * 16, 0);
    for (int t = 0; t < ntp; ++t) {
      int32_t* d = tp_desc.data() + (size_t)t * 16;
      d[0] = tp_I[t]; d[1] = tp_J[t]; d[2] = tp_dst[t];
      d[3] = (tp_trans[t] ? 1 : 0) | (tile_factored[tp_I[t]] ? 2 : 0) | (tile_factored[tp_J[t]] ? 4 : 0);
      d[4] = tp_chunk0[t]; d[5] = tp_chunk0[t + 1];
      for (int u = 0; u < 8; ++u) d[8 + u] = tp_chunk0[t] + u < tp_chunk0[t + 1] ? tp_chunk_list[(size_t)tp_chunk0[t] + u] : -1;
    }
    up.upload_const(&sv.tp_desc, tp_desc);
  }
  if ((rc = s_alloc(s, &sv.schur_part, (size_t)std::max(sv.nchunk, 1) * (kTile * kTile + kTile)))) return rc;
  up.upload_ref(&s->d_upd, s->upd);
  // the write-once cells of the persistent Cholesky driver — factor tiles | partial tiles | W | z, y | published X — live in ONE
  // allocation: one memset re-arms them before a launch (five launches before)
  {
    const size_t nLf = (size_t)sv.nslots * kTile * kTile, nPart = (size_t)std::max(parts, 1) * (kTile * kTile + kTile), nW = (size_t)nt * kTile * kTile, nZ = 3 * (size_t)sv.npad + 8;   // z | y | z2 | {s eta}
    s->ncells = nLf + nPart + nW + nZ + nW;
    s->cell_off[0] = 0; s->cell_off[1] = nLf; s->cell_off[2] = nLf + nPart; s->cell_off[3] = nLf + nPart + nW; s->cell_off[4] = nLf + nPart + nW + nZ;
    for (int b = 0; b < 2; ++b) {
      if ((rc = s_alloc(s, &s->cells[b], s->ncells))) return rc;
      HIP_TRY(hipMemsetAsync(s->cells[b], 0xFF, s->ncells * sizeof(double), h->stream));   // both sets start out armed
    }
    HIP_TRY(dev_stream_acquire(&s->mstream));
    for (int b = 0; b < 2; ++b) HIP_TRY(dev_event_acquire(&s->ev_armed[b], false));
    HIP_TRY(dev_event_acquire(&s->ev_released, false));
    HIP_TRY(dev_event_acquire(&s->ev_fork, false));
    HIP_TRY(dev_event_acquire(&s->ev_join, false));
    double* cells = s->cells[0];
    sv.Lf = cells; sv.chol_part = cells + nLf; sv.Winv = sv.chol_part + nPart; sv.zv = sv.Winv + nW; sv.yv = sv.zv + sv.npad; sv.Xpub = sv.zv + nZ;
    sv.zv2 = nullptr; sv.ceta = nullptr; sv.border2 = nullptr; sv.rt = nullptr;   // (set with the border, below)
  }
  up.upload_ref(&s->d_tasks, s->tasks);
  if ((rc = s_alloc(s, &s->d_dag_sync, 4))) return rc;
  HIP_TRY(hipMemsetAsync(s->d_dag_sync, 0, 4 * sizeof(unsigned int), h->stream));   // (the persistent kernel leaves its counters at zero behind every launch)
  if ((rc = s_alloc(s, &s->zy2, 2 * (size_t)sv.npad))) return rc;
  up.upload_ref(&s->d_diag_info, s->diag_info);
  up.upload_ref(&s->d_diag_ptr, s->diag_ptr);
  up.upload_ref(&s->d_diag_list, s->diag_list);
  up.upload_ref(&s->d_sub_info, s->sub_info);
  up.upload_ref(&s->d_sub_ptr, s->sub_ptr);
  up.upload_ref(&s->d_sub_list, s->sub_list);
  up.upload_ref(&s->d_sub_col, s->sub_col);
  up.upload_ref(&s->d_diag_own, s->diag_own);
  up.upload_ref(&s->d_diag_fuse, s->diag_fuse);
  up.upload_ref(&s->d_sub_pub, s->sub_pub);
  up.upload_ref(&s->d_sub_own, s->sub_own);
  up.upload_ref(&s->d_back_info, s->back_info);
  up.upload_ref(&s->d_back_ptr, s->back_ptr);
  up.upload_ref(&s->d_back_list, s->back_list);
  {
    // second right-hand side: per DIAG item the index of its column among the separators' tile columns (ascending new order: the order of top_tiles below)
    s->diag_toprow.assign(diag_colnew.size(), -1);
    if (sharded) {
      std::vector<int32_t> trow_of_col((size_t)nt, -1);
      int32_t cnt = 0;
      for (int j = 0; j < nt; ++j) if (cpart[j] < 0) trow_of_col[j] = cnt++;
      for (size_t d = 0; d < diag_colnew.size(); ++d) s->diag_toprow[d] = trow_of_col[diag_colnew[d]];
    }
    up.upload_ref(&s->d_fwd_full, s->fwd_full); up.upload_ref(&s->d_diag_toprow, s->diag_toprow);
    if (sharded) { up.upload_ref(&s->d_fwd_a, s->fwd_a); up.upload_ref(&s->d_fwd_b, s->fwd_b); }
  }

  // ---- sharded factorisation: what the exchange between the two launches needs ----
  std::vector<int32_t> top_slots, top_info, asm_ptr(1, 0), asm_list, top_tiles, top_fill;   // (alive until the uploads have finished)
  std::vector<uint8_t> row_mine((size_t)nt, 1);
  std::vector<double> frame_lead;
  sv.frame_lead = nullptr;
  if (sharded) {
    std::vector<uint8_t> has_pair((size_t)sv.nslots, 0);
    for (int t = 0; t < ntp; ++t) has_pair[tp_dst[t]] = 1;
    // the separators' tiles, column by column: {slot, has a tile pair (else: fill only, zero in S), index among the separator tiles of its row of the rhs or -1}
    for (int j = 0; j < nt; ++j) if (cpart[j] < 0) {
      const int trow = (int)top_tiles.size();
      top_tiles.push_back(perm[j]);
      auto add = [&](int32_t slot, int rhs_row) {
        top_slots.push_back(slot); top_info.push_back(has_pair[slot]); top_info.push_back(rhs_row);
        const std::vector<int32_t>& a = asm_of_slot[slot];
        for (size_t q = 0; q + 1 < a.size(); q += 2) if (a[q] == h->rank) asm_list.push_back(a[q + 1]);   // this rank's partial tiles, in list order
        asm_ptr.push_back((int32_t)asm_list.size());
        if (!has_pair[slot]) top_fill.push_back(slot);
      };
      add(slot_base[j], trow);
      for (size_t u = 0; u < col[j].size(); ++u) add(slot_base[j] + 1 + (int32_t)u, -1);   // (rows below a separator column are separators too: fill only reaches ancestors)
    }
    s->ntop_slots = (int)top_slots.size(); s->ntop_tiles = (int)top_tiles.size(); s->ntop_fill = (int)top_fill.size();
    // rows of y this rank contributes to the gather (and whose residual it can check: every tile of those rows is complete here):
    // its own part; the separators' rows come from rank 0
    for (int t = 0; t < nt; ++t) { const int p = tord.part_of[t]; row_mine[t] = p == h->rank || (p < 0 && h->rank == 0); }
    // who adds the replicated terms (damping, gradient, identity padding) of a camera-side frame to its partial S: the rank that
    // owns the frame's part, rank 0 for the separators
    frame_lead.assign((size_t)nt * FT, 0.0);
    for (int a = 0; a < nt * FT; ++a) frame_lead[a] = row_mine[a / FT] ? 1.0 : 0.0;
    up.upload(&s->d_top_slots, top_slots); up.upload(&s->d_top_info, top_info); up.upload(&s->d_asm_ptr, asm_ptr); up.upload(&s->d_asm_list, asm_list);
    up.upload(&s->d_top_tiles, top_tiles); up.upload(&s->d_row_mine, row_mine); up.upload(&s->d_top_fill, top_fill);
    { std::vector<uint8_t> row_check((size_t)nt, 0); for (int t = 0; t < nt; ++t) row_check[t] = tord.part_of[t] == h->rank; up.upload(&s->d_row_check, row_check); }
    { std::vector<uint8_t> row_sep((size_t)nt, 0); for (int t = 0; t < nt; ++t) row_sep[t] = tord.part_of[t] < 0; up.upload(&s->d_row_sep, row_sep); }
    up.upload_const(&sv.frame_lead, frame_lead);
    up.upload_ref(&s->d_tasks_a, s->tasks_a); up.upload_ref(&s->d_tasks_b, s->tasks_b);
    up.upload_ref(&s->d_diag_info_sh, s->diag_info_sh); up.upload_ref(&s->d_sub_info_sh, s->sub_info_sh);
    // (+ with a second right-hand side: the parts' share of the separators' rows of it, and of the two dots — behind the tiles and the rhs rows)
    if ((rc = s_alloc(s, &s->topx_buf, (size_t)s->ntop_slots * kTile * kTile + (size_t)s->ntop_tiles * kTile + (two_rhs ? (size_t)s->ntop_tiles * kTile + 8 : 0)))) return rc;
    if ((rc = s_alloc(s, &s->ybuf, (size_t)sv.npad))) return rc;
    if (two_rhs) HIP_TRY(hipMemsetAsync(s->topx_buf + (size_t)s->ntop_slots * kTile * kTile + (size_t)s->ntop_tiles * kTile, 0, ((size_t)s->ntop_tiles * kTile + 8) * sizeof(double), h->stream));
  }
  HIP_TRY(up.finish());
  tick("uploads");
  const size_t REC = 2 + 2 * (size_t)dp.K;
  h->dp.obs_slot = s->d_obs_slot;
  // The point-side passes recompute the records (lm_record.hpp) from the observations in slot order; problems with several
  // intrinsics parameter blocks (per-frame f.cam) keep the point-major copy.  RSBA_RECORDS=1 forces the copy.
  sv.slot_xy = nullptr; h->dp.rec = nullptr; h->dp.rec_alt = nullptr; h->dp.rec_candidate = 0;   // (recompute: settled with the group layout above)
  if (recompute) {
    double2* sxy = nullptr;
    if ((rc = s_alloc(s, &sxy, (size_t)N))) return rc;
    HIP_TRY(launch_slot_xy(h->dp, sxy, h->stream));
    sv.slot_xy = sxy;
  } else {
    if ((rc = s_alloc(s, &h->dp.rec, (size_t)N * REC))) return rc;
    // ... and a second set for a candidate's records (device_state.hpp: rec_alt): with it the candidate is evaluated in LM mode like everybody
    // else's, and problems that keep records — several intrinsics blocks (per-frame f.cam, CeresHandler.h:260,277) — run the loop whose
    // decisions are taken on the device.  RSBA_RECORDS_ALT=0: one set, candidates residual-only, the host decides (round 5's form; A/B)
    const char* e = std::getenv("RSBA_RECORDS_ALT");
    if (!(e && e[0] == '0')) { if ((rc = s_alloc(s, &h->dp.rec_alt, (size_t)N * REC))) return rc; }
  }
  sv.fused_sweep = sv.slot_xy && !h->dp.calibrated && sv.CD == 12 && sv.all_real_factored != 0 && sv.NPF > 0 && sv.nvgroups > 0 && sv.NIB == 1 && !std::getenv("RSBA_NO_FUSED_SWEEP");
  if (N > 0) {
    // camera (and intrinsics border) blocks inside the evaluation kernel: per (64-observation wave, frame it touches)
    // the 16 x 16 blocks on and below the diagonal of [Ji | Jc | r]^T [Ji | Jc | r]
    const int64_t nwaves = (int64_t)eval_num_blocks(N) * (kEvalBlock / 64);
    std::vector<int32_t> wave_seg_base((size_t)nwaves + 1, 0), frame_rank(FR, 0);
    { int rk = 0; for (int f = 0; f < FR; ++f) { frame_rank[f] = rk; if (frame_ptr[f + 1] > frame_ptr[f]) ++rk; } }
    for (int64_t w = 0; w < nwaves; ++w) {
      const int64_t a = w * 64, b = std::min<int64_t>(a + 64, N);
      wave_seg_base[w + 1] = wave_seg_base[w] + (a < N ? frame_rank[of[b - 1]] - frame_rank[of[a]] + 1 : 0);
    }
    int32_t *d_base = nullptr, *d_rank = nullptr;
    if ((rc = s_upload(s, &d_base, wave_seg_base))) return rc;
    if ((rc = s_upload(s, &d_rank, frame_rank))) return rc;
    const int nblk = cam_part_blocks((dp.K - 3) + 1);   // (device_state.hpp)
    if ((rc = s_alloc(s, &h->dp.cam_part, (size_t)std::max(wave_seg_base[nwaves], 1) * nblk * 256))) return rc;
    h->dp.wave_seg_base = d_base; h->dp.frame_rank = d_rank;
  }
  const size_t ucross_len = h->prior_frames.empty() ? 0 : (size_t)FR * CD * CD;
  if ((rc = s_alloc(s, &sv.U, (size_t)ucross_base + ucross_len))) return rc;
  if (ucross_len) { s->ucross = sv.U + ucross_base; HIP_TRY(hipMemset(s->ucross, 0, ucross_len * sizeof(double))); }   // stays zero on the other ranks
  if (ucross_len && h->prior_free) {   // the ratio is one more camera-side unknown: a 1-wide dense border of S, handled by a second solve
    if ((rc = s_alloc(s, &s->border, (size_t)sv.npad))) return rc;
    if ((rc = s_alloc(s, &s->ratio4, kRtSize))) return rc;
    HIP_TRY(hipMemset(s->ratio4, 0, kRtSize * sizeof(double)));
    HIP_TRY(hipMemset(s->border, 0, (size_t)sv.npad * sizeof(double)));
    sv.zv2 = sv.zv + 2 * sv.npad; sv.ceta = sv.zv + 3 * sv.npad; sv.border2 = s->border; sv.rt = s->ratio4;
    if (lead) s->num_reduced_params += 1;
  }
  if ((rc = s_alloc(s, &sv.gc, (size_t)F * CD))) return rc;
  if ((rc = s_alloc(s, &sv.intr_part, (size_t)FR * 54))) return rc;
  if ((rc = s_alloc(s, &sv.trial_intr, 9 * (size_t)std::max(dp.NI, 1)))) return rc;
  HIP_TRY(hipMemcpy(sv.trial_intr, dp.intr, 9 * (size_t)dp.NI * sizeof(double), hipMemcpyDeviceToDevice));
  if ((rc = s_alloc(s, &sv.V, (size_t)M * 6))) return rc;
  if ((rc = s_alloc(s, &sv.gp, (size_t)M * 3))) return rc;
  if ((rc = s_alloc(s, &sv.diag_c, (size_t)F * CD))) return rc;
  if ((rc = s_alloc(s, &sv.diag_p, (size_t)M * 3))) return rc;
  if ((rc = s_alloc(s, &sv.Linv, (size_t)M * 6))) return rc;
  if ((rc = s_alloc(s, &sv.z, (size_t)M * 3))) return rc;
  const size_t pm_doubles = (size_t)pt_total + kGroupFull;   // (+ the all-zero group)
  sv.zero_off = (uint32_t)pt_total;
  sv.lerp_rot = dp.interp_rotation && dp.shutter != 0;
  if ((rc = s_alloc(s, &sv.Pm, pm_doubles))) return rc;
  HIP_TRY(hipMemsetAsync(sv.Pm, 0, pm_doubles * sizeof(double), h->stream));   // rows of frames that do not see the point stay zero for good: nothing ever writes them
  if ((rc = s_alloc(s, &sv.schur_next, 9 * 16))) return rc;
  HIP_TRY(hipMemsetAsync(sv.schur_next, 0, 9 * 16 * sizeof(unsigned), h->stream));   // (every launch leaves the counters at zero: its last workgroup)
  if ((rc = s_alloc(s, &sv.schur_mfma_count, 1))) return rc;
  HIP_TRY(hipMemsetAsync(sv.schur_mfma_count, 0, sizeof(unsigned long long), h->stream));

  if ((rc = s_alloc(s, &sv.S, (size_t)sv.nslots * kTile * kTile + (size_t)sv.npad))) return rc;
  sv.rhs = sv.S + (size_t)sv.nslots * kTile * kTile;   // one buffer = exchange payload (2)
  HIP_TRY(hipMemsetAsync(sv.S, 0, ((size_t)sv.nslots * kTile * kTile + (size_t)sv.npad) * sizeof(double), h->stream));   // fill-only tiles stay zero for good
  if ((rc = s_alloc(s, &sv.udiag, (size_t)F * CD))) return rc;
  if ((rc = s_alloc(s, &sv.xbuf, 2 * (size_t)F * CD + 3 + kMaxRankSlots))) return rc;   // (+ the ranks' gradient maxima)
  if ((rc = s_alloc(s, &sv.yp, (size_t)M * 3))) return rc;
  if ((rc = s_alloc(s, &sv.trial_poses, (size_t)FR * CD))) return rc;
  if ((rc = s_alloc(s, &sv.trial_points, (size_t)M * 3))) return rc;
  const size_t nb = std::max<size_t>((N + 255) / 256, ((size_t)sv.n + 3 * (size_t)M + 255) / 256);
  if ((rc = s_alloc(s, &sv.partial, 2 * std::max(nb, ((size_t)M + 15) / 16 + 1) + 2))) return rc;   // (the point sweeps leave one partial per workgroup: 16 - 64 points)
  if ((rc = s_alloc(s, &sv.partial_c, 2 * (((size_t)sv.n + 3 * (size_t)M + 255) / 256) + 2))) return rc;
  if ((rc = s_alloc(s, &sv.scalars, 16))) return rc;
  if ((rc = s_alloc(s, &s->d_ctl, kCtlSize))) return rc;
  HIP_TRY(hipMemset(s->d_ctl, 0, kCtlSize * sizeof(double)));
  sv.ctl = s->d_ctl;   // (in the device copies of the plan: the persistent Cholesky looks at the status word — zero while the host decides; launches by value get null then)
  if ((rc = s_alloc(s, &sv.chol_fail, 1))) return rc;
  if ((rc = s_alloc(s, &s->d_gpose, (size_t)F * CD))) return rc;
  if ((rc = s_alloc(s, &s->d_gpoint, (size_t)M * 3))) return rc;
  if (h->allreduce && h->world > 1) { if ((rc = s_alloc(s, &s->merge_buf, 4 * (size_t)M))) return rc; }
  if (dp.pp_count > 0) {
    const size_t n6 = 6 * (size_t)dp.pp_count;
    if ((rc = s_alloc(s, &s->pp.v0, n6))) return rc;
    if ((rc = s_alloc(s, &s->pp.g0, n6))) return rc;
    if ((rc = s_alloc(s, &s->pp.cross, n6))) return rc;
    if ((rc = s_alloc(s, &s->pp.diag, n6))) return rc;
    std::vector<int32_t> tds(nt);
    for (int t = 0; t < nt; ++t) tds[t] = slot_base[iperm[t]];
    if ((rc = s_upload_const(s, &s->pp.tile_diag_slot, tds))) return rc;
  }
  HIP_TRY(hipMemset(sv.scalars, 0, 16 * sizeof(double)));
  HIP_TRY(hipMemset(sv.chol_fail, 0, sizeof(int)));
  CholPlan& pl = s->plan;
  pl.upd = s->d_upd; pl.diag_info = s->d_diag_info; pl.diag_ptr = s->d_diag_ptr; pl.diag_list = s->d_diag_list;
  pl.sub_info = s->d_sub_info; pl.sub_ptr = s->d_sub_ptr; pl.sub_list = s->d_sub_list; pl.sub_col = s->d_sub_col; pl.diag_own = s->d_diag_own; pl.sub_own = s->d_sub_own;
  pl.diag_fuse = s->d_diag_fuse; pl.sub_pub = s->d_sub_pub;
  pl.back_info = s->d_back_info; pl.back_ptr = s->d_back_ptr; pl.back_list = s->d_back_list;
  pl.tasks = s->d_tasks; pl.ntasks = (int)(s->tasks.size() / 2); pl.ndiag = (int)(s->diag_info.size() / 4);
  pl.ticket = s->d_dag_sync;
  pl.nslots = sv.nslots; pl.nparts = parts;
  pl.fwd_range = s->d_fwd_full; pl.diag_toprow = s->d_diag_toprow; pl.fwd2_minus = nullptr; pl.fwd2_partial = nullptr; pl.eta_tiles = nullptr; pl.eta_extra = nullptr; pl.eta_partial = nullptr;
  if (sharded) {
    s->plan_a = pl; s->plan_a.tasks = s->d_tasks_a; s->plan_a.ntasks = (int)(s->tasks_a.size() / 2);
    s->plan_b = pl; s->plan_b.tasks = s->d_tasks_b; s->plan_b.ntasks = (int)(s->tasks_b.size() / 2);
    if (two_rhs) {
      double* tail = s->topx_buf + (size_t)s->ntop_slots * kTile * kTile + (size_t)s->ntop_tiles * kTile;
      s->plan_a.fwd_range = s->d_fwd_a; s->plan_a.fwd2_partial = tail; s->plan_a.eta_tiles = s->d_row_check; s->plan_a.eta_partial = tail + (size_t)s->ntop_tiles * kTile;
      s->plan_b.fwd_range = s->d_fwd_b; s->plan_b.fwd2_minus = tail; s->plan_b.eta_tiles = s->d_row_sep; s->plan_b.eta_extra = tail + (size_t)s->ntop_tiles * kTile;
    }
    s->plan_b.diag_info = s->d_diag_info_sh; s->plan_b.sub_info = s->d_sub_info_sh;   // (the parts' partial tiles have been summed in by the exchange)
  }
  int cus = 0;
  HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device));
  // One persistent workgroup per CU — or two, for a WIDE task graph.  Claimed tasks that wait for their inputs hold a workgroup and the
  // schedule is short of them (192 instead of 256 workgroups cost 5 % at C4); the kernel is built so that two fit a CU (78 KB of LDS,
  // <= 256 registers per lane).  Through round 4 two per CU slowed the solve down by orders of magnitude: twice the waves polling AND every
  // operand load going to the memory side.  With the operand tiles looked ahead at through L2 (cholesky.hip, Frag::load) that is gone:
  // 512 workgroups are stable (C5 7.54 - 7.64 ms per LM iteration in six runs against 7.86 - 7.96 with 256, 384 in between) where a level of
  // the elimination tree holds more tasks than there are CUs (C5: 380 per level), and change nothing where the chain dominates (C4: 116 per
  // level, 1.62 ms either way).  RSBA_CHOL_WGS overrides, capped at two per CU.
  s->dag_workgroups = std::max(1, std::min(pl.ntasks, std::max(cus, 1)));
  const int64_t my_tasks = s->sharded ? (int64_t)s->plan_a.ntasks + s->plan_b.ntasks : (int64_t)pl.ntasks;   // (what THIS rank runs)
  if (s->nlev > 0 && my_tasks > (int64_t)s->nlev * std::max(cus, 1)) { s->dag_workgroups = std::max(1, std::min(pl.ntasks, 2 * std::max(cus, 1))); s->dag_one_per_cu = false; }
  // A small plan (100 cameras: 267 tasks, ~15 per elimination level) is served better by a quarter as many workgroups as tasks — fewer
  // pollers around the chain: 0.428 -> 0.418 ms per iteration, three runs each — and leaves the rest of the chip to other streams.
  if (pl.ntasks <= 512) s->dag_workgroups = std::max(1, std::min(s->dag_workgroups, std::max(64, pl.ntasks / 4)));
  if (const char* e = std::getenv("RSBA_CHOL_WGS")) { s->dag_workgroups = std::max(1, std::min(std::min(pl.ntasks, 2 * std::max(cus, 1)), std::atoi(e))); s->dag_one_per_cu = s->dag_workgroups <= cus; }
  if (std::getenv("RSBA_CHOL_TRACE")) {
    if ((rc = s_alloc(s, &s->d_trace, 8 * (size_t)pl.ntasks))) return rc;
    HIP_TRY(hipMemset(s->d_trace, 0, 8 * (size_t)pl.ntasks * sizeof(long long)));
  }
  pl.trace = s->d_trace;
  tick("allocations");
  if (dbg_plan)
    std::fprintf(stderr, "[rsba plan] tiles %d, factor tiles %d, levels %d, tasks %d (partials %d); tile pairs %d, entries %lld, schur chunks %d\n", nt, sv.nslots,
                 s->nlev, pl.ntasks, parts, sv.ntp, (long long)s->num_pairs, sv.nchunk);
  const char* lv = std::getenv("RSBA_CHOL_LEVELS");
  s->use_levels = lv && lv[0] == '1';
  {
    DagArgs host_args{sv, pl};
    {
      std::vector<int32_t> st2(2 * (size_t)sv.nslots);
      for (int k = 0; k < nt; ++k) {
        st2[2 * (size_t)slot_base[k]] = perm[k]; st2[2 * (size_t)slot_base[k] + 1] = perm[k];
        for (size_t u = 0; u < col[k].size(); ++u) { st2[2 * (size_t)(slot_base[k] + 1 + u)] = perm[col[k][u]]; st2[2 * (size_t)(slot_base[k] + 1 + u) + 1] = perm[k]; }
      }
      if ((rc = s_upload(s, &s->d_slot_tiles, st2))) return rc;
    }
    if ((rc = s_alloc(s, &s->d_verify, 2 * (size_t)sv.npad))) return rc;
    if ((rc = s_alloc(s, &s->verify_b, (size_t)sv.npad))) return rc;
    HIP_TRY(dev_stream_acquire(&s->vstream));
    HIP_TRY(dev_event_acquire(&s->ev_solved, false));
    HIP_TRY(dev_event_acquire(&s->ev_verified, false));
    HIP_TRY(hipMemset(s->d_verify, 0, 2 * (size_t)sv.npad * sizeof(double)));   // the check kernel leaves it zero again
    { const char* v = std::getenv("RSBA_CHOL_VERIFY"); s->verify_dag = !(v && v[0] == '0'); }
    { const char* v = test_hook("RSBA_CHOL_TEST_CORRUPT"); s->test_corrupt_once = v && v[0] == '1'; }
    for (int b = 0; b < 2; ++b) {   // one device copy of {sv, plan} per set of cells
      double* c = s->cells[b];
      host_args.sv.Lf = c + s->cell_off[0]; host_args.sv.chol_part = c + s->cell_off[1]; host_args.sv.Winv = c + s->cell_off[2];
      host_args.sv.zv = c + s->cell_off[3]; host_args.sv.yv = host_args.sv.zv + sv.npad; host_args.sv.Xpub = c + s->cell_off[4];
      if (sv.zv2) { host_args.sv.zv2 = host_args.sv.zv + 2 * sv.npad; host_args.sv.ceta = host_args.sv.zv + 3 * sv.npad; }
      if ((rc = s_alloc(s, &s->d_dag_args2[b], 1))) return rc;
      HIP_TRY(hipMemcpy(s->d_dag_args2[b], &host_args, sizeof host_args, hipMemcpyHostToDevice));
      if (sharded) {
        DagArgs a = host_args, bb = host_args;
        a.pl = s->plan_a; bb.pl = s->plan_b;
        if ((rc = s_alloc(s, &s->d_dag_args_a[b], 1))) return rc;
        if ((rc = s_alloc(s, &s->d_dag_args_b[b], 1))) return rc;
        HIP_TRY(hipMemcpy(s->d_dag_args_a[b], &a, sizeof a, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(s->d_dag_args_b[b], &bb, sizeof bb, hipMemcpyHostToDevice));
      }
    }
    s->d_dag_args = s->d_dag_args2[0];
  }
  {
    rsba_plan_stats& ps = s->stats;
    ps.tiles = nt; ps.factor_tiles = sv.nslots; ps.levels = s->nlev; ps.tasks = pl.ntasks;
    ps.schur_entries = nent; ps.schur_chunks = sv.nchunk;
    // block products of the Schur complement that are not structurally zero: per entry (frames present on the I side) x (on the J side),
    // summed in the pass that forms the entries' block masks
    ps.schur_block_products = 0;
    for (int64_t v : products_part) ps.schur_block_products += v;
    // tile factorisation: per DIAG item its contributors (lower half of L L^T: T^3 each) + potrf and inverse (T^3 / 3 each);
    // per SUB item 2 T^3 per contributor + the product with W (T^3); forward / backward solve 2 T^2 per factor tile, twice
    const int64_t T3 = (int64_t)kTile * kTile * kTile;
    ps.cholesky_flops = T3 * ((int64_t)(s->diag_list.size() / 2) + 2 * (int64_t)(s->sub_list.size() / 2) + (int64_t)(s->sub_info.size() / 4)) +
                        2 * T3 / 3 * (int64_t)(s->diag_info.size() / 4) + 4 * (int64_t)kTile * kTile * ((int64_t)sv.nslots + nt);
    ps.exchange_doubles = (int64_t)s->exch_tiles * kTile * kTile + sv.npad;   // exchange (2) of a sharded solve: the plan's tile pairs | rhs (the fill-in tiles of the factor's layout stay home)
    ps.schur_groups = sv.ngroups;
    ps.schur_group_bytes = pt_total * (int64_t)sizeof(double);
    ps.schur_factored_groups = dev_plan ? dpo.factored_groups : 0;
    for (int64_t g = 0; g < sv.ngroups && !dev_plan; ++g) ps.schur_factored_groups += tile_factored[g_tile[(size_t)g]];
    ps.sharded_factorisation = sharded ? 1 : 0;
    if (sharded) {   // ... or, when every rank factors its own part: the separators' tiles | their rows of the rhs, and the gather of the step
      ps.exchange_doubles = (int64_t)s->ntop_slots * kTile * kTile + (int64_t)s->ntop_tiles * kTile + sv.npad;
      ps.separator_tiles = s->ntop_tiles; ps.separator_factor_tiles = s->ntop_slots;
      ps.local_tasks = (int64_t)(s->tasks_a.size() / 2); ps.separator_tasks = (int64_t)(s->tasks_b.size() / 2);
      // the two dependency chains: elimination levels inside this rank's part, and levels that hold a separator column
      int lmax = -1; std::vector<uint8_t> sep_level((size_t)nlev, 0);
      for (int j = 0; j < nt; ++j) { if (cpart[j] == h->rank) lmax = std::max(lmax, level[j]); else if (cpart[j] < 0) sep_level[level[j]] = 1; }
      ps.local_levels = lmax + 1; ps.separator_levels = 0;
      for (uint8_t b : sep_level) ps.separator_levels += b;
    }
  }
  sv.ctl = nullptr;
  tick("statistics");
  HIP_TRY(hipStreamSynchronize(h->stream));   // the plan's one-time fills and scatters are done whatever stream the solves will run on
  tick("device fills");
  if (dbg_plan) std::fprintf(stderr, "[rsba plan] host phases:%s\n", phases.c_str());
  return RSBA_OK;
}

// A plan that failed half-way (out of memory, an unsupported size) must not be taken for a finished one by the next call: the
// half-built solver is torn down again, so that a retry builds — and fails — afresh instead of launching kernels on null tables.
int32_t build_solver(rsba_handle* h) {
  if (h->solver) return RSBA_OK;
  const int32_t rc = build_solver_impl(h);
  if (rc != RSBA_OK) {
    const std::string why = rsba_last_error();   // (the teardown must not lose what went wrong)
    rsba_destroy_solver(h);
    return rsba_set_error(rc, why.c_str());
  }
  return RSBA_OK;
}

int32_t reset_scales(rsba_handle* h) {
  const DeviceProblem& dp = h->dp;
  HIP_TRY(hipMemcpyAsync(dp.scale_pose, h->d_mask_pose, h->mask_pose.size() * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(dp.scale_point, h->d_mask_point, h->mask_point.size() * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(dp.scale_intr, h->d_mask_intr, h->mask_intr.size() * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  if (dp.pp_count > 0) {   // the priorPoses blocks are always free: scale 1
    static const std::vector<double> ones(1 << 16, 1.0);
    for (size_t o = 0; o < 6 * (size_t)dp.pp_count; o += ones.size())
      HIP_TRY(hipMemcpyAsync(dp.pp_scale + o, ones.data(), std::min(ones.size(), 6 * (size_t)dp.pp_count - o) * sizeof(double), hipMemcpyHostToDevice, h->stream));
  }
  return RSBA_OK;
}

// all-reduce across the ranks of a point-partitioned solve (no-op for a single GPU); kind: RSBA_EXCHANGE_* (statistics)
int32_t exchange(rsba_handle* h, double* buf, int64_t count, int op, int kind) {
  if (!h->allreduce) return RSBA_OK;   // (a one-rank exchange still goes through its transport: identity, but the path is exercised)
  (void)rsba_set_error(RSBA_OK, "");   // a transport that fails says why through rsba_set_error (the RCCL one: the ncclResult string); keep what it said
  ++h->x_calls[kind]; h->x_doubles[kind] += count;
  PhaseTimer* xt = h->solver && h->solver->xtimer.on ? &h->solver->xtimer : nullptr;
  hipEvent_t a = nullptr;
  if (xt) { a = xt->get(); (void)hipEventRecord(a, h->stream); }
  const int failed = h->allreduce(h->allreduce_ctx, buf, count, op, h->stream);
  if (xt) { hipEvent_t b = xt->get(); (void)hipEventRecord(b, h->stream); xt->pending.push_back({kind, a, b}); }
  if (failed != 0) {
    const std::string why = rsba_last_error();
    char where[112];
    std::snprintf(where, sizeof where, " (all-reduce %s of %lld doubles, op %d, rank %d of %d)", rsba_exchange_name(kind), (long long)count, op, h->rank, h->world);
    return rsba_set_error(RSBA_ERR_COMM, ((why.empty() ? std::string("the all-reduce callback reported a failure") : why) + where).c_str());
  }
  return RSBA_OK;
}

// The ratio's column of the normal equations (and its own h, g) is formed by EVERY rank over ALL priors, from replicated poses — also where
// a sharded plan has shared the priors' blocks out (h->prior_split: dp.prior_of then lists this rank's share).
DeviceProblem all_priors(const rsba_handle* h) {
  DeviceProblem d = h->dp;
  if (h->prior_split) d.prior_of = h->prior_of_all;
  return d;
}

// r, J (loss-corrected, masked, scaled) and the normal-equation blocks at the current parameters;
// exchange (1): per-camera gradient blocks + diag(U) + cost scalars.  Results: sv.gc / sv.udiag global,
// scalars[kCost, kFixedCost, kEvalFailed].
// have_eval: the LM-mode evaluation at these parameters (per-wave camera blocks, cost incl. the prior blocks' in d_cost2) has just
// been run — the trust-region loop evaluates its candidates that way when nothing would be lost by it (see rsba_solve).
int32_t linearize(rsba_handle* h, bool have_eval = false, bool want_gradmax = false) {
  Solver* s = h->solver;
  if (!have_eval) {
    PhaseScope ps(h, RSBA_PHASE_EVAL_LM);
    HIP_TRY(launch_eval(h->dp, kLmJacobian, h->stream));   // (the cost reduction overwrites dp.fail_count: nothing to clear)
    HIP_TRY(launch_cost_reduce(h->dp, h->d_cost2, h->stream));
  }
  {
    PhaseScope ps(h, RSBA_PHASE_CAMERA_BLOCKS);
    HIP_TRY(launch_camera_blocks(h->dp, s->sv, h->stream));
    HIP_TRY(launch_intr_blocks(h->dp, s->sv, h->stream));
  }
  const bool my_priors = s->ucross && (s->sv.lead || h->prior_split);   // motion priors: replicated terms from the lead rank — or, sharded factorisation, every rank its share
  if (my_priors || s->border) {
    PhaseScope ps(h, RSBA_PHASE_PRIORS);
    if (my_priors) {
      if (!have_eval) HIP_TRY(launch_prior_cost(h->dp, h->d_cost2, h->prior_invalid, h->stream));
      HIP_TRY(launch_prior_blocks(h->dp, s->sv, s->ucross, h->stream));
    }
    if (s->border) HIP_TRY(launch_prior_border(all_priors(h), s->sv, s->border, s->ratio4, h->stream));   // every rank: from replicated poses
  }
  if (h->dp.pp_count > 0 || h->dp.pp_spherical >= 0) {   // per-pose priors: replicated terms, contributed by the lead rank;
    PhaseScope ps(h, RSBA_PHASE_PRIORS);                   // the linearisation of the priorPoses coordinates (v0, g0, cross) on every rank: each one steps them itself
    if (s->sv.lead && !have_eval) HIP_TRY(launch_pose_prior_cost(h->dp, h->d_cost2, h->stream));
    HIP_TRY(launch_pose_prior_blocks(h->dp, s->sv, s->pp, h->stream));
  }
  {
    PhaseScope ps(h, RSBA_PHASE_POINT_BLOCKS);
    HIP_TRY(launch_point_blocks(h->dp, s->sv, h->stream));
  }
  PhaseScope ps(h, RSBA_PHASE_EXCHANGE);
  s->gradmax_done = false;
  if (!h->allreduce) { HIP_TRY(launch_local_linearize(h->dp, s->sv, h->d_cost2, h->stream)); return RSBA_OK; }   // one rank: nothing to sum
  // max |g_i| (the gradient test that follows an accepted step) rides in this exchange instead of taking a MAX all-reduce of its own —
  // every collective is tens of microseconds of latency on a node: each rank's maximum over ITS points (they are nobody else's) in a
  // slot of its own behind the payload, the other ranks' slots zero, so the SUM delivers all of them; the cameras' maximum is taken
  // from the summed gradient behind the exchange.  (Not with per-pose priors: their blocks' maximum is the lead rank's alone.)
  const bool ride = want_gradmax && h->world <= kMaxRankSlots && h->dp.pp_count == 0;   // (a SphericalPrior has no coordinates of its own: its gradient is part of the summed camera gradient)
  HIP_TRY(launch_pack_linearize(h->dp, s->sv, h->d_cost2, h->stream, ride ? h->world : 0));
  if (ride) HIP_TRY(launch_gradient_max_points(h->dp, s->sv, h->rank, h->stream));
  int32_t rc = exchange(h, s->sv.xbuf, 2 * s->sv.n + 3 + (ride ? h->world : 0), 0, RSBA_EXCHANGE_CAMERA);
  if (rc) return rc;
  HIP_TRY(launch_unpack_linearize(h->dp, s->sv, h->stream));
  if (ride) { HIP_TRY(launch_gradient_max_cameras(h->dp, s->sv, h->world, h->stream)); s->gradmax_done = true; }
  return RSBA_OK;
}

int32_t gradient_max(rsba_handle* h) {
  Solver* s = h->solver;
  if (s->gradmax_done) { s->gradmax_done = false; return RSBA_OK; }   // (came with the camera exchange of the linearisation just before)
  PhaseScope ps(h, RSBA_PHASE_OTHER);
  HIP_TRY(launch_gradient_max(h->dp, s->sv, h->stream));
  if (s->sv.lead) HIP_TRY(launch_pose_prior_gradmax(h->dp, s->sv, s->pp, h->stream));
  return exchange(h, s->sv.scalars + kGradMax, 1, 1, RSBA_EXCHANGE_SCALARS);
}

// reduced camera system S and rhs at the given trust-region radius (point elimination), summed over the ranks
int32_t reduce_system(rsba_handle* h, double radius) {
  Solver* s = h->solver; const SolverDev& sv = s->sv; hipStream_t st = h->stream;
  { PhaseScope ps(h, RSBA_PHASE_POINT_FACTOR); HIP_TRY(launch_point_factor(h->dp, sv, radius, st, s->clamp_with_factor ? s->clamp_lo_hi : nullptr)); }
  {
    PhaseScope ps(h, RSBA_PHASE_PROJECT);
    // A shared intrinsics block at scale (4k cameras: projection 1.03 ms of stores, virtual-record sweep 0.57 ms of fp64): the two passes
    // read the same point factors and write different records — side by side, the sweep on the arming stream (idle here), two events.
    const bool beside = sv.NPF > 0 && sv.nvgroups > 0 && s->mstream && s->ev_fork && h->dp.N >= 2000000 && !std::getenv("RSBA_NO_SIDE_SWEEP") && !project_covers_virtual_records(h->dp, sv);   // (round 5: one fused sweep does both where it applies)
    if (beside) {
      HIP_TRY(hipEventRecord(s->ev_fork, st));
      HIP_TRY(hipStreamWaitEvent(s->mstream, s->ev_fork, 0));
      HIP_TRY(launch_virtual_records(h->dp, sv, s->mstream));
      HIP_TRY(hipEventRecord(s->ev_join, s->mstream));
    }
    HIP_TRY(launch_project(h->dp, sv, st));
    if (beside) HIP_TRY(hipStreamWaitEvent(st, s->ev_join, 0));
    else HIP_TRY(launch_virtual_records(h->dp, sv, st));
  }
  {
    PhaseScope ps(h, RSBA_PHASE_SCHUR);
    // (S needs no clearing: the merge kernel overwrites every tile a tile pair maps to, and the fill-only tiles — zeroed when the
    // plan was built — are never written by anyone: the factorisation keeps its own tiles, the multi-GPU exchange adds zeros)
    HIP_TRY(launch_schur_blocks(h->dp, sv, radius, st));
    ++s->schur_launches;
    if (sv.lead || sv.frame_lead) HIP_TRY(launch_pose_prior_reduce(h->dp, sv, s->pp, radius, st));   // the priorPoses blocks leave the system like points (each on the rank that holds its pose's tile)
  }
  // exchange (2): partial reduced camera systems -> the full one on every rank (then factored redundantly) — unless every rank
  // factors its own part: then only the separators travel, between the two launches of the factorisation (solve_reduced_system)
  if (s->sharded && !s->sharded_off && !s->use_levels) return RSBA_OK;
  PhaseScope ps(h, RSBA_PHASE_EXCHANGE);
  if (s->sharded && s->ntop_fill) HIP_TRY(launch_zero_tiles(sv.S, s->d_top_fill, s->ntop_fill, st));   // (a sharded solve left its reduced values in the fill-only separator tiles)
  if (h->allreduce && s->exch_slots) {   // only the tiles that can be non-zero travel (the fill-in tiles of the layout are zero on every rank)
    const int64_t count = (int64_t)s->exch_tiles * kTile * kTile + sv.npad;
    HIP_TRY(launch_exchange_pack(sv, s->exch_slots, s->exch_tiles, s->exch_buf, false, st));
    if (int32_t rc = exchange(h, s->exch_buf, count, 0, RSBA_EXCHANGE_SYSTEM)) return rc;
    HIP_TRY(launch_exchange_pack(sv, s->exch_slots, s->exch_tiles, s->exch_buf, true, st));
    return RSBA_OK;
  }
  return exchange(h, sv.S, (int64_t)sv.nslots * kTile * kTile + sv.npad, 0, RSBA_EXCHANGE_SYSTEM);
}

// S y = rhs: left-looking tile Cholesky (forward solve rides along), then the backward solve: one persistent DAG
// launch, or — the schedule it is checked against — one launch per (level, kind).  S and rhs are left as they are
// (the factor has its own tiles); y lands in sv.yv.
// the solver's stream waits for a verification still running beside it (before its flag is read, and before the cells it reads are re-armed)
int32_t await_verification(rsba_handle* h) {
  Solver* s = h->solver;
  if (s->verify_pending) { HIP_TRY(hipStreamWaitEvent(h->stream, s->ev_verified, 0)); s->verify_pending = false; }
  return RSBA_OK;
}
int32_t solve_reduced_system(rsba_handle* h, bool rhs_stays = false) {
  Solver* s = h->solver; SolverDev& sv = s->sv; hipStream_t st = h->stream;
  PhaseScope ps(h, RSBA_PHASE_CHOLESKY);
  if (int32_t rc = await_verification(h)) return rc;
  if (!s->use_levels) {
    // The set of cells the last solve used is released here — everything that reads it has been enqueued on this stream or has been
    // waited for above — and re-armed (every cell empty: all ones) on the side stream while THIS solve runs on the other set.
    const int used = s->cur_cells, now = used ^ 1;
    HIP_TRY(hipEventRecord(s->ev_released, st));
    HIP_TRY(hipStreamWaitEvent(s->mstream, s->ev_released, 0));
    HIP_TRY(hipMemsetAsync(s->cells[used], 0xFF, s->ncells * sizeof(double), s->mstream));
    HIP_TRY(hipEventRecord(s->ev_armed[used], s->mstream));
    s->arm_pending[used] = true;
    if (s->arm_pending[now]) { HIP_TRY(hipStreamWaitEvent(st, s->ev_armed[now], 0)); s->arm_pending[now] = false; }
    s->cur_cells = now;
    double* c = s->cells[now];
    sv.Lf = c + s->cell_off[0]; sv.chol_part = c + s->cell_off[1]; sv.Winv = c + s->cell_off[2]; sv.zv = c + s->cell_off[3]; sv.yv = sv.zv + sv.npad; sv.Xpub = c + s->cell_off[4];
    if (sv.zv2) { sv.zv2 = sv.zv + 2 * sv.npad; sv.ceta = sv.zv + 3 * sv.npad; }
    s->d_dag_args = s->d_dag_args2[now];
    if (s->sharded && !s->sharded_off) {
      // launch A: the columns of this rank's part, from its own partial S — complete for them: every point that sees one of its tiles is here
      HIP_TRY(launch_chol_dag(sv, s->plan_a, s->d_dag_args_a[now], std::min(s->dag_workgroups, std::max(1, s->plan_a.ntasks)), s->dag_one_per_cu, st));
      // exchange (2'): the separators' tiles, each rank's share less what its part subtracts from them, summed over the ranks
      const int64_t count = (int64_t)s->ntop_slots * kTile * kTile + (int64_t)s->ntop_tiles * kTile + (s->two_rhs ? (int64_t)s->ntop_tiles * kTile + 8 : 0);   // (+ the second right-hand side's share: launch A left it behind the rows of the first)
      ps.stop();
      {
        PhaseScope pe(h, RSBA_PHASE_EXCHANGE);
        HIP_TRY(launch_top_assemble(sv, s->d_top_slots, s->d_top_info, s->d_asm_ptr, s->d_asm_list, s->d_top_tiles, s->ntop_slots, s->topx_buf, st));
        if (int32_t rc = exchange(h, s->topx_buf, count, 0, RSBA_EXCHANGE_SYSTEM)) return rc;
        HIP_TRY(launch_top_unpack(sv, s->d_top_slots, s->d_top_info, s->d_top_tiles, s->ntop_slots, s->topx_buf, st));
      }
      ps.start();
      // launch B: the separators (every rank alike), forward and backward, then this rank's part backward
      HIP_TRY(launch_chol_dag(sv, s->plan_b, s->d_dag_args_b[now], std::min(s->dag_workgroups, std::max(1, s->plan_b.ntasks)), s->dag_one_per_cu, st));
      // exchange (4): the camera step — every rank contributes the rows of its part, rank 0 the separators'
      ps.stop();
      {
        PhaseScope pe(h, RSBA_PHASE_EXCHANGE);
        HIP_TRY(launch_step_rows(sv.yv, s->d_row_mine, sv.npad, s->ybuf, st));
        if (int32_t rc = exchange(h, s->ybuf, sv.npad, 0, RSBA_EXCHANGE_STEP)) return rc;
        HIP_TRY(hipMemcpyAsync(sv.yv, s->ybuf, (size_t)sv.npad * sizeof(double), hipMemcpyDeviceToDevice, st));
      }
      ps.start();
    } else
    HIP_TRY(launch_chol_dag(sv, s->plan, s->d_dag_args, s->dag_workgroups, s->dag_one_per_cu, st));
    if (s->test_corrupt_once) { s->test_corrupt_once = false; HIP_TRY(hipMemsetAsync(sv.yv + (sv.n / 2 / 6) * 6 + 1, 0, sizeof(double), st)); }   // test hook: one entry of the solution (a pose coordinate in mid-video) lost
    if (s->verify_dag) {
      // (rhs_stays: nobody writes sv.rhs before the check has been waited for — the LM iteration without a free ratio; otherwise the check gets a copy)
      const double* b_rhs = sv.rhs;
      if (s->two_rhs) { HIP_TRY(launch_border_combine(s->verify_b, sv.rhs, s->border, 0.0, sv.npad, st, s->ratio4 + kRtC)); b_rhs = s->verify_b; }   // what was solved for: g - (s eta) b
      else if (!rhs_stays) { HIP_TRY(hipMemcpyAsync(s->verify_b, sv.rhs, (size_t)sv.npad * sizeof(double), hipMemcpyDeviceToDevice, st)); b_rhs = s->verify_b; }
      HIP_TRY(hipEventRecord(s->ev_solved, st));
      HIP_TRY(hipStreamWaitEvent(s->vstream, s->ev_solved, 0));
      HIP_TRY(launch_chol_verify(sv, s->d_slot_tiles, b_rhs, s->d_verify, s->d_verify + sv.npad, 1e-7, sv.scalars + kDagSuspect, s->vstream,
                                 s->sharded && !s->sharded_off ? s->d_row_check : nullptr));   // (a rank of a sharded factorisation holds the whole of its part's rows of S, nothing else)
      HIP_TRY(hipEventRecord(s->ev_verified, s->vstream));
      s->verify_pending = true;
    }
  } else {
    for (int l = 0; l < s->nlev; ++l) {
      const int d0 = s->lev_diag_ptr[l], d1 = s->lev_diag_ptr[l + 1], t0 = s->lev_sub_ptr[l], t1 = s->lev_sub_ptr[l + 1];
      const int u0 = s->lev_upd_ptr[l], u1 = s->lev_upd_ptr[l + 1];
      HIP_TRY(launch_chol_level(sv, s->plan, kTaskUpdate, u0, u1 - u0, st));
      HIP_TRY(launch_chol_level(sv, s->plan, kTaskDiag, d0, d1 - d0, st));
      if (s->two_rhs) HIP_TRY(launch_chol_level(sv, s->plan, kTaskFwd2, d0, d1 - d0, st));
      HIP_TRY(launch_chol_level(sv, s->plan, kTaskSub, t0, t1 - t0, st));
    }
    if (s->two_rhs) HIP_TRY(launch_chol_level(sv, s->plan, kTaskEta, 0, 1, st));
    for (int l = s->nlev - 1; l >= 0; --l) {
      const int d0 = s->lev_diag_ptr[l], d1 = s->lev_diag_ptr[l + 1];
      HIP_TRY(launch_chol_level(sv, s->plan, kTaskBack, d0, d1 - d0, st));
    }
  }
  return RSBA_OK;
}

// S v = b2 for one more right-hand side, through the factor tiles the last solve_reduced_system left: forward and backward
// substitution only (cholesky.hip chol_solve_kernel).  *v_out points at the solution ([npad], valid until the next call).  The result
// is checked like the first one (same sticky flag), here on the solver's own stream: the check is 20 us, the solve 0.2 ms.
int32_t solve_again(rsba_handle* h, const double* b2, const double** v_out) {
  Solver* s = h->solver; const SolverDev& sv = s->sv; hipStream_t st = h->stream;
  PhaseScope ps(h, RSBA_PHASE_CHOLESKY);
  if (!s->use_levels) HIP_TRY(launch_chol_solve(sv, s->plan, s->d_dag_args, b2, s->zy2, s->d_dag_sync + 2, s->dag_workgroups, st));
  else {   // the same tasks, one launch per level: no polling (what a suspect persistent result is redone with)
    for (int l = 0; l < s->nlev; ++l) HIP_TRY(launch_chol_solve_level(sv, s->plan, false, s->lev_diag_ptr[l], s->lev_diag_ptr[l + 1] - s->lev_diag_ptr[l], b2, s->zy2, st));
    for (int l = s->nlev - 1; l >= 0; --l) HIP_TRY(launch_chol_solve_level(sv, s->plan, true, s->lev_diag_ptr[l], s->lev_diag_ptr[l + 1] - s->lev_diag_ptr[l], b2, s->zy2, st));
  }
  *v_out = s->zy2 + sv.npad;
  if (!s->use_levels && s->verify_dag) {
    if (int32_t rc = await_verification(h)) return rc;   // (the accumulators of the check are shared)
    SolverDev sv2 = sv;
    sv2.yv = s->zy2 + sv.npad;
    HIP_TRY(launch_chol_verify(sv2, s->d_slot_tiles, b2, s->d_verify, s->d_verify + sv.npad, 1e-7, sv.scalars + kDagSuspect, st));
  }
  return RSBA_OK;
}

// ratio (free interFrameRatio only): in {h_s + D/radius, g_s, scale of the ratio}, out the ratio's scaled step eta.
// The ratio's column b of the damped normal equations is a 1-wide dense border of S.  With S = L L^T, z = L^-1 g, z2 = L^-1 b:
//   eta = (g_s - s z2.z) / (h_s + D - s^2 z2.z2),   L^T y = z - (s eta) z2
// — both forward solves inside the factorisation's own launch (FWD2 tasks beside the DIAG tasks, cholesky.hip), the ETA task between
// the phases, ONE backward solve.  (Until round 4 the second right-hand side took a launch of its own: S u = g, S v = b, y = u - s eta v.)
// ratio == nullptr with a two-column plan: the device-side loop — the ratio's scalars are on the device already (ratio_prepare_ctl).
struct RatioStep { double diag, gs, scale, eta; };
int32_t factor_and_solve(rsba_handle* h, double radius, RatioStep* ratio = nullptr) {
  Solver* s = h->solver; const SolverDev& sv = s->sv; hipStream_t st = h->stream;
  if (ratio) HIP_TRY(launch_ratio_prepare(s->ratio4, ratio->diag, ratio->gs, ratio->scale, st));
  int32_t rc = reduce_system(h, radius);
  if (rc) return rc;
  if ((rc = solve_reduced_system(h, /*rhs_stays=*/!s->two_rhs))) return rc;
  if (ratio) {
    HIP_TRY(hipMemcpyAsync(&ratio->eta, s->ratio4 + kRtEta, sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  s->sv.step = sv.yv;   // the camera step is read where the solve left it (the cells of this solve stay armed until the next one)
  PhaseScope ps(h, RSBA_PHASE_BACK_SUBSTITUTE);
  HIP_TRY(launch_back_substitute(h->dp, sv, st));
  return RSBA_OK;
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

void rsba_release_plan_scratch() {
  std::lock_guard<std::mutex> lk(g_plan_scratch_mutex);
  g_plan_scratch.reset();
}

void rsba_destroy_solver(rsba_handle* h) {
  if (!h || !h->solver) return;
  const bool dbg = std::getenv("RSBA_DEBUG_PLAN") != nullptr;
  const double td0 = dbg ? now_s() : 0.0;
  // streams, events and the pinned block go back to the pool (devmem.hpp): idle first — the main stream too, whose last waits name these events
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->solver->vstream) { (void)hipStreamSynchronize(h->solver->vstream); dev_stream_release(h->solver->vstream); }
  if (h->solver->mstream) { (void)hipStreamSynchronize(h->solver->mstream); dev_stream_release(h->solver->mstream); }
  for (hipEvent_t e : {h->solver->ev_armed[0], h->solver->ev_armed[1], h->solver->ev_released, h->solver->ev_solved, h->solver->ev_verified, h->solver->ev_fork, h->solver->ev_join}) dev_event_release(e, false);
  dev_pinned_release(h->solver->h_ctl);
  if (const char* path = h->solver->sv.schur_trace ? std::getenv("RSBA_SCHUR_TRACE") : nullptr) {   // debugging aid: stamps of the last Schur launch
    std::vector<long long> tr(8 * (size_t)h->solver->sv.nchunk);
    if (hipMemcpy(tr.data(), h->solver->sv.schur_trace, tr.size() * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess)
      if (FILE* f = std::fopen(path, "wb")) { std::fwrite(tr.data(), sizeof(long long), tr.size(), f); std::fclose(f); }
  }
  const double td1 = dbg ? now_s() : 0.0;
  if (h->stream) (void)hipStreamSynchronize(h->stream);   // (the side streams above are idle too: the blocks go back to the cache, devmem.hpp)
  const double td2 = dbg ? now_s() : 0.0;
  for (void* p : h->solver->allocs) dev_free(p);
  const double td3 = dbg ? now_s() : 0.0;
  delete h->solver;
  if (dbg) std::fprintf(stderr, "[rsba destroy] plan: streams + events to the pool %.2f ms; stream sync %.2f ms; blocks to the cache %.2f ms; host state %.2f ms\n", 1e3 * (td1 - td0), 1e3 * (td2 - td1), 1e3 * (td3 - td2), 1e3 * (now_s() - td3));
  h->solver = nullptr;
  if (h->prior_split) { h->dp.prior_of = h->prior_of_all; h->prior_invalid = h->prior_invalid_all; h->prior_split = false; }   // the rank's share of the priors was a table of the plan
  h->dp.rec = nullptr; h->dp.rec_alt = nullptr; h->dp.rec_candidate = 0; h->dp.obs_slot = nullptr; h->dp.cam_part = nullptr; h->dp.wave_seg_base = nullptr; h->dp.frame_rank = nullptr;
}

// gradient of Problem::Evaluate: loss-corrected J^T r on the masked tangent space, [F*P*6 | M*3 | NI*9]
int32_t rsba_gradient(rsba_handle* h, double* g) {
  int32_t rc = build_solver(h);
  if (rc) return rc;
  Solver* s = h->solver; const DeviceProblem& dp = h->dp;
  if ((rc = reset_scales(h))) return rc;
  if ((rc = linearize(h))) return rc;
  HIP_TRY(launch_unscaled_gradient(dp, s->sv, s->d_gpose, s->d_gpoint, h->stream));
  const size_t npose = (size_t)dp.F * dp.P * 6, npt = (size_t)dp.M * 3;
  HIP_TRY(hipMemcpyAsync(g, s->d_gpose, npose * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipMemcpyAsync(g + npose, s->d_gpoint, npt * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  std::fill(g + npose + npt, g + npose + npt + (size_t)dp.NI * 9, 0.0);
  for (int c = 0; c < s->sv.NIB; ++c)   // the 9 coordinates of block c sit at the front of its pseudo frames
    HIP_TRY(hipMemcpyAsync(g + npose + npt + (size_t)c * 9, s->d_gpose + npose + (size_t)c * s->sv.NPF * s->sv.CD, 9 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return RSBA_OK;
}

extern "C" int32_t rsba_set_exchange(rsba_handle* h, rsba_allreduce_fn fn, void* ctx, int32_t rank, int32_t world) {
  if (!h || world < 1 || rank < 0 || rank >= world || (world > 1 && !fn)) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "bad exchange arguments");
  if (h->solver) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "rsba_set_exchange must precede the first solve / gradient call");
  h->allreduce = fn; h->allreduce_ctx = ctx; h->rank = rank; h->world = world;
  return RSBA_OK;
}

extern "C" int32_t rsba_get_block_structure(rsba_handle* h, uint8_t* mask, int64_t* frame_obs_count) {
  if (!h || !mask) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "null argument");
  const int F = h->dp.F; const int64_t N = h->dp.N;
  std::fill(mask, mask + (size_t)F * F, (uint8_t)0);
  if (frame_obs_count) std::fill(frame_obs_count, frame_obs_count + F, (int64_t)0);
  // observations grouped by point: frames of one point pairwise share it
  std::vector<int64_t> ptr((size_t)h->dp.M + 1, 0);
  for (int64_t i = 0; i < N; ++i) ptr[h->obs_point[i] + 1]++;
  for (int j = 0; j < h->dp.M; ++j) ptr[j + 1] += ptr[j];
  std::vector<int32_t> fr(N);
  { std::vector<int64_t> fill(ptr.begin(), ptr.end() - 1); for (int64_t i = 0; i < N; ++i) fr[fill[h->obs_point[i]]++] = h->obs_frame[i]; }
  for (int j = 0; j < h->dp.M; ++j)
    for (int64_t x = ptr[j]; x < ptr[j + 1]; ++x) for (int64_t y = ptr[j]; y <= x; ++y) {
      const int a = std::max(fr[x], fr[y]), b = std::min(fr[x], fr[y]);
      mask[(size_t)a * F + b] = 1;
    }
  if (frame_obs_count) for (int64_t i = 0; i < N; ++i) frame_obs_count[h->obs_frame[i]]++;
  return RSBA_OK;
}

extern "C" int32_t rsba_set_block_structure(rsba_handle* h, const uint8_t* mask, const int64_t* frame_obs_count) {
  if (!h || !mask) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "null argument");
  if (h->solver) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "rsba_set_block_structure must precede the first solve / gradient call");
  const int F = h->dp.F;
  h->union_mask.assign(mask, mask + (size_t)F * F);
  if (frame_obs_count) h->frame_obs_total.assign(frame_obs_count, frame_obs_count + F); else h->frame_obs_total.clear();
  return RSBA_OK;
}

extern "C" int32_t rsba_get_phase_times(rsba_handle* h, rsba_phase_times* out) {
  if (!h || !out) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "null argument");
  std::memset(out, 0, sizeof *out);
  if (!h->solver) return RSBA_OK;
  for (int p = 0; p < RSBA_NUM_PHASES; ++p) { out->ms[p] = h->solver->timer.ms[p]; out->calls[p] = h->solver->timer.calls[p]; }
  return RSBA_OK;
}
extern "C" const char* rsba_exchange_name(int32_t kind) {
  static const char* names[RSBA_NUM_EXCHANGES] = {"setup", "(1) camera blocks", "(2) reduced system", "(3) step scalars", "(4) camera step", "point merge"};
  return kind >= 0 && kind < RSBA_NUM_EXCHANGES ? names[kind] : "?";
}
extern "C" int32_t rsba_get_exchange_stats(rsba_handle* h, rsba_exchange_stats* out) {
  if (!h || !out) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "null argument");
  std::memset(out, 0, sizeof *out);
  out->rank = h->rank; out->world = h->world;
  for (int k = 0; k < RSBA_NUM_EXCHANGES; ++k) { out->calls[k] = h->x_calls[k]; out->doubles[k] = h->x_doubles[k]; if (h->solver) out->ms[k] = h->solver->xtimer.ms[k]; }
  return RSBA_OK;
}
extern "C" const char* rsba_phase_name(int32_t phase) {
  static const char* names[RSBA_NUM_PHASES] = {"eval_lm", "camera_blocks", "point_blocks", "point_factor", "project", "schur", "cholesky",
                                               "back_substitute", "candidate", "eval_trial", "priors", "exchange", "other"};
  return phase >= 0 && phase < RSBA_NUM_PHASES ? names[phase] : "?";
}
extern "C" int32_t rsba_get_plan_stats(rsba_handle* h, rsba_plan_stats* out) {
  if (!h || !out) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "null argument");
  HIP_TRY(hipSetDevice(h->device));
  int32_t rc = build_solver(h);
  if (rc) return rc;
  *out = h->solver->stats;
  unsigned long long issued = 0;
  HIP_TRY(hipMemcpyAsync(&issued, h->solver->sv.schur_mfma_count, sizeof issued, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  out->schur_mfma_issued = (int64_t)issued; out->schur_launches = h->solver->schur_launches;
  return RSBA_OK;
}

extern "C" int32_t rsba_sync_block_structure(rsba_handle* h) {
  if (!h) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "null handle");
  if (h->solver) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "rsba_sync_block_structure must precede the first solve / gradient call");
  if (!h->allreduce) return RSBA_OK;
  HIP_TRY(hipSetDevice(h->device));
  const size_t F = (size_t)h->dp.F, M = (size_t)h->dp.M, count = F * F + F + M;
  std::vector<uint8_t> mask(F * F); std::vector<int64_t> cnt(F);
  int32_t rc = rsba_get_block_structure(h, mask.data(), cnt.data());
  if (rc) return rc;
  // one all-reduce (sum) of [mask | per-frame counts | per-point "observed here" flags], as doubles: the exchange's type
  std::vector<double> host(count, 0.0);
  for (size_t i = 0; i < F * F; ++i) host[i] = mask[i];
  for (size_t f = 0; f < F; ++f) host[F * F + f] = (double)cnt[f];
  for (int64_t i = 0; i < h->dp.N; ++i) host[F * F + F + (size_t)h->obs_point[i]] = 1.0;
  double* dev = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dev), count * sizeof(double)));
  hipError_t e = hipMemcpyAsync(dev, host.data(), count * sizeof(double), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) { rc = exchange(h, dev, (int64_t)count, 0, RSBA_EXCHANGE_SETUP); if (rc) { (void)hipFree(dev); return rc; } }
  if (e == hipSuccess) e = hipMemcpyAsync(host.data(), dev, count * sizeof(double), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  (void)hipFree(dev);
  if (e != hipSuccess) return rsba_set_error(RSBA_ERR_HIP, hipGetErrorString(e));
  for (size_t j = 0; j < M; ++j)
    if (host[F * F + F + j] > 1.0) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "a point has observations on more than one rank: partition the observations by point");
  for (size_t i = 0; i < F * F; ++i) mask[i] = host[i] != 0.0;
  for (size_t f = 0; f < F; ++f) cnt[f] = (int64_t)host[F * F + f];
  return rsba_set_block_structure(h, mask.data(), cnt.data());
}

extern "C" int32_t rsba_normal_equations(rsba_handle* h, double* U, double* gc, double* V, double* gp) {
  if (!h) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "null handle");
  HIP_TRY(hipSetDevice(h->device));
  int32_t rc = build_solver(h);
  if (rc) return rc;
  Solver* s = h->solver; const DeviceProblem& dp = h->dp; const int CD = s->sv.CD;
  if ((rc = reset_scales(h))) return rc;
  if ((rc = linearize(h))) return rc;
  if (U) HIP_TRY(hipMemcpyAsync(U, s->sv.U, (size_t)dp.F * CD * CD * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (gc) HIP_TRY(hipMemcpyAsync(gc, s->sv.gc, (size_t)dp.F * CD * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  std::vector<double> v6;
  if (V) { v6.resize((size_t)dp.M * 6); HIP_TRY(hipMemcpyAsync(v6.data(), s->sv.V, v6.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream)); }
  if (gp) HIP_TRY(hipMemcpyAsync(gp, s->sv.gp, (size_t)dp.M * 3 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (V) for (int j = 0; j < dp.M; ++j) {
    const double* v = &v6[(size_t)j * 6]; double* o = V + (size_t)j * 9;
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[1]; o[4] = v[3]; o[5] = v[4]; o[6] = v[2]; o[7] = v[4]; o[8] = v[5];
  }
  return RSBA_OK;
}

// Covariance of one frame's pose block(s): the (frame, frame) block of (J^T J)^-1, J = loss-corrected Jacobian of the
// problem at the current parameters on the tangent space of its parameterizations — what ceres::Covariance returns
// for the blocks (p0,p0), (p0,p1), (p1,p1) that VideoSfMHandler::BA asks for (VideoSfMHandler.cc:602-621).
// It is the same block of the inverse of the reduced camera system: S without damping and without Jacobi scaling
// (radius 1e300: fixed coordinates keep a vanishing, decoupled diagonal instead of an exact zero), one solve per
// unit vector through the factorisation.  Fixed coordinates have zero covariance.
extern "C" int32_t rsba_pose_covariance(rsba_handle* h, int32_t frame, double* cov) {
  if (!h || !cov) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "null argument");
  if (frame < 0 || frame >= h->dp.F) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "frame out of range");
  HIP_TRY(hipSetDevice(h->device));
  int32_t rc = build_solver(h);
  if (rc) return rc;
  Solver* s = h->solver; SolverDev& sv = s->sv; hipStream_t st = h->stream; const int CD = sv.CD;
  if ((rc = reset_scales(h))) return rc;
  if ((rc = linearize(h))) return rc;
  HIP_TRY(launch_clamp_diagonal(h->dp, sv, 1e-6, 1e32, st));   // only its floor matters: the decoupled diagonal of fixed coordinates
  HIP_TRY(launch_pose_prior_clamp(h->dp, s->pp, 1e-6, 1e32, st));
  HIP_TRY(hipMemsetAsync(sv.chol_fail, 0, sizeof(int), st));
  // (CD + 1 right-hand sides through ONE factorisation: the substitution-only solves walk every column's factor tiles, so a sharded
  // plan takes its replicated form here — the whole of S summed onto every rank)
  struct ShardedOff { Solver* s; bool was; ~ShardedOff() { s->sharded_off = was; } } sharded_guard{s, s->sharded_off};
  s->sharded_off = true;
  if ((rc = reduce_system(h, 1e300))) return rc;
  if (s->two_rhs) HIP_TRY(launch_ratio_prepare(s->ratio4, 1.0, 0.0, 0.0, st));   // (a plan that carries the ratio's column through its factorisation: s eta = 0 here — the plain solves S y = e_k)
  std::vector<double> col((size_t)CD * CD, 0.0);
  const double one = 1.0;
  // CD (+1 with the border) solves through the factorisation; the DAG driver's verification flag is sticky, so one read after
  // the last solve covers them all — a suspect result is thrown away and the solves are repeated on the level schedule
  const bool levels_before = s->use_levels;
  std::vector<double> vf; double hb[3] = {0.0, 0.0, 0.0};
  for (int attempt = 0; attempt < 2; ++attempt) {
  HIP_TRY(hipMemsetAsync(sv.scalars + kDagSuspect, 0, sizeof(double), st));
  for (int k = 0; k < CD; ++k) {
    HIP_TRY(hipMemsetAsync(sv.rhs, 0, (size_t)sv.npad * sizeof(double), st));
    HIP_TRY(hipMemcpyAsync(sv.rhs + (size_t)frame * CD + k, &one, sizeof(double), hipMemcpyHostToDevice, st));
    const double* y = sv.yv;
    if (k == 0) { if ((rc = solve_reduced_system(h))) return rc; y = sv.yv; }   // the factorisation, once; every other column is a pair of substitutions
    else if ((rc = solve_again(h, sv.rhs, &y))) return rc;
    HIP_TRY(hipMemcpyAsync(&col[(size_t)k * CD], y + (size_t)frame * CD, (size_t)CD * sizeof(double), hipMemcpyDeviceToHost, st));
  }
  // A free interFrameRatio is one more parameter block of J^T J, coupled to every pose through its column b (the 1-wide
  // border of the reduced system, diagonal entry h): by the block inverse the pose block of the bordered system is
  // S^-1 + v v^T / (h - b.v) with S v = b — what ceres::Covariance returns for the problem CeresHandler builds by default.
  if (s->border) {
    const double* v = nullptr;
    if ((rc = solve_again(h, s->border, &v))) return rc;
    HIP_TRY(launch_border_dots(s->border, v, v, sv.npad, s->ratio4 + 2, st));
    vf.resize(CD);
    HIP_TRY(hipMemcpyAsync(vf.data(), v + (size_t)frame * CD, (size_t)CD * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(hb, s->ratio4, sizeof hb, hipMemcpyDeviceToHost, st));     // {h, g, b.v}
  }
  double suspect = 0.0;
  if ((rc = await_verification(h))) return rc;
  HIP_TRY(hipMemcpyAsync(&suspect, sv.scalars + kDagSuspect, sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (suspect == 0.0 || s->use_levels) break;
  s->use_levels = true; ++s->dag_fallbacks;
  }
  s->use_levels = levels_before;
  int fail = 0, nfail = 0;
  HIP_TRY(hipMemcpyAsync(&fail, sv.chol_fail, sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(&nfail, h->dp.fail_count, sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (nfail) return rsba_set_error(RSBA_ERR_EVALUATION_FAILED, "residual and Jacobian evaluation failed");
  if (fail) return rsba_set_error(RSBA_ERR_UNSUPPORTED, "J^T J is rank deficient (fix the gauge): no covariance, as ceres::Covariance::Compute returns false");
  double border_scale = 0.0;
  if (s->border) {
    const double schur = hb[0] - hb[2];    // the ratio's own pivot of the bordered system
    if (!(schur > 0.0) || !std::isfinite(schur)) return rsba_set_error(RSBA_ERR_UNSUPPORTED, "J^T J is rank deficient in the interFrameRatio: no covariance");
    border_scale = 1.0 / schur;
  }
  for (int a = 0; a < CD; ++a) for (int b = 0; b < CD; ++b) {
    const double ma = h->mask_pose[(size_t)frame * CD + a], mb = h->mask_pose[(size_t)frame * CD + b];
    cov[(size_t)a * CD + b] = (ma != 0.0 && mb != 0.0) ? col[(size_t)b * CD + a] + (s->border ? vf[a] * vf[b] * border_scale : 0.0) : 0.0;
  }
  return RSBA_OK;
}

extern "C" int32_t rsba_solve(rsba_handle* h, const rsba_solver_options* opt, rsba_solver_summary* sum, rsba_iteration* trace, int32_t trace_cap) {
  if (!h || !opt || !sum) return rsba_set_error(RSBA_ERR_INVALID_ARGUMENT, "null argument");
  HIP_TRY(hipSetDevice(h->device));
  const double t_start = now_s();
  std::memset(sum, 0, sizeof *sum);
  int32_t rc = build_solver(h);
  if (rc) return rc;
  Solver* s = h->solver; SolverDev& sv = s->sv; DeviceProblem& dp = h->dp; hipStream_t st = h->stream;
  sum->termination_type = RSBA_NO_CONVERGENCE;
  s->timer.on = opt->profile_phases != 0;
  if (s->timer.on) s->timer.reset();
  s->xtimer.on = s->timer.on && h->allreduce;
  if (s->xtimer.on) s->xtimer.reset();
  struct TimerGuard {   // whichever way this call returns, later rsba_gradient / covariance calls must not keep queueing phase records
    PhaseTimer& t; PhaseTimer& x;
    ~TimerGuard() { for (PhaseTimer* q : {&t, &x}) if (q->on) { q->on = false; q->pending.clear(); q->next = 0; } }
  } timer_guard{s->timer, s->xtimer};
  { const char* lv = std::getenv("RSBA_CHOL_LEVELS"); s->use_levels = opt->level_scheduled_cholesky != 0 || (lv && lv[0] == '1'); }
  bool any_rank_needs_host = false;
  {
    // problem-size figures of the whole (all-rank) problem
    const double npri = sv.lead ? (double)h->prior_frames.size() + (double)h->pp_blocks.size() + (dp.pp_spherical >= 0 ? 1.0 : 0.0) : 0.0;
    // (+ how many ranks cannot run the loop without the host — no observations, or phase timers on: every rank must take the same form of the loop)
    const bool host_form_only = dp.N == 0 || s->timer.on || test_hook("RSBA_DEVICE_LM_OFF_ON_THIS_RANK") != nullptr;
    double cnt[4] = {(double)dp.N + npri, (double)(s->num_reduced_blocks + s->num_priors_reduced), (double)s->num_reduced_params, host_form_only ? 1.0 : 0.0};
    if (h->allreduce) {
      HIP_TRY(hipMemcpyAsync(sv.scalars + 8, cnt, sizeof cnt, hipMemcpyHostToDevice, st));
      if ((rc = exchange(h, sv.scalars + 8, 4, 0, RSBA_EXCHANGE_SETUP))) return rc;
      HIP_TRY(hipMemcpyAsync(cnt, sv.scalars + 8, sizeof cnt, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemsetAsync(sv.scalars + 8, 0, 4 * sizeof(double), st));   // slots 8-11 ride in the per-iteration sum from here on
      HIP_TRY(hipStreamSynchronize(st));
    }
    sum->num_residual_blocks = (int32_t)cnt[0]; sum->num_residual_blocks_reduced = (int32_t)cnt[1]; sum->num_parameters_reduced = (int32_t)cnt[2];
    any_rank_needs_host = cnt[3] != 0.0;
  }
  int ntrace = 0;
  auto push = [&](const rsba_iteration& it) {
    if (trace && ntrace < trace_cap) trace[ntrace] = it;
    ++ntrace; sum->num_iterations = ntrace;
    if (opt->minimizer_progress_to_stdout)
      std::printf("%4d  cost % .6e  change % .3e  |grad| %.3e  |step| %.3e  rho % .3e  radius %.3e  %s\n", it.iteration, it.cost, it.cost_change,
                  it.gradient_max_norm, it.step_norm, it.relative_decrease, it.trust_region_radius, it.step_is_successful ? "ok" : (it.iteration ? "rejected" : ""));
  };
  double host_sc[16]; double cost2[2]; int nfail = 0, cfail = 0;
  // free interFrameRatio (the reference's default for the motion priors, CeresHandler.h:161,172,175): one more unknown of
  // the LM, kept on the host — value, Jacobi scale, LM diagonal, and {h, g} = its column's J^T J and J^T r from the device.
  // The candidate is projected onto the lower bound (ParameterBlock::Plus); Ceres' projected line search is not restated.
  const bool free_ratio = s->border != nullptr;
  const double ratio_lb = dp.prior_kind == 2 ? 2.220446049250313e-16 : 0.0;
  double ratio = dp.prior_ratio, ratio_scale = 1.0, ratio_diag = 0.0, ratio_hg[2] = {0.0, 0.0}, ratio_new = dp.prior_ratio;
  auto read_back = [&]() -> int32_t {
    HIP_TRY(hipMemcpyAsync(host_sc, sv.scalars, sizeof host_sc, hipMemcpyDeviceToHost, st));
    if (free_ratio) HIP_TRY(hipMemcpyAsync(ratio_hg, s->ratio4, sizeof ratio_hg, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (s->timer.on) s->timer.resolve();
    if (s->xtimer.on) s->xtimer.resolve();
    cost2[0] = host_sc[kCost]; cost2[1] = host_sc[kFixedCost];
    nfail = host_sc[kEvalFailed] != 0.0; cfail = host_sc[kSolveFailed] != 0.0;
    return RSBA_OK;
  };
  auto finish = [&](int32_t term) -> int32_t {
    sum->termination_type = term;
    sum->is_solution_usable = term != RSBA_FAILURE;
    (void)reset_scales(h);
    const size_t npose = (size_t)dp.F * dp.P * 6, npt = (size_t)dp.M * 3;
    if (h->allreduce && h->world > 1) {   // every rank leaves with the complete point array: each point from its owner
      HIP_TRY(launch_own_points(dp, sv, s->merge_buf, st));
      int32_t rc2 = exchange(h, s->merge_buf, 4 * (int64_t)dp.M, 0, RSBA_EXCHANGE_POINTS);
      if (rc2) return rc2;
      HIP_TRY(launch_merge_points(dp, s->merge_buf, st));
    }
    HIP_TRY(hipMemcpyAsync(h->desc.poses, dp.poses, npose * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h->desc.points, dp.points, npt * sizeof(double), hipMemcpyDeviceToHost, st));
    if (!dp.calibrated) HIP_TRY(hipMemcpyAsync(h->desc.intrinsics, dp.intr, (size_t)dp.NI * 9 * sizeof(double), hipMemcpyDeviceToHost, st));
    if (dp.pp_count > 0 && h->pp_host) HIP_TRY(hipMemcpyAsync(h->pp_host, dp.pp_value, 6 * (size_t)dp.pp_count * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (s->timer.on) { s->timer.resolve(); s->timer.on = false; }
    if (s->xtimer.on) { s->xtimer.resolve(); s->xtimer.on = false; }
    h->prior_ratio_result = dp.prior_ratio;
    sum->total_time_s = now_s() - t_start;
    if (const char* path = h->solver->d_trace ? std::getenv("RSBA_CHOL_TRACE") : nullptr) {   // debugging aid, off by default
      Solver* sl = h->solver;
      std::vector<long long> tr(8 * (size_t)sl->plan.ntasks);
      HIP_TRY(hipMemcpy(tr.data(), sl->d_trace, tr.size() * sizeof(long long), hipMemcpyDeviceToHost));
      if (FILE* f = std::fopen(path, "wb")) {
        const int32_t n = sl->plan.ntasks;
        std::fwrite(&n, sizeof(n), 1, f);
        std::fwrite(sl->tasks.data(), sizeof(int32_t), sl->tasks.size(), f);
        std::fwrite(tr.data(), sizeof(long long), tr.size(), f);
        std::fclose(f);
      }
    }
    return RSBA_OK;
  };

  // ---- iteration 0: initial evaluation ----
  double t0 = now_s();
  if ((rc = reset_scales(h))) return rc;
  if ((rc = linearize(h, false, true))) return rc;
  if ((rc = gradient_max(h))) return rc;
  if ((rc = read_back())) return rc;
  sum->residual_jacobian_time_s += now_s() - t0;
  if (nfail) { sum->termination_type = RSBA_FAILURE; (void)finish(RSBA_FAILURE); return rsba_set_error(RSBA_ERR_EVALUATION_FAILED, "initial residual and Jacobian evaluation failed"); }
  double cost = cost2[0]; const double fixed = cost2[1];
  auto with_ratio_gradient = [&](double g) { return free_ratio ? std::max(g, std::fabs(ratio - std::max(ratio_lb, ratio - ratio_hg[1]))) : g; };
  double gmax = with_ratio_gradient(host_sc[kGradMax]);
  sum->fixed_cost = fixed; sum->initial_cost = cost + fixed; sum->final_cost = cost + fixed;
  double radius = opt->initial_trust_region_radius, decrease_factor = 2.0; bool reuse_diagonal = false;
  rsba_iteration it; std::memset(&it, 0, sizeof it);
  it.cost = cost + fixed; it.gradient_max_norm = gmax; it.trust_region_radius = radius;
  if (gmax <= opt->gradient_tolerance) { push(it); return finish(RSBA_CONVERGENCE); }
  if (opt->jacobi_scaling) {
    // EstimateScale from the first Jacobian, then the Jacobian is column-scaled for good; here the
    // scales feed the evaluation kernel, so re-linearise once with them
    HIP_TRY(launch_jacobi_scale(dp, sv, st));
    HIP_TRY(launch_pose_prior_scale(dp, s->pp, st));
    if (free_ratio) ratio_scale = 1.0 / (1.0 + std::sqrt(ratio_hg[0]));
    if ((rc = linearize(h))) return rc;
  }
  push(it);

  // current <-> candidate parameter buffers (intrinsics only when they are a parameter block)
  auto swap_params = [&]() {
    std::swap(dp.poses, sv.trial_poses); std::swap(dp.points, sv.trial_points);
    if (sv.NPF > 0) std::swap(dp.intr, sv.trial_intr);
    if (dp.pp_count > 0) std::swap(dp.pp_value, dp.pp_trial);
  };
  // A candidate is evaluated in LM mode straight away (residuals, Jacobian, per-wave camera blocks) when the problem keeps no
  // records (they would be overwritten and a rejected step needs the old ones): an accepted step — the rule — then re-uses that
  // evaluation for its linearisation instead of evaluating twice, a rejected one has computed Jacobians for nothing.
  bool speculate = dp.rec == nullptr || dp.rec_alt != nullptr;   // (records: only with a second set for the candidate's)
  if (const char* e = std::getenv("RSBA_SPECULATE")) speculate = speculate && e[0] != '0';
  int invalid_streak = 0, iteration = 0;
  const size_t pose_bytes = (size_t)dp.F * dp.P * 6 * sizeof(double), point_bytes = (size_t)dp.M * 3 * sizeof(double);
  (void)pose_bytes; (void)point_bytes;
  // ---- trust-region control on the device ----
  // The loop body below, decisions included, as a sequence of launches that never waits for the host: radius, accept / reject and the
  // convergence tests live in HBM (s->d_ctl), two single-thread kernels take the decisions by the same rules in the same order, the
  // kernels of an iteration read the radius there and skip themselves where the host form would not have launched them (a rejected
  // candidate is not linearised).  What the reference calls per frame — windowedBA over ~100 cameras (VideoSfMClient.cc:241-246) —
  // is where this counts: an iteration there is 0.5 ms, and the host form's 22 dependent launches, two read-backs and their gaps were
  // 0.09 ms of it.  Here an iteration is 13 launches on this stream (the small steps share launches: kernels_normal.hip) and no wait.
  // Every problem this call takes, on one rank or several (every rank takes the same form: settled with the problem-size exchange) — rounds 3 - 6
  // added them kind by kind: motion priors, a free interFrameRatio, GoodPosePrior blocks, the SphericalPrior, several intrinsics blocks (their
  // candidates' records go to a second set), GoodPosePrior blocks on several ranks.  What goes through the host form: phase timing, a rank that
  // asks for it (no observations), RSBA_DEVICE_LM=0 / RSBA_RECORDS_ALT=0, and a suspect factorisation (the level schedule repeats the iteration).
  bool device_ctl = speculate && !s->use_levels &&
                    !any_rank_needs_host && opt->max_num_iterations > 0;
  const bool has_pp = dp.pp_count > 0 || dp.pp_spherical >= 0;
  if (const char* e = std::getenv("RSBA_DEVICE_LM")) device_ctl = device_ctl && e[0] != '0';   // A/B switch: 0 = the host decides
  if (device_ctl) ++s->stats.device_loop_solves; else ++s->stats.host_loop_solves;
  if (device_ctl) {
    const int cap = opt->max_num_iterations + 2;
    if (cap > s->trace_it_cap) { if ((rc = s_alloc(s, &s->d_trace_it, (size_t)cap))) return rc; s->trace_it_cap = cap; }
    const LmRules R{opt->max_num_iterations, opt->max_num_consecutive_invalid_steps, opt->max_trust_region_radius, opt->min_trust_region_radius, opt->min_relative_decrease,
                    opt->function_tolerance, opt->gradient_tolerance, opt->parameter_tolerance};
    if (!s->h_ctl) {
      static_assert((size_t)(Solver::kCtlRing + 1) * kCtlSize * sizeof(double) <= 4096, "one pooled pinned block");
      HIP_TRY(dev_pinned_acquire(reinterpret_cast<void**>(&s->h_ctl), (size_t)(Solver::kCtlRing + 1) * kCtlSize * sizeof(double)));
      HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&s->h_ctl_dev), s->h_ctl, 0));
      std::fill(s->h_ctl, s->h_ctl + (size_t)(Solver::kCtlRing + 1) * kCtlSize, 0.0);
    }
    double* const hc0 = s->h_ctl + (size_t)Solver::kCtlRing * kCtlSize;   // the initial state (pinned: the upload does not wait for the host)
    std::fill(hc0, hc0 + kCtlSize, 0.0);
    hc0[kCtlRadius] = radius; hc0[kCtlDecrease] = decrease_factor; hc0[kCtlCost] = cost; hc0[kCtlFixed] = fixed; hc0[kCtlGmax] = gmax; hc0[kCtlFinalCost] = sum->final_cost;
    HIP_TRY(hipMemcpyAsync(s->d_ctl, hc0, kCtlSize * sizeof(double), hipMemcpyHostToDevice, st));
    struct CtlGuard {   // whichever way this block is left, the host form finds the state it expects: nobody skips, the radius (and the ratio) come by value
      rsba_handle* h; Solver* s;
      ~CtlGuard() { (void)hipStreamSynchronize(h->stream); s->sv.ctl = nullptr; h->dp.ctl = nullptr; h->dp.prior_ratio_ptr = nullptr; s->clamp_with_factor = false; (void)hipMemsetAsync(s->d_ctl, 0, kCtlSize * sizeof(double), h->stream); }
    } ctl_guard{h, s};
    sv.ctl = s->d_ctl; dp.ctl = s->d_ctl;
    if (free_ratio) {   // the ratio joins the state on the device: value, Jacobi scale, lower bound ({h, g} of the last linearisation are there)
      HIP_TRY(launch_ratio_init(s->ratio4, ratio, ratio_scale, ratio_lb, st));
      dp.prior_ratio_ptr = s->ratio4 + kRtRatio;
    }
    s->clamp_with_factor = true; s->clamp_lo_hi[0] = opt->min_lm_diagonal; s->clamp_lo_hi[1] = opt->max_lm_diagonal;
    HIP_TRY(launch_begin_solve(sv, st));   // (from here on the last kernel of an iteration clears the two flags for the next)
    // The last kernel of an iteration writes the state to a slot of pinned host memory and stamps it; the host polls the stamp (no
    // event, no copy in the stream) and enqueues the next iteration the moment it shows.  RSBA_LM_AHEAD = k keeps k iterations
    // enqueued beyond the one whose outcome the host has seen (those behind a termination fall through: every kernel looks at the
    // status word); measured at 100 and 1 000 cameras the queue does not need it — 0.492 / 0.498 / 0.500 ms per iteration for
    // k = 0 / 1 / 2 (profiles/r04/iteration_gaps.txt) — so the default enqueues nothing that might not be wanted.
    int ahead = 0;
    if (const char* e = std::getenv("RSBA_LM_AHEAD")) ahead = std::min(Solver::kCtlRing - 2, std::max(0, std::atoi(e)));   // (a slot is written again only after the host has moved on from it)
    const double* hc = hc0;
    int enqueued = 0, looked = 0;
    bool stopped = false;
    t0 = now_s();
    const double seq0 = s->ctl_seq;
    auto look = [&]() -> int32_t {   // the state behind iteration `looked`: wait for its stamp (the deciding kernel writes it last)
      const double* slot = s->h_ctl + (size_t)(looked % Solver::kCtlRing) * kCtlSize;
      const double want = seq0 + (double)(looked + 1);
      for (unsigned spins = 0;; ++spins) {
        if (__atomic_load_n(reinterpret_cast<const uint64_t*>(slot + kCtlSeq), __ATOMIC_ACQUIRE) == *reinterpret_cast<const uint64_t*>(&want)) break;
        if ((spins & 0xFFFu) == 0xFFFu) {   // now and then: is the stream still alive?  (an idle stream whose stamp never came is an error, not a wait)
          const hipError_t q = hipStreamQuery(st);
          if (q == hipSuccess) { if (__atomic_load_n(reinterpret_cast<const uint64_t*>(slot + kCtlSeq), __ATOMIC_ACQUIRE) == *reinterpret_cast<const uint64_t*>(&want)) break; return rsba_set_error(RSBA_ERR_HIP, "the trust-region state of an iteration never reached the host"); }
          if (q != hipErrorNotReady) return rsba_set_error(RSBA_ERR_HIP, hipGetErrorString(q));
        }
        __builtin_ia32_pause();
      }
      hc = slot;
      ++looked;
      stopped = hc[kCtlStatus] != 0.0;
      return RSBA_OK;
    };
    const bool multi = h->allreduce != nullptr;   // several ranks: the same loop with the three exchanges of an iteration enqueued between its kernels (RCCL: stream-ordered, no host wait)
    while (!stopped && enqueued < opt->max_num_iterations) {
      // (thirteen launches on this stream; the steps the host form spreads over twenty-two, in its order: kernels_normal.hip, "the same steps in
      // fewer launches".  The diagonal's clamp rides in the point factor's launch — after a rejected step it recomputes what is there.)
      if (free_ratio) HIP_TRY(launch_ratio_prepare_ctl(s->ratio4, s->d_ctl, opt->min_lm_diagonal, opt->max_lm_diagonal, st));   // the ratio's damped pivot and gradient for the ETA task of the factorisation
      if (dp.pp_count > 0) HIP_TRY(launch_pose_prior_clamp(dp, s->pp, opt->min_lm_diagonal, opt->max_lm_diagonal, st));   // (the priorPoses coordinates' LM diagonal: recomputed from what the last accepted linearisation left — the same numbers after a rejected step)
      if ((rc = factor_and_solve(h, 1.0))) return rc;   // (the radius argument is ignored: the kernels read ctl)
      if (free_ratio) HIP_TRY(launch_ratio_candidate(s->ratio4, s->d_ctl, st));
      HIP_TRY(launch_candidate_and_model_cost(dp, sv, st));
      if (s->ucross && (sv.lead || h->prior_split)) HIP_TRY(launch_prior_model(dp, sv, sv.scalars + kModelCostChange, 0.0, st, free_ratio ? s->ratio4 + kRtC : nullptr));   // motion priors: their share of the model cost change (a free ratio's step included) ...
      if (has_pp) HIP_TRY(launch_pose_prior_step(dp, sv, s->pp, 1.0, st));   // per-pose priors: the candidate priorPoses values, their share of the three sums (the lead rank's to add)
      swap_params();
      { DeviceProblem dq = dp; dq.rec_candidate = 1; HIP_TRY(launch_eval(dq, kLmJacobian, st)); }   // (a problem that keeps records: the candidate's go to the other set)
      const bool extra_cost = s->ucross != nullptr || has_pp;   // prior blocks add their cost behind the observations': the cost is reduced by a launch of its own then
      if (extra_cost) {                                                                                // ... their cost at the candidate, behind the observations' ...
        HIP_TRY(launch_cost_reduce(dp, h->d_cost2, st));
        if (s->ucross && (sv.lead || h->prior_split)) {
          DeviceProblem dq = dp;
          if (free_ratio) dq.prior_ratio_ptr = s->ratio4 + kRtRatioEval;   // (... at the candidate's ratio)
          HIP_TRY(launch_prior_cost(dq, h->d_cost2, h->prior_invalid, st));
        }
        if (has_pp && sv.lead) HIP_TRY(launch_pose_prior_cost(dp, h->d_cost2, st));   // (replicated blocks: the lead rank's share of the summed cost)
      }
      swap_params();
      if ((rc = await_verification(h))) return rc;
      const bool my_priors = s->ucross && (sv.lead || h->prior_split);
      if (!multi) HIP_TRY(launch_lm_verdict_step(dp, sv, h->d_cost2, s->d_ctl, R, s->d_trace_it, cap, st, /*cost_reduced=*/extra_cost));
      else {   // several ranks: the scalars of the step are summed over the ranks between the reduction and the decision — exchange (3), enqueued like a kernel
        if (!extra_cost) HIP_TRY(launch_cost_reduce(dp, h->d_cost2, st));
        HIP_TRY(launch_pack_trial(dp, sv, h->d_cost2, st));
        if ((rc = exchange(h, sv.scalars, 12, 0, RSBA_EXCHANGE_SCALARS))) return rc;
        HIP_TRY(launch_lm_decide_step(sv, s->d_ctl, R, s->d_trace_it, cap, st));
      }
      bool fused = false;
      HIP_TRY(launch_linearize_blocks(dp, sv, st, &fused));   // camera blocks, the accepted candidate's copy over x, point blocks: side by side in one launch
      if (!fused) HIP_TRY(launch_camera_blocks(dp, sv, st, /*take_candidate=*/true, /*padding_is_zero=*/true));   // (the initial linearisation zeroed the pseudo frames' padding)
      if (my_priors) HIP_TRY(launch_prior_blocks(dp, sv, s->ucross, st));                             // ... and their blocks of an accepted step's linearisation
      if (free_ratio) HIP_TRY(launch_prior_border(all_priors(h), sv, s->border, s->ratio4, st));      // (the ratio's column at the accepted point: every rank, from replicated poses)
      if (has_pp) { HIP_TRY(launch_pose_prior_take(dp, sv, st)); HIP_TRY(launch_pose_prior_blocks(dp, sv, s->pp, st)); }   // per-pose priors: the accepted values, their blocks
      HIP_TRY(launch_intr_blocks(dp, sv, st));
      if (!fused) HIP_TRY(launch_point_blocks(dp, sv, st));
      s->ctl_seq += 1.0;
      double* const slot = s->h_ctl_dev + (size_t)(enqueued % Solver::kCtlRing) * kCtlSize;
      if (!multi) {
        HIP_TRY(launch_lm_linearize_gradient(dp, sv, h->d_cost2, st));
        const int ngm = (int)((sv.n + 3 * (int64_t)dp.M + 255) / 256);
        if (dp.pp_count > 0) HIP_TRY(launch_pose_prior_gradmax(dp, sv, s->pp, st, sv.partial + ngm));   // (one more partial maximum for the verdict)
        HIP_TRY(launch_lm_verdict_gradient(dp, sv, s->d_ctl, R, s->d_trace_it, cap, slot, s->ctl_seq, st, false, dp.pp_count > 0 ? 1 : 0));
      } else {   // exchange (1): the camera gradient, diag(U), the cost — and every rank's gradient maximum over its points — whether or not the candidate
                 // was accepted (the host does not know): the unpacking skips itself after a rejected one, the maximum comes out as it was
        const bool ride = h->world <= kMaxRankSlots && dp.pp_count == 0;   // (the priorPoses blocks' maximum is the lead rank's alone: a MAX exchange of its own then, as in the host form)
        HIP_TRY(launch_pack_linearize(dp, sv, h->d_cost2, st, ride ? h->world : 0));
        if (ride) HIP_TRY(launch_gradient_max_points(dp, sv, h->rank, st));
        if ((rc = exchange(h, sv.xbuf, 2 * sv.n + 3 + (ride ? h->world : 0), 0, RSBA_EXCHANGE_CAMERA))) return rc;
        HIP_TRY(launch_unpack_linearize(dp, sv, st));
        if (ride) HIP_TRY(launch_gradient_max_cameras(dp, sv, h->world, st));
        else {
          HIP_TRY(launch_gradient_max(dp, sv, st));
          if (sv.lead) HIP_TRY(launch_pose_prior_gradmax(dp, sv, s->pp, st));
          if ((rc = exchange(h, sv.scalars + kGradMax, 1, 1, RSBA_EXCHANGE_SCALARS))) return rc;
        }
        HIP_TRY(launch_lm_verdict_gradient(dp, sv, s->d_ctl, R, s->d_trace_it, cap, slot, s->ctl_seq, st, /*gradmax_done=*/true));
      }
      ++enqueued;
      if (enqueued - looked > ahead) { if ((rc = look())) return rc; }
    }
    while (!stopped && looked < enqueued) { if ((rc = look())) return rc; }
    // (iterations enqueued behind the termination fall through; the stream is drained below, before anything of the loop is read or torn down)
    {
      const int have = std::min((int)hc[kCtlNumTrace], cap);
      std::vector<rsba_iteration> recs((size_t)std::max(have, 1));
      if (have > 0) HIP_TRY(hipMemcpyAsync(recs.data(), s->d_trace_it, (size_t)have * sizeof(rsba_iteration), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      for (int k = 0; k < have; ++k) push(recs[(size_t)k]);
    }
    sum->linear_solver_time_s += now_s() - t0;
    if (free_ratio) {   // the ratio's state back to the host (the stream is idle)
      double rt[kRtSize];
      HIP_TRY(hipMemcpy(rt, s->ratio4, sizeof rt, hipMemcpyDeviceToHost));
      ratio = rt[kRtRatio]; ratio_new = ratio; ratio_diag = rt[kRtDiag]; ratio_hg[0] = rt[kRtH]; ratio_hg[1] = rt[kRtG];
      dp.prior_ratio = ratio;
    }
    radius = hc[kCtlRadius]; decrease_factor = hc[kCtlDecrease]; cost = hc[kCtlCost]; gmax = hc[kCtlGmax];
    if (dp.rec_alt && hc[kCtlRecSel] != 0.0) std::swap(dp.rec, dp.rec_alt);   // (the set that holds the current point's records is dp.rec again, as the host form has it)
    iteration = (int)hc[kCtlIteration]; invalid_streak = (int)hc[kCtlInvalidStreak];
    sum->num_successful_steps = (int)hc[kCtlSuccessful]; sum->num_unsuccessful_steps = (int)hc[kCtlUnsuccessful]; sum->final_cost = hc[kCtlFinalCost];
    const double status = hc[kCtlStatus];
    if (status > 0.0) return finish((int32_t)status - 1);
    if (status == -2.0) { (void)finish(RSBA_FAILURE); return rsba_set_error(RSBA_ERR_EVALUATION_FAILED, "residual and Jacobian evaluation failed"); }
    // status -1: the persistent driver's solution of the last iteration does not satisfy its system.  Nothing of that iteration has
    // touched x or the state: the host form below repeats it, and finishes the problem, on the level schedule.
    s->use_levels = true; ++s->dag_fallbacks; ++sum->num_dag_fallbacks;
    s->sharded_off = true;   // (a sharded factorisation goes back to the replicated one: the level schedule needs the whole of S on every rank)
    reuse_diagonal = true;   // (the diagonal is in place)
  }
  while (true) {
    if (iteration >= opt->max_num_iterations) return finish(RSBA_NO_CONVERGENCE);
    t0 = now_s();
    if (!reuse_diagonal) {
      PhaseScope ps(h, RSBA_PHASE_OTHER);
      HIP_TRY(launch_clamp_diagonal(dp, sv, opt->min_lm_diagonal, opt->max_lm_diagonal, st));
      HIP_TRY(launch_pose_prior_clamp(dp, s->pp, opt->min_lm_diagonal, opt->max_lm_diagonal, st));
      ratio_diag = std::min(std::max(ratio_scale * ratio_scale * ratio_hg[0], opt->min_lm_diagonal), opt->max_lm_diagonal);
    }
    HIP_TRY(launch_begin_solve(sv, st));   // chol_fail = 0 and the sticky verification flag of the DAG Cholesky = 0 (any solve of this iteration may raise it): one launch
    RatioStep rs{ratio_scale * ratio_scale * ratio_hg[0] + ratio_diag / radius, ratio_scale * ratio_hg[1], ratio_scale, 0.0};
    if ((rc = factor_and_solve(h, radius, free_ratio ? &rs : nullptr))) return rc;
    reuse_diagonal = true;
    const double ratio_step = free_ratio ? ratio_scale * rs.eta : 0.0;      // the ratio's step in its own units is -ratio_step
    ratio_new = free_ratio ? std::max(ratio_lb, ratio - ratio_step) : ratio;
    { PhaseScope ps(h, RSBA_PHASE_BACK_SUBSTITUTE); HIP_TRY(launch_model_cost_change(dp, sv, st)); }
    if (s->ucross && (sv.lead || h->prior_split)) { PhaseScope ps(h, RSBA_PHASE_PRIORS); HIP_TRY(launch_prior_model(dp, sv, sv.scalars + kModelCostChange, std::isfinite(ratio_step) ? ratio_step : 0.0, st)); }
    { PhaseScope ps(h, RSBA_PHASE_CANDIDATE); HIP_TRY(launch_candidate(dp, sv, st)); }
    if (dp.pp_count > 0 || dp.pp_spherical >= 0) { PhaseScope ps(h, RSBA_PHASE_PRIORS); HIP_TRY(launch_pose_prior_step(dp, sv, s->pp, radius, st)); }   // every rank: candidate priorPoses values (scalars from the lead rank only)
    // residuals only at the candidate (T = double path)
    swap_params();
    {
      PhaseScope ps(h, RSBA_PHASE_EVAL_TRIAL);
      { DeviceProblem dq = dp; dq.rec_candidate = 1; HIP_TRY(launch_eval(dq, speculate ? kLmJacobian : kResidualOnly, st)); }
      HIP_TRY(launch_cost_reduce(dp, h->d_cost2, st));
    }
    if (s->ucross && (sv.lead || h->prior_split)) {
      PhaseScope ps(h, RSBA_PHASE_PRIORS);
      dp.prior_ratio = std::isfinite(ratio_new) ? ratio_new : ratio;
      HIP_TRY(launch_prior_cost(dp, h->d_cost2, h->prior_invalid, st));
      dp.prior_ratio = ratio;
    }
    if (sv.lead) HIP_TRY(launch_pose_prior_cost(dp, h->d_cost2, st));
    swap_params();
    // exchange (3): model decrease, |step|^2, |x|^2, (skip the max slot), trial cost, -, failure flags
    {
      PhaseScope ps(h, RSBA_PHASE_EXCHANGE);
      if ((rc = await_verification(h))) return rc;   // (its flag rides in the scalars below)
      HIP_TRY(launch_pack_trial(dp, sv, h->d_cost2, st));
      if ((rc = exchange(h, sv.scalars, 12, 0, RSBA_EXCHANGE_SCALARS))) return rc;   // one sum: ... and the verification flag of the Cholesky driver (every rank decides alike); slot kGradMax comes back as it was (pack_trial: the lead rank's copy, zeros from the others)
    }
    if ((rc = read_back())) return rc;
    if (host_sc[kDagSuspect] != 0.0 && !s->use_levels) {
      // the persistent driver's solution does not satisfy the system it was given: nothing of this iteration has touched
      // x yet — repeat it, and finish this problem, on the level schedule
      s->use_levels = true; ++s->dag_fallbacks; ++sum->num_dag_fallbacks;
      s->sharded_off = true;   // (a sharded factorisation goes back to the replicated one: the level schedule needs the whole of S on every rank)
      continue;   // (the flag is cleared at the top of the iteration)
    }
    cost2[1] = 0.0;   // the trial evaluation reports the total in kCost
    sum->linear_solver_time_s += now_s() - t0;
    ++iteration;
    std::memset(&it, 0, sizeof it); it.iteration = iteration;
    if (free_ratio) { host_sc[kStepSq] += (ratio - ratio_new) * (ratio - ratio_new); host_sc[kXSq] += ratio * ratio; }
    const double model_cost_change = host_sc[kModelCostChange];
    const bool solved = !cfail && std::isfinite(model_cost_change) && std::isfinite(host_sc[kStepSq]);
    const bool valid = solved && model_cost_change >= 0.0;
    it.model_cost_change = solved ? model_cost_change : 0.0;
    if (!valid) {
      if (++invalid_streak >= opt->max_num_consecutive_invalid_steps) { it.cost = cost + fixed; it.trust_region_radius = radius; push(it); return finish(RSBA_FAILURE); }
      radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;   // StepIsInvalid == StepRejected(0)
      ++sum->num_unsuccessful_steps;
      it.gradient_max_norm = gmax;
    } else {
      invalid_streak = 0; it.step_is_valid = 1;
      const double new_cost = nfail ? std::numeric_limits<double>::max() : (cost2[0] + cost2[1]) - fixed;
      it.step_norm = std::sqrt(host_sc[kStepSq]);
      const double x_norm = std::sqrt(host_sc[kXSq]);
      if (it.step_norm <= opt->parameter_tolerance * (x_norm + opt->parameter_tolerance)) { it.cost = cost + fixed; it.trust_region_radius = radius; push(it); return finish(RSBA_CONVERGENCE); }
      it.cost_change = cost - new_cost;
      if (std::fabs(it.cost_change) < opt->function_tolerance * cost) { it.cost = cost + fixed; it.trust_region_radius = radius; push(it); return finish(RSBA_CONVERGENCE); }
      it.relative_decrease = it.cost_change / model_cost_change;
      if (it.relative_decrease > opt->min_relative_decrease) {
        it.step_is_successful = 1; ++sum->num_successful_steps;
        const double t3 = 2.0 * it.relative_decrease - 1.0;   // (the cube by two multiplications — what the deciding kernel of the device-side loop computes, bit for bit)
        radius = radius / std::max(1.0 / 3.0, 1.0 - t3 * t3 * t3);
        radius = std::min(opt->max_trust_region_radius, radius); decrease_factor = 2.0; reuse_diagonal = false;
        swap_params();   // x = x_plus_delta
        if (speculate && dp.rec_alt) std::swap(dp.rec, dp.rec_alt);   // ... and its records, where the problem keeps them
        if (free_ratio) { ratio = ratio_new; dp.prior_ratio = ratio; }
        if (sv.NPF > 0) HIP_TRY(hipMemcpyAsync(sv.trial_intr, dp.intr, 9 * (size_t)dp.NI * sizeof(double), hipMemcpyDeviceToDevice, st));   // constant coordinates stay in sync
        t0 = now_s();
        if ((rc = linearize(h, speculate, true))) return rc;
        if ((rc = gradient_max(h))) return rc;
        if ((rc = read_back())) return rc;
        sum->residual_jacobian_time_s += now_s() - t0;
        if (nfail) { push(it); (void)finish(RSBA_FAILURE); return rsba_set_error(RSBA_ERR_EVALUATION_FAILED, "residual and Jacobian evaluation failed"); }
        cost = cost2[0]; gmax = with_ratio_gradient(host_sc[kGradMax]);
        it.gradient_max_norm = gmax;
        sum->final_cost = std::min(sum->final_cost, cost + fixed);
        if (gmax <= opt->gradient_tolerance) { it.cost = cost + fixed; it.trust_region_radius = radius; push(it); return finish(RSBA_CONVERGENCE); }
      } else {
        ++sum->num_unsuccessful_steps; it.gradient_max_norm = gmax;
        radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
      }
    }
    it.cost = cost + fixed; it.trust_region_radius = radius;
    push(it);
    if (radius < opt->min_trust_region_radius) return finish(RSBA_CONVERGENCE);
  }
}
