// C-ABI implementation (include/rsba_amd.h).  Host-side glue only: every number on the path is
// produced by the HIP kernels; there is no CPU evaluation path in this library.
#include "../../include/rsba_amd.h"

#include <algorithm>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <numeric>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "handle.hpp"
#include "devmem.hpp"

using namespace rsba;

namespace {
thread_local std::string g_last_error;
int32_t fail(int32_t code, const std::string& msg) { g_last_error = msg; return code; }
}  // namespace

int32_t rsba_set_error(int32_t code, const char* msg) { return fail(code, msg); }

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? RSBA_ERR_OUT_OF_MEMORY : RSBA_ERR_HIP, \
                                      std::string(#expr) + ": " + hipGetErrorString(e_));          \
  } while (0)

template <class T>
static int32_t dev_alloc(rsba_handle* h, T** p, size_t count) {
  void* q = nullptr;
  HIP_TRY(rsba::dev_malloc(&q, std::max<size_t>(count, 1) * sizeof(T)));
  h->allocs.push_back(q);
  *p = static_cast<T*>(q);
  return RSBA_OK;
}
template <class T>
static int32_t dev_upload(rsba_handle* h, T** p, const T* src, size_t count) {
  int32_t rc = dev_alloc(h, p, count);
  if (rc) return rc;
  if (count) HIP_TRY(hipMemcpy(*p, src, count * sizeof(T), hipMemcpyHostToDevice));
  return RSBA_OK;
}

// device scratch that outlives the call: (re)allocated only when it has to grow
template <class T>
static int32_t grow(rsba_handle* h, T** p, int64_t* cap, int64_t need) {
  if (need <= *cap && *p) return RSBA_OK;
  if (*p) { (void)hipStreamSynchronize(h->stream); rsba::dev_free(*p); h->allocs.erase(std::remove(h->allocs.begin(), h->allocs.end(), static_cast<void*>(*p)), h->allocs.end()); *p = nullptr; }
  int32_t rc = dev_alloc(h, p, (size_t)need);
  if (rc == RSBA_OK) *cap = need;
  return rc;
}

// The two host copies of the observation indices a handle keeps (h->obs_frame, h->obs_point: 41 MB each at 4k cameras) come out of a small
// process-wide pool: a fresh 41 MB vector is 10 000 page faults when it is filled (9 ms of a 4k-camera rsba_create for the two) and an
// munmap when it goes (11 ms of rsba_destroy) — windowedBA builds and destroys a handle per frame (VideoSfMHandler.cc:185-214).
// rsba_release_host_scratch() empties it.
namespace {
struct IndexPool { std::mutex m; std::vector<std::vector<int32_t>> idle; };
IndexPool& index_pool() { static IndexPool* p = new IndexPool(); return *p; }
std::vector<int32_t> index_vector_take(size_t n) {
  IndexPool& p = index_pool();
  std::lock_guard<std::mutex> lk(p.m);
  size_t best = p.idle.size();
  for (size_t i = 0; i < p.idle.size(); ++i)
    if (p.idle[i].capacity() >= n && (best == p.idle.size() || p.idle[i].capacity() < p.idle[best].capacity())) best = i;
  if (best == p.idle.size()) return {};
  std::vector<int32_t> v = std::move(p.idle[best]);
  p.idle.erase(p.idle.begin() + (std::ptrdiff_t)best);
  return v;
}
void index_vector_give(std::vector<int32_t>&& v) {
  if (v.capacity() < (size_t)1 << 18) return;   // (small ones are the allocator's business)
  IndexPool& p = index_pool();
  std::lock_guard<std::mutex> lk(p.m);
  if (p.idle.size() < 4) { v.clear(); p.idle.push_back(std::move(v)); }
}
void index_pool_release() { IndexPool& p = index_pool(); std::lock_guard<std::mutex> lk(p.m); p.idle.clear(); p.idle.shrink_to_fit(); }
}  // namespace

namespace rsba {
hipError_t allow_dynamic_lds_impl(const void* kernel, size_t bytes) {
  struct Entry { const void* fn; unsigned long long devices; size_t bytes; };
  static Entry table[64];
  static int used = 0;
  static std::mutex mu;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mu);
  Entry* ent = nullptr;
  for (int i = 0; i < used; ++i) if (table[i].fn == kernel) { ent = &table[i]; break; }
  if (ent && dev >= 0 && dev < 64 && (ent->devices >> dev & 1ull) && ent->bytes >= bytes) return hipSuccess;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return e;
  if (!ent && used < 64) { ent = &table[used++]; ent->fn = kernel; ent->devices = 0; ent->bytes = 0; }
  if (ent && dev >= 0 && dev < 64) { if (bytes > ent->bytes) { ent->bytes = bytes; ent->devices = 0; } ent->devices |= 1ull << dev; }
  return hipSuccess;
}
}  // namespace rsba

extern "C" {

int32_t rsba_abi_version(void) { return RSBA_AMD_ABI_VERSION; }

const char* rsba_status_string(int32_t s) {
  switch (s) {
    case RSBA_OK: return "ok";
    case RSBA_ERR_INVALID_ARGUMENT: return "invalid argument";
    case RSBA_ERR_NO_DEVICE: return "no HIP device (this library has no CPU fallback)";
    case RSBA_ERR_HIP: return "HIP error";
    case RSBA_ERR_EVALUATION_FAILED: return "evaluation failed (a residual functor returned false)";
    case RSBA_ERR_OUT_OF_MEMORY: return "out of device memory";
    case RSBA_ERR_UNSUPPORTED: return "unsupported configuration";
    case RSBA_ERR_COMM: return "collective exchange failed";
  }
  return "unknown status";
}

const char* rsba_last_error(void) { return g_last_error.c_str(); }

int32_t rsba_device_count(int32_t* count) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (count) *count = (e == hipSuccess) ? n : 0;
  if (e != hipSuccess || n <= 0) return fail(RSBA_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  return RSBA_OK;
}

// h->order (internal observation -> caller's index) of a handle whose observations came in frame-major order is the identity and
// is written out on first use only
static void ensure_order(rsba_handle* h) {
  if (h->identity_order && (int64_t)h->order.size() != h->dp.N) { h->order.resize(h->dp.N); std::iota(h->order.begin(), h->order.end(), (int64_t)0); }
}

int32_t rsba_create(const rsba_problem_desc* d, int32_t device, rsba_handle** out) {
  if (!d || !out) return fail(RSBA_ERR_INVALID_ARGUMENT, "null argument");
  *out = nullptr;
  if (d->poses_per_frame != 1 && d->poses_per_frame != 2) return fail(RSBA_ERR_INVALID_ARGUMENT, "poses_per_frame must be 1 or 2");
  if (d->num_frames <= 0 || d->num_points <= 0 || d->num_intrinsics <= 0 || d->num_observations < 0)
    return fail(RSBA_ERR_INVALID_ARGUMENT, "empty frames / points / intrinsics");
  if (!d->poses || !d->points || !d->intrinsics || (d->num_observations && (!d->obs_xy || !d->obs_frame || !d->obs_point)))
    return fail(RSBA_ERR_INVALID_ARGUMENT, "null array");
  if (d->shutter < 0 || d->shutter > 2) return fail(RSBA_ERR_INVALID_ARGUMENT, "shutter");
  if (d->shutter != RSBA_SHUTTER_GLOBAL && d->poses_per_frame == 2 && d->scanlines[0] == d->scanlines[1])
    return fail(RSBA_ERR_INVALID_ARGUMENT, "scanlines[0] == scanlines[1]");
  if (d->num_intrinsics > 1 && !d->frame_intrinsics) return fail(RSBA_ERR_INVALID_ARGUMENT, "frame_intrinsics required when num_intrinsics > 1");
  const int64_t N = d->num_observations;
  const bool dbg = std::getenv("RSBA_DEBUG_PLAN") != nullptr;
  auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_mark = dbg ? now_ms() : 0.0; char stages[256] = ""; int stages_n = 0;
  auto stage = [&](const char* name) { if (!dbg) return; const double t = now_ms(); stages_n += std::snprintf(stages + stages_n, sizeof stages - stages_n, " %s %.2f;", name, t - t_mark); t_mark = t; };
  bool frame_major = true;   // (one pass: index ranges and whether the list is already in frame-major order)
  // ... on a few threads for large lists (10 M observations: 80 MB of indices), and — the list being frame-major as a rule — beside the two
  // host copies of the indices the plan keeps (h->obs_frame, h->obs_point): two more threads
  std::vector<int32_t> of = index_vector_take((size_t)N), op = index_vector_take((size_t)N);
  of.clear(); op.clear();
  {
    const uint32_t nf = (uint32_t)d->num_frames, np = (uint32_t)d->num_points;
    const int T = N >= ((int64_t)1 << 20) ? 4 : 1;
    std::vector<uint8_t> ok((size_t)T, 1), mono((size_t)T, 1);
    auto check = [&](int t) {
      const int64_t i0 = N * t / T, i1 = N * (t + 1) / T;
      bool in_range = true, fm = true; int32_t prev = i0 > 0 ? d->obs_frame[i0 - 1] : 0;
      for (int64_t i = i0; i < i1; ++i) {
        const int32_t f = d->obs_frame[i];
        in_range = in_range && (uint32_t)f < nf && (uint32_t)d->obs_point[i] < np;
        fm = fm && f >= prev; prev = f;
      }
      ok[(size_t)t] = in_range; mono[(size_t)t] = fm;
    };
    if (T == 1) check(0);
    else {
      std::vector<std::thread> pool;
      for (int t = 0; t < T; ++t) pool.emplace_back(check, t);
      pool.emplace_back([&] { of.assign(d->obs_frame, d->obs_frame + N); });
      pool.emplace_back([&] { op.assign(d->obs_point, d->obs_point + N); });
      for (auto& th : pool) th.join();
    }
    for (int t = 0; t < T; ++t) { frame_major = frame_major && mono[(size_t)t]; if (!ok[(size_t)t]) { index_vector_give(std::move(of)); index_vector_give(std::move(op)); return fail(RSBA_ERR_INVALID_ARGUMENT, "observation index out of range"); } }
  }
  if (d->frame_intrinsics) for (int f = 0; f < d->num_frames; ++f)
    if (d->frame_intrinsics[f] < 0 || d->frame_intrinsics[f] >= d->num_intrinsics) return fail(RSBA_ERR_INVALID_ARGUMENT, "frame_intrinsics out of range");

  stage("index check + host copies");
  int32_t ndev = 0;
  int32_t rc = rsba_device_count(&ndev);
  if (rc) return rc;
  if (device < 0 || device >= ndev) return fail(RSBA_ERR_INVALID_ARGUMENT, "device ordinal out of range");
  HIP_TRY(hipSetDevice(device));

  rsba_handle* h = new rsba_handle();
  h->device = device;
  h->desc = *d;
  auto bail = [&](int32_t code) { rsba_destroy(h); return code; };
  if (rsba::dev_stream_acquire(&h->own_stream) != hipSuccess) { delete h; return fail(RSBA_ERR_HIP, "hipStreamCreate"); }
  h->stream = h->own_stream;

  // frame-major order: the order CeresHandler::Add produces is already frame-major
  // (VideoSfMHandler.cc:587-590); anything else is stably sorted once here.
  // (in that case nothing is permuted: the caller's arrays are uploaded as they are and h->order is only written out when
  // somebody asks for it — a fresh handle per call is windowedBA's pattern, VideoSfMHandler.cc:185-214)
  h->identity_order = frame_major;
  std::vector<double> xy;
  if (frame_major) { if ((int64_t)of.size() != N) { of.assign(d->obs_frame, d->obs_frame + N); op.assign(d->obs_point, d->obs_point + N); } }
  else {
    h->order.resize(N);
    std::iota(h->order.begin(), h->order.end(), (int64_t)0);
    std::stable_sort(h->order.begin(), h->order.end(), [&](int64_t a, int64_t b) { return d->obs_frame[a] < d->obs_frame[b]; });
    xy.resize(2 * (size_t)N); of.resize(N); op.resize(N);
    for (int64_t i = 0; i < N; ++i) {
      const int64_t u = h->order[i];
      xy[2 * i] = d->obs_xy[2 * u]; xy[2 * i + 1] = d->obs_xy[2 * u + 1];
      of[i] = d->obs_frame[u]; op[i] = d->obs_point[u];
    }
  }

  DeviceProblem& dp = h->dp;
  std::memset(&dp, 0, sizeof dp);
  dp.pp_spherical = -1;
  dp.shutter = d->shutter; dp.scan0 = d->scanlines[0]; dp.scan1 = d->scanlines[1];
  dp.interp_rotation = d->interpolate_rotation != 0; dp.calibrated = d->calibrated != 0; dp.P = d->poses_per_frame;
  dp.F = d->num_frames; dp.M = d->num_points; dp.NI = d->num_intrinsics; dp.N = N;
  dp.ntiles = std::max<int64_t>((N + kEvalBlock - 1) / kEvalBlock, 1);
  dp.K = (dp.calibrated ? 0 : 9) + 6 * dp.P + 3;
  dp.huber_a = d->huber_a;
  const size_t npose = (size_t)dp.F * dp.P * 6;

  stage("handle");
  double2* dxy = nullptr; int32_t *dof = nullptr, *dop = nullptr, *dfi = nullptr;
  if ((rc = dev_alloc(h, reinterpret_cast<double**>(&dxy), 2 * (size_t)N))) return bail(rc);
  if ((rc = dev_alloc(h, &dof, (size_t)N))) return bail(rc);
  if ((rc = dev_alloc(h, &dop, (size_t)N))) return bail(rc);
  {   // the three observation arrays in ONE staged upload (devmem.hpp: pinned ring, a few copying threads; small lists: hipMemcpy)
    const rsba::UploadSeg segs[3] = {{dxy, frame_major ? d->obs_xy : xy.data(), 2 * (size_t)N * sizeof(double)}, {dof, of.data(), (size_t)N * sizeof(int32_t)}, {dop, op.data(), (size_t)N * sizeof(int32_t)}};
    const hipError_t ue = rsba::dev_upload_staged(segs, 3);
    if (ue != hipSuccess) return bail(fail(ue == hipErrorOutOfMemory ? RSBA_ERR_OUT_OF_MEMORY : RSBA_ERR_HIP, std::string("upload of the observations: ") + hipGetErrorString(ue)));
  }
  stage("observations to the device");
  std::vector<int32_t> fi(dp.F, 0);
  if (d->frame_intrinsics) std::copy(d->frame_intrinsics, d->frame_intrinsics + dp.F, fi.begin());
  if ((rc = dev_upload(h, &dfi, fi.data(), (size_t)dp.F))) return bail(rc);
  h->frame_intr = fi;
  dp.xy = dxy; dp.obs_frame = dof; dp.obs_point = dop; dp.frame_intr = dfi;
  h->obs_frame = std::move(of); h->obs_point = std::move(op);
  if ((rc = dev_upload(h, &dp.poses, d->poses, npose))) return bail(rc);
  if ((rc = dev_upload(h, &dp.points, d->points, (size_t)dp.M * 3))) return bail(rc);
  if ((rc = dev_upload(h, &dp.intr, d->intrinsics, (size_t)dp.NI * 9))) return bail(rc);

  // column scales: 0 at fixed coordinates, 1 elsewhere until the Jacobi scale is estimated
  h->mask_pose.assign(npose, 1.0); h->mask_point.assign((size_t)dp.M * 3, 1.0); h->mask_intr.assign((size_t)dp.NI * 9, 1.0);
  if (d->pose_fixed_mask) for (size_t b = 0; b < (size_t)dp.F * dp.P; ++b) for (int k = 0; k < 6; ++k)
    if (d->pose_fixed_mask[b] & (1u << k)) h->mask_pose[6 * b + k] = 0.0;
  if (d->point_constant) for (int j = 0; j < dp.M; ++j) if (d->point_constant[j]) for (int k = 0; k < 3; ++k) h->mask_point[3 * (size_t)j + k] = 0.0;
  if (d->intrinsics_constant) for (int c = 0; c < dp.NI; ++c) if (d->intrinsics_constant[c]) for (int k = 0; k < 9; ++k) h->mask_intr[9 * (size_t)c + k] = 0.0;
  if ((rc = dev_upload(h, &dp.scale_pose, h->mask_pose.data(), npose))) return bail(rc);
  if ((rc = dev_upload(h, &dp.scale_point, h->mask_point.data(), (size_t)dp.M * 3))) return bail(rc);
  if ((rc = dev_upload(h, &dp.scale_intr, h->mask_intr.data(), (size_t)dp.NI * 9))) return bail(rc);
  if ((rc = dev_upload(h, &h->d_mask_pose, h->mask_pose.data(), npose))) return bail(rc);
  if ((rc = dev_upload(h, &h->d_mask_point, h->mask_point.data(), (size_t)dp.M * 3))) return bail(rc);
  if ((rc = dev_upload(h, &h->d_mask_intr, h->mask_intr.data(), (size_t)dp.NI * 9))) return bail(rc);

  stage("parameters + masks");
  if ((rc = dev_alloc(h, &dp.res, 2 * (size_t)kEvalBlock * dp.ntiles))) return bail(rc);
  dp.jac = nullptr;   // [2 K x 256 x ntiles] what CostFunction::Evaluate materialises — 480 B per observation (4 GB at 4k cameras): allocated when a raw
                      // evaluation is first asked for (ensure_jacobians).  BA() never is: the LM iteration forms its blocks without writing the Jacobian.
  const int nb = std::max(eval_num_blocks(N), 1);
  if ((rc = dev_alloc(h, &dp.cost_partial, (size_t)nb))) return bail(rc);
  if ((rc = dev_alloc(h, &dp.fixed_partial, (size_t)nb))) return bail(rc);
  if ((rc = dev_alloc(h, &dp.fail_partial, (size_t)nb))) return bail(rc);
  if ((rc = dev_alloc(h, &dp.fail_count, 1))) return bail(rc);
  if ((rc = dev_alloc(h, &h->d_cost2, 2))) return bail(rc);
  if (hipMemset(dp.fail_count, 0, sizeof(int)) != hipSuccess) return bail(fail(RSBA_ERR_HIP, "hipMemset"));
  if (rsba::dev_event_acquire(&h->ev0, true) != hipSuccess || rsba::dev_event_acquire(&h->ev1, true) != hipSuccess) return bail(fail(RSBA_ERR_HIP, "hipEventCreate"));
  stage("outputs");
  if (dbg) std::fprintf(stderr, "[rsba create] (ms):%s\n", stages);
  *out = h;
  return RSBA_OK;
}

void rsba_destroy(rsba_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  const bool dbg = std::getenv("RSBA_DEBUG_PLAN") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = dbg ? now() : 0.0;
  rsba_destroy_solver(h);
  const double t1 = dbg ? now() : 0.0;
  if (h->stream) (void)hipStreamSynchronize(h->stream);   // nothing on the device touches the blocks any more: they go back to the cache (devmem.hpp)
  if (h->own_stream && h->own_stream != h->stream) (void)hipStreamSynchronize(h->own_stream);
  for (void* p : h->allocs) rsba::dev_free(p);
  const double t2 = dbg ? now() : 0.0;
  rsba::dev_event_release(h->ev0, true);      // (both streams are idle: synchronised above)
  rsba::dev_event_release(h->ev1, true);
  if (h->own_stream) { (void)hipStreamSynchronize(h->own_stream); rsba::dev_stream_release(h->own_stream); }
  const double t3 = dbg ? now() : 0.0;
  index_vector_give(std::move(h->obs_frame)); index_vector_give(std::move(h->obs_point));
  delete h;
  if (dbg) std::fprintf(stderr, "[rsba destroy] handle: plan %.2f ms; blocks to the cache %.2f ms; events + stream to the pool %.2f ms; host state %.2f ms\n", t1 - t0, t2 - t1, t3 - t2, now() - t3);
}

int32_t rsba_set_stream(rsba_handle* h, void* s) {
  if (!h) return fail(RSBA_ERR_INVALID_ARGUMENT, "null handle");
  h->stream = s ? static_cast<hipStream_t>(s) : h->own_stream;
  return RSBA_OK;
}

int32_t rsba_upload_parameters(rsba_handle* h, const double* poses, const double* points, const double* intr) {
  if (!h) return fail(RSBA_ERR_INVALID_ARGUMENT, "null handle");
  HIP_TRY(hipSetDevice(h->device));
  const DeviceProblem& dp = h->dp;
  if (poses) HIP_TRY(hipMemcpyAsync(dp.poses, poses, (size_t)dp.F * dp.P * 6 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (points) HIP_TRY(hipMemcpyAsync(dp.points, points, (size_t)dp.M * 3 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (intr) HIP_TRY(hipMemcpyAsync(dp.intr, intr, (size_t)dp.NI * 9 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (dp.pp_count > 0 && h->pp_host) HIP_TRY(hipMemcpyAsync(dp.pp_value, h->pp_host, 6 * (size_t)dp.pp_count * sizeof(double), hipMemcpyHostToDevice, h->stream));   // the priorPoses blocks are parameters too
  HIP_TRY(hipStreamSynchronize(h->stream));
  return RSBA_OK;
}

int32_t rsba_download_parameters(rsba_handle* h, double* poses, double* points, double* intr) {
  if (!h) return fail(RSBA_ERR_INVALID_ARGUMENT, "null handle");
  HIP_TRY(hipSetDevice(h->device));
  const DeviceProblem& dp = h->dp;
  if (poses) HIP_TRY(hipMemcpyAsync(poses, dp.poses, (size_t)dp.F * dp.P * 6 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (points) HIP_TRY(hipMemcpyAsync(points, dp.points, (size_t)dp.M * 3 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (intr) HIP_TRY(hipMemcpyAsync(intr, dp.intr, (size_t)dp.NI * 9 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return RSBA_OK;
}

static int32_t ensure_jacobians(rsba_handle* h) {
  if (h->dp.jac) return RSBA_OK;
  HIP_TRY(hipSetDevice(h->device));
  return dev_alloc(h, &h->dp.jac, 2 * (size_t)h->dp.K * kEvalBlock * h->dp.ntiles);
}

int32_t rsba_evaluate_device(rsba_handle* h, int32_t with_jacobians) {
  if (!h) return fail(RSBA_ERR_INVALID_ARGUMENT, "null handle");
  if (with_jacobians) { if (int32_t rc = ensure_jacobians(h)) return rc; }
  HIP_TRY(launch_eval(h->dp, with_jacobians ? kRawJacobian : kResidualOnly, h->stream));
  return RSBA_OK;
}

int32_t rsba_get_device_view(rsba_handle* h, rsba_device_view* v) {
  if (!h || !v) return fail(RSBA_ERR_INVALID_ARGUMENT, "null argument");
  std::memset(v, 0, sizeof *v);
  // (jacobians stays NULL until an evaluation WITH Jacobians has run — rsba_evaluate_device(h, 1), rsba_evaluate with a Jacobian output,
  // rsba_time_evaluate(…, 1): the buffer is 480 B per observation, 4 GB at 4k cameras, and BA() never needs it; a view must not be what allocates it)
  v->residuals = h->dp.res; v->jacobians = h->dp.jac; v->tile = kEvalBlock; v->jacobian_cols = h->dp.K;
  ensure_order(h);
  v->order_host = h->order.data(); v->poses = h->dp.poses; v->points = h->dp.points; v->intrinsics = h->dp.intr;
  return RSBA_OK;
}

int32_t rsba_time_evaluate(rsba_handle* h, int32_t with_jacobians, int32_t warmup, int32_t iters, double* avg_ms) {
  if (!h || !avg_ms || iters <= 0) return fail(RSBA_ERR_INVALID_ARGUMENT, "bad argument");
  HIP_TRY(hipSetDevice(h->device));
  const EvalMode mode = with_jacobians ? kRawJacobian : kResidualOnly;
  if (with_jacobians) { if (int32_t rc = ensure_jacobians(h)) return rc; }
  for (int i = 0; i < warmup; ++i) HIP_TRY(launch_eval(h->dp, mode, h->stream));
  HIP_TRY(hipEventRecord(h->ev0, h->stream));
  for (int i = 0; i < iters; ++i) HIP_TRY(launch_eval(h->dp, mode, h->stream));
  HIP_TRY(hipEventRecord(h->ev1, h->stream));
  HIP_TRY(hipEventSynchronize(h->ev1));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  *avg_ms = (double)ms / iters;
  return RSBA_OK;
}

int32_t rsba_set_motion_priors(rsba_handle* h, int32_t kind, double scale, double inter_frame_ratio, const int32_t* frames, int32_t count) {
  if (!h || kind < 0 || kind > 2 || count < 0 || (count > 0 && !frames)) return fail(RSBA_ERR_INVALID_ARGUMENT, "bad motion prior arguments");
  if (h->solver) return fail(RSBA_ERR_INVALID_ARGUMENT, "rsba_set_motion_priors must precede the first solve / gradient call");
  HIP_TRY(hipSetDevice(h->device));
  DeviceProblem& dp = h->dp;
  if (kind == 0 || count == 0) { dp.prior_of = nullptr; dp.prior_kind = 0; h->prior_frames.clear(); h->prior_invalid = 0; return RSBA_OK; }
  if (dp.P != 2) return fail(RSBA_ERR_INVALID_ARGUMENT, "motion priors need two poses per frame (CeresHandler.h:151)");
  std::vector<int32_t> flags((size_t)dp.F + 1, 0);
  for (int32_t k = 0; k < count; ++k) {
    if (frames[k] < 1 || frames[k] >= dp.F || (k > 0 && frames[k] <= frames[k - 1])) return fail(RSBA_ERR_INVALID_ARGUMENT, "prior frames must be strictly increasing in [1, F)");
    flags[frames[k]] = 1;
  }
  int32_t* d = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), flags.size() * sizeof(int32_t)));
  h->allocs.push_back(d);
  HIP_TRY(hipMemcpy(d, flags.data(), flags.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  double* part = nullptr; unsigned* ticket = nullptr;
  const size_t nwaves = ((size_t)dp.F + 63) / 64;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&part), 2 * nwaves * sizeof(double)));
  h->allocs.push_back(part);
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ticket), sizeof(unsigned)));
  h->allocs.push_back(ticket);
  HIP_TRY(hipMemset(ticket, 0, sizeof(unsigned)));
  dp.prior_partial = part; dp.prior_ticket = ticket;
  dp.prior_of = d; dp.prior_kind = kind; dp.prior_scale = scale; dp.prior_ratio = inter_frame_ratio;
  h->prior_frames.assign(frames, frames + count);
  h->prior_ratio_result = inter_frame_ratio;
  // RsConstVeloPrior returns ratio >= 0, RsConstAccelerationPrior ratio >= _EPS (video_bundler_rs_inter.h:92,:157)
  const bool valid = kind == 1 ? inter_frame_ratio >= 0.0 : inter_frame_ratio >= 2.220446049250313e-16;
  h->prior_invalid = valid ? 0 : count;
  return RSBA_OK;
}

int32_t rsba_set_pose_priors(rsba_handle* h, double rotation, double position, const int32_t* pose_blocks, double* prior_values, int32_t count,
                             int32_t spherical_pose_block) {
  if (!h || count < 0 || (count > 0 && (!pose_blocks || !prior_values))) return fail(RSBA_ERR_INVALID_ARGUMENT, "bad pose prior arguments");
  if (h->solver) return fail(RSBA_ERR_INVALID_ARGUMENT, "rsba_set_pose_priors must precede the first solve / gradient call");
  DeviceProblem& dp = h->dp;
  const int nblocks = dp.F * dp.P;
  if (spherical_pose_block >= nblocks) return fail(RSBA_ERR_INVALID_ARGUMENT, "spherical pose block out of range");
  std::vector<uint8_t> used((size_t)nblocks, 0);
  for (int32_t k = 0; k < count; ++k) {
    if (pose_blocks[k] < 0 || pose_blocks[k] >= nblocks || used[pose_blocks[k]]) return fail(RSBA_ERR_INVALID_ARGUMENT, "pose prior blocks must be distinct pose blocks f * P + q");
    used[pose_blocks[k]] = 1;
  }
  HIP_TRY(hipSetDevice(h->device));
  dp.pp_count = count; dp.pp_rotation = rotation; dp.pp_position = position; dp.pp_spherical = spherical_pose_block < 0 ? -1 : spherical_pose_block;
  h->pp_blocks.assign(pose_blocks, pose_blocks + count); h->pp_host = prior_values;
  if (count > 0) {
    int32_t* d_blocks = nullptr;
    int32_t rc = dev_upload(h, &d_blocks, pose_blocks, (size_t)count);
    if (rc) return rc;
    dp.pp_block = d_blocks;
    if ((rc = dev_upload(h, &dp.pp_value, prior_values, 6 * (size_t)count))) return rc;
    if ((rc = dev_upload(h, &dp.pp_trial, prior_values, 6 * (size_t)count))) return rc;
    std::vector<double> ones(6 * (size_t)count, 1.0);
    if ((rc = dev_upload(h, &dp.pp_scale, ones.data(), ones.size()))) return rc;
  }
  return RSBA_OK;
}

void rsba_release_host_scratch(void) { rsba_release_plan_scratch(); index_pool_release(); rsba::dev_release_cache(); }

int32_t rsba_set_global_shutter_frames(rsba_handle* h, const uint8_t* is_global) {
  if (!h) return fail(RSBA_ERR_INVALID_ARGUMENT, "null handle");
  if (h->solver) return fail(RSBA_ERR_INVALID_ARGUMENT, "rsba_set_global_shutter_frames must precede the first solve / gradient call");
  DeviceProblem& dp = h->dp;
  if (dp.P != 2) return fail(RSBA_ERR_INVALID_ARGUMENT, "one-pose frames are flagged inside problems with poses_per_frame = 2");
  HIP_TRY(hipSetDevice(h->device));
  // the masks the caller gave (pose_fixed_mask) with the second pose slot of every flagged frame held constant on top
  std::vector<double> mask(h->mask_pose_caller.empty() ? h->mask_pose : h->mask_pose_caller);
  if (h->mask_pose_caller.empty()) h->mask_pose_caller = h->mask_pose;
  if (!is_global) dp.frame_global = nullptr;
  else {
    uint8_t* d_flags = nullptr;
    int32_t rc = dev_upload(h, &d_flags, is_global, (size_t)dp.F);
    if (rc) return rc;
    dp.frame_global = d_flags;
    for (int f = 0; f < dp.F; ++f) if (is_global[f]) for (int k = 0; k < 6; ++k) mask[((size_t)f * 2 + 1) * 6 + k] = 0.0;
  }
  h->mask_pose = mask;
  HIP_TRY(hipMemcpy(h->d_mask_pose, mask.data(), mask.size() * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dp.scale_pose, mask.data(), mask.size() * sizeof(double), hipMemcpyHostToDevice));
  return RSBA_OK;
}

int32_t rsba_set_inter_frame_ratio_free(rsba_handle* h, int32_t is_free) {
  if (!h) return fail(RSBA_ERR_INVALID_ARGUMENT, "null handle");
  if (h->solver) return fail(RSBA_ERR_INVALID_ARGUMENT, "rsba_set_inter_frame_ratio_free must precede the first solve / gradient call");
  h->prior_free = is_free != 0;
  h->dp.prior_free = h->prior_free ? 1 : 0;
  return RSBA_OK;
}
int32_t rsba_get_inter_frame_ratio(rsba_handle* h, double* ratio) {
  if (!h || !ratio) return fail(RSBA_ERR_INVALID_ARGUMENT, "null argument");
  *ratio = h->dp.prior_of ? h->dp.prior_ratio : 0.0;
  return RSBA_OK;
}

int32_t rsba_evaluate(rsba_handle* h, double* cost, double* residuals, double* jacobians, double* gradient, int64_t* num_failed) {
  if (!h) return fail(RSBA_ERR_INVALID_ARGUMENT, "null handle");
  HIP_TRY(hipSetDevice(h->device));
  DeviceProblem& dp = h->dp;
  const int64_t N = dp.N; const int K = dp.K;
  if (jacobians) { if (int32_t rc = ensure_jacobians(h)) return rc; }
  HIP_TRY(hipMemsetAsync(dp.fail_count, 0, sizeof(int), h->stream));
  HIP_TRY(launch_eval(dp, jacobians ? kRawJacobian : kResidualOnly, h->stream));
  HIP_TRY(launch_cost_reduce(dp, h->d_cost2, h->stream));
  if (dp.prior_of && (h->rank == 0 || h->prior_split)) HIP_TRY(launch_prior_cost(dp, h->d_cost2, h->prior_invalid, h->stream));
  if (h->rank == 0) HIP_TRY(launch_pose_prior_cost(dp, h->d_cost2, h->stream));
  double c2[2] = {0, 0}; int nfail = 0;
  HIP_TRY(hipMemcpyAsync(c2, h->d_cost2, sizeof c2, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipMemcpyAsync(&nfail, dp.fail_count, sizeof nfail, hipMemcpyDeviceToHost, h->stream));
  if ((residuals || jacobians) && N > 0) {
    if (!h->d_order) {
      HIP_TRY(rsba::dev_malloc(reinterpret_cast<void**>(&h->d_order), (size_t)N * sizeof(int64_t)));
      h->allocs.push_back(h->d_order);
      ensure_order(h);
      HIP_TRY(hipMemcpyAsync(h->d_order, h->order.data(), (size_t)N * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
      HIP_TRY(rsba::dev_malloc(reinterpret_cast<void**>(&h->d_rows), (size_t)N * (2 + 2 * (size_t)K) * sizeof(double)));
      h->allocs.push_back(h->d_rows);
    }
    double* d_res = h->d_rows; double* d_jac = h->d_rows + 2 * (size_t)N;
    HIP_TRY(launch_untile(dp, h->d_order, jacobians != nullptr, d_res, d_jac, h->stream));
    if (residuals) HIP_TRY(hipMemcpyAsync(residuals, d_res, 2 * (size_t)N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (jacobians) HIP_TRY(hipMemcpyAsync(jacobians, d_jac, 2 * (size_t)K * N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (cost) *cost = c2[0] + c2[1];
  if (num_failed) *num_failed = nfail;
  if (gradient) {
    int32_t rc = rsba_gradient(h, gradient);
    if (rc) return rc;
  }
  if (nfail) return fail(RSBA_ERR_EVALUATION_FAILED, std::to_string(nfail) + " residual blocks failed to evaluate");
  return RSBA_OK;
}

int32_t rsba_validate_observations(rsba_handle* h, double sq_threshold, double min_distance, uint8_t* valid) {
  if (!h || !valid) return fail(RSBA_ERR_INVALID_ARGUMENT, "null argument");
  HIP_TRY(hipSetDevice(h->device));
  const int64_t N = h->dp.N;
  if (N == 0) return RSBA_OK;
  int32_t rc = grow(h, &h->d_flags, &h->flags_cap, N);
  if (rc) return rc;
  HIP_TRY(launch_validate(h->dp, sq_threshold, min_distance, h->d_flags, h->stream));
  const uint8_t* src = h->d_flags;
  if (!h->identity_order) {   // back to the caller's observation order on the device
    if (!h->d_order) {
      if ((rc = dev_alloc(h, &h->d_order, (size_t)N))) return rc;
      HIP_TRY(hipMemcpyAsync(h->d_order, h->order.data(), (size_t)N * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    }
    if ((rc = grow(h, &h->d_flags_out, &h->flags_out_cap, N))) return rc;
    HIP_TRY(launch_scatter_flags(h->d_flags, h->d_order, N, h->d_flags_out, h->stream));
    src = h->d_flags_out;
  }
  HIP_TRY(hipMemcpyAsync(valid, src, (size_t)N, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return RSBA_OK;
}

int32_t rsba_reproject(rsba_handle* h, const int32_t* frames, const int32_t* points, int64_t n, double* xy_out, uint8_t* ok_out) {
  if (!h || (n > 0 && (!frames || !points || !xy_out || !ok_out))) return fail(RSBA_ERR_INVALID_ARGUMENT, "null argument");
  for (int64_t i = 0; i < n; ++i)
    if (frames[i] < 0 || frames[i] >= h->dp.F || points[i] < 0 || points[i] >= h->dp.M) return fail(RSBA_ERR_INVALID_ARGUMENT, "index out of range");
  if (n <= 0) return RSBA_OK;
  HIP_TRY(hipSetDevice(h->device));
  int32_t rc = grow(h, &h->d_pairs, &h->pairs_cap, 2 * n);
  if (rc) return rc;
  if ((rc = grow(h, &h->d_pair_xy, &h->pair_xy_cap, 2 * n))) return rc;
  if ((rc = grow(h, &h->d_flags_out, &h->flags_out_cap, n))) return rc;
  uint8_t* dok = h->d_flags_out;
  HIP_TRY(hipMemcpyAsync(h->d_pairs, frames, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->d_pairs + n, points, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(launch_reproject(h->dp, h->d_pairs, h->d_pairs + n, n, h->d_pair_xy, dok, h->stream));
  HIP_TRY(hipMemcpyAsync(xy_out, h->d_pair_xy, (size_t)n * 16, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipMemcpyAsync(ok_out, dok, (size_t)n, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return RSBA_OK;
}

void rsba_default_solver_options(rsba_solver_options* o) {
  if (!o) return;
  // Ceres 1.9 Solver::Options defaults; iteration cap as CeresHandler.h:405
  o->max_num_iterations = 50; o->jacobi_scaling = 1; o->max_num_consecutive_invalid_steps = 5; o->minimizer_progress_to_stdout = 0;
  o->initial_trust_region_radius = 1e4; o->max_trust_region_radius = 1e16; o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3; o->min_lm_diagonal = 1e-6; o->max_lm_diagonal = 1e32;
  o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10; o->parameter_tolerance = 1e-8;
  o->level_scheduled_cholesky = 0; o->profile_phases = 0;
}

}  // extern "C"
