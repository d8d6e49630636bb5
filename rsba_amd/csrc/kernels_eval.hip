// K1  rs_residual_jacobian — the metric's "residual + Jacobian evaluation" (SURVEY §2.1 K1/K3/K8).
//
// One lane per observation over the frame-major observation list.  A 256-lane workgroup touches
// 1-2 camera blocks, so their 6P pose doubles (and column scales) are staged once through LDS;
// observation records (24 B) are coalesced reads, points are a 24-B gather that hits L2 (the point
// array of the 1k-camera scene is 2.4 MB), and every output component is one coalesced 512-B store
// per wave into the component-major res/jac arrays.  HBM-bound: 280 B/observation, ~0.6 kflop.
#include "device_state.hpp"
#include "lm_record.hpp"
#include "obs_math.hpp"

namespace rsba {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// value held by the neighbouring lane (lane ^ 1), via DPP quad_perm [1,0,3,2]
__device__ __forceinline__ double swap_pair(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xB1, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xB1, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

template <bool CAL, int P, int MODE>
__global__ __launch_bounds__(kEvalBlock) void eval_kernel(const DeviceProblem dp) {
  if (MODE == kLmJacobian && lm_stopped(dp.ctl)) return;   // (device-side trust region: the solve is over, iterations enqueued ahead fall through)
  constexpr int CD = 6 * P;
  constexpr int K = ObsOut<CAL, P>::K;
  constexpr int OFF_POSE = CAL ? 0 : 9;
  constexpr int OFF_PT = OFF_POSE + CD;
  __shared__ double s_pose[kStageFrames * CD];
  __shared__ double s_scale[MODE == kLmJacobian ? kStageFrames * CD : 1];
  __shared__ double s_red[3][kEvalBlock / 64];
  constexpr bool CAMBLK = (MODE == kLmJacobian);          // camera (and intrinsics) blocks formed here (dp.cam_part)
  constexpr int NCOL = (K - 3) + 1;                        // [Ji (9, uncalibrated only) | Jc (CD) | r]
  constexpr int NCB = (NCOL + 15) / 16;                    // 16-column blocks of it: 1, or 2 for rolling shutter + intrinsics
  constexpr int TP = 16 * NCB + 1;                         // pitch of the operand transposition buffer
  constexpr int RECW = 2 + 2 * K, RPITCH = RECW | 1;       // point-major record, and its pitch in the staging buffer
  constexpr int TRW = (MODE == kLmJacobian) ? ((64 * TP > 32 * RPITCH) ? 64 * TP : 32 * RPITCH) : 1;   // doubles per wave
  __shared__ double s_tr[(MODE == kLmJacobian) ? (kEvalBlock / 64) * TRW : 1];

  const int tid = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * kEvalBlock;
  const int64_t i = base + tid;
  const int64_t last = (base + kEvalBlock - 1 < dp.N) ? base + kEvalBlock - 1 : dp.N - 1;
  const int f_lo = dp.obs_frame[base];
  const int f_hi = dp.obs_frame[last];
  const bool staged = (f_hi - f_lo) < kStageFrames;
  if (staged) {
    const int nvals = (f_hi - f_lo + 1) * CD;
    for (int k = tid; k < nvals; k += kEvalBlock) {
      s_pose[k] = dp.poses[(size_t)f_lo * CD + k];
      if (MODE == kLmJacobian) s_scale[k] = dp.scale_pose[(size_t)f_lo * CD + k];
    }
    __syncthreads();
  }

  double cost = 0.0, fixed = 0.0, nfail = 0.0;   // nfail: this lane's functor returned false (counted like the cost: no atomics)
  {
    // lanes past the end of the list recompute the last observation (they must stay converged for the
    // lane-pair exchange below); their results land in the tile padding and are not counted
    const bool valid = i < dp.N;
    const int64_t ic = valid ? i : dp.N - 1;
    const double2 xy = dp.xy[ic];
    const int f = dp.obs_frame[ic];
    const int j = dp.obs_point[ic];
    double pose[CD];
    if (staged) {
#pragma unroll
      for (int k = 0; k < CD; ++k) pose[k] = s_pose[(f - f_lo) * CD + k];
    } else {
#pragma unroll
      for (int k = 0; k < CD; ++k) pose[k] = dp.poses[(size_t)f * CD + k];
    }
    ObsOut<CAL, P> o;
    double half_rho = 0.0;
    if (MODE == kLmJacobian) {
      double psc[CD];
      if (staged) {
#pragma unroll
        for (int k = 0; k < CD; ++k) psc[k] = s_scale[(f - f_lo) * CD + k];
      } else {
#pragma unroll
        for (int k = 0; k < CD; ++k) psc[k] = dp.scale_pose[(size_t)f * CD + k];
      }
      bool dropped;
      lm_observation<CAL, P>(dp, f, j, xy.x, xy.y, pose, psc, o, half_rho, dropped);   // loss-corrected, masked, scaled (lm_record.hpp)
      if (valid && !o.ok) nfail = 1.0;
      if (!valid) half_rho = 0.0;
      if (dropped) { fixed = half_rho; half_rho = 0.0; }
    } else {
      double X[3], cam[9];
#pragma unroll
      for (int k = 0; k < 3; ++k) X[k] = dp.points[(size_t)j * 3 + k];
      const int ci = (dp.NI == 1) ? 0 : dp.frame_intr[f];
#pragma unroll
      for (int k = 0; k < 9; ++k) cam[k] = dp.intr[(size_t)ci * 9 + k];
      const Model m = {(dp.frame_global && dp.frame_global[f]) ? (int)kGlobal : dp.shutter, dp.scan0, dp.scan1, dp.interp_rotation};
      eval_observation<CAL, P, MODE != kResidualOnly>(m, cam, pose, X, xy.x, xy.y, o);
      if (valid && !o.ok) nfail = 1.0;
      // Ceres 1.9 ResidualBlock::Evaluate: cost = rho0/2 from the uncorrected residual
      const double s = o.r[0] * o.r[0] + o.r[1] * o.r[1];
      double rho[3] = {s, 1.0, 0.0};
      if (dp.huber_a > 0.0) huber_rho(dp.huber_a, s, rho);
      half_rho = (o.ok && valid) ? 0.5 * rho[0] : 0.0;
    }

    if (MODE == kLmJacobian && dp.rec) {   // point-major copy of the corrected record (device_state.hpp); calibrated problems recompute it instead (dp.rec == nullptr)
      constexpr int REC = 2 + 2 * K;
      constexpr int KC = K - 3;
      // record layout [r0 r1 | Jp row0 (3) Jp row1 (3) | Jc row0 (KC) Jc row1 (KC)]: entry k of the record
      auto rec_entry = [&](int k) -> double {
        if (k < 2) return o.r[k];
        if (k < 8) return o.J[(k - 2) / 3][OFF_PT + (k - 2) % 3];
        return o.J[(k - 8) / KC][(k - 8) % KC];
      };
      // The records are scattered by slot.  Written one lane per record, every 16-B store of a wave hits 64
      // different cache lines, an eighth of a line at a time; staged through LDS (half a wave at a time), each
      // record leaves as one contiguous run written by REC/2 neighbouring lanes.
      {
        const int lane = tid & 63, wv = tid >> 6;
        double* st = s_tr + wv * TRW;
        double* const recw = lm_records(dp, dp.rec_candidate != 0);
        const int my_slot = valid ? dp.obs_slot[ic] : -1;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if ((lane >> 5) == h) {
#pragma unroll
            for (int k = 0; k < REC; ++k) st[(lane & 31) * RPITCH + k] = rec_entry(k);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
          for (int q = 0; q < (16 * REC + 63) / 64; ++q) {
            const int idx = lane + 64 * q;                       // double2 index within the half's 32 records
            const int rcd = idx / (REC / 2), part = idx % (REC / 2);
            const int slot = __shfl(my_slot, 32 * h + (rcd < 32 ? rcd : 31), 64);
            if (idx < 16 * REC && slot >= 0)
              *reinterpret_cast<double2*>(recw + (size_t)slot * REC + 2 * part) = make_double2(st[rcd * RPITCH + 2 * part], st[rcd * RPITCH + 2 * part + 1]);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
      }
    }
    cost = half_rho;

    if (CAMBLK && dp.cam_part) {
      // K2a in place: G = [Ji | Jc | r]^T [Ji | Jc | r] over the observations of ONE frame holds U_f = Jc^T Jc, g_f = Jc^T r
      // and, with intrinsics as a parameter block, the border blocks Ji^T Jc, Ji^T Ji, Ji^T r of the same frame.
      // MFMA wants lane (i = lane & 15, g = lane >> 4) to hold column i of Jacobian row k = 4 step + g, while the rows
      // live one observation per lane: they change lanes through LDS, half a wave (64 rows) at a time.  A wave
      // that straddles frames does one pass per frame, rows of the other frames masked to zero.  One 16 x 16 product per four rows
      // for up to 16 columns; TWO for the 22 columns of rolling shutter + intrinsics (device_state.hpp: cam_part_blocks — the two operands
      // of a product select different columns, so that two blocks hold everything on and below the diagonal of G; three products
      // (0,0) (1,0) (1,1) until round 6: 47 % of their work multiplied padding).
      typedef double dbl4 __attribute__((ext_vector_type(4)));
      constexpr int NBLK = cam_part_blocks(NCOL);
      constexpr bool TWO = NCB == 2;
      static_assert(NCB <= 2, "operands of up to 24 columns");
      const int lane = tid & 63, wv = tid >> 6, ci = lane & 15, cg = lane >> 4;
      // (two-product form) the columns this lane supplies: block 0 = columns 0 .. 15 against columns 0 .. 7, 16 .. ; block 1 = columns 8 .. against themselves; NCOL = a zero column
      const int col_b0 = ci < 8 ? ci : (ci + 8 < NCOL ? ci + 8 : NCOL), col_a1 = ci + 8 < NCOL ? ci + 8 : NCOL;
      double* tr = s_tr + wv * TRW;
      const int my_frame = valid ? f : -1;
      const int64_t wave_id = (base >> 6) + wv;
      int seg = dp.wave_seg_base[wave_id];
      int done = 0;
      if (NCOL < 16 * NCB) {   // the padding columns of the operand rows are zero and stay zero: written once per wave, not with every half of every frame's pass (20 of a row's 32 cells with rolling shutter + intrinsics)
        double* row = tr + 2 * (lane & 31) * TP;
#pragma unroll
        for (int c = NCOL; c < 16 * NCB; ++c) { row[c] = 0.0; row[TP + c] = 0.0; }
      }
      while (done < 64) {
        const int fr = __builtin_amdgcn_readlane(my_frame, done);
        if (fr < 0) break;
        const unsigned long long in_seg = __ballot(my_frame == fr);
        const int seg_end = done + __builtin_popcountll(in_seg);     // observations of a frame are consecutive
        dbl4 d[NBLK];
#pragma unroll
        for (int q = 0; q < NBLK; ++q) d[q] = dbl4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if ((lane >> 5) == h) {
            const bool keep = o.ok && valid && my_frame == fr;
            double* row = tr + 2 * (lane & 31) * TP;
#pragma unroll
            for (int c = 0; c < K - 3; ++c) { row[c] = keep ? o.J[0][c] : 0.0; row[TP + c] = keep ? o.J[1][c] : 0.0; }
            row[K - 3] = keep ? o.r[0] : 0.0; row[TP + K - 3] = keep ? o.r[1] : 0.0;
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          // rows of this half that belong to the segment: observations [max(done, 32h), min(seg_end, 32h + 32))
          const int o_lo = done > 32 * h ? done : 32 * h, o_hi = seg_end < 32 * h + 32 ? seg_end : 32 * h + 32;
          if (o_lo < o_hi) {
            const int s_lo = (2 * (o_lo - 32 * h)) >> 2, s_hi = (2 * (o_hi - 32 * h) + 3) >> 2;
            for (int step = s_lo; step < s_hi; ++step) {
              const double* xr = tr + (4 * step + cg) * TP;
              if constexpr (TWO) {
                const double x0 = xr[ci], xb = xr[col_b0], x1 = xr[col_a1];
                d[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, xb, d[0], 0, 0, 0);
                d[NBLK - 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x1, d[NBLK - 1], 0, 0, 0);
              } else {
                const double x0 = xr[ci];
                d[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x0, d[0], 0, 0, 0);
              }
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        double* part = dp.cam_part + (size_t)seg * (NBLK * 256);
#pragma unroll
        for (int q = 0; q < NBLK; ++q)
#pragma unroll
          for (int v = 0; v < 4; ++v) part[q * 256 + (cg + 4 * v) * 16 + ci] = d[q][v];
        ++seg;
        done = seg_end;
      }
    }

    // Tiled component-major stores, 16 B per lane: lanes 2m / 2m+1 exchange one value per component
    // pair (DPP quad_perm [1,0,3,2]); the even lane then stores component c of observations (2m, 2m+1)
    // and the odd lane component c+1 of the same two — 16-B stores sustain ~6 % more of the HBM write
    // stream than 8-B ones on this part (tools/hbm_calib.hip: tiled_x2 vs write_tiled).
    if (!(CAMBLK && dp.cam_part)) {
    const bool odd = tid & 1;
    double* rt = dp.res + (size_t)blockIdx.x * (2 * kEvalBlock) + (tid & ~1) + (odd ? kEvalBlock : 0);
    {
      const double r0 = o.ok ? o.r[0] : 0.0, r1 = o.ok ? o.r[1] : 0.0;
      const double got = swap_pair(odd ? r0 : r1);
      *reinterpret_cast<double2*>(rt) = odd ? make_double2(got, r1) : make_double2(r0, got);
    }
    if (MODE != kResidualOnly) {
      double* jt = dp.jac + (size_t)blockIdx.x * (2 * K * kEvalBlock) + (tid & ~1) + (odd ? kEvalBlock : 0);
#pragma unroll
      for (int c = 0; c < 2 * K; c += 2) {
        const double v0 = o.ok ? o.J[c / K][c % K] : 0.0, v1 = o.ok ? o.J[(c + 1) / K][(c + 1) % K] : 0.0;
        const double got = swap_pair(odd ? v0 : v1);
        *reinterpret_cast<double2*>(jt + (size_t)c * kEvalBlock) = odd ? make_double2(got, v1) : make_double2(v0, got);
      }
    }
    }
  }

  // deterministic cost: fixed-order wave and workgroup sums, one partial per workgroup
  cost = wave_sum(cost);
  nfail = wave_sum(nfail);
  if (MODE == kLmJacobian) fixed = wave_sum(fixed);
  if ((tid & 63) == 0) { s_red[0][tid >> 6] = cost; s_red[1][tid >> 6] = fixed; s_red[2][tid >> 6] = nfail; }
  __syncthreads();
  if (tid == 0) {
    double c = 0.0, fx = 0.0, nf = 0.0;
#pragma unroll
    for (int w = 0; w < kEvalBlock / 64; ++w) { c += s_red[0][w]; fx += s_red[1][w]; nf += s_red[2][w]; }
    dp.cost_partial[blockIdx.x] = c;
    dp.fixed_partial[blockIdx.x] = (MODE == kLmJacobian) ? fx : 0.0;
    dp.fail_partial[blockIdx.x] = nf;
  }
}

// Fixed-order reduction of the per-workgroup partials: out[0] = cost, out[1] = fixed cost.
__global__ __launch_bounds__(256) void reduce_cost_kernel(const double* __restrict__ cost_partial, const double* __restrict__ fixed_partial,
                                                          const double* __restrict__ fail_partial, int n, double* out, int* fail_count) {
  __shared__ double s_red[3][4];
  double c = 0.0, f = 0.0, nf = 0.0;
  // (sixteen strides' loads issued together, added in stride order: the sums of the one-load-at-a-time loop, bit for bit — 4k cameras
  // have 40 000 partials, 157 dependent round trips for this one workgroup: 66 us)
  int k = threadIdx.x;
  for (; k + 15 * 256 < n; k += 16 * 256) {
    double vc[16], vf[16], vn[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { vc[u] = cost_partial[k + 256 * u]; vf[u] = fixed_partial[k + 256 * u]; vn[u] = fail_partial[k + 256 * u]; }
#pragma unroll
    for (int u = 0; u < 16; ++u) { c += vc[u]; f += vf[u]; nf += vn[u]; }
  }
  for (; k < n; k += 256) { c += cost_partial[k]; f += fixed_partial[k]; nf += fail_partial[k]; }
  c = wave_sum(c); f = wave_sum(f); nf = wave_sum(nf);
  if ((threadIdx.x & 63) == 0) { s_red[0][threadIdx.x >> 6] = c; s_red[1][threadIdx.x >> 6] = f; s_red[2][threadIdx.x >> 6] = nf; }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
    out[1] = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
    *fail_count = (int)(s_red[2][0] + s_red[2][1] + s_red[2][2] + s_red[2][3]);
  }
}

// Host-returning evaluations (rsba_evaluate): convert the device layout to the caller's on the device, so that the
// results leave in one contiguous D2H copy each instead of being re-ordered element by element on the host.
__global__ __launch_bounds__(256) void untile_kernel(const DeviceProblem dp, const int64_t* __restrict__ order, int with_jac,
                                                     double* __restrict__ res_rows, double* __restrict__ jac_rows) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= dp.N) return;
  const int64_t u = order[i];
  const int K = dp.K;
  const double* rt = dp.res + (size_t)(i >> 8) * (2 * kEvalBlock) + (i & 255);
  res_rows[2 * u] = rt[0]; res_rows[2 * u + 1] = rt[kEvalBlock];
  if (with_jac) {
    const double* jt = dp.jac + (size_t)(i >> 8) * (2 * (size_t)K * kEvalBlock) + (i & 255);
    double* dst = jac_rows + (size_t)u * 2 * K;
    for (int c = 0; c < 2 * K; ++c) dst[c] = jt[(size_t)c * kEvalBlock];
  }
}

hipError_t launch_untile(const DeviceProblem& dp, const int64_t* order, bool with_jacobians, double* res_rows, double* jac_rows, hipStream_t st) {
  if (dp.N <= 0) return hipSuccess;
  hipLaunchKernelGGL(untile_kernel, dim3((unsigned)((dp.N + 255) / 256)), dim3(256), 0, st, dp, order, with_jacobians ? 1 : 0, res_rows, jac_rows);
  return hipGetLastError();
}

hipError_t launch_cost_reduce(const DeviceProblem& dp, double* out2, hipStream_t st) {
  hipLaunchKernelGGL(reduce_cost_kernel, dim3(1), dim3(256), 0, st, dp.cost_partial, dp.fixed_partial, dp.fail_partial, eval_num_blocks(dp.N), out2, dp.fail_count);
  return hipGetLastError();
}

int eval_num_blocks(int64_t n) { return (int)((n + kEvalBlock - 1) / kEvalBlock); }

template <bool CAL, int P>
static hipError_t launch_mode(const DeviceProblem& dp, EvalMode mode, hipStream_t st) {
  const dim3 grid(eval_num_blocks(dp.N)), block(kEvalBlock);
  switch (mode) {
    case kResidualOnly: hipLaunchKernelGGL((eval_kernel<CAL, P, kResidualOnly>), grid, block, 0, st, dp); break;
    case kRawJacobian: hipLaunchKernelGGL((eval_kernel<CAL, P, kRawJacobian>), grid, block, 0, st, dp); break;
    case kLmJacobian: hipLaunchKernelGGL((eval_kernel<CAL, P, kLmJacobian>), grid, block, 0, st, dp); break;
  }
  return hipGetLastError();
}

hipError_t launch_eval(const DeviceProblem& dp, EvalMode mode, hipStream_t st) {
  if (dp.N <= 0) return hipSuccess;
  if (dp.calibrated) return dp.P == 2 ? launch_mode<true, 2>(dp, mode, st) : launch_mode<true, 1>(dp, mode, st);
  return dp.P == 2 ? launch_mode<false, 2>(dp, mode, st) : launch_mode<false, 1>(dp, mode, st);
}

}  // namespace rsba
