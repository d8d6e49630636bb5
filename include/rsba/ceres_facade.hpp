// Header-only C++ facade: the subset of the Ceres-Solver API that rsba programs against
// (SURVEY.md §8b lists every call site), mapped onto the C ABI in include/rsba_amd.h.
//
//   namespace ceres = rsba_amd::ceres;      // a CeresHandler-shaped program then compiles unchanged
//
// What is mapped (reference call sites, paths relative to /root/reference/src/rsba/):
//   ceres::CostFunction / AddResidualBlock / SetParameterBlockConstant / SetParameterization +
//   SubsetParameterization / HuberLoss / Problem::Evaluate / Solver::Options / Solve / Summary
//     — CeresHandler.h:78-90, 208-301, 335-382, 386-387, 394-426; VideoSfMHandler.cc:579-596, 627-630.
//
// Accepted cost functions: rsba's two hot-path functors — the typed cost objects of reprojection_costs.hpp
// (RsBundleAdjustment / ReprojectionError factories) — and, between the frames those observe, the motion priors of
// motion_priors.hpp (RsConstVeloPrior / RsConstAccelerationPrior; the interFrameRatio block constant or, as in the
// reference's default, free with CeresHandler's lower bound).  There is NO host-side evaluation: every residual, Jacobian and solve goes through librsba_amd's HIP
// kernels; so do the per-pose priors of pose_priors.hpp (GoodPosePrior, SphericalPrior).  A cost function of any other type
// is rejected with an error, not evaluated on the CPU.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../rsba_amd.h"
#include "session.hpp"

namespace rsba_amd {
namespace ceres {

enum LinearSolverType { DENSE_NORMAL_CHOLESKY, DENSE_QR, SPARSE_NORMAL_CHOLESKY, DENSE_SCHUR, SPARSE_SCHUR, ITERATIVE_SCHUR, CGNR };
enum TerminationType { CONVERGENCE, NO_CONVERGENCE, FAILURE, USER_SUCCESS, USER_FAILURE };
enum Ownership { DO_NOT_TAKE_OWNERSHIP, TAKE_OWNERSHIP };

// ceres::LossFunction / ceres::HuberLoss (CeresHandler.h:80,88)
class LossFunction {
 public:
  virtual ~LossFunction() {}
  virtual void Evaluate(double sq_norm, double out[3]) const = 0;
};
class HuberLoss : public LossFunction {
 public:
  explicit HuberLoss(double a) : a_(a), b_(a * a) {}
  // rho(s) = s for s <= a^2, 2 a sqrt(s) - a^2 beyond (closed form; the solve applies it on the device)
  void Evaluate(double s, double rho[3]) const override {
    if (s > b_) { const double r = std::sqrt(s); rho[0] = 2.0 * a_ * r - b_; rho[1] = std::max(std::numeric_limits<double>::min(), a_ / r); rho[2] = -rho[1] / (2.0 * s); }
    else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
  }
  double a() const { return a_; }
 private:
  double a_, b_;
};

// ceres::LocalParameterization / ceres::SubsetParameterization (CeresHandler.h:355,367,378)
class LocalParameterization {
 public:
  virtual ~LocalParameterization() {}
  virtual int GlobalSize() const = 0;
  virtual int LocalSize() const = 0;
};
class SubsetParameterization : public LocalParameterization {
 public:
  SubsetParameterization(int size, const std::vector<int>& constant_parameters) : size_(size), mask_(0) {
    for (int c : constant_parameters) if (c >= 0 && c < 32) mask_ |= (1u << c);
    local_ = size_;
    for (int c = 0; c < size_ && c < 32; ++c) if (mask_ & (1u << c)) --local_;
  }
  int GlobalSize() const override { return size_; }
  int LocalSize() const override { return local_; }
  unsigned constancy_mask() const { return mask_; }
 private:
  int size_, local_;
  unsigned mask_;
};

// ceres::CostFunction (virtual Evaluate, row-major num_residuals x block_size Jacobians, false = failed)
class CostFunction {
 public:
  virtual ~CostFunction() {}
  virtual bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const = 0;
  const std::vector<int32_t>& parameter_block_sizes() const { return sizes_; }
  int num_residuals() const { return num_residuals_; }
 protected:
  std::vector<int32_t> sizes_;
  int num_residuals_ = 0;
};

// The one family of cost functions this library accelerates.  Created by the factories in
// reprojection_costs.hpp, which keep the reference's names and argument lists.
class ReprojectionCost : public CostFunction {
 public:
  // blocks: [cam 9]? pose0 [pose1]? point
  ReprojectionCost(bool rolling, bool with_cam, const double observed[2], const double* fixed_cam,
                   const Session* sess, const SfmOptions* opt)
      : rolling_(rolling), with_cam_(with_cam), sess_(sess), opt_(opt) {
    obs_[0] = observed[0]; obs_[1] = observed[1];
    if (fixed_cam) std::memcpy(cam_, fixed_cam, sizeof cam_); else std::memset(cam_, 0, sizeof cam_);
    num_residuals_ = 2;
    if (with_cam) sizes_.push_back(NUM_CAM_PARAMS);
    sizes_.push_back(NUM_POSE_PARAMS);
    if (rolling) sizes_.push_back(NUM_POSE_PARAMS);
    sizes_.push_back(NUM_POINT_PARAMS);
  }
  bool rolling() const { return rolling_; }
  bool with_cam() const { return with_cam_; }
  const double* observed() const { return obs_; }
  const double* fixed_cam() const { return cam_; }
  // the RS functor reads session / option fields through references at evaluation time (VideoSfmBaRs.h:82-83)
  int shutter() const { return rolling_ && sess_ ? sess_->rs : GLOBAL; }
  void scanlines(int32_t out[2]) const { out[0] = (rolling_ && sess_ && sess_->scanlines.size() >= 2) ? sess_->scanlines[0] : 0; out[1] = (rolling_ && sess_ && sess_->scanlines.size() >= 2) ? sess_->scanlines[1] : 1; }
  bool interpolate_rotation() const { return opt_ ? opt_->model.interpolateRotation : true; }

  ~ReprojectionCost() override { if (h_) rsba_destroy(h_); }
  ReprojectionCost(const ReprojectionCost&) = delete;
  ReprojectionCost& operator=(const ReprojectionCost&) = delete;

  // CostFunction::Evaluate for one block: a one-observation problem through the same HIP kernel.  The device problem is
  // built on the first call and kept: later calls only upload the parameter blocks they are given (it is rebuilt if the
  // session / option fields the functor reads at evaluation time have changed).
  bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const override {
    int b = 0;
    double cam[NUM_CAM_PARAMS], poses[2 * NUM_POSE_PARAMS], point[NUM_POINT_PARAMS];
    std::memcpy(cam, with_cam_ ? parameters[b++] : cam_, sizeof cam);
    std::memcpy(poses, parameters[b++], NUM_POSE_PARAMS * sizeof(double));
    if (rolling_) std::memcpy(poses + NUM_POSE_PARAMS, parameters[b++], NUM_POSE_PARAMS * sizeof(double));
    std::memcpy(point, parameters[b++], sizeof point);
    int32_t key[4]; key[0] = shutter(); scanlines(key + 1); key[3] = interpolate_rotation();
    std::lock_guard<std::mutex> lock(mu_);
    if (h_ && std::memcmp(key, key_, sizeof key) != 0) { rsba_destroy(h_); h_ = nullptr; }
    if (!h_) {
      rsba_problem_desc d; std::memset(&d, 0, sizeof d);
      const int32_t zero = 0;
      d.shutter = key[0]; d.scanlines[0] = key[1]; d.scanlines[1] = key[2]; d.interpolate_rotation = key[3];
      d.calibrated = !with_cam_; d.poses_per_frame = rolling_ ? 2 : 1;
      d.num_frames = d.num_points = d.num_intrinsics = 1; d.num_observations = 1;
      d.poses = poses; d.points = point; d.intrinsics = cam; d.obs_xy = obs_; d.obs_frame = &zero; d.obs_point = &zero;
      if (rsba_create(&d, 0, &h_) != RSBA_OK) { h_ = nullptr; return false; }
      std::memcpy(key_, key, sizeof key);
    } else if (rsba_upload_parameters(h_, poses, point, cam) != RSBA_OK) return false;
    const int K = (with_cam_ ? 9 : 0) + (rolling_ ? 12 : 6) + 3;
    std::vector<double> J(2 * (size_t)K);
    int64_t nfail = 0; double cost = 0;
    const int32_t st = rsba_evaluate(h_, &cost, residuals, jacobians ? J.data() : nullptr, nullptr, &nfail);
    if (st != RSBA_OK) return false;
    if (jacobians) {
      int col = 0;
      for (size_t blk = 0; blk < sizes_.size(); ++blk) {
        const int n = sizes_[blk];
        if (jacobians[blk]) for (int r = 0; r < 2; ++r) for (int c = 0; c < n; ++c) jacobians[blk][r * n + c] = J[(size_t)r * K + col + c];
        col += n;
      }
    }
    return true;
  }

 private:
  bool rolling_, with_cam_;
  double obs_[2], cam_[NUM_CAM_PARAMS];
  const Session* sess_;
  const SfmOptions* opt_;
  mutable rsba_handle* h_ = nullptr;   // the one-observation device problem behind Evaluate()
  mutable int32_t key_[4] = {0, 0, 0, 0};
  mutable std::mutex mu_;
};

struct CRSMatrix {
  int num_rows = 0, num_cols = 0;
  std::vector<int> cols, rows;
  std::vector<double> values;
};

// Frame-to-frame motion prior blocks (video_bundler_rs_inter.h:55-173): typed handles like ReprojectionCost, created
// by the factories in motion_priors.hpp.  Blocks: interFrameRatio[1], f.poses[0], f.poses[1], f-1.poses[0], f-1.poses[1];
// 12 residuals.  They are evaluated inside a Problem (device path, rsba_set_motion_priors); a stand-alone Evaluate of
// one block is not provided.
class MotionPriorCost : public CostFunction {
 public:
  MotionPriorCost(int kind, double scale) : kind_(kind), scale_(scale) {
    num_residuals_ = 12;
    sizes_ = {1, NUM_POSE_PARAMS, NUM_POSE_PARAMS, NUM_POSE_PARAMS, NUM_POSE_PARAMS};
  }
  int kind() const { return kind_; }       // 1 RsConstVeloPrior, 2 RsConstAccelerationPrior
  double scale() const { return scale_; }
  // not provided stand-alone (a `false` here would read as "the functor failed"): prior blocks are evaluated inside a Problem
  bool Evaluate(double const* const*, double*, double**) const override {
    throw std::logic_error("rsba_amd: MotionPriorCost::Evaluate is not provided stand-alone; add the block to a ceres::Problem (Problem::Evaluate / Solve)");
  }
 private:
  int kind_; double scale_;
};

// Per-pose prior blocks of CeresHandler (CeresHandler.h:24-73): typed handles created by the factories in pose_priors.hpp.
//   kind 0  GoodPosePrior(rotation, position): blocks priorPose[6], pose[6]; 6 residuals
//   kind 1  SphericalPrior: block pose[6]; 2 residuals
// Evaluated inside a Problem (device path, rsba_set_pose_priors); the reference attaches them without a loss function.
class PosePriorCost : public CostFunction {
 public:
  PosePriorCost(int kind, double rotation, double position) : kind_(kind), rotation_(rotation), position_(position) {
    num_residuals_ = kind == 0 ? 6 : 2;
    if (kind == 0) sizes_ = {NUM_POSE_PARAMS, NUM_POSE_PARAMS}; else sizes_ = {NUM_POSE_PARAMS};
  }
  int kind() const { return kind_; }
  double rotation() const { return rotation_; }
  double position() const { return position_; }
  bool Evaluate(double const* const*, double*, double**) const override {
    throw std::logic_error("rsba_amd: PosePriorCost::Evaluate is not provided stand-alone; add the block to a ceres::Problem (Problem::Evaluate / Solve)");
  }
 private:
  int kind_; double rotation_, position_;
};

class Problem;
struct Solver {
  struct Options {
    LinearSolverType linear_solver_type = SPARSE_NORMAL_CHOLESKY;
    bool minimizer_progress_to_stdout = false;
    int max_num_iterations = 50;
    int min_linear_solver_iterations = 1;
    int num_threads = 1, num_linear_solver_threads = 1;
    bool jacobi_scaling = true;
    bool use_nonmonotonic_steps = false;
    int max_num_consecutive_invalid_steps = 5;
    double initial_trust_region_radius = 1e4, max_trust_region_radius = 1e16, min_trust_region_radius = 1e-32;
    double min_relative_decrease = 1e-3, min_lm_diagonal = 1e-6, max_lm_diagonal = 1e32;
    double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
    int device = 0;   // extension: HIP device ordinal (one process per GPU)
    bool level_scheduled_cholesky = false;   // extension: see rsba_solver_options
  };
  struct Summary {
    TerminationType termination_type = FAILURE;
    std::string message;
    double initial_cost = 0, final_cost = 0, fixed_cost = 0;
    int num_successful_steps = 0, num_unsuccessful_steps = 0;
    int num_residual_blocks = 0, num_residual_blocks_reduced = 0, num_parameters_reduced = 0;
    int num_dag_fallbacks = 0;   // extension: see rsba_solver_summary
    double total_time_in_seconds = 0, jacobian_evaluation_time_in_seconds = 0, linear_solver_time_in_seconds = 0;
    std::vector<rsba_iteration> iterations;
    bool IsSolutionUsable() const { return termination_type == CONVERGENCE || termination_type == NO_CONVERGENCE || termination_type == USER_SUCCESS; }
    std::string BriefReport() const {
      std::ostringstream o;
      o << "rsba_amd report: iterations: " << iterations.size() << ", initial cost: " << initial_cost << ", final cost: " << final_cost
        << ", termination: " << (termination_type == CONVERGENCE ? "CONVERGENCE" : termination_type == NO_CONVERGENCE ? "NO_CONVERGENCE" : "FAILURE");
      return o.str();
    }
    std::string FullReport() const {
      std::ostringstream o;
      o << "Solver Summary (rsba_amd, MI355X)\n"
        << "Residual blocks      " << num_residual_blocks << " (reduced " << num_residual_blocks_reduced << ")\n"
        << "Effective parameters " << num_parameters_reduced << "\n"
        << "Linear solver        SPARSE_SCHUR (on-device point elimination + tile-sparse Cholesky)\n"
        << "Cost: initial " << initial_cost << "  final " << final_cost << "  change " << (initial_cost - final_cost) << "\n"
        << "Minimizer iterations " << iterations.size() << " (successful " << num_successful_steps << ", unsuccessful " << num_unsuccessful_steps << ")\n"
        << "Time (s): residual+jacobian " << jacobian_evaluation_time_in_seconds << "  linear solver " << linear_solver_time_in_seconds
        << "  total " << total_time_in_seconds << "\n"
        << "Termination: " << (termination_type == CONVERGENCE ? "CONVERGENCE" : termination_type == NO_CONVERGENCE ? "NO_CONVERGENCE" : "FAILURE")
        << (message.empty() ? "" : " (" + message + ")") << "\n";
      return o.str();
    }
  };
};

// ceres::Problem with default options: takes ownership of cost / loss / parameterization objects; the
// same HuberLoss* may be passed many times and is freed once (CeresHandler.h:252,259,270,276).
class Problem {
 public:
  struct EvaluateOptions {};
  Problem() {}
  Problem(const Problem&) = delete;
  Problem& operator=(const Problem&) = delete;
  ~Problem() {
    for (CostFunction* c : costs_) delete c;
    for (LossFunction* l : losses_) delete l;
    for (LocalParameterization* p : params_) delete p;
  }

  void AddResidualBlock(CostFunction* cost, LossFunction* loss, double* x0) { add(cost, loss, {x0}); }
  void AddResidualBlock(CostFunction* cost, LossFunction* loss, double* x0, double* x1) { add(cost, loss, {x0, x1}); }
  void AddResidualBlock(CostFunction* cost, LossFunction* loss, double* x0, double* x1, double* x2) { add(cost, loss, {x0, x1, x2}); }
  void AddResidualBlock(CostFunction* cost, LossFunction* loss, double* x0, double* x1, double* x2, double* x3) { add(cost, loss, {x0, x1, x2, x3}); }
  void AddResidualBlock(CostFunction* cost, LossFunction* loss, double* x0, double* x1, double* x2, double* x3, double* x4) { add(cost, loss, {x0, x1, x2, x3, x4}); }

  void SetParameterBlockConstant(double* values) { constant_.insert(values); }
  void SetParameterBlockVariable(double* values) { constant_.erase(values); }
  void SetParameterization(double* values, LocalParameterization* p) { if (p) params_.insert(p); parameterization_[values] = p; }
  void SetParameterLowerBound(double* values, int index, double bound) { lower_bounds_[std::make_pair(values, index)] = bound; }
  int NumParameterBlocks() const { return (int)block_sizes_.size(); }
  int NumResidualBlocks() const { return (int)blocks_.size(); }
  int NumResiduals() const { int n = 0; for (const Block& b : blocks_) n += b.cost->num_residuals(); return n; }

  // Problem::Evaluate (CeresHandler.h:386-387): cost = 1/2 sum rho(|r|^2); residuals in block order;
  // gradient over the parameter blocks in order of first appearance (full ambient size, zeros at fixed
  // coordinates).  The CRS Jacobian output is not provided.
  bool Evaluate(const EvaluateOptions&, double* cost, std::vector<double>* residuals, std::vector<double>* gradient, CRSMatrix* jacobian) {
    if (jacobian) return false;
    Flat f;
    if (!flatten(&f, nullptr)) return false;
    rsba_handle* h = nullptr;
    if (residuals && !f.prior_frames.empty()) return false;   // per-block residual output covers the reprojection blocks only
    if (gradient && !f.pp_blocks.empty()) return false;   // the gradient over the priorPoses blocks is not part of the device output
    if (create_handle(f, 0, &h) != RSBA_OK) return false;
    std::vector<double> g;
    if (gradient) g.resize(f.poses.size() + f.points.size() + f.intr.size());
    if (residuals) residuals->assign(2 * blocks_.size(), 0.0);
    int64_t nfail = 0;
    const int32_t st = rsba_evaluate(h, cost, residuals ? residuals->data() : nullptr, nullptr, gradient ? g.data() : nullptr, &nfail);
    rsba_destroy(h);
    if (st != RSBA_OK) return false;
    if (gradient) {
      gradient->clear();
      for (double* p : block_order_) {
        const Slot s = f.slot_of[p];
        if (s.kind == 3 || s.kind == 4) { gradient->insert(gradient->end(), (size_t)block_sizes_[p], 0.0); continue; }   // constant data block
        const double* src = s.kind == 0 ? &g[(size_t)s.index * 6] : s.kind == 1 ? &g[f.poses.size() + (size_t)s.index * 3] : &g[f.poses.size() + f.points.size() + (size_t)s.index * 9];
        gradient->insert(gradient->end(), src, src + block_sizes_[p]);
      }
    }
    return true;
  }

 private:
  friend void Solve(const Solver::Options&, Problem*, Solver::Summary*);
  friend class Covariance;
  struct Block { CostFunction* cost; LossFunction* loss; std::vector<double*> x; };
  struct Slot { int kind; int index; };   // kind 0 pose block, 1 point, 2 intrinsics, 3 scalar (interFrameRatio), 4 priorPoses block of a GoodPosePrior
  struct Flat {
    rsba_problem_desc desc;
    std::vector<double> poses, points, intr, xy;
    std::vector<int32_t> obs_frame, obs_point, frame_intr;
    std::vector<uint8_t> pose_mask, point_const, intr_const;
    std::vector<uint8_t> frame_global;                     // two-pose problems that contain one-pose frames (CeresHandler.h:266-285): 1 per such frame; empty = none
    std::vector<double*> pose_ptr, point_ptr, intr_ptr;   // where each flat block came from (nullptr: data, not a block)
    std::map<double*, Slot> slot_of;
    // motion priors (constant interFrameRatio): lowered to rsba_set_motion_priors
    int prior_kind = 0; double prior_scale = 0, prior_ratio = 1;
    std::vector<int32_t> prior_frames;
    double* ratio_ptr = nullptr;       // the interFrameRatio block; ratio_free: it is a (lower-bounded) parameter of the solve
    bool ratio_free = false;
    // per-pose priors: lowered to rsba_set_pose_priors
    std::vector<int32_t> pp_blocks; std::vector<double> pp_values; std::vector<double*> pp_ptr;   // GoodPosePrior: pose block, priorPoses values / where they came from
    double pp_rotation = 0, pp_position = 0; int32_t spherical_block = -1;
  };
  // device problem of a flattened graph: rsba_create + the prior blocks
  static int32_t create_handle(const Flat& f, int device, rsba_handle** h) {
    int32_t st = rsba_create(&f.desc, device, h);
    if (st != RSBA_OK) return st;
    if (!f.frame_global.empty()) {
      st = rsba_set_global_shutter_frames(*h, f.frame_global.data());
      if (st != RSBA_OK) { rsba_destroy(*h); *h = nullptr; return st; }
    }
    if (!f.prior_frames.empty()) {
      st = rsba_set_motion_priors(*h, f.prior_kind, f.prior_scale, f.prior_ratio, f.prior_frames.data(), (int32_t)f.prior_frames.size());
      if (st == RSBA_OK && f.ratio_free) st = rsba_set_inter_frame_ratio_free(*h, 1);
      if (st != RSBA_OK) { rsba_destroy(*h); *h = nullptr; return st; }
    }
    if (!f.pp_blocks.empty() || f.spherical_block >= 0) {
      st = rsba_set_pose_priors(*h, f.pp_rotation, f.pp_position, f.pp_blocks.data(), const_cast<double*>(f.pp_values.data()), (int32_t)f.pp_blocks.size(), f.spherical_block);
      if (st != RSBA_OK) { rsba_destroy(*h); *h = nullptr; }
    }
    return st;
  }

  void add(CostFunction* cost, LossFunction* loss, std::vector<double*> x) {
    costs_.insert(cost);
    if (loss) losses_.insert(loss);
    const std::vector<int32_t>& sz = cost->parameter_block_sizes();
    for (size_t i = 0; i < x.size() && i < sz.size(); ++i)
      if (!block_sizes_.count(x[i])) { block_sizes_[x[i]] = sz[i]; block_order_.push_back(x[i]); }
    blocks_.push_back(Block{cost, loss, std::move(x)});
  }

  unsigned mask_of(double* p, int size) const {
    if (constant_.count(p)) return size >= 32 ? 0xffffffffu : ((1u << size) - 1u);
    auto it = parameterization_.find(p);
    if (it != parameterization_.end()) if (auto* s = dynamic_cast<SubsetParameterization*>(it->second)) return s->constancy_mask();
    return 0;
  }

  // AoS pointer graph -> the flat SoA description of include/rsba_amd.h.  Returns false (with a
  // message) for anything outside the accelerated path.
  bool flatten(Flat* f, std::string* why) {
    auto fail = [&](const char* m) { if (why) *why = m; return false; };
    if (blocks_.empty()) return fail("no residual blocks");
    const ReprojectionCost* first = nullptr; LossFunction* loss0 = nullptr;
    for (const Block& b : blocks_) if ((first = dynamic_cast<const ReprojectionCost*>(b.cost))) { loss0 = b.loss; break; }
    if (!first) return fail("only rsba's reprojection cost functions (plus motion priors between their frames) are accelerated");
    // CeresHandler::Add picks the functor per frame (f.poses.size() == 2: RsBundleAdjustment, else ReprojectionError on getPose(...),
    // CeresHandler.h:245-286): a session may mix them.  The flat problem then has two pose slots per frame; a one-pose frame uses
    // the first (flagged for rsba_set_global_shutter_frames), its second slot is a constant copy that is not written back.
    bool rolling = false;
    for (const Block& b : blocks_) if (const ReprojectionCost* c = dynamic_cast<const ReprojectionCost*>(b.cost)) rolling = rolling || c->rolling();
    const bool with_cam = first->with_cam();
    const int P = rolling ? 2 : 1;
    const ReprojectionCost* first_rolling = nullptr;      // shutter / scanlines / interpolateRotation of the session come from a rolling-shutter block
    for (const Block& b : blocks_) if (const ReprojectionCost* c = dynamic_cast<const ReprojectionCost*>(b.cost)) if (c->rolling()) { first_rolling = c; break; }
    if (first_rolling) first = first_rolling;
    std::map<std::pair<double*, double*>, int> frame_of;
    std::map<double*, int> point_of, intr_of;
    std::map<std::vector<double>, int> intr_by_value;
    int32_t sl[2]; first->scanlines(sl);
    for (const Block& b : blocks_) {
      if (dynamic_cast<const MotionPriorCost*>(b.cost) || dynamic_cast<const PosePriorCost*>(b.cost)) continue;   // later passes, once every frame has its number
      const ReprojectionCost* c = dynamic_cast<const ReprojectionCost*>(b.cost);
      if (!c) return fail("only rsba's reprojection cost functions (plus motion priors between their frames) are accelerated");
      if (c->with_cam() != with_cam) return fail("residual blocks with and without an intrinsics parameter block in one problem are not supported");
      const bool one_pose = rolling && !c->rolling();
      if (b.loss != loss0) return fail("all residual blocks must share one loss function (as CeresHandler does)");
      if (b.x.size() != c->parameter_block_sizes().size()) return fail("wrong number of parameter blocks");
      int32_t s2[2]; c->scanlines(s2);
      if (!one_pose && (c->shutter() != first->shutter() || s2[0] != sl[0] || s2[1] != sl[1] || c->interpolate_rotation() != first->interpolate_rotation()))
        return fail("residual blocks disagree on shutter / scanlines / interpolateRotation");
      size_t k = 0;
      double* camp = with_cam ? b.x[k++] : nullptr;
      double* p0 = b.x[k++];
      double* p1 = c->rolling() ? b.x[k++] : nullptr;
      double* pt = b.x[k++];
      int ci;
      if (with_cam) {
        auto it = intr_of.find(camp);
        if (it == intr_of.end()) { ci = (int)f->intr_ptr.size(); intr_of[camp] = ci; f->intr_ptr.push_back(camp); f->intr.insert(f->intr.end(), camp, camp + 9); f->slot_of[camp] = Slot{2, ci}; }
        else ci = it->second;
      } else {
        std::vector<double> v(c->fixed_cam(), c->fixed_cam() + 9);
        auto it = intr_by_value.find(v);
        if (it == intr_by_value.end()) { ci = (int)f->intr_ptr.size(); intr_by_value[v] = ci; f->intr_ptr.push_back(nullptr); f->intr.insert(f->intr.end(), v.begin(), v.end()); }
        else ci = it->second;
      }
      const auto key = std::make_pair(p0, p1);
      int fi;
      auto itf = frame_of.find(key);
      if (itf == frame_of.end()) {
        fi = (int)f->frame_intr.size(); frame_of[key] = fi; f->frame_intr.push_back(ci);
        f->pose_ptr.push_back(p0); f->poses.insert(f->poses.end(), p0, p0 + 6); f->slot_of[p0] = Slot{0, fi * P};
        if (rolling && p1) { f->pose_ptr.push_back(p1); f->poses.insert(f->poses.end(), p1, p1 + 6); f->slot_of[p1] = Slot{0, fi * P + 1}; }
        else if (rolling) {   // a one-pose frame in a two-pose problem: the second slot is data (a copy of the pose), not a block
          f->pose_ptr.push_back(nullptr); f->poses.insert(f->poses.end(), p0, p0 + 6);
          if (f->frame_global.size() < f->frame_intr.size()) f->frame_global.resize(f->frame_intr.size(), 0);
          f->frame_global[fi] = 1;
        }
      } else { fi = itf->second; if (f->frame_intr[fi] != ci) return fail("one frame observed through two different intrinsics"); }
      int pi;
      auto itp = point_of.find(pt);
      if (itp == point_of.end()) { pi = (int)f->point_ptr.size(); point_of[pt] = pi; f->point_ptr.push_back(pt); f->points.insert(f->points.end(), pt, pt + 3); f->slot_of[pt] = Slot{1, pi}; }
      else pi = itp->second;
      f->xy.push_back(c->observed()[0]); f->xy.push_back(c->observed()[1]);
      f->obs_frame.push_back(fi); f->obs_point.push_back(pi);
    }
    // motion priors (CeresHandler.h:147-185): blocks (ratio, f.p0, f.p1, f-1.p0, f-1.p1) between consecutive frames
    std::vector<int> prior_prev;
    for (const Block& b : blocks_) {
      const MotionPriorCost* c = dynamic_cast<const MotionPriorCost*>(b.cost);
      if (!c) continue;
      if (!rolling || b.x.size() != 5) return fail("motion priors need rolling-shutter frames with two poses");
      if (b.loss != loss0) return fail("all residual blocks must share one loss function (as CeresHandler does)");
      const bool ratio_free = !constant_.count(b.x[0]);                 // CeresHandler.h:175: free unless the option differs from 1
      if (ratio_free) {
        // the bound CeresHandler sets (:161 _EPS for the acceleration prior, :172 0 for the velocity prior) is the one the
        // device path applies; anything else is refused rather than silently replaced
        auto lb = lower_bounds_.find(std::make_pair(b.x[0], 0));
        const double want = c->kind() == 2 ? std::numeric_limits<double>::epsilon() : 0.0;
        if (lb == lower_bounds_.end() || lb->second != want) return fail("a free interFrameRatio needs the lower bound CeresHandler sets (0 / DBL_EPSILON)");
      }
      if (!f->prior_frames.empty() && (f->ratio_ptr != b.x[0] || f->ratio_free != ratio_free)) return fail("motion priors of one problem must share the interFrameRatio block");
      f->ratio_ptr = b.x[0]; f->ratio_free = ratio_free;
      // A pose pair that no reprojection block uses is a frame of its own, without observations: the frame before the
      // window of a windowed BA (CeresHandler::Add(startFrame) links startFrame to startFrame - 1, VideoSfMHandler.cc:779),
      // or a frame whose observations were all rejected.  Ceres optimises such a prior-only block; so does the device path.
      auto frame_index = [&](double* p0, double* p1) {
        const auto key = std::make_pair(p0, p1);
        auto it = frame_of.find(key);
        if (it != frame_of.end()) return it->second;
        const int fi = (int)f->frame_intr.size(); frame_of[key] = fi; f->frame_intr.push_back(0);
        f->pose_ptr.push_back(p0); f->poses.insert(f->poses.end(), p0, p0 + 6); f->slot_of[p0] = Slot{0, fi * P};
        f->pose_ptr.push_back(p1); f->poses.insert(f->poses.end(), p1, p1 + 6); f->slot_of[p1] = Slot{0, fi * P + 1};
        return fi;
      };
      const int prev = frame_index(b.x[3], b.x[4]), cur = frame_index(b.x[1], b.x[2]);
      if (prev == cur) return fail("a motion prior links a frame with itself");
      if (f->prior_frames.empty()) { f->prior_kind = c->kind(); f->prior_scale = c->scale(); f->prior_ratio = *b.x[0]; }
      else if (f->prior_kind != c->kind() || f->prior_scale != c->scale() || f->prior_ratio != *b.x[0]) return fail("motion priors of one problem must share kind, scale and ratio");
      f->prior_frames.push_back(cur); prior_prev.push_back(prev);
      f->slot_of[b.x[0]] = Slot{3, 0};
    }
    if (!f->prior_frames.empty()) {
      // The device path lists a prior by its frame f and links it to frame f - 1: number the frames so that every chain of
      // priors runs through consecutive indices (the identity when frames were added in order, as CeresHandler does).
      const int nf = (int)f->frame_intr.size();
      std::vector<int> next_of(nf, -1), prev_of(nf, -1);
      for (size_t k = 0; k < f->prior_frames.size(); ++k) {
        const int cur = f->prior_frames[k], prev = prior_prev[k];
        if (prev_of[cur] != -1) return fail("two motion priors on one frame");
        if (next_of[prev] != -1) return fail("two motion priors refer back to the same frame");
        prev_of[cur] = prev; next_of[prev] = cur;
      }
      std::vector<int> order; order.reserve(nf);
      for (int fi = 0; fi < nf; ++fi) if (prev_of[fi] == -1) for (int x = fi; x != -1; x = next_of[x]) order.push_back(x);
      if ((int)order.size() != nf) return fail("motion priors form a cycle");
      std::vector<int> new_of(nf);
      for (int k = 0; k < nf; ++k) new_of[order[k]] = k;
      bool identity = true;
      for (int k = 0; k < nf; ++k) identity = identity && order[k] == k;
      if (!identity) {
        std::vector<int32_t> fi2(nf); std::vector<double*> pp2(f->pose_ptr.size()); std::vector<double> po2(f->poses.size());
        if (!f->frame_global.empty()) f->frame_global.resize((size_t)nf, 0);
        std::vector<uint8_t> fg2(f->frame_global.size());
        for (int k = 0; k < nf; ++k) {
          const int o = order[k];
          fi2[k] = f->frame_intr[o];
          if (!fg2.empty()) fg2[k] = f->frame_global[o];
          for (int q = 0; q < P; ++q) {
            pp2[(size_t)k * P + q] = f->pose_ptr[(size_t)o * P + q];
            std::memcpy(&po2[((size_t)k * P + q) * 6], &f->poses[((size_t)o * P + q) * 6], 6 * sizeof(double));
            if (pp2[(size_t)k * P + q]) f->slot_of[pp2[(size_t)k * P + q]] = Slot{0, k * P + q};
          }
        }
        f->frame_global.swap(fg2);
        f->frame_intr.swap(fi2); f->pose_ptr.swap(pp2); f->poses.swap(po2);
        for (int32_t& x : f->obs_frame) x = new_of[x];
        for (int32_t& x : f->prior_frames) x = new_of[x];
      }
    }
    std::sort(f->prior_frames.begin(), f->prior_frames.end());
    // per-pose priors (CeresHandler.h:127-130, 188-204): GoodPosePrior over (priorPoses[i], poses[i]), SphericalPrior over poses[0]
    for (const Block& b : blocks_) {
      const PosePriorCost* c = dynamic_cast<const PosePriorCost*>(b.cost);
      if (!c) continue;
      if (b.loss != nullptr) return fail("pose priors take no loss function (CeresHandler passes nullptr)");
      if (b.x.size() != c->parameter_block_sizes().size()) return fail("wrong number of parameter blocks");
      double* pose = b.x.back();
      auto it = f->slot_of.find(pose);
      if (it == f->slot_of.end() || it->second.kind != 0) return fail("pose prior on a pose block that no reprojection block or motion prior uses");
      if (c->kind() == 1) {
        if (f->spherical_block >= 0) return fail("two SphericalPrior blocks in one problem");
        f->spherical_block = it->second.index;
      } else {
        if (constant_.count(b.x[0])) return fail("a constant priorPoses block is not supported (CeresHandler leaves them free)");
        if (f->slot_of.count(b.x[0])) return fail("a priorPoses block is used in another role");
        if (!f->pp_blocks.empty() && (f->pp_rotation != c->rotation() || f->pp_position != c->position())) return fail("GoodPosePrior blocks of one problem must share their weights");
        f->pp_rotation = c->rotation(); f->pp_position = c->position();
        f->pp_blocks.push_back(it->second.index); f->pp_ptr.push_back(b.x[0]);
        f->pp_values.insert(f->pp_values.end(), b.x[0], b.x[0] + 6);
      }
    }
    for (size_t k = 0; k < f->pp_ptr.size(); ++k) f->slot_of[f->pp_ptr[k]] = Slot{4, (int)k};
    const int nscalar = (f->prior_frames.empty() ? 0 : 1) + (int)f->pp_ptr.size();
    // a pose pointer may not serve as pose0 of one frame and pose1 of another
    int npose_blocks = 0;
    for (double* p : f->pose_ptr) npose_blocks += p != nullptr;   // (the second slot of a one-pose frame is data, not a block)
    if ((int)f->slot_of.size() - nscalar != npose_blocks + (int)f->point_ptr.size() + (with_cam ? (int)f->intr_ptr.size() : 0))
      return fail("a parameter block is used in two different roles");
    for (double* p : f->pose_ptr) f->pose_mask.push_back(p ? (uint8_t)(mask_of(p, 6) & 0x3f) : (uint8_t)0x3f);
    if (!f->frame_global.empty()) f->frame_global.resize(f->frame_intr.size(), 0);
    for (double* p : f->point_ptr) f->point_const.push_back(constant_.count(p) ? 1 : 0);
    for (double* p : f->intr_ptr) f->intr_const.push_back(p && constant_.count(p) ? 1 : 0);
    for (const auto& lb : lower_bounds_) { auto it = f->slot_of.find(lb.first.first); if (it != f->slot_of.end() && it->second.kind != 3) return fail("bounds on pose / point / intrinsics blocks are not supported"); }
    for (double* q : f->pp_ptr) if (parameterization_.count(q)) return fail("a parameterization on a priorPoses block is not supported");
    rsba_problem_desc& d = f->desc; std::memset(&d, 0, sizeof d);
    d.shutter = first->shutter(); d.scanlines[0] = sl[0]; d.scanlines[1] = sl[1];
    d.interpolate_rotation = first->interpolate_rotation(); d.calibrated = !with_cam; d.poses_per_frame = P;
    d.num_frames = (int32_t)f->frame_intr.size(); d.num_points = (int32_t)f->point_ptr.size(); d.num_intrinsics = (int32_t)f->intr_ptr.size();
    d.num_observations = (int64_t)f->obs_frame.size();
    d.poses = f->poses.data(); d.points = f->points.data(); d.intrinsics = f->intr.data(); d.frame_intrinsics = f->frame_intr.data();
    d.obs_xy = f->xy.data(); d.obs_frame = f->obs_frame.data(); d.obs_point = f->obs_point.data();
    d.pose_fixed_mask = f->pose_mask.data(); d.point_constant = f->point_const.data(); d.intrinsics_constant = f->intr_const.data();
    const HuberLoss* hl = dynamic_cast<const HuberLoss*>(loss0);
    if (loss0 && !hl) return fail("only ceres::HuberLoss is supported");
    d.huber_a = hl ? hl->a() : 0.0;
    return true;
  }

  // results of a solve go back into the caller's blocks, as ceres::Solve mutates them in place
  void scatter(const Flat& f) {
    const int nposeblk = (int)f.pose_ptr.size();
    for (int b = 0; b < nposeblk; ++b) if (f.pose_ptr[b]) std::memcpy(f.pose_ptr[b], &f.poses[(size_t)b * 6], 6 * sizeof(double));
    for (size_t j = 0; j < f.point_ptr.size(); ++j) std::memcpy(f.point_ptr[j], &f.points[j * 3], 3 * sizeof(double));
    for (size_t c = 0; c < f.intr_ptr.size(); ++c) if (f.intr_ptr[c]) std::memcpy(f.intr_ptr[c], &f.intr[c * 9], 9 * sizeof(double));
    for (size_t k = 0; k < f.pp_ptr.size(); ++k) std::memcpy(f.pp_ptr[k], &f.pp_values[k * 6], 6 * sizeof(double));   // the priorPoses blocks are solved for too
  }

  std::vector<Block> blocks_;
  std::set<CostFunction*> costs_;
  std::set<LossFunction*> losses_;
  std::set<LocalParameterization*> params_;
  std::set<double*> constant_;
  std::map<double*, LocalParameterization*> parameterization_;
  std::map<std::pair<double*, int>, double> lower_bounds_;
  std::map<double*, int> block_sizes_;
  std::vector<double*> block_order_;
};

// ceres::Solve(options, &problem, &summary) (CeresHandler.h:419).  Synchronous; never throws; status in
// the summary.  Any linear_solver_type is served by the one exact Schur-complement solver.
inline void Solve(const Solver::Options& options, Problem* problem, Solver::Summary* summary) {
  *summary = Solver::Summary();
  Problem::Flat f;
  std::string why;
  if (!problem->flatten(&f, &why)) { summary->termination_type = FAILURE; summary->message = why; return; }
  rsba_handle* h = nullptr;
  int32_t st = Problem::create_handle(f, options.device, &h);
  if (st != RSBA_OK) { summary->termination_type = FAILURE; summary->message = std::string(rsba_status_string(st)) + ": " + rsba_last_error(); return; }
  rsba_solver_options o; rsba_default_solver_options(&o);
  o.max_num_iterations = options.max_num_iterations; o.jacobi_scaling = options.jacobi_scaling;
  o.max_num_consecutive_invalid_steps = options.max_num_consecutive_invalid_steps;
  o.minimizer_progress_to_stdout = options.minimizer_progress_to_stdout;
  o.initial_trust_region_radius = options.initial_trust_region_radius; o.max_trust_region_radius = options.max_trust_region_radius;
  o.min_trust_region_radius = options.min_trust_region_radius; o.min_relative_decrease = options.min_relative_decrease;
  o.min_lm_diagonal = options.min_lm_diagonal; o.max_lm_diagonal = options.max_lm_diagonal;
  o.function_tolerance = options.function_tolerance; o.gradient_tolerance = options.gradient_tolerance; o.parameter_tolerance = options.parameter_tolerance;
  o.level_scheduled_cholesky = options.level_scheduled_cholesky ? 1 : 0;
  rsba_solver_summary s;
  std::vector<rsba_iteration> trace((size_t)options.max_num_iterations + 2);
  st = rsba_solve(h, &o, &s, trace.data(), (int32_t)trace.size());
  double solved_ratio = 0.0;
  if (f.ratio_free) (void)rsba_get_inter_frame_ratio(h, &solved_ratio);
  rsba_destroy(h);
  summary->termination_type = s.termination_type == RSBA_CONVERGENCE ? CONVERGENCE : s.termination_type == RSBA_NO_CONVERGENCE ? NO_CONVERGENCE : FAILURE;
  if (st != RSBA_OK) { summary->termination_type = FAILURE; summary->message = std::string(rsba_status_string(st)) + ": " + rsba_last_error(); }
  summary->initial_cost = s.initial_cost; summary->final_cost = s.final_cost; summary->fixed_cost = s.fixed_cost;
  summary->num_successful_steps = s.num_successful_steps; summary->num_unsuccessful_steps = s.num_unsuccessful_steps;
  summary->num_residual_blocks = s.num_residual_blocks; summary->num_residual_blocks_reduced = s.num_residual_blocks_reduced;
  summary->num_parameters_reduced = s.num_parameters_reduced; summary->num_dag_fallbacks = s.num_dag_fallbacks;
  summary->total_time_in_seconds = s.total_time_s; summary->jacobian_evaluation_time_in_seconds = s.residual_jacobian_time_s;
  summary->linear_solver_time_in_seconds = s.linear_solver_time_s;
  trace.resize(std::min<size_t>(trace.size(), (size_t)std::max(0, s.num_iterations)));
  summary->iterations = trace;
  if (summary->IsSolutionUsable()) {
    problem->scatter(f);
    if (f.ratio_free && f.ratio_ptr) *f.ratio_ptr = solved_ratio;      // a free ratio block is solved for like every parameter block
  }
}

// ceres::Covariance for pose blocks (the use rsba makes of it, VideoSfMHandler.cc:602-621): Compute() takes pairs of
// pose blocks of the problem, GetCovarianceBlock() returns the row-major 6 x 6 block between two poses of ONE
// frame.  Each frame asked for costs one call of rsba_pose_covariance (CD solves through the device factorisation).
// Compute() returns false — like Ceres — when J^T J is rank deficient, and for blocks outside the accelerated
// path (points, intrinsics, poses of two different frames).
class Covariance {
 public:
  struct Options { int device = 0; };
  Covariance() {}
  explicit Covariance(const Options& o) : options_(o) {}

  bool Compute(const std::vector<std::pair<const double*, const double*>>& blocks, Problem* problem) {
    ready_.clear();
    Problem::Flat f;
    if (!problem->flatten(&f, nullptr)) return false;
    const int P = f.desc.poses_per_frame, CD = 6 * P;
    std::set<int> frames;
    for (const auto& b : blocks) {
      auto i0 = f.slot_of.find(const_cast<double*>(b.first)), i1 = f.slot_of.find(const_cast<double*>(b.second));
      if (i0 == f.slot_of.end() || i1 == f.slot_of.end() || i0->second.kind != 0 || i1->second.kind != 0) return false;
      if (i0->second.index / P != i1->second.index / P) return false;
      frames.insert(i0->second.index / P);
    }
    rsba_handle* h = nullptr;
    if (Problem::create_handle(f, options_.device, &h) != RSBA_OK) return false;
    bool ok = true;
    for (int fr : frames) {
      std::vector<double> cov((size_t)CD * CD);
      if (rsba_pose_covariance(h, fr, cov.data()) != RSBA_OK) { ok = false; break; }
      for (int q = 0; q < P; ++q) if (f.pose_ptr[(size_t)fr * P + q]) index_[f.pose_ptr[(size_t)fr * P + q]] = std::make_pair(fr, q);
      ready_[fr] = std::move(cov);
    }
    rsba_destroy(h);
    poses_per_frame_ = P;
    if (!ok) ready_.clear();
    return ok;
  }

  bool GetCovarianceBlock(const double* p0, const double* p1, double* out) const {
    auto i0 = index_.find(p0), i1 = index_.find(p1);
    if (i0 == index_.end() || i1 == index_.end() || i0->second.first != i1->second.first) return false;
    auto it = ready_.find(i0->second.first);
    if (it == ready_.end()) return false;
    const int CD = 6 * poses_per_frame_, r0 = 6 * i0->second.second, c0 = 6 * i1->second.second;
    for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) out[a * 6 + b] = it->second[(size_t)(r0 + a) * CD + c0 + b];
    return true;
  }

 private:
  Options options_;
  int poses_per_frame_ = 1;
  std::map<const double*, std::pair<int, int>> index_;   // pose block -> (frame, pose within the frame)
  std::map<int, std::vector<double>> ready_;             // frame -> [CD][CD]
};

}  // namespace ceres
}  // namespace rsba_amd
