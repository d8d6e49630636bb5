"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so).

ORACLE — TEST INFRASTRUCTURE ONLY.  Import this from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never from rsba_amd/.  PARITY UNPINNED beyond the reference's mat_test.cc cases
(see rsba_oracle_math.hpp).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("rsba_oracle.cpp", "rsba_oracle.h", "rsba_oracle_math.hpp", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", _HERE] + (["-B"] if force else []), check=True, capture_output=True)
    return _LIB_PATH


class OrcProblem(C.Structure):
    _fields_ = [
        ("shutter", C.c_int32), ("scanlines", C.c_int32 * 2), ("interpolate_rotation", C.c_int32),
        ("calibrated", C.c_int32), ("poses_per_frame", C.c_int32),
        ("num_frames", C.c_int32), ("num_points", C.c_int32), ("num_intrinsics", C.c_int32),
        ("num_observations", C.c_int64),
        ("poses", C.c_void_p), ("points", C.c_void_p), ("intrinsics", C.c_void_p),
        ("frame_intrinsics", C.c_void_p), ("obs_xy", C.c_void_p), ("obs_frame", C.c_void_p),
        ("obs_point", C.c_void_p), ("pose_fixed_mask", C.c_void_p), ("point_constant", C.c_void_p),
        ("intrinsics_constant", C.c_void_p), ("huber_a", C.c_double),
        ("prior_kind", C.c_int32), ("num_priors", C.c_int32), ("prior_frames", C.c_void_p),
        ("prior_scale", C.c_double), ("inter_frame_ratio", C.c_double),
        ("num_pose_priors", C.c_int32), ("pose_prior_block", C.c_void_p), ("pose_prior_values", C.c_void_p),
        ("pose_prior_rotation", C.c_double), ("pose_prior_position", C.c_double),
        ("spherical_pose_block", C.c_int32), ("has_spherical", C.c_int32),
        ("no_validate", C.c_int32), ("ratio_free", C.c_int32), ("frame_global", C.c_void_p),
    ]


class OrcOptions(C.Structure):
    _fields_ = [
        ("max_num_iterations", C.c_int32), ("jacobi_scaling", C.c_int32),
        ("max_num_consecutive_invalid_steps", C.c_int32), ("num_threads", C.c_int32),
        ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double),
        ("min_trust_region_radius", C.c_double), ("min_relative_decrease", C.c_double),
        ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
        ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
    ]


class OrcIteration(C.Structure):
    _fields_ = [
        ("iteration", C.c_int32), ("step_is_valid", C.c_int32), ("step_is_successful", C.c_int32), ("pad", C.c_int32),
        ("cost", C.c_double), ("cost_change", C.c_double), ("gradient_max_norm", C.c_double), ("step_norm", C.c_double),
        ("relative_decrease", C.c_double), ("trust_region_radius", C.c_double), ("model_cost_change", C.c_double),
    ]


class OrcSummary(C.Structure):
    _fields_ = [
        ("termination_type", C.c_int32), ("num_successful_steps", C.c_int32), ("num_unsuccessful_steps", C.c_int32),
        ("num_iterations", C.c_int32), ("num_residual_blocks", C.c_int32), ("num_residual_blocks_reduced", C.c_int32),
        ("num_parameters_reduced", C.c_int32), ("pad", C.c_int32),
        ("initial_cost", C.c_double), ("final_cost", C.c_double), ("fixed_cost", C.c_double),
    ]


_lib = None


def select_build(which: str = "host") -> str:
    """bench.py's cpu_baseline only: switch to the AVX-512 build of the same source when the host supports it ("host"), so that the
    reported baseline is the box's best; "v3" switches back.  Returns the library path in use.  The parity tests never call this."""
    global _lib, _LIB_PATH
    build()
    path = os.path.join(_HERE, "_build", "liboracle.so")
    if which == "host":
        try:
            flags = next(l for l in open("/proc/cpuinfo") if l.startswith("flags")).split()
        except Exception:  # noqa: BLE001
            flags = []
        v4 = os.path.join(_HERE, "_build", "liboracle_v4.so")
        if all(f in flags for f in ("avx512f", "avx512bw", "avx512cd", "avx512dq", "avx512vl")) and os.path.exists(v4):
            path = v4
    if path != _LIB_PATH or _lib is None:
        _LIB_PATH, _lib = path, None
    return path


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_evaluate_blocks.restype = C.c_int64
        _lib.orc_evaluate_blocks_ceres_style.restype = C.c_int64
        _lib.orc_evaluate_residuals.restype = C.c_int64
        _lib.orc_norm3.restype = C.c_double
        _lib.orc_huber.argtypes = [C.c_double, C.c_double, C.c_void_p]
        # the checker's many small parallel regions crawl with a 256-thread team on a 2-socket host:
        # keep its default team small (the cpu_baseline leg passes its own thread count explicitly)
        _lib.orc_set_num_threads(C.c_int32(int(os.environ.get("RSBA_ORACLE_THREADS", min(os.cpu_count() or 1, 16)))))
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def desc(prob) -> OrcProblem:
    """Build the C descriptor over the problem's own numpy arrays (no copies: solves write in place)."""
    d = OrcProblem()
    d.shutter = int(prob.shutter)
    d.scanlines[0], d.scanlines[1] = int(prob.scanlines[0]), int(prob.scanlines[1])
    d.interpolate_rotation = int(bool(prob.interpolate_rotation))
    d.calibrated = int(bool(prob.calibrated))
    d.poses_per_frame = prob.poses_per_frame
    d.num_frames, d.num_points, d.num_intrinsics = prob.num_frames, prob.num_points, prob.num_intrinsics
    d.num_observations = prob.num_observations
    d.poses, d.points, d.intrinsics = _ptr(prob.poses), _ptr(prob.points), _ptr(prob.intrinsics)
    d.frame_intrinsics = _ptr(prob.frame_intrinsics)
    d.obs_xy, d.obs_frame, d.obs_point = _ptr(prob.obs_xy), _ptr(prob.obs_frame), _ptr(prob.obs_point)
    d.pose_fixed_mask = _ptr(prob.pose_fixed_mask)
    d.point_constant = _ptr(prob.point_constant)
    d.intrinsics_constant = _ptr(prob.intrinsics_constant)
    d.huber_a = float(prob.huber_a)
    d.prior_kind = int(prob.prior_kind) if prob.prior_frames is not None and len(prob.prior_frames) else 0
    d.num_priors = 0 if d.prior_kind == 0 else len(prob.prior_frames)
    d.prior_frames = _ptr(prob.prior_frames) if d.prior_kind else None
    d.prior_scale, d.inter_frame_ratio = float(prob.prior_scale), float(prob.inter_frame_ratio)
    d.ratio_free = int(bool(getattr(prob, 'ratio_free', False)))
    d.frame_global = _ptr(getattr(prob, 'frame_global', None))
    pb = getattr(prob, "pose_prior_block", None)
    if pb is not None and len(pb):
        d.num_pose_priors = len(pb)
        d.pose_prior_block, d.pose_prior_values = _ptr(prob.pose_prior_block), _ptr(prob.pose_prior_values)
        d.pose_prior_rotation, d.pose_prior_position = float(prob.pose_prior_rotation), float(prob.pose_prior_position)
    sp = getattr(prob, "spherical_pose_block", None)
    if sp is not None and sp >= 0:
        d.spherical_pose_block, d.has_spherical = int(sp), 1
    d._keep = prob  # keep arrays alive
    return d


def evaluate_blocks(prob, jac: bool = True, threads: int = 0, d: OrcProblem | None = None):
    """-> residuals [N,2], jacobians [N,2,K] (or None), ok [N] bool"""
    d = d or desc(prob)
    n, k = prob.num_observations, prob.jacobian_cols
    r = np.zeros((n, 2))
    J = np.zeros((n, 2, k)) if jac else None
    ok = np.zeros(n, dtype=np.uint8)
    if jac:
        lib().orc_evaluate_blocks(C.byref(d), _ptr(r), _ptr(J), _ptr(ok), C.c_int32(threads))
    else:
        lib().orc_evaluate_residuals(C.byref(d), _ptr(r), _ptr(ok), C.c_int32(threads))
    return r, J, ok.astype(bool)


def evaluate(prob, gradient: bool = True):
    """Problem::Evaluate -> (ok, cost, gradient dict or None)"""
    d = desc(prob)
    cost = C.c_double(0.0)
    npose = prob.num_frames * prob.poses_per_frame * 6
    g = np.zeros(npose + 3 * prob.num_points + 9 * prob.num_intrinsics) if gradient else None
    rc = lib().orc_evaluate(C.byref(d), C.byref(cost), _ptr(g))
    gd = None
    if gradient:
        gd = dict(poses=g[:npose].reshape(prob.poses.shape), points=g[npose:npose + 3 * prob.num_points].reshape(-1, 3),
                  intrinsics=g[npose + 3 * prob.num_points:].reshape(-1, 9))
    return rc == 0, cost.value, gd


def normal_equations(prob):
    d = desc(prob)
    cd = 6 * prob.poses_per_frame
    U = np.zeros((prob.num_frames, cd, cd)); gc = np.zeros((prob.num_frames, cd))
    V = np.zeros((prob.num_points, 3, 3)); gp = np.zeros((prob.num_points, 3))
    rc = lib().orc_normal_equations(C.byref(d), _ptr(U), _ptr(gc), _ptr(V), _ptr(gp))
    assert rc == 0, rc
    return U, gc, V, gp


def pose_covariance(prob, frame: int):
    """ceres::Covariance blocks of one frame -> ([CD, CD] array, ok)"""
    d = desc(prob)
    cd = 6 * prob.poses_per_frame
    cov = np.zeros((cd, cd))
    ok = lib().orc_pose_covariance(C.byref(d), C.c_int32(frame), _ptr(cov))
    return cov, bool(ok)


def pnp_task(cam, shutter, scanlines, object_points, image_points, subset, init_poses, max_iter=10, reprojection_error=8.0, drop_coincident=True):
    """One RANSAC hypothesis of solveRsPnPRansac (rsba_oracle.h: orc_pnp_task) -> dict or None if skipped"""
    cam = np.ascontiguousarray(cam, dtype=np.float64); sl = np.ascontiguousarray(scanlines, dtype=np.int32)
    op = np.ascontiguousarray(object_points, dtype=np.float32).reshape(-1, 3); ip = np.ascontiguousarray(image_points, dtype=np.float32).reshape(-1, 2)
    sub = np.ascontiguousarray(subset, dtype=np.int32); init = np.ascontiguousarray(init_poses, dtype=np.float64).reshape(12)
    poses = np.zeros((2, 6)); mask = np.zeros(len(op), dtype=np.uint8)
    usable, cnt, cost = C.c_int32(0), C.c_int32(0), C.c_double(0.0)
    done = lib().orc_pnp_task(_ptr(cam), C.c_int32(int(shutter)), _ptr(sl), _ptr(op), _ptr(ip), C.c_int32(len(op)), _ptr(sub), C.c_int32(len(sub)),
                              C.c_int32(int(drop_coincident)), _ptr(init), C.c_int32(int(max_iter)), C.c_double(float(reprojection_error)), _ptr(poses), C.byref(usable), C.byref(cost),
                              C.byref(cnt), _ptr(mask))
    if not done:
        return None
    return dict(poses=poses, usable=bool(usable.value), final_cost=cost.value, num_inliers=cnt.value, mask=mask.astype(bool))


def default_options(**kw) -> OrcOptions:
    o = OrcOptions()
    lib().orc_default_options(C.byref(o))
    for k, v in kw.items():
        assert hasattr(o, k), k
        setattr(o, k, v)
    return o


def solve(prob, options: OrcOptions | None = None, trace_cap: int = 256):
    """ceres::Solve restated; prob's parameter arrays are overwritten.  -> (summary, [iteration records])"""
    d = desc(prob)
    o = options or default_options()
    s = OrcSummary()
    tr = (OrcIteration * trace_cap)()
    lib().orc_solve(C.byref(d), C.byref(o), C.byref(s), tr, C.c_int32(trace_cap))
    if d.ratio_free:
        prob.inter_frame_ratio = float(d.inter_frame_ratio)      # a free ratio block is a parameter: solved for in place
    n = min(s.num_iterations, trace_cap)
    return s, [tr[i] for i in range(n)]


# ---- scalar helpers for the known-answer tests -------------------------------------------------
def _v(x):
    return np.ascontiguousarray(x, dtype=np.float64)


def angle_axis_rotate(w, p):
    out = np.zeros(3); lib().orc_angle_axis_rotate(_ptr(_v(w)), _ptr(_v(p)), _ptr(out)); return out


def angle_axis_rotate_inplace(w, p):
    """Rotate p in place (mat_test.cc:52 uses the aliasing form)."""
    lib().orc_angle_axis_rotate(_ptr(_v(w)), _ptr(p), _ptr(p)); return p


def lerp_rotation(r0, r1, tau):
    out = np.zeros(3); lib().orc_lerp_rotation(_ptr(_v(r0)), _ptr(_v(r1)), C.c_double(tau), _ptr(out)); return out


def distort(cam, img):
    out = np.zeros(2); lib().orc_distort(_ptr(_v(cam)), _ptr(_v(img)), _ptr(out)); return out


def undistort(cam, img):
    out = np.zeros(2); ok = lib().orc_undistort(_ptr(_v(cam)), _ptr(_v(img)), _ptr(out)); return bool(ok), out


def w2c(pose, X):
    out = np.zeros(3); lib().orc_w2c(_ptr(_v(pose)), _ptr(_v(X)), _ptr(out)); return out


def c2w(pose, pt):
    out = np.zeros(3); lib().orc_c2w(_ptr(_v(pose)), _ptr(_v(pt)), _ptr(out)); return out


def w2i(cam, pose, X, validate=True):
    out = np.zeros(2); ok = lib().orc_w2i(_ptr(_v(cam)), _ptr(_v(pose)), _ptr(_v(X)), _ptr(out), C.c_int32(int(validate))); return bool(ok), out


def direction_world(pose, X):
    out = np.zeros(3); ok = lib().orc_direction_world(_ptr(_v(pose)), _ptr(_v(X)), _ptr(out)); return bool(ok), out


def c2direction(pose, pt):
    out = np.zeros(3); ok = lib().orc_c2direction(_ptr(_v(pose)), _ptr(_v(pt)), _ptr(out)); return bool(ok), out


def direction_pixel(cam, pose, xy):
    out = np.zeros(3); ok = lib().orc_direction_pixel(_ptr(_v(cam)), _ptr(_v(pose)), _ptr(_v(xy)), _ptr(out)); return bool(ok), out


def ray_intersect(p2, d1, d2):
    out = np.zeros(3); ok = lib().orc_ray_intersect(_ptr(_v(p2)), _ptr(_v(d1)), _ptr(_v(d2)), _ptr(out)); return bool(ok), out


def triangulate(c1, d1, c2, d2):
    out = np.zeros(3); ok = lib().orc_triangulate(_ptr(_v(c1)), _ptr(_v(d1)), _ptr(_v(c2)), _ptr(_v(d2)), _ptr(out)); return bool(ok), out


def validate(cam, pose, xy, X, thr):
    return bool(lib().orc_validate(_ptr(_v(cam)), _ptr(_v(pose)), _ptr(_v(xy)), _ptr(_v(X)), C.c_double(thr)))


def ray_dist(cam, pose, obs, cam2, pose2, obs2):
    out = np.zeros(3)
    ok = lib().orc_ray_dist(_ptr(_v(cam)), _ptr(_v(pose)), _ptr(_v(obs)), _ptr(_v(cam2)), _ptr(_v(pose2)), _ptr(_v(obs2)), _ptr(out))
    return bool(ok), out


def norm3(v):
    return lib().orc_norm3(_ptr(_v(v)))


def interpolate_rs(p0, p1, shutter, scan, obs, interp_rotation=True):
    out = np.zeros(6); sc = np.asarray(scan, dtype=np.int32)
    lib().orc_interpolate_rs(_ptr(_v(p0)), _ptr(_v(p1)), C.c_int32(shutter), _ptr(sc), _ptr(_v(obs)), C.c_int32(int(interp_rotation)), _ptr(out))
    return out


def huber(a, s):
    out = np.zeros(3); lib().orc_huber(C.c_double(a), C.c_double(s), _ptr(out)); return out


def reproject(cam, poses, shutter, scan, X, sq_threshold, interp_rotation=True):
    poses = _v(poses).reshape(-1, 6); out = np.zeros(2); sc = np.asarray(scan, dtype=np.int32)
    ok = lib().orc_reproject(_ptr(_v(cam)), _ptr(poses), C.c_int32(len(poses)), C.c_int32(shutter), _ptr(sc),
                             C.c_int32(int(interp_rotation)), _ptr(_v(X)), C.c_double(sq_threshold), _ptr(out))
    return bool(ok), out


def validate_obs(cam, poses, shutter, scan, X, obs, sq_threshold, min_dist, interp_rotation=True):
    poses = _v(poses).reshape(-1, 6); sc = np.asarray(scan, dtype=np.int32)
    return bool(lib().orc_validate_obs(_ptr(_v(cam)), _ptr(poses), C.c_int32(len(poses)), C.c_int32(shutter), _ptr(sc),
                                       C.c_int32(int(interp_rotation)), _ptr(_v(X)), _ptr(_v(obs)), C.c_double(sq_threshold), C.c_double(min_dist)))


def scanline_pose_index(nposes, shutter, obs):
    """struct/VideoSfM.cc:83-97: the pose a frame with more than two poses ("fullDoF") lends to an observation."""
    return int(lib().orc_scanline_pose_index(C.c_int32(nposes), C.c_int32(shutter), _ptr(_v(obs))))


def add_loop_pose_blocks(frame_poses, obs_frame, obs_xy, shutter):
    """Which pose blocks CeresHandler::Add hands to Ceres for the observations of a session whose frames carry 1, 2 or MORE poses
    (CeresHandler.h:245-286), as a list `blocks` of (frame, pose index or None) in the order Ceres first sees them and, per
    observation, the index into it.  A two-pose frame is one entry (frame, None): RsBundleAdjustment over both poses; a one-pose
    frame is (frame, 0); a frame with more contributes getPose's pick per observation — poses nobody picks never reach Ceres."""
    blocks, index, which = [], {}, []
    for f, xy in zip(obs_frame, obs_xy):
        n = len(frame_poses[f])
        key = (int(f), None) if n == 2 else (int(f), 0 if n == 1 else scanline_pose_index(n, shutter, xy))
        if key not in index:
            index[key] = len(blocks); blocks.append(key)
        which.append(index[key])
    return blocks, np.asarray(which, dtype=np.int32)


class CeresStyleEvaluator:
    """CPU baseline: one heap-allocated cost object per observation, evaluated through Dual<K> by an
    OpenMP parallel-for over residual blocks — how Ceres' evaluator runs rsba's functors
    (src/rsba/CeresHandler.h:408-415 sets num_threads = hardware_concurrency)."""

    def __init__(self, prob, threads: int = 0):
        self.prob, self.threads = prob, threads
        self.d = desc(prob)
        n, k = prob.num_observations, prob.jacobian_cols
        self.r = np.zeros((n, 2))
        self.J = np.zeros((n, 2 * k))   # per block: [block0 2xN0 row-major | block1 2xN1 | ...]
        lib().orc_evaluate_blocks_ceres_style(C.byref(self.d), None, None, C.c_int32(threads))   # build cost objects

    def run(self) -> int:
        return lib().orc_evaluate_blocks_ceres_style(C.byref(self.d), _ptr(self.r), _ptr(self.J), C.c_int32(self.threads))
