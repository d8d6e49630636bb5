"""ctypes binding of the C ABI in include/rsba_amd.h (rsba_amd/_lib/librsba_amd.so).

There is no CPU fallback: if the library is missing, fails to load, or no HIP device is present,
everything here raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from .problem import BAProblem

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RSBA_AMD_LIB") or os.path.join(_HERE, "_lib", "librsba_amd.so")   # override: kernel experiments (tools/)

EXPORTS = [
    "rsba_abi_version", "rsba_status_string", "rsba_last_error", "rsba_device_count", "rsba_create", "rsba_destroy",
    "rsba_set_stream", "rsba_upload_parameters", "rsba_download_parameters", "rsba_evaluate_device", "rsba_evaluate",
    "rsba_get_device_view", "rsba_time_evaluate", "rsba_default_solver_options", "rsba_solve", "rsba_normal_equations",
    "rsba_set_exchange", "rsba_get_block_structure", "rsba_set_block_structure",
    "rsba_validate_observations", "rsba_reproject", "rsba_pose_covariance", "rsba_set_motion_priors",
    "rsba_pnp_tasks", "rsba_pnp_inliers", "rsba_set_inter_frame_ratio_free", "rsba_get_inter_frame_ratio",
    "rsba_sync_block_structure", "rsba_rccl_get_unique_id", "rsba_rccl_comm_create", "rsba_rccl_comm_destroy", "rsba_set_exchange_rccl",
    "rsba_get_phase_times", "rsba_phase_name", "rsba_get_plan_stats", "rsba_validate_frame", "rsba_reproject_frame", "rsba_set_pose_priors", "rsba_set_global_shutter_frames", "rsba_release_host_scratch",
    "rsba_partition_points", "rsba_get_exchange_stats", "rsba_exchange_name", "rsba_rccl_describe",
]
NUM_EXCHANGES = 6
NUM_PHASES = 13


class RsbaError(RuntimeError):
    def __init__(self, status: int, detail: str):
        super().__init__(f"rsba_amd status {status}: {detail}")
        self.status = status


class ProblemDesc(C.Structure):
    _fields_ = [
        ("shutter", C.c_int32), ("scanlines", C.c_int32 * 2), ("interpolate_rotation", C.c_int32),
        ("calibrated", C.c_int32), ("poses_per_frame", C.c_int32),
        ("num_frames", C.c_int32), ("num_points", C.c_int32), ("num_intrinsics", C.c_int32),
        ("num_observations", C.c_int64),
        ("poses", C.c_void_p), ("points", C.c_void_p), ("intrinsics", C.c_void_p),
        ("frame_intrinsics", C.c_void_p), ("obs_xy", C.c_void_p), ("obs_frame", C.c_void_p),
        ("obs_point", C.c_void_p), ("pose_fixed_mask", C.c_void_p), ("point_constant", C.c_void_p),
        ("intrinsics_constant", C.c_void_p), ("huber_a", C.c_double),
    ]


class SolverOptions(C.Structure):
    _fields_ = [
        ("max_num_iterations", C.c_int32), ("jacobi_scaling", C.c_int32),
        ("max_num_consecutive_invalid_steps", C.c_int32), ("minimizer_progress_to_stdout", C.c_int32),
        ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double),
        ("min_trust_region_radius", C.c_double), ("min_relative_decrease", C.c_double),
        ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
        ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
        ("level_scheduled_cholesky", C.c_int32), ("profile_phases", C.c_int32),
    ]


class Iteration(C.Structure):
    _fields_ = [
        ("iteration", C.c_int32), ("step_is_valid", C.c_int32), ("step_is_successful", C.c_int32), ("reserved", C.c_int32),
        ("cost", C.c_double), ("cost_change", C.c_double), ("gradient_max_norm", C.c_double), ("step_norm", C.c_double),
        ("relative_decrease", C.c_double), ("trust_region_radius", C.c_double), ("model_cost_change", C.c_double),
    ]


class SolverSummary(C.Structure):
    _fields_ = [
        ("termination_type", C.c_int32), ("num_successful_steps", C.c_int32), ("num_unsuccessful_steps", C.c_int32),
        ("num_iterations", C.c_int32), ("num_residual_blocks", C.c_int32), ("num_residual_blocks_reduced", C.c_int32),
        ("num_parameters_reduced", C.c_int32), ("is_solution_usable", C.c_int32),
        ("initial_cost", C.c_double), ("final_cost", C.c_double), ("fixed_cost", C.c_double),
        ("total_time_s", C.c_double), ("residual_jacobian_time_s", C.c_double), ("linear_solver_time_s", C.c_double),
        ("num_dag_fallbacks", C.c_int32), ("reserved", C.c_int32),
    ]


class DeviceView(C.Structure):
    _fields_ = [
        ("residuals", C.c_void_p), ("jacobians", C.c_void_p), ("tile", C.c_int64), ("jacobian_cols", C.c_int32),
        ("reserved", C.c_int32), ("order_host", C.c_void_p), ("poses", C.c_void_p), ("points", C.c_void_p),
        ("intrinsics", C.c_void_p),
    ]


class PhaseTimes(C.Structure):
    _fields_ = [("ms", C.c_double * NUM_PHASES), ("calls", C.c_int32 * NUM_PHASES), ("reserved", C.c_int32)]


class PlanStats(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ("tiles", "factor_tiles", "levels", "tasks", "schur_entries", "schur_chunks", "schur_block_products",
                                         "cholesky_flops", "exchange_doubles", "schur_groups", "schur_mfma_issued", "schur_launches",
                                         "sharded_factorisation", "separator_tiles", "separator_factor_tiles", "local_tasks", "separator_tasks",
                                         "local_levels", "separator_levels", "schur_group_bytes", "schur_factored_groups", "device_loop_solves", "host_loop_solves")]


class ExchangeStats(C.Structure):
    _fields_ = [("calls", C.c_int64 * NUM_EXCHANGES), ("doubles", C.c_int64 * NUM_EXCHANGES), ("ms", C.c_double * NUM_EXCHANGES), ("rank", C.c_int32), ("world", C.c_int32)]


def build(force: bool = False) -> str:
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    subprocess.run(["make", "-C", src, "-j4"] + (["-B"] if force else []), check=True, capture_output=True)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RsbaError(-1, f"{LIB_PATH} is missing: run __graft_entry__.build() (no CPU fallback exists)")
        try:
            # torch ships its own libamdhip64 with the same SONAME; importing it first makes this
            # library share that runtime, so device pointers / streams are interchangeable with torch.
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is optional for the C ABI itself
            pass
        _lib = C.CDLL(LIB_PATH)
        _lib.rsba_status_string.restype = C.c_char_p
        _lib.rsba_last_error.restype = C.c_char_p
        _lib.rsba_destroy.restype = None
        _lib.rsba_release_host_scratch.restype = None
        _lib.rsba_default_solver_options.restype = None
        _lib.rsba_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        _lib.rsba_destroy.argtypes = [C.c_void_p]
        _lib.rsba_phase_name.restype = C.c_char_p
        _lib.rsba_rccl_comm_destroy.restype = None
        _lib.rsba_rccl_comm_destroy.argtypes = [C.c_void_p]
    return _lib


def _check(status: int):
    if status != 0:
        L = lib()
        raise RsbaError(status, f"{L.rsba_status_string(status).decode()} — {L.rsba_last_error().decode()}")


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def device_count() -> int:
    n = C.c_int32(0)
    _check(lib().rsba_device_count(C.byref(n)))
    return n.value


def make_desc(prob: BAProblem) -> ProblemDesc:
    prob.__post_init__()   # (arrays assigned after construction may have another dtype or layout: the pointers below must be what the C side reads; a no-op when they already are)
    d = ProblemDesc()
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
    return d


def partition_points(prob: BAProblem, world: int):
    """rsba_partition_points: owner[num_points] of a sharded solve whose reduced camera system can be factored where it is formed
    (host only; every rank computes the same answer from the whole problem) and the number of tile columns in the shared separators."""
    d = make_desc(prob)
    owner = np.zeros(prob.num_points, dtype=np.int32)
    ntop = C.c_int32(0)
    _check(lib().rsba_partition_points(C.byref(d), C.c_int32(world), _ptr(owner), C.byref(ntop)))
    return owner, int(ntop.value)


def release_host_scratch():
    """rsba_release_host_scratch: give the symbolic phase's host scratch (kept between handles) back to the allocator."""
    lib().rsba_release_host_scratch()


class DeviceProblem:
    """A problem resident in HBM (the ceres::Problem of CeresHandler, src/rsba/CeresHandler.h:78)."""

    def __init__(self, prob: BAProblem, device: int = 0):
        self.prob = prob
        self._desc = make_desc(prob)
        # the handle keeps these HOST POINTERS (rsba_solve writes the solved parameters back through them, as ceres::Solve writes into the
        # blocks it was given): the arrays stay alive with this object, and arrays assigned to the problem later are copied INTO them
        self._bound = (prob.poses, prob.points, prob.intrinsics)
        self._h = C.c_void_p()
        _check(lib().rsba_create(C.byref(self._desc), C.c_int32(device), C.byref(self._h)))
        if prob.prior_kind and prob.prior_frames is not None and len(prob.prior_frames):
            self.set_motion_priors(prob.prior_kind, prob.prior_scale, prob.inter_frame_ratio, prob.prior_frames)
            if getattr(prob, "ratio_free", False):
                _check(lib().rsba_set_inter_frame_ratio_free(self._h, C.c_int32(1)))
        if getattr(prob, "frame_global", None) is not None:      # one-pose frames inside a two-pose session (CeresHandler.h:266-285)
            fg = np.ascontiguousarray(prob.frame_global, dtype=np.uint8)
            assert fg.shape == (prob.num_frames,)
            self._frame_global = fg
            _check(lib().rsba_set_global_shutter_frames(self._h, _ptr(fg)))
        npp = 0 if prob.pose_prior_block is None else len(prob.pose_prior_block)
        if npp or prob.spherical_pose_block >= 0:
            _check(lib().rsba_set_pose_priors(self._h, C.c_double(float(prob.pose_prior_rotation)), C.c_double(float(prob.pose_prior_position)),
                                              _ptr(prob.pose_prior_block) if npp else None, _ptr(prob.pose_prior_values) if npp else None, C.c_int32(npp),
                                              C.c_int32(int(prob.spherical_pose_block))))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().rsba_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_motion_priors(self, kind: int, scale: float, inter_frame_ratio: float, frames):
        """Frame-to-frame motion priors with a constant interFrameRatio (CeresHandler.h:147-185; rsba_amd.h)."""
        fr = np.ascontiguousarray(frames, dtype=np.int32)
        _check(lib().rsba_set_motion_priors(self._h, C.c_int32(int(kind)), C.c_double(float(scale)), C.c_double(float(inter_frame_ratio)),
                                            fr.ctypes.data_as(C.c_void_p), C.c_int32(len(fr))))

    def inter_frame_ratio(self) -> float:
        """current interFrameRatio of the motion priors (the solved value when it is a free parameter)"""
        v = C.c_double(0.0)
        _check(lib().rsba_get_inter_frame_ratio(self._h, C.byref(v)))
        return v.value

    def set_stream(self, raw_stream: int | None):
        _check(lib().rsba_set_stream(self._h, C.c_void_p(raw_stream or 0)))

    def _rebind(self, take_values: bool):
        """The problem's parameter arrays must BE the arrays bound at rsba_create (the handle holds their addresses).  An array assigned to
        the problem since then is copied into the bound one (take_values) and the attribute is pointed back at it; a different shape is an error."""
        p = self.prob
        for name, bound in zip(("poses", "points", "intrinsics"), self._bound):
            cur = getattr(p, name)
            if cur is bound:
                continue
            cur = np.asarray(cur, dtype=np.float64)
            # the same SHAPE, or the flat form of it: anything else with as many elements ((F*2, 6), (6, 2, F) ...) would be re-read in another element order
            if cur.shape != bound.shape and cur.shape != (bound.size,):
                raise ValueError(f"{name}: {cur.shape} cannot replace the {bound.shape} array this handle was created with")
            if take_values:
                np.copyto(bound, cur.reshape(bound.shape))
            else:
                import warnings
                warnings.warn(f"{name}: the array assigned to the problem since the handle was created is replaced by the handle's own (results are written there)", stacklevel=3)
            setattr(p, name, bound)

    def upload_parameters(self):
        self._rebind(take_values=True)
        p = self.prob
        _check(lib().rsba_upload_parameters(self._h, _ptr(p.poses), _ptr(p.points), _ptr(p.intrinsics)))

    def download_parameters(self):
        self._rebind(take_values=False)
        p = self.prob
        _check(lib().rsba_download_parameters(self._h, _ptr(p.poses), _ptr(p.points), _ptr(p.intrinsics)))

    def evaluate_device(self, with_jacobians: bool = True):
        """One residual(+Jacobian) evaluation of every observation; asynchronous, results stay in HBM."""
        _check(lib().rsba_evaluate_device(self._h, C.c_int32(int(with_jacobians))))

    def time_evaluate(self, with_jacobians: bool = True, warmup: int = 2, iters: int = 10) -> float:
        ms = C.c_double(0.0)
        _check(lib().rsba_time_evaluate(self._h, C.c_int32(int(with_jacobians)), C.c_int32(warmup), C.c_int32(iters), C.byref(ms)))
        return ms.value

    def evaluate(self, residuals: bool = True, jacobians: bool = True, gradient: bool = False, allow_failed: bool = True):
        """Problem::Evaluate -> dict(cost, residuals [N,2], jacobians [N,2,K], gradient{...}, num_failed)"""
        p = self.prob
        n, k = p.num_observations, p.jacobian_cols
        cost = C.c_double(0.0)
        nf = C.c_int64(0)
        r = np.zeros((n, 2)) if residuals else None
        J = np.zeros((n, 2, k)) if jacobians else None
        npose = p.num_frames * p.poses_per_frame * 6
        g = np.zeros(npose + 3 * p.num_points + 9 * p.num_intrinsics) if gradient else None
        st = lib().rsba_evaluate(self._h, C.byref(cost), _ptr(r), _ptr(J), _ptr(g), C.byref(nf))
        if st != 0 and not (allow_failed and st == 4):
            _check(st)
        out = dict(cost=cost.value, residuals=r, jacobians=J, num_failed=nf.value)
        if gradient:
            out["gradient"] = dict(poses=g[:npose].reshape(p.poses.shape), points=g[npose:npose + 3 * p.num_points].reshape(-1, 3),
                                   intrinsics=g[npose + 3 * p.num_points:].reshape(-1, 9))
        return out

    def normal_equations(self):
        """-> U [F,CD,CD], gc [F,CD], V [M,3,3], gp [M,3] (loss-corrected, masked, undamped)"""
        p = self.prob
        cd = 6 * p.poses_per_frame
        U = np.zeros((p.num_frames, cd, cd)); gc = np.zeros((p.num_frames, cd))
        V = np.zeros((p.num_points, 3, 3)); gp = np.zeros((p.num_points, 3))
        _check(lib().rsba_normal_equations(self._h, _ptr(U), _ptr(gc), _ptr(V), _ptr(gp)))
        return U, gc, V, gp

    def validate_observations(self, sq_threshold: float, min_distance: float = 0.0) -> np.ndarray:
        """vision::sfm::validate for every observation (struct/VideoSfM.cc:159-169) -> bool [N]"""
        out = getattr(self, "_valid_buf", None)
        if out is None or len(out) != self.prob.num_observations:
            out = self._valid_buf = np.ones(self.prob.num_observations, dtype=np.uint8)   # touched once: no page faults inside the timed copy
        _check(lib().rsba_validate_observations(self._h, C.c_double(sq_threshold), C.c_double(min_distance), _ptr(out)))
        return out.view(np.bool_).copy()

    def reproject(self, frames, points):
        """vision::sfm::reproject (struct/VideoSfM.cc:139-155) for (frame, point) pairs -> xy [n,2], ok [n]"""
        fr = np.ascontiguousarray(frames, dtype=np.int32); pt = np.ascontiguousarray(points, dtype=np.int32)
        xy = np.zeros((len(fr), 2)); ok = np.zeros(len(fr), dtype=np.uint8)
        _check(lib().rsba_reproject(self._h, _ptr(fr), _ptr(pt), C.c_int64(len(fr)), _ptr(xy), _ptr(ok)))
        return xy, ok.astype(bool)

    def pose_covariance(self, frame: int) -> np.ndarray:
        """ceres::Covariance blocks of one frame (VideoSfMHandler.cc:602-621) -> [CD, CD]"""
        cd = 6 * self.prob.poses_per_frame
        out = np.zeros((cd, cd))
        try:
            _check(lib().rsba_pose_covariance(self._h, C.c_int32(frame), _ptr(out)))
        finally:
            self._exchanged()
        return out

    def set_exchange_rccl(self, comm, rank: int, world: int):
        """Native transport of the multi-GPU exchange: ncclAllReduce on the solver's stream (rsba_amd.h)."""
        _check(lib().rsba_set_exchange_rccl(self._h, C.c_void_p(comm), C.c_int32(rank), C.c_int32(world)))

    def sync_block_structure(self):
        """All ranks: union of the co-visibility structures over the installed exchange (before the first solve)."""
        try:
            _check(lib().rsba_sync_block_structure(self._h))
        finally:
            self._exchanged()

    def phase_times(self) -> dict:
        """HIP-event time per phase of the last solve run with profile_phases=1 -> {name: (ms, calls)}"""
        t = PhaseTimes()
        _check(lib().rsba_get_phase_times(self._h, C.byref(t)))
        return {lib().rsba_phase_name(C.c_int32(p)).decode(): (t.ms[p], t.calls[p]) for p in range(NUM_PHASES)}

    def exchange_stats(self) -> dict:
        """Per kind of collective: calls and bytes since the handle was created, HIP-event ms of the last solve with profile_phases."""
        L = lib()
        st = ExchangeStats()
        _check(L.rsba_get_exchange_stats(self._h, C.byref(st)))
        L.rsba_exchange_name.restype = C.c_char_p
        return {"rank": st.rank, "world": st.world,
                "collectives": {L.rsba_exchange_name(C.c_int32(k)).decode(): {"calls": int(st.calls[k]), "bytes": int(st.doubles[k]) * 8, "ms": float(st.ms[k])} for k in range(NUM_EXCHANGES)}}

    def plan_stats(self) -> dict:
        st = PlanStats()
        try:
            _check(lib().rsba_get_plan_stats(self._h, C.byref(st)))
        finally:
            self._exchanged()
        return {k: int(getattr(st, k)) for k, _ in PlanStats._fields_}

    def _exchanged(self):
        """After an API call that may have run collectives: the hook of a transport that makes ranks sharing a GPU take turns
        (rsba_amd.distributed.attach(serialize=True)); nothing otherwise."""
        hook = getattr(self, "_after_exchange", None)
        if hook is not None:
            hook()

    def device_view(self) -> DeviceView:
        v = DeviceView()
        _check(lib().rsba_get_device_view(self._h, C.byref(v)))
        return v

    def solve(self, options: SolverOptions | None = None, trace_cap: int = 256):
        """ceres::Solve; the BAProblem's parameter arrays are overwritten.  -> (summary, [iterations])"""
        o = options or default_options()
        s = SolverSummary()
        tr = (Iteration * trace_cap)()
        try:
            st = lib().rsba_solve(self._h, C.byref(o), C.byref(s), tr, C.c_int32(trace_cap))
        finally:
            self._exchanged()
        _check(st)
        self._rebind(take_values=False)   # (the solved parameters are in the arrays bound at create: the problem's attributes are those arrays again)
        if getattr(self.prob, "ratio_free", False) and self.prob.prior_kind:
            self.prob.inter_frame_ratio = self.inter_frame_ratio()       # a free ratio block is solved for, like every parameter
        return s, [tr[i] for i in range(min(s.num_iterations, trace_cap))]


def validate_frame(cam, poses, shutter, scanlines, points, obs_xy, sq_threshold, min_distance=0.0, interpolate_rotation=True, device=0):
    """vision::sfm::validate for the (point, observation) pairs of one frame, no handle (rsba_amd.h: rsba_validate_frame)"""
    cam = np.ascontiguousarray(cam, dtype=np.float64); ps = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 6)
    sl = np.ascontiguousarray(scanlines, dtype=np.int32); X = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    xy = np.ascontiguousarray(obs_xy, dtype=np.float64).reshape(-1, 2)
    out = np.zeros(len(X), dtype=np.uint8)
    _check(lib().rsba_validate_frame(C.c_int32(device), _ptr(cam), _ptr(ps), C.c_int32(len(ps)), C.c_int32(int(shutter)), _ptr(sl), C.c_int32(int(interpolate_rotation)),
                                     _ptr(X), _ptr(xy), C.c_int64(len(X)), C.c_double(sq_threshold), C.c_double(min_distance), _ptr(out)))
    return out.astype(bool)


def reproject_frame(cam, poses, shutter, scanlines, points, interpolate_rotation=True, device=0):
    """vision::sfm::reproject of points into one frame, no handle (rsba_amd.h: rsba_reproject_frame) -> xy [n,2], ok [n]"""
    cam = np.ascontiguousarray(cam, dtype=np.float64); ps = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 6)
    sl = np.ascontiguousarray(scanlines, dtype=np.int32); X = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    xy = np.zeros((len(X), 2)); ok = np.zeros(len(X), dtype=np.uint8)
    _check(lib().rsba_reproject_frame(C.c_int32(device), _ptr(cam), _ptr(ps), C.c_int32(len(ps)), C.c_int32(int(shutter)), _ptr(sl), C.c_int32(int(interpolate_rotation)),
                                      _ptr(X), C.c_int64(len(X)), _ptr(xy), _ptr(ok)))
    return xy, ok.astype(bool)


def rccl_unique_id() -> bytes:
    """== ncclGetUniqueId (one rank; hand the bytes to the others)"""
    buf = C.create_string_buffer(128)
    _check(lib().rsba_rccl_get_unique_id(buf))
    return buf.raw


def rccl_comm_create(uid: bytes, rank: int, world: int, device: int) -> int:
    """== ncclCommInitRank; collective.  -> ncclComm_t as an integer"""
    comm = C.c_void_p()
    _check(lib().rsba_rccl_comm_create(C.c_char_p(uid), C.c_int32(rank), C.c_int32(world), C.c_int32(device), C.byref(comm)))
    return comm.value


def rccl_describe(comm: int | None = None) -> dict:
    """ncclGetVersion / ncclCommCount / ncclCommUserRank as the library's own RCCL sees them (-1 where a symbol is missing)."""
    v, n, r = C.c_int32(-1), C.c_int32(-1), C.c_int32(-1)
    _check(lib().rsba_rccl_describe(C.c_void_p(comm) if comm else None, C.byref(v), C.byref(n), C.byref(r)))
    return {"rccl_version": int(v.value), "comm_ranks": int(n.value), "comm_rank": int(r.value)}


def rccl_comm_destroy(comm: int):
    lib().rsba_rccl_comm_destroy(C.c_void_p(comm))


def pnp_tasks(cam, shutter, scanlines, object_points, image_points, subsets, init_poses, max_num_iterations=10,
              reprojection_error=8.0, device=0, drop_coincident=True):
    """The RANSAC hypotheses of solveRsPnPRansac, batched (rsba_amd.h: rsba_pnp_tasks).
    object_points [n,3] float32, image_points [n,2] float32, subsets [H,m] int32, init_poses [12] or [H,12].
    -> dict(poses [H,2,6], status [H], final_cost [H], num_inliers [H])"""
    cam = np.ascontiguousarray(cam, dtype=np.float64); sl = np.ascontiguousarray(scanlines, dtype=np.int32)
    op = np.ascontiguousarray(object_points, dtype=np.float32).reshape(-1, 3); ip = np.ascontiguousarray(image_points, dtype=np.float32).reshape(-1, 2)
    sub = np.ascontiguousarray(subsets, dtype=np.int32); H, m = sub.shape
    init = np.ascontiguousarray(init_poses, dtype=np.float64).reshape(-1, 12)
    assert len(init) in (1, H)
    poses = np.zeros((H, 2, 6)); status = np.zeros(H, dtype=np.uint8); cost = np.zeros(H); inl = np.zeros(H, dtype=np.int32)
    _check(lib().rsba_pnp_tasks(C.c_int32(device), _ptr(cam), C.c_int32(int(shutter)), _ptr(sl), _ptr(op), _ptr(ip), C.c_int32(len(op)),
                                _ptr(sub), C.c_int32(m), C.c_int32(H), _ptr(init), C.c_int32(12 if len(init) == H else 0),
                                C.c_int32(int(max_num_iterations)), C.c_int32(int(drop_coincident)), C.c_float(float(reprojection_error)), _ptr(poses), _ptr(status),
                                _ptr(cost), _ptr(inl)))
    return dict(poses=poses, status=status, final_cost=cost, num_inliers=inl)


def pnp_inliers(cam, shutter, scanlines, object_points, image_points, poses, reprojection_error=8.0, device=0):
    """Inlier flags [n] of one pose pair (rsba_amd.h: rsba_pnp_inliers)."""
    cam = np.ascontiguousarray(cam, dtype=np.float64); sl = np.ascontiguousarray(scanlines, dtype=np.int32)
    op = np.ascontiguousarray(object_points, dtype=np.float32).reshape(-1, 3); ip = np.ascontiguousarray(image_points, dtype=np.float32).reshape(-1, 2)
    ps = np.ascontiguousarray(poses, dtype=np.float64).reshape(12)
    mask = np.zeros(len(op), dtype=np.uint8)
    _check(lib().rsba_pnp_inliers(C.c_int32(device), _ptr(cam), C.c_int32(int(shutter)), _ptr(sl), _ptr(op), _ptr(ip), C.c_int32(len(op)),
                                  _ptr(ps), C.c_float(float(reprojection_error)), _ptr(mask)))
    return mask.astype(bool)


def default_options(**kw) -> SolverOptions:
    o = SolverOptions()
    lib().rsba_default_solver_options(C.byref(o))
    for k, v in kw.items():
        assert hasattr(o, k), k
        setattr(o, k, v)
    return o
