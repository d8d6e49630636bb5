"""Frame-to-frame motion priors (constant interFrameRatio; CeresHandler.h:147-185,
video_bundler_rs_inter.h:55-173) on the device against the oracle's restatement and the independent
numpy/scipy minima in tests/golden/prior_solves.json.  Tolerances as in test_gpu_solve.py."""
import numpy as np
import pytest

from helpers import load_golden, problem_from_solve_case
from test_gpu_solve import check_normal_equations, compare_solves, small_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from rsba_amd import capi
    assert capi.device_count() >= 1
    return capi


def with_priors(p, kind, scale, ratio, frames=None):
    p.prior_kind, p.prior_scale, p.inter_frame_ratio = kind, scale, ratio
    p.prior_frames = np.arange(1, p.num_frames, dtype=np.int32) if frames is None else np.asarray(frames, dtype=np.int32)
    return p


@pytest.mark.parametrize("kind,scale,ratio,huber", [(1, 6.0, 0.8, 0.0), (2, 25.0, 1.25, 2.0), (1, 3.0, 0.0, 0.0), (1, 40.0, 2.5, 1.5)])
def test_cost_gradient_and_diagonal_blocks(capi, oracle, kind, scale, ratio, huber):
    p = with_priors(small_scene(outlier_ratio=0.1 if huber else 0.0), kind, scale, ratio)
    p.huber_a = huber
    check_normal_equations(capi, oracle, p)


def test_priors_on_some_frames_and_next_to_constant_frames(capi, oracle):
    from rsba_amd.problem import apply_gauge_masks
    p = small_scene()
    apply_gauge_masks(p, fix_first_n_cameras=3)           # frames 0..2 constant: prior 2 is all-constant (fixed cost), prior 3 half
    with_priors(p, 1, 8.0, 0.7, frames=[1, 2, 3, 7, 8, 15])
    check_normal_equations(capi, oracle, p)
    compare_solves(capi, oracle, p, iters=15)


@pytest.mark.parametrize("idx", [0, 1])
def test_golden_prior_solves(capi, oracle, idx):
    c = load_golden("prior_solves.json")[idx]
    p = problem_from_solve_case(c)
    with capi.DeviceProblem(p.copy()) as dp:
        out = dp.evaluate(residuals=False, jacobians=False)
    assert abs(out["cost"] - c["expected"]["initial_cost"]) <= 1e-9 * out["cost"]
    s, s_ref, p_dev, p_cpu = compare_solves(capi, oracle, p, iters=200, tight=True, final_tol=1e-8)
    assert s.num_residual_blocks == p.num_observations + len(c["prior_frames"])
    assert abs(s.final_cost - c["expected"]["final_cost"]) <= 1e-8 * c["expected"]["final_cost"]
    ptol = 1e-3 if c["huber_a"] > 0 else 1e-5
    assert np.max(np.abs(p_dev.poses - np.array(c["expected"]["poses"]))) <= ptol
    assert np.max(np.abs(p_dev.points - np.array(c["expected"]["points"]))) <= 10 * ptol


@pytest.mark.parametrize("kind,scale,ratio,huber,shared_intrinsics", [(1, 10.0, 0.8, 0.0, False), (2, 30.0, 1.3, 2.0, False), (1, 10.0, 0.9, 2.0, True)])
def test_solve_trajectory_matches_oracle(capi, oracle, kind, scale, ratio, huber, shared_intrinsics):
    p = with_priors(small_scene(frames=30, points=1500, outlier_ratio=0.05 if huber else 0.0), kind, scale, ratio)
    p.huber_a = huber
    if shared_intrinsics:
        p.calibrated = False
    s, s_ref, p_dev, p_cpu = compare_solves(capi, oracle, p, iters=30)
    assert s.num_residual_blocks == s_ref.num_residual_blocks == p.num_observations + p.num_frames - 1
    # both stop at Ceres' default tolerances: the last step's size bounds how far apart they may end (as in test_gpu_solve.py)
    assert np.max(np.abs(p_dev.poses - p_cpu.poses)) <= 1e-5


def test_level_schedule_and_dag_agree_with_priors(capi):
    p = with_priors(small_scene(frames=60, points=3000), 2, 20.0, 1.1)
    res = []
    for levels in (0, 1):
        q = p.copy()
        with capi.DeviceProblem(q) as dp:
            s, _ = dp.solve(capi.default_options(max_num_iterations=8, level_scheduled_cholesky=levels))
        res.append((s.final_cost, q.poses.copy()))
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1])


def test_invalid_ratio_fails_the_evaluation(capi):
    p = with_priors(small_scene(), 2, 5.0, 0.0)        # RsConstAccelerationPrior returns ratio >= _EPS
    with capi.DeviceProblem(p) as dp:
        with pytest.raises(capi.RsbaError):
            dp.solve(capi.default_options(max_num_iterations=3))


def test_priors_need_two_poses_per_frame(capi):
    p = small_scene(rolling=False)
    with capi.DeviceProblem(p) as dp:
        with pytest.raises(capi.RsbaError):
            dp.set_motion_priors(1, 1.0, 0.8, [1, 2])


def test_covariance_sees_the_priors(capi, oracle):
    p = with_priors(small_scene(), 1, 50.0, 0.8)
    cov_ref, ok = oracle.pose_covariance(p, 5)
    assert ok
    with capi.DeviceProblem(p) as dp:
        cov = dp.pose_covariance(5)
    assert np.max(np.abs(cov - cov_ref)) <= 1e-7 * np.max(np.abs(cov_ref))
    q = small_scene()
    cov0, ok = oracle.pose_covariance(q, 5)
    assert np.trace(cov_ref) < np.trace(cov0)          # the priors add information


def test_full_size_cost_is_additive_with_an_independent_numpy_model(capi):
    """BASELINE config C4 (1k frames, 2M observations): Problem::Evaluate with a motion prior on every frame equals the
    plain cost plus the priors' cost from an independent numpy model of their physical meaning (tests/golden/make_golden.py)."""
    import importlib.util, os
    from rsba_amd.scene import make_config
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
    p = make_config("C4").problem
    with capi.DeviceProblem(p) as dp:
        plain = dp.evaluate(residuals=False, jacobians=False)["cost"]
    for kind, scale, ratio, huber in ((1, 6.0, 0.8, 0.0), (2, 25.0, 1.25, 0.05)):
        q = with_priors(p.copy(), kind, scale, ratio)
        q.huber_a = huber
        pr = mg.np_prior_residuals(q.poses, kind, scale, ratio, list(range(1, q.num_frames)))
        s = np.sum(pr * pr, axis=1)
        rho = np.where(s <= huber * huber, s, 2 * huber * np.sqrt(s) - huber * huber) if huber > 0 else s
        with capi.DeviceProblem(q) as dq:
            got = dq.evaluate(residuals=False, jacobians=False)["cost"]
        if huber > 0:   # the observation blocks share the loss: their part under the same loss, from the device itself
            q0 = q.copy(); q0.prior_kind = 0; q0.prior_frames = None
            with capi.DeviceProblem(q0) as d0:
                base = d0.evaluate(residuals=False, jacobians=False)["cost"]
            assert np.count_nonzero(s > huber * huber) > 0
        else:
            base = plain
        assert abs(got - (base + 0.5 * float(np.sum(rho)))) <= 1e-12 * got


# ---- the free, lower-bounded interFrameRatio (the reference's default: option left at 1) ----
@pytest.mark.parametrize("idx", [0, 1])
def test_free_ratio_reaches_the_independent_bounded_minimum(capi, oracle, idx):
    c = load_golden("free_ratio_solves.json")[idx]
    p = problem_from_solve_case(c)
    s, s_ref, p_dev, p_cpu = compare_solves(capi, oracle, p, iters=300, tight=True, final_tol=1e-8)
    assert abs(s.final_cost - c["expected"]["final_cost"]) <= 1e-8 * c["expected"]["final_cost"]
    assert abs(p_dev.inter_frame_ratio - c["expected"]["ratio"]) <= 1e-5 and abs(p_dev.inter_frame_ratio - p_cpu.inter_frame_ratio) <= 1e-7
    assert np.max(np.abs(p_dev.poses - np.array(c["expected"]["poses"]))) <= 1e-5


@pytest.mark.parametrize("kind,huber,shared_intrinsics", [(1, 0.0, False), (2, 2.0, False), (1, 2.0, True)])
def test_free_ratio_trajectory_matches_oracle(capi, oracle, kind, huber, shared_intrinsics):
    p = with_priors(small_scene(frames=30, points=1500, outlier_ratio=0.05 if huber else 0.0), kind, 10.0 if kind == 1 else 30.0, 1.0)
    p.ratio_free = True
    p.huber_a = huber
    if shared_intrinsics:
        p.calibrated = False
    s, s_ref, p_dev, p_cpu = compare_solves(capi, oracle, p, iters=30)
    assert s.num_parameters_reduced == s_ref.num_parameters_reduced
    assert abs(p_dev.inter_frame_ratio - p_cpu.inter_frame_ratio) <= 1e-5 and abs(p_dev.inter_frame_ratio - 1.0) > 1e-3
    assert np.max(np.abs(p_dev.poses - p_cpu.poses)) <= 1e-5


def test_free_ratio_level_schedule_equals_dag(capi):
    p = with_priors(small_scene(frames=60, points=3000), 1, 20.0, 1.0)
    p.ratio_free = True
    res = []
    for levels in (0, 1):
        q = p.copy()
        with capi.DeviceProblem(q) as dp:
            s, _ = dp.solve(capi.default_options(max_num_iterations=8, level_scheduled_cholesky=levels))
        res.append((s.final_cost, q.poses.copy(), q.inter_frame_ratio))
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1]) and res[0][2] == res[1][2] != 1.0
