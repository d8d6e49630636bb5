"""SURVEY §8f row f1, second half: the per-pose prior blocks of CeresHandler::Add — GoodPosePrior (CeresHandler.h:52-73,
188-204; its priorPoses blocks are FREE parameter blocks in the reference) and SphericalPrior (:36-50, 127-130) — on the
device against the oracle's Dual-number restatement and against what is known independently: the priors are linear, so
their cost is a closed form; and because the priorPoses blocks are free, the minimum of a problem with GoodPosePrior blocks
is the minimum without them (tests/golden/tiny_solves.json: scipy) with every prior sitting on its pose."""
import numpy as np
import pytest

from helpers import load_golden, problem_from_solve_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from rsba_amd import capi as C
    return C


def scene(frames=8, points=300, seed=71, huber=0.0):
    from rsba_amd.problem import apply_gauge_masks
    from rsba_amd.scene import make_scene
    p = make_scene(frames, points, rolling=True, seed=seed, outlier_ratio=0.04 if huber else 0.0).problem
    p.huber_a = huber
    apply_gauge_masks(p, fix_first_n_cameras=1)
    return p


def with_good_pose_priors(p, rotation=3.0, position=5.0, sigma=0.01, seed=5, blocks=None):
    rng = np.random.default_rng(seed)
    q = p.copy()
    q.pose_prior_block = np.arange(2, 2 * p.num_frames, dtype=np.int32) if blocks is None else np.asarray(blocks, dtype=np.int32)
    q.pose_prior_values = p.poses.reshape(-1, 6)[q.pose_prior_block] + rng.normal(0, sigma, (len(q.pose_prior_block), 6))
    q.pose_prior_rotation, q.pose_prior_position = rotation, position
    return q


@pytest.mark.parametrize("huber", [0.0, 2.0])
def test_good_pose_prior_cost_and_gradient(capi, oracle, huber):
    p = scene(huber=huber)
    q = with_good_pose_priors(p)
    with capi.DeviceProblem(p) as d0, capi.DeviceProblem(q) as d1:
        a, b = d0.evaluate(residuals=False, jacobians=False, gradient=True), d1.evaluate(residuals=False, jacobians=False, gradient=True)
    r = (q.pose_prior_values - q.poses.reshape(-1, 6)[q.pose_prior_block]) * np.array([3.0, 3, 3, 5, 5, 5])
    assert abs((b["cost"] - a["cost"]) - 0.5 * np.sum(r * r)) <= 1e-9 * 0.5 * np.sum(r * r)          # no loss function on the prior blocks
    g = np.zeros((2 * p.num_frames, 6)); g[q.pose_prior_block] = -r * np.array([3.0, 3, 3, 5, 5, 5])   # d/dpose of 1/2 |W (prior - pose)|^2
    assert np.max(np.abs((b["gradient"]["poses"] - a["gradient"]["poses"]).reshape(-1, 6) - g)) <= 1e-9 * np.abs(g).max()
    ok, cost_ref, g_ref = oracle.evaluate(q)
    assert ok and abs(b["cost"] - cost_ref) <= 1e-12 * cost_ref
    assert np.max(np.abs(b["gradient"]["poses"] - g_ref["poses"])) <= 1e-11 * np.abs(g_ref["poses"]).max()
    assert b["num_failed"] == 0


@pytest.mark.parametrize("huber,subset", [(0.0, False), (2.0, True)])
def test_good_pose_prior_solve_matches_the_oracle(capi, oracle, huber, subset):
    p = scene(frames=10, points=400, seed=72, huber=huber)
    q = with_good_pose_priors(p, blocks=[2, 3, 7, 8, 13, 19] if subset else None)
    qd, qc = q.copy(), q.copy()
    with capi.DeviceProblem(qd) as dp:
        s, tr = dp.solve(capi.default_options(max_num_iterations=25))
    s_ref, tr_ref = oracle.solve(qc, oracle.default_options(max_num_iterations=25))
    assert s.num_residual_blocks == s_ref.num_residual_blocks and s.num_residual_blocks_reduced == s_ref.num_residual_blocks_reduced
    assert s.num_parameters_reduced == s_ref.num_parameters_reduced
    assert abs(s.initial_cost - s_ref.initial_cost) <= 1e-12 * s_ref.initial_cost
    for a, b in list(zip(tr, tr_ref))[:5]:
        assert a.step_is_successful == b.step_is_successful and abs(a.cost - b.cost) <= 1e-9 * b.cost, (a.iteration, a.cost, b.cost)
        assert abs(a.trust_region_radius - b.trust_region_radius) <= 1e-6 * b.trust_region_radius
        if a.iteration:
            assert abs(a.step_norm - b.step_norm) <= 1e-6 * b.step_norm and abs(a.model_cost_change - b.model_cost_change) <= 1e-6 * abs(b.model_cost_change)
            assert abs(a.gradient_max_norm - b.gradient_max_norm) <= 1e-6 * b.gradient_max_norm
    assert s.termination_type == s_ref.termination_type and abs(s.final_cost - s_ref.final_cost) <= 1e-6 * s_ref.final_cost
    assert np.max(np.abs(qd.poses - qc.poses)) <= 1e-5 and np.max(np.abs(qd.pose_prior_values - qc.pose_prior_values)) <= 1e-5
    assert not np.array_equal(qd.pose_prior_values, q.pose_prior_values)            # the priorPoses blocks are solved for, in place


@pytest.mark.parametrize("idx", [0, 2])
def test_free_prior_poses_leave_the_minimum_where_it_was(capi, idx):
    """Independent pin: scipy's minimum of the tiny scenes (computed without any prior).  With GoodPosePrior blocks whose
    priorPoses are free, the minimum is the same and every prior ends on its pose."""
    c = load_golden("tiny_solves.json")[idx]
    p = problem_from_solve_case(c)
    q = with_good_pose_priors(p, rotation=2.0, position=4.0, sigma=0.02)
    with capi.DeviceProblem(q) as dp:
        s, _ = dp.solve(capi.default_options(max_num_iterations=300, function_tolerance=1e-14, parameter_tolerance=1e-14, gradient_tolerance=1e-12))
    assert abs(s.final_cost - c["expected"]["final_cost"]) <= 1e-7 * c["expected"]["final_cost"]
    assert np.max(np.abs(q.pose_prior_values - q.poses.reshape(-1, 6)[q.pose_prior_block])) <= 1e-5
    assert np.max(np.abs(q.poses - np.array(c["expected"]["poses"]))) <= (1e-3 if c["huber_a"] > 0 else 1e-4)


def test_good_pose_prior_functor_failure(capi):
    """GoodPosePrior returns residuals[0] < 1: a prior that far from its pose fails the evaluation like any functor."""
    p = scene()
    q = with_good_pose_priors(p)
    q.pose_prior_values[3, 0] += 1.0                     # rotation weight 3: residual[0] = 3 * (1.0 + noise) >= 1
    with capi.DeviceProblem(q) as dp:
        out = dp.evaluate(residuals=False, jacobians=False)
        assert out["num_failed"] == 1
        with pytest.raises(capi.RsbaError) as e:
            dp.solve()
    assert e.value.status == 4


def spherical_scene(seed=71):
    p = scene(seed=seed)
    p.poses[0] = 0.0
    p.poses[1] = 0.0
    p.poses[1, :, 3:] += 1e-4                            # CeresHandler::Add's start of frame 1 (:121-125)
    p.spherical_pose_block = 2
    return p


def test_spherical_prior_cost_gradient_and_failure(capi, oracle):
    p = spherical_scene()
    plain = p.copy(); plain.spherical_pose_block = -1
    with capi.DeviceProblem(p) as d1, capi.DeviceProblem(plain) as d0:
        b, a = d1.evaluate(residuals=False, jacobians=False, gradient=True), d0.evaluate(residuals=False, jacobians=False, gradient=True)
    pose = p.poses[1, 0]
    r0, r1 = np.sum(pose[:3] ** 2), 1e20 * (1.0 - np.abs(pose[3]) - np.abs(pose[4]) - np.abs(pose[5]))
    assert abs(b["cost"] - 0.5 * (r0 * r0 + r1 * r1)) <= 1e-12 * b["cost"] and b["cost"] > 1e39
    ok, cost_ref, g_ref = oracle.evaluate(p)
    assert ok and abs(b["cost"] - cost_ref) <= 1e-13 * cost_ref
    assert np.max(np.abs(b["gradient"]["poses"] - g_ref["poses"])) <= 1e-12 * np.abs(g_ref["poses"]).max()
    assert np.allclose(b["gradient"]["poses"][1, 0, 3:], -1e20 * r1, rtol=1e-9)         # d/dc of 1/2 r1^2, sign(c) = +1
    assert np.array_equal(b["gradient"]["points"], a["gradient"]["points"])
    q = p.copy(); q.poses[1, 0, :3] = [0.8, 0.5, 0.4]                                       # |rot|^2 >= 1: the functor returns false
    with capi.DeviceProblem(q) as dp:
        assert dp.evaluate(residuals=False, jacobians=False)["num_failed"] >= 1


def test_spherical_prior_solve(capi, oracle):
    """The 1e20-weighted residual pins |c|_1 of frame 1's first pose to 1 (the scale gauge of a session started at the
    origin).  While the residual is far above rounding both solvers walk the same path (each LM step shrinks it by ~3e-5:
    the damping).  Once it is met, its value is rounding noise times 1e20 — 0 if 1 - |cx| - |cy| - |cz| happens to cancel
    exactly, +-5.5e3 (half an ulp of 1, times 1e20) if not — and which of the two a solver lands on is luck of the last bit:
    with the 5.5e3 the cost carries 1.5e7 that no representable step can remove and the trust region collapses (real Ceres
    does the same).  So from there only the state is compared: the constraint holds to the last bit or the one before."""
    p = spherical_scene()
    pd, pc = p.copy(), p.copy()
    with capi.DeviceProblem(pd) as dp:
        s, tr = dp.solve(capi.default_options(max_num_iterations=30))
    s_ref, tr_ref = oracle.solve(pc, oracle.default_options(max_num_iterations=30))
    assert s.is_solution_usable and s.num_residual_blocks == s_ref.num_residual_blocks and s.num_parameters_reduced == s_ref.num_parameters_reduced
    assert abs(s.initial_cost - s_ref.initial_cost) <= 1e-12 * s_ref.initial_cost
    for a, b in list(zip(tr, tr_ref))[:3]:                   # residual 1e20 x {1, 3e-5, 4e-10}: still well above rounding
        assert a.step_is_successful == b.step_is_successful and abs(a.cost - b.cost) <= 1e-5 * b.cost, (a.iteration, a.cost, b.cost)
        assert abs(a.trust_region_radius - b.trust_region_radius) <= 1e-9 * b.trust_region_radius
    for q in (pd, pc):
        assert abs(1.0 - np.abs(q.poses[1, 0, 3:]).sum()) <= 1.2e-16
    assert np.max(np.abs(pd.poses[1, 0, 3:] - pc.poses[1, 0, 3:])) <= 1e-9
    rest = pd.copy(); rest.spherical_pose_block = -1
    start = spherical_scene(); start.spherical_pose_block = -1
    assert oracle.evaluate(rest, gradient=False)[1] < 0.01 * oracle.evaluate(start, gradient=False)[1]      # the reprojection part came down with it
