"""CPU oracle against the committed golden vectors (tests/golden/*.json, produced by the independent
mpmath / scipy implementation in tests/golden/make_golden.py)."""
import numpy as np
import pytest

from helpers import batch_cases, load_golden, problem_from_case, problem_from_solve_case, rel_err

# fp64 evaluation of a ~100-flop chain with |J| ~ 1e3: forward error a few ulp of the largest term
R_TOL = 1e-11   # residuals (pixels), relative to max(1,|r|)
J_TOL = 1e-10   # Jacobian entries, relative to max(1,|J|)
# |w| within a few 1e-8 of the DBL_EPSILON branch point of AngleAxisRotatePoint: the Rodrigues branch
# forms 1 - cos(theta) with theta ~ 2e-8 in fp64 (one ulp of 1.0 = 50 % of the value), so Ceres' own
# Jet result carries ~1e-16/theta * |J| of noise there.  The golden is the exact derivative.
J_TOL_TINY_ANGLE = 1e-6


def j_tol(c):
    return J_TOL_TINY_ANGLE if c["name"].startswith("tiny") else J_TOL


def test_per_observation_residual_and_jacobian(oracle):
    cases = load_golden("per_observation.json")
    assert len(cases) >= 40
    for c in cases:
        prob = problem_from_case(c)
        r, J, ok = oracle.evaluate_blocks(prob)
        assert bool(ok[0]) == c["ok"], c["name"]
        if not c["ok"]:
            continue
        assert rel_err(r[0], c["residual"]) <= R_TOL, (c["name"], r[0], c["residual"])
        assert J.shape[2] == len(c["jacobian"][0])
        assert rel_err(J[0], c["jacobian"]) <= j_tol(c), (c["name"], np.abs(J[0] - np.array(c["jacobian"])).max())
        # residual-only path (T = double) agrees with the dual-number path
        r2, _, ok2 = oracle.evaluate_blocks(prob, jac=False)
        assert ok2[0] and rel_err(r2, r) <= 1e-13   # (FMA contraction differs between the two instantiations)


def test_batched_cases_match_single(oracle):
    cases = load_golden("per_observation.json")
    n = 0
    for prob, idxs in batch_cases(cases):
        r, J, ok = oracle.evaluate_blocks(prob)
        for row, i in enumerate(idxs):
            c = cases[i]
            assert bool(ok[row]) == c["ok"]
            if c["ok"]:
                assert rel_err(r[row], c["residual"]) <= R_TOL and rel_err(J[row], c["jacobian"]) <= j_tol(c)
                n += 1
    assert n == sum(1 for c in cases if c["ok"])


def test_vertical_shutter_quirk_reads_x(oracle):
    """VideoSfmBaRs.h:31 passes (x, x): with VERTICAL shutter tau still comes from observed_x."""
    cases = [c for c in load_golden("per_observation.json") if c["shutter"] == 2 and c["ok"]]
    assert cases
    for c in cases:
        prob = problem_from_case(c)
        r, _, _ = oracle.evaluate_blocks(prob, jac=False)
        # same observation with HORIZONTAL shutter gives the identical residual ...
        prob.shutter = 1
        r_h, _, _ = oracle.evaluate_blocks(prob, jac=False)
        assert np.array_equal(r, r_h)
        # ... whereas the non-functor twin (struct/VideoSfM.cc:103-133) really uses y for VERTICAL
        p_true_v = oracle.interpolate_rs(c["pose0"], c["pose1"], 2, c["scanlines"], c["obs"], c["interpolate_rotation"])
        p_quirk = oracle.interpolate_rs(c["pose0"], c["pose1"], 2, c["scanlines"], [c["obs"][0], c["obs"][0]], c["interpolate_rotation"])
        assert not np.allclose(p_true_v, p_quirk)


def test_huber(oracle):
    for c in load_golden("huber.json"):
        rho = oracle.huber(c["a"], c["s"])
        assert rel_err(rho, c["rho"]) <= 1e-14, c


@pytest.mark.parametrize("idx", [0, 1, 2])
def test_tiny_solve_reaches_independent_minimum(oracle, idx):
    """LM restatement vs scipy's trust-region-reflective minimiser on the same tiny scene:
    same initial cost, same minimum (tolerances tightened so both sit at the minimum)."""
    c = load_golden("tiny_solves.json")[idx]
    prob = problem_from_solve_case(c)
    ok, cost0, _ = oracle.evaluate(prob, gradient=False)
    assert ok and abs(cost0 - c["expected"]["initial_cost"]) <= 1e-9 * cost0
    opts = oracle.default_options(max_num_iterations=200, function_tolerance=1e-14, parameter_tolerance=1e-14, gradient_tolerance=1e-12)
    s, trace = oracle.solve(prob, opts)
    assert s.termination_type in (0, 1)
    assert abs(s.final_cost - c["expected"]["final_cost"]) <= 1e-8 * c["expected"]["final_cost"], (s.final_cost, c["expected"]["final_cost"])
    # parameters agree with the independent minimiser (gauge fixed -> unique minimum)
    # (the Huber case's scipy run stops at |g|inf ~ 1e-3 — finite-difference Jacobian of a C1 cost — so
    # its parameters are only good to ~1e-3; its cost is still second-order accurate)
    ptol = 1e-3 if c["huber_a"] > 0 else 1e-5
    assert np.max(np.abs(prob.poses - np.array(c["expected"]["poses"]))) <= ptol
    assert np.max(np.abs(prob.points - np.array(c["expected"]["points"]))) <= 10 * ptol
    # default Ceres tolerances stop within 1e-6 relative of that minimum
    prob2 = problem_from_solve_case(c)
    s2, _ = oracle.solve(prob2, oracle.default_options(max_num_iterations=50))
    assert s2.termination_type == 0
    assert abs(s2.final_cost - c["expected"]["final_cost"]) <= 2e-6 * c["expected"]["final_cost"]


@pytest.mark.parametrize("idx", [0, 1])
def test_motion_prior_solve_reaches_independent_minimum(oracle, idx):
    """Frame-to-frame motion priors (constant interFrameRatio): the restated functors + the LM restatement
    against an independent numpy model of the priors minimised by scipy (tests/golden/make_golden.py priors)."""
    c = load_golden("prior_solves.json")[idx]
    prob = problem_from_solve_case(c)
    ok, cost0, g = oracle.evaluate(prob)
    assert ok and abs(cost0 - c["expected"]["initial_cost"]) <= 1e-9 * cost0
    # gradient incl. the prior blocks against central differences of the cost
    rng = np.random.default_rng(1)
    for _ in range(6):
        f, q, k = rng.integers(1, prob.num_frames - 1), rng.integers(0, 2), rng.integers(0, 6)
        h = 1e-6
        p1 = prob.copy(); p1.poses[f, q, k] += h
        p2 = prob.copy(); p2.poses[f, q, k] -= h
        fd = (oracle.evaluate(p1, gradient=False)[1] - oracle.evaluate(p2, gradient=False)[1]) / (2 * h)
        assert abs(fd - g["poses"][f, q, k]) <= 1e-5 * max(1.0, abs(fd))
    opts = oracle.default_options(max_num_iterations=200, function_tolerance=1e-14, parameter_tolerance=1e-14, gradient_tolerance=1e-12)
    s, trace = oracle.solve(prob, opts)
    assert s.termination_type in (0, 1)
    assert s.num_residual_blocks == prob.num_observations + len(c["prior_frames"])
    assert abs(s.final_cost - c["expected"]["final_cost"]) <= 1e-8 * c["expected"]["final_cost"], (s.final_cost, c["expected"]["final_cost"])
    ptol = 1e-3 if c["huber_a"] > 0 else 1e-5
    assert np.max(np.abs(prob.poses - np.array(c["expected"]["poses"]))) <= ptol
    assert np.max(np.abs(prob.points - np.array(c["expected"]["points"]))) <= 10 * ptol
    # the priors matter: without them the same scene ends at a different (lower) cost
    prob0 = problem_from_solve_case(c); prob0.prior_kind = 0; prob0.prior_frames = None
    s0, _ = oracle.solve(prob0, opts)
    assert s0.final_cost < (1 - 1e-4) * s.final_cost


@pytest.mark.parametrize("idx", [0, 1])
def test_free_inter_frame_ratio_reaches_independent_minimum(oracle, idx):
    """The reference's default for the motion priors (opt.ceres.interFrameRatio left at 1): the ratio is a free parameter
    block bounded below (CeresHandler.h:161,172,175).  Restatement: one more unknown of the LM, candidate projected onto
    the bound — against scipy's bounded minimiser on the independent numpy model (make_golden.py free_ratio)."""
    c = load_golden("free_ratio_solves.json")[idx]
    prob = problem_from_solve_case(c)
    ok, cost0, _ = oracle.evaluate(prob, gradient=False)
    assert ok and abs(cost0 - c["expected"]["initial_cost"]) <= 1e-9 * cost0
    s, _ = oracle.solve(prob, oracle.default_options(max_num_iterations=300, function_tolerance=1e-14, parameter_tolerance=1e-14, gradient_tolerance=1e-12))
    assert s.termination_type in (0, 1)
    assert abs(s.final_cost - c["expected"]["final_cost"]) <= 1e-8 * c["expected"]["final_cost"]
    assert abs(prob.inter_frame_ratio - c["expected"]["ratio"]) <= 1e-5 and prob.inter_frame_ratio > 2.0     # it moved from 1 to the scene's gap / exposure
    assert np.max(np.abs(prob.poses - np.array(c["expected"]["poses"]))) <= 1e-5
    # one more effective parameter than with the ratio held constant
    fixed = problem_from_solve_case(c); fixed.ratio_free = False
    s1, _ = oracle.solve(fixed, oracle.default_options(max_num_iterations=5))
    assert s.num_parameters_reduced == s1.num_parameters_reduced + 1 and fixed.inter_frame_ratio == 1.0


@pytest.mark.parametrize("idx", [0, 1])
def test_pnp_refinement_reaches_independent_minimum(oracle, idx):
    """RS-PnP: the oracle's hypothesis task (RsBA<float> blocks over the two poses through the LM
    restatement) against scipy's minimum of an independent numpy model (tests/golden/make_golden.py pnp)."""
    c = load_golden("pnp_solves.json")[idx]
    X, xy = np.array(c["object_points"], dtype=np.float32), np.array(c["image_points"], dtype=np.float32)
    sub = np.arange(len(X), dtype=np.int32)
    r0 = oracle.pnp_task(c["cam"], c["shutter"], c["scanlines"], X, xy, sub, c["init_poses"], 0, 3.0)
    assert np.array_equal(r0["poses"], np.array(c["init_poses"]))            # zero iterations: nothing moves
    r = oracle.pnp_task(c["cam"], c["shutter"], c["scanlines"], X, xy, sub, c["init_poses"], 100, 3.0)
    assert r["usable"]
    # the task runs at Ceres' default tolerances, as the reference does (function_tolerance 1e-6): it stops within that of the minimum
    assert 0 <= r["final_cost"] - c["expected"]["final_cost"] <= 2e-6 * c["expected"]["final_cost"]
    assert np.max(np.abs(r["poses"] - np.array(c["expected"]["poses"]))) <= 1e-3
    assert r["num_inliers"] >= 0.95 * len(X)                                  # 0.5 px noise, 3 px threshold


def test_gradient_matches_finite_differences(oracle):
    c = load_golden("tiny_solves.json")[2]   # Huber case
    prob = problem_from_solve_case(c)
    ok, cost, g = oracle.evaluate(prob)
    assert ok
    rng = np.random.default_rng(0)
    for _ in range(6):
        f, q, k = rng.integers(1, prob.num_frames - 1), rng.integers(0, 2), rng.integers(0, 6)
        h = 1e-6
        p1 = prob.copy(); p1.poses[f, q, k] += h
        p2 = prob.copy(); p2.poses[f, q, k] -= h
        fd = (oracle.evaluate(p1, gradient=False)[1] - oracle.evaluate(p2, gradient=False)[1]) / (2 * h)
        assert abs(fd - g["poses"][f, q, k]) <= 1e-5 * max(1.0, abs(fd))
    # fixed coordinates carry zero gradient
    assert np.all(g["poses"][0] == 0) and np.all(g["poses"][-1, -1, 3:] == 0)
