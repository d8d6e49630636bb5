"""Parity of the on-device LM / Schur / Cholesky solve (rsba_solve through the C ABI) with the CPU
oracle's restatement of ceres::Solve(SPARSE_SCHUR), and with the independent scipy minima in
tests/golden/tiny_solves.json.  Tolerances follow SURVEY Appendix C.6:
  (a) single evaluation at identical x: cost / gradient / normal-equation blocks <= 1e-11 relative
  (b) identical LM iterations from identical state: per-iteration cost <= 1e-9 relative
  (c) final cost at tightened tolerances <= 1e-8, at default tolerances <= 1e-6 relative (the contract)."""
import numpy as np
import pytest

from helpers import load_golden, problem_from_solve_case, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from rsba_amd import capi
    assert capi.device_count() >= 1
    return capi


def small_scene(rolling=True, frames=20, points=900, seed=21, **kw):
    from rsba_amd.problem import apply_gauge_masks
    from rsba_amd.scene import make_scene
    p = make_scene(frames, points, rolling=rolling, seed=seed, **kw).problem
    apply_gauge_masks(p, fix_first_n_cameras=1)
    p.pose_fixed_mask[-1, -1] |= 0b111000
    return p


def scaled_err(a, b):
    return float(np.max(np.abs(a - b)) / max(1.0, float(np.max(np.abs(b)))))


def check_normal_equations(capi, oracle, p, tol=1e-11):
    U_ref, gc_ref, V_ref, gp_ref = oracle.normal_equations(p)
    with capi.DeviceProblem(p) as dp:
        U, gc, V, gp = dp.normal_equations()
        out = dp.evaluate(residuals=False, jacobians=False, gradient=True)
    assert scaled_err(U, U_ref) <= tol and scaled_err(gc, gc_ref) <= tol
    assert scaled_err(V, V_ref) <= tol and scaled_err(gp, gp_ref) <= tol
    ok, cost_ref, g_ref = oracle.evaluate(p)
    assert abs(out["cost"] - cost_ref) <= 1e-12 * cost_ref
    assert scaled_err(out["gradient"]["poses"], g_ref["poses"]) <= tol
    assert scaled_err(out["gradient"]["points"], g_ref["points"]) <= tol


@pytest.mark.parametrize("rolling", [True, False])
@pytest.mark.parametrize("huber", [0.0, 2.0])
def test_normal_equation_blocks(capi, oracle, rolling, huber):
    p = small_scene(rolling=rolling, outlier_ratio=0.1 if huber else 0.0)
    p.huber_a = huber
    p.point_constant = np.zeros(p.num_points, dtype=np.uint8); p.point_constant[::11] = 1
    check_normal_equations(capi, oracle, p)


def compare_solves(capi, oracle, p, iters=50, traj_tol=1e-9, final_tol=1e-6, tight=False, expect_same_path=True):
    kw = dict(max_num_iterations=iters)
    if tight:
        kw.update(function_tolerance=1e-14, parameter_tolerance=1e-14, gradient_tolerance=1e-12)
    p_dev, p_cpu = p.copy(), p.copy()
    with capi.DeviceProblem(p_dev) as dp:
        s, tr = dp.solve(capi.default_options(**kw))
    s_ref, tr_ref = oracle.solve(p_cpu, oracle.default_options(**kw))
    assert s.is_solution_usable and s.termination_type == s_ref.termination_type or tight
    assert abs(s.initial_cost - s_ref.initial_cost) <= 1e-12 * s_ref.initial_cost
    assert s.num_residual_blocks_reduced == s_ref.num_residual_blocks_reduced
    assert s.num_parameters_reduced == s_ref.num_parameters_reduced
    assert abs(s.fixed_cost - s_ref.fixed_cost) <= 1e-12 * max(1.0, s_ref.fixed_cost)
    # (b) the first iterations replay the oracle's trajectory
    if expect_same_path:
        for a, b in list(zip(tr, tr_ref))[:4]:
            assert a.iteration == b.iteration and a.step_is_successful == b.step_is_successful
            assert abs(a.cost - b.cost) <= traj_tol * b.cost, (a.iteration, a.cost, b.cost)
            assert abs(a.trust_region_radius - b.trust_region_radius) <= 1e-6 * b.trust_region_radius
    # (c) final cost
    assert abs(s.final_cost - s_ref.final_cost) <= final_tol * s_ref.final_cost, (s.final_cost, s_ref.final_cost, s.num_iterations, s_ref.num_iterations)
    return s, s_ref, p_dev, p_cpu


@pytest.mark.parametrize("idx", [0, 1, 2])
def test_tiny_solves_reach_the_independent_minimum(capi, oracle, idx):
    c = load_golden("tiny_solves.json")[idx]
    p = problem_from_solve_case(c)
    s, s_ref, p_dev, p_cpu = compare_solves(capi, oracle, p, iters=200, tight=True, final_tol=1e-9)
    assert abs(s.final_cost - c["expected"]["final_cost"]) <= 1e-8 * c["expected"]["final_cost"]
    ptol = 1e-3 if c["huber_a"] > 0 else 1e-5
    assert np.max(np.abs(p_dev.poses - np.array(c["expected"]["poses"]))) <= ptol
    # (parameters agree as far as the flat directions of the cost allow; the Huber scene is flatter.  Both solves stop on
    # their tolerances somewhere along those directions: the distance was observed at 0.9-1.1e-7 from run to run — the
    # oracle's OpenMP reductions are not order-deterministic — so the bound leaves a factor of a few)
    qtol = 1e-5 if c["huber_a"] > 0 else 5e-7
    assert np.max(np.abs(p_dev.poses - p_cpu.poses)) <= qtol and np.max(np.abs(p_dev.points - p_cpu.points)) <= 10 * qtol
    # default Ceres tolerances: within the 1e-6 contract of the minimum
    p2 = problem_from_solve_case(c)
    with capi.DeviceProblem(p2) as dp:
        s2, _ = dp.solve(capi.default_options(max_num_iterations=50))
    assert s2.termination_type == 0
    assert abs(s2.final_cost - c["expected"]["final_cost"]) <= 2e-6 * c["expected"]["final_cost"]


@pytest.mark.parametrize("rolling", [True, False])
def test_small_scene_solve(capi, oracle, rolling):
    compare_solves(capi, oracle, small_scene(rolling=rolling), iters=30)


def test_huber_solve(capi, oracle):
    p = small_scene(outlier_ratio=0.05, seed=33)
    p.huber_a = 2.0
    compare_solves(capi, oracle, p, iters=40)


@pytest.mark.parametrize("variant", ["free_gauge", "fix_rotation", "fix_position", "const3d", "window", "fix_scale"])
def test_constness_rules(capi, oracle, variant):
    """CeresHandler::Add's constant / subset rules (src/rsba/CeresHandler.h:288-300, 342-382)."""
    from rsba_amd.problem import apply_gauge_masks
    from rsba_amd.scene import make_scene
    p = make_scene(14, 500, seed=17).problem
    if variant == "free_gauge":
        apply_gauge_masks(p)                      # reference default: nothing fixed, S rank-deficient up to damping
    elif variant == "fix_rotation":
        apply_gauge_masks(p, fix_first_n_cameras=1, fix_rotation=True)
    elif variant == "fix_position":
        apply_gauge_masks(p, fix_first_n_cameras=1, fix_position=True)
    elif variant == "const3d":
        apply_gauge_masks(p, fix_first_n_cameras=0, const3d=True)
    elif variant == "window":
        apply_gauge_masks(p, fix_first_n_cameras=0, start_frame=9)
        p.pose_fixed_mask[:9] = 0x3F              # frames before the window are not part of the problem's free set
    elif variant == "fix_scale":
        apply_gauge_masks(p, fix_scale=True)
    compare_solves(capi, oracle, p, iters=15, final_tol=1e-6, expect_same_path=(variant != "free_gauge"))


@pytest.mark.parametrize("rolling", [True, False])
@pytest.mark.parametrize("huber", [0.0, 2.0])
def test_shared_intrinsics_as_parameter_block(capi, oracle, rolling, huber):
    """opt.model.calibrated = false with the shared sess.cam block (CeresHandler.h:256-264, 273-280):
    BASELINE config C5's model (Huber + shared intrinsics) at a size the oracle solves in seconds."""
    p = small_scene(rolling=rolling, frames=18, points=800, seed=51, outlier_ratio=0.05 if huber else 0.0)
    p.calibrated = False
    p.huber_a = huber
    p.intrinsics = p.intrinsics * (1.0 + 1e-3 * np.array([[1, -1, 20, -20, 10, 10, -10, 0.5, -0.5]]))   # start off the true calibration
    check_normal_equations_uncalibrated(capi, oracle, p)
    # the pose blocks U_f = Jc^T Jc, g_f and the point blocks directly: with rolling shutter the operand [Ji | Jc | r] has 22 columns and the pose
    # columns 9 .. 20 straddle the two 16 x 16 products the evaluation kernel leaves per wave (device_state.hpp: cam_part_entry — block 0, block 0
    # transposed and block 1 all hold entries of U_f)
    if huber == 0.0:   # (no loss: the corrected Jacobian is the oracle's raw one)
        r, J, ok = oracle.evaluate_blocks(p)
        cd = 6 * p.poses_per_frame
        Jc = np.where(ok[:, None, None], J[:, :, 9:9 + cd], 0.0); Jp = np.where(ok[:, None, None], J[:, :, 9 + cd:], 0.0); rr = np.where(ok[:, None], r, 0.0)
        if p.pose_fixed_mask is not None:   # the columns of fixed coordinates are zero in what the solver linearises (gauge: the first camera)
            free = ((p.pose_fixed_mask.reshape(p.num_frames, -1)[:, :, None] >> np.arange(6)) & 1) == 0          # [F, P, 6]
            Jc = Jc * free.reshape(p.num_frames, cd)[p.obs_frame][:, None, :]
        U_ref = np.zeros((p.num_frames, cd, cd)); gc_ref = np.zeros((p.num_frames, cd)); V_ref = np.zeros((p.num_points, 3, 3)); gp_ref = np.zeros((p.num_points, 3))
        np.add.at(U_ref, p.obs_frame, np.einsum("nrk,nrl->nkl", Jc, Jc)); np.add.at(gc_ref, p.obs_frame, np.einsum("nrk,nr->nk", Jc, rr))
        np.add.at(V_ref, p.obs_point, np.einsum("nrk,nrl->nkl", Jp, Jp)); np.add.at(gp_ref, p.obs_point, np.einsum("nrk,nr->nk", Jp, rr))
        with capi.DeviceProblem(p) as dp:
            U, gc, V, gp = dp.normal_equations()
        assert scaled_err(U, U_ref) <= 1e-11 and scaled_err(gc, gc_ref) <= 1e-11 and scaled_err(V, V_ref) <= 1e-11 and scaled_err(gp, gp_ref) <= 1e-11
    s, s_ref, p_dev, p_cpu = compare_solves(capi, oracle, p, iters=40)
    assert np.max(np.abs(p_dev.intrinsics - p_cpu.intrinsics) / np.maximum(1.0, np.abs(p_cpu.intrinsics))) <= 1e-6
    assert not np.array_equal(p_dev.intrinsics, p.intrinsics)


def test_constant_shared_intrinsics_block(capi, oracle):
    p = small_scene(frames=10, points=300, seed=52)
    p.calibrated = False
    p.intrinsics_constant = np.array([1], dtype=np.uint8)
    s, s_ref, p_dev, p_cpu = compare_solves(capi, oracle, p, iters=20)
    assert np.array_equal(p_dev.intrinsics, p.intrinsics)


@pytest.mark.parametrize("layout", ["per_frame", "mixed", "mixed_global_shutter", "unsorted"])
def test_per_frame_intrinsics_blocks_in_the_solve(capi, oracle, layout):
    """CeresHandler::Add uses f.cam as the intrinsics parameter block of a frame that carries one and sess.cam otherwise
    (CeresHandler.h:256-264, 273-280): several 9-blocks, each coupled to the frames that use it.  per_frame: one block per
    frame; mixed: the first half of the video shares block 0, the rest have their own, one of them constant; the global-
    shutter variant has 6-wide camera blocks (two pseudo frames per intrinsics block)."""
    rolling = layout != "mixed_global_shutter"
    p = small_scene(rolling=rolling, frames=16, points=800, seed=57, outlier_ratio=0.03)
    p.calibrated = False
    p.huber_a = 2.0
    rng = np.random.default_rng(3)
    F = p.num_frames
    if layout in ("per_frame", "unsorted"):
        fi = np.arange(F, dtype=np.int32)
    else:
        fi = np.concatenate([np.zeros(F // 2, dtype=np.int32), np.arange(1, F - F // 2 + 1, dtype=np.int32)])
    ni = int(fi.max()) + 1
    p.intrinsics = np.tile(p.intrinsics[:1], (ni, 1)) * (1.0 + 1e-3 * rng.normal(size=(ni, 9)) * np.array([[1, 1, 20, 20, 10, 10, 10, 0.5, 0.5]]))
    p.frame_intrinsics = fi
    if layout == "unsorted":
        p.frame_intrinsics = rng.permutation(F).astype(np.int32)
        perm = rng.permutation(p.num_observations)
        p.obs_xy, p.obs_frame, p.obs_point = p.obs_xy[perm].copy(), p.obs_frame[perm].copy(), p.obs_point[perm].copy()
    if layout.startswith("mixed"):
        p.intrinsics_constant = np.zeros(ni, dtype=np.uint8); p.intrinsics_constant[2] = 1
    start = p.intrinsics.copy()
    check_normal_equations_uncalibrated(capi, oracle, p)
    s, s_ref, p_dev, p_cpu = compare_solves(capi, oracle, p, iters=25)
    assert np.max(np.abs(p_dev.intrinsics - p_cpu.intrinsics) / np.maximum(1.0, np.abs(p_cpu.intrinsics))) <= 1e-6
    moved = np.any(p_dev.intrinsics != start, axis=1)
    if layout.startswith("mixed"):
        assert not moved[2] and moved[[0, 1, 3]].all()
    else:
        assert moved.all()


def check_normal_equations_uncalibrated(capi, oracle, p, tol=1e-11):
    ok, cost_ref, g_ref = oracle.evaluate(p)
    with capi.DeviceProblem(p) as dp:
        out = dp.evaluate(residuals=False, jacobians=False, gradient=True)
    assert abs(out["cost"] - cost_ref) <= 1e-12 * cost_ref
    for k in ("poses", "points", "intrinsics"):
        assert scaled_err(out["gradient"][k], g_ref[k]) <= tol, k


def test_point_observed_twice_in_one_frame(capi, oracle):
    """Two residual blocks on the same (frame, point) pair: the elimination must carry the cross terms."""
    p = small_scene(frames=12, points=300, seed=61)
    rng = np.random.default_rng(3)
    dup = rng.choice(p.num_observations, size=150, replace=False)
    p.obs_xy = np.concatenate([p.obs_xy, p.obs_xy[dup] + rng.normal(0, 0.7, (len(dup), 2))])
    p.obs_frame = np.concatenate([p.obs_frame, p.obs_frame[dup]])
    p.obs_point = np.concatenate([p.obs_point, p.obs_point[dup]])
    check_normal_equations(capi, oracle, p)
    compare_solves(capi, oracle, p, iters=20)


def test_behind_camera_initial_failure(capi):
    p = small_scene(frames=6, points=100)
    p.points[3, 2] = -5.0
    with capi.DeviceProblem(p) as dp:
        with pytest.raises(capi.RsbaError) as e:
            dp.solve()
    assert e.value.status == 4


def test_config_c1_solve(capi, oracle):
    from rsba_amd.scene import make_config
    compare_solves(capi, oracle, make_config("C1").problem, iters=25)


def test_config_c2_solve_matches_oracle(capi, oracle):
    """BASELINE config C3: the 100-frame RS scene, full LM loop on the device, cost within 1e-6."""
    from rsba_amd.scene import make_config
    p = make_config("C2").problem
    s, s_ref, p_dev, p_cpu = compare_solves(capi, oracle, p, iters=12)
    assert np.sqrt(s.final_cost / s.num_residual_blocks_reduced) < 0.6   # "average reprojection error" (VideoSfMHandler.cc:627-628)


@pytest.mark.parametrize("config,shared_intrinsics", [("C2", False), ("C2", True), ("C4", False)])
def test_dag_cholesky_equals_the_level_schedule(capi, monkeypatch, config, shared_intrinsics):
    """The persistent task-DAG Cholesky (tickets + per-tile flags + agent-coherent loads/stores) runs the same task
    bodies in the same summation order as one launch per (level, kind): the two solves must agree to the bit.  A
    stale read or a flag raised before its data would show up here."""
    from rsba_amd.scene import make_config
    out = {}
    monkeypatch.delenv("RSBA_CHOL_LEVELS", raising=False)
    for mode in ("0", "1"):
        p = make_config(config).problem
        if shared_intrinsics:
            p.calibrated = False
            p.huber_a = 2.0
        with capi.DeviceProblem(p) as dp:
            s, _ = dp.solve(capi.default_options(max_num_iterations=6, level_scheduled_cholesky=int(mode)))
        out[mode] = (s.final_cost, s.num_iterations, p.poses.copy(), p.points.copy(), p.intrinsics.copy())
    a, b = out["0"], out["1"]
    assert a[0] == b[0] and a[1] == b[1]
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])


def test_shared_intrinsics_with_many_points(capi, oracle):
    """The diagonal tile pair of the intrinsics pseudo tile has one Schur chunk per 512 points: above 32 chunks its
    partial tiles are pre-reduced in groups before the merge.  20 000 points reach that path; the oracle still
    solves the 40-frame reduced system densely in seconds."""
    p = small_scene(rolling=True, frames=40, points=20000, seed=53, outlier_ratio=0.02)
    p.calibrated = False
    p.huber_a = 2.0
    p.intrinsics = p.intrinsics * (1.0 + 1e-3 * np.array([[1, -1, 20, -20, 10, 10, -10, 0.5, -0.5]]))
    s, s_ref, p_dev, p_cpu = compare_solves(capi, oracle, p, iters=8)
    assert np.max(np.abs(p_dev.intrinsics - p_cpu.intrinsics) / np.maximum(1.0, np.abs(p_cpu.intrinsics))) <= 1e-6


def run_hook_case(case, tmp_path):
    """A case of tests/hook_worker.py in a process of its own, on the instrumented library (the release library has no fault-injection switches)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(str(tmp_path), "hook.json")
    env = dict(os.environ, RSBA_AMD_LIB=os.path.join(root, "rsba_amd", "_lib", "librsba_amd_hooks.so"))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "hook_worker.py"), case, out], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = json.load(open(out))
    assert res["library"].endswith("librsba_amd_hooks.so")
    return res


def test_a_wrong_dag_result_is_caught_and_redone_on_the_level_schedule(capi, tmp_path):
    """The persistent Cholesky driver's solution is verified against the system it was given (res = rhs - S y against
    |rhs| + |S||y|).  RSBA_CHOL_TEST_CORRUPT=1 (instrumented library) makes the first DAG solve lose one entry of y: the check must notice, the
    iteration is repeated — and the problem finished — on the level schedule, and the result is the level schedule's, bit
    for bit.  Without the hook no solve of this suite trips the check."""
    out = run_hook_case("corrupt_dag", tmp_path)
    assert out["corrupt"]["fallbacks"] == 1 and out["levels"]["fallbacks"] == 0 and out["plain"]["fallbacks"] == 0
    for a, b in ((out["corrupt"], out["levels"]), (out["plain"], out["levels"])):
        assert a["final_cost"] == b["final_cost"] and a["iters"] == b["iters"] and a["poses"] == b["poses"] and a["points"] == b["points"]


def test_the_release_library_has_no_fault_injection(capi, monkeypatch):
    """The switches above exist in the instrumented build only (rsba_amd/csrc/test_hooks.hpp): the library the product ships neither fails its
    plan nor loses an entry of a solve because a variable is set."""
    assert capi.LIB_PATH.endswith("librsba_amd.so")
    p, q = small_scene(frames=12, points=500), small_scene(frames=12, points=500)
    with capi.DeviceProblem(q) as dp:
        s_ref, _ = dp.solve(capi.default_options(max_num_iterations=5))
    for hook in ("RSBA_TEST_FAIL_PLAN", "RSBA_TEST_FAIL_DEVICE_PLAN", "RSBA_CHOL_TEST_CORRUPT"):
        monkeypatch.setenv(hook, "1")
    with capi.DeviceProblem(p) as dp:
        s, _ = dp.solve(capi.default_options(max_num_iterations=5))
    assert s.num_dag_fallbacks == 0 and s.final_cost == s_ref.final_cost and np.array_equal(p.poses, q.poses)


@pytest.mark.parametrize("shared_intrinsics", [False, True])
def test_recomputed_records_equal_the_stored_ones(capi, monkeypatch, shared_intrinsics):
    """The point-side passes of the solve recompute each observation's LM record (rsba_amd/csrc/lm_record.hpp) instead of reading
    the point-major copy the evaluation kernel can leave for them (RSBA_RECORDS=1, and always with several intrinsics blocks):
    the same function of the same inputs — the two solves may differ by the rounding of differently fused multiply-adds only."""
    from rsba_amd.problem import apply_gauge_masks
    from rsba_amd.scene import make_scene
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("RSBA_RECORDS", mode)
        p = make_scene(40, 3000, seed=23, outlier_ratio=0.05).problem
        p.huber_a = 2.0
        p.calibrated = not shared_intrinsics
        apply_gauge_masks(p, fix_first_n_cameras=1)
        p.pose_fixed_mask[-1, -1] |= 0b111000
        with capi.DeviceProblem(p) as dp:
            s, tr = dp.solve(capi.default_options(max_num_iterations=8, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0))
        res[mode] = (s, tr, p.poses.copy(), p.points.copy())
    a, b = res["0"], res["1"]
    assert a[0].num_iterations == b[0].num_iterations
    for x, y in zip(a[1], b[1]):
        assert x.step_is_successful == y.step_is_successful
        assert abs(x.cost - y.cost) <= 1e-11 * y.cost
    assert np.max(np.abs(a[2] - b[2])) <= 1e-8 and np.max(np.abs(a[3] - b[3])) <= 1e-7


def mixed_session(seed=31):
    """A rolling-shutter session in which every third frame has ONE pose (CeresHandler::Add gives such a frame the global-shutter
    functor, CeresHandler.h:245-286): two pose slots per frame, the one-pose frames flagged, their second slot constant."""
    from rsba_amd.problem import apply_gauge_masks
    from rsba_amd.scene import make_scene
    p = make_scene(30, 2000, seed=seed).problem
    apply_gauge_masks(p, fix_first_n_cameras=1)
    p.pose_fixed_mask[-1, -1] |= 0b111000
    p.frame_global = (np.arange(p.num_frames) % 3 == 2).astype(np.uint8)
    p.pose_fixed_mask[p.frame_global == 1, 1] = 0x3F
    return p


def test_one_pose_frames_inside_a_rolling_shutter_session(capi, oracle):
    p = mixed_session()
    r_ref, J_ref, ok_ref = oracle.evaluate_blocks(p)
    with capi.DeviceProblem(p) as dp:
        out = dp.evaluate()
        valid = dp.validate_observations(25.0, 0.0)
    gs = p.frame_global[p.obs_frame] == 1
    assert gs.sum() > 1000 and ok_ref.all()
    assert np.max(np.abs(out["residuals"] - r_ref) / np.maximum(1.0, np.abs(r_ref))) <= 1e-11
    assert np.max(np.abs(out["jacobians"] - J_ref) / np.maximum(1.0, np.abs(J_ref))) <= 1e-9
    assert np.all(out["jacobians"][gs][:, :, 6:12] == 0.0) and np.any(out["jacobians"][~gs][:, :, 6:12] != 0.0)   # the second pose slot of a one-pose frame is no parameter
    ref_valid = np.array([oracle.validate_obs(p.intrinsics[0], p.poses[f, :1] if p.frame_global[f] else p.poses[f], p.shutter, p.scanlines, p.points[j], xy, 25.0, 0.0)
                          for f, j, xy in zip(p.obs_frame[:3000], p.obs_point[:3000], p.obs_xy[:3000])])
    assert np.array_equal(valid[:3000], ref_valid)
    pd, pc = p.copy(), p.copy()
    with capi.DeviceProblem(pd) as dp:
        s, tr = dp.solve(capi.default_options(max_num_iterations=12))
    s_ref, tr_ref = oracle.solve(pc, oracle.default_options(max_num_iterations=12))
    assert s.num_residual_blocks_reduced == s_ref.num_residual_blocks_reduced and s.num_parameters_reduced == s_ref.num_parameters_reduced
    assert abs(s.initial_cost - s_ref.initial_cost) <= 1e-12 * s_ref.initial_cost
    for a, b in zip(tr[:6], tr_ref[:6]):
        assert abs(a.cost - b.cost) <= 1e-9 * b.cost
    assert abs(s.final_cost - s_ref.final_cost) <= 1e-6 * s_ref.final_cost
    assert np.array_equal(pd.poses[p.frame_global == 1, 1], p.poses[p.frame_global == 1, 1])      # untouched
    assert np.max(np.abs(pd.poses - pc.poses)) <= 1e-5


def test_fresh_handles_share_the_plan_scratch_and_plan_alike(capi):
    """The symbolic phase's host scratch outlives a handle (a fresh handle per call is windowedBA's pattern, VideoSfMHandler.cc:185-214):
    a big problem, a smaller one in the scratch the big one left, the big one again, and once more after rsba_release_host_scratch —
    every plan and every solve must come out as from a first-ever handle (no stale entries from the previous problem's arrays)."""
    big, small = small_scene(frames=40, points=6000, seed=5), small_scene(frames=14, points=500, seed=6)
    opt = capi.default_options(max_num_iterations=6)

    def run(p):
        q = p.copy()
        with capi.DeviceProblem(q) as dp:
            st = dp.plan_stats()
            s, tr = dp.solve(opt)
        return tuple(st[k] for k in ("tiles", "factor_tiles", "tasks", "schur_entries", "schur_chunks", "schur_block_products", "schur_groups")), [t.cost for t in tr], q.poses, q.points

    first_big, first_small = run(big), None
    capi.release_host_scratch()
    first_small = run(small)                      # (a first-ever plan of the small one)
    for p, ref in ((big, first_big), (small, first_small), (big, first_big)):
        got = run(p)
        assert got[0] == ref[0] and got[1] == ref[1]
        assert np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3])
    capi.release_host_scratch()
    got = run(big)
    assert got[0] == first_big[0] and got[1] == first_big[1] and np.array_equal(got[3], first_big[3])


def test_ragged_scenes_and_degenerate_programs(capi, oracle):
    """Edges of the solve: frames without observations, points with one observation or none; a problem without residual blocks; a
    problem whose parameter blocks are all constant (everything goes to fixed_cost, no iteration); one frame seeing three points (rank
    deficient: the LM damping alone keeps it solvable).  Summary and trajectory as the oracle's restatement of ceres::Solve has them."""
    from rsba_amd.problem import BAProblem
    p = small_scene(frames=12, points=300, seed=2)
    q = p.copy()
    keep = ~np.isin(q.obs_frame, [3, 7])
    rng = np.random.default_rng(0)
    for j in rng.choice(q.num_points, 40, replace=False):
        idx = np.flatnonzero((q.obs_point == j) & keep); keep[idx[1:]] = False
    keep &= ~np.isin(q.obs_point, np.arange(10))
    q.obs_xy, q.obs_frame, q.obs_point = q.obs_xy[keep].copy(), q.obs_frame[keep].copy(), q.obs_point[keep].copy()
    compare_solves(capi, oracle, q, iters=12)

    empty = p.copy()
    empty.obs_xy, empty.obs_frame, empty.obs_point = np.zeros((0, 2)), np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.int32)
    const = p.copy()
    const.pose_fixed_mask[:] = 0x3f; const.point_constant = np.ones(const.num_points, dtype=np.uint8)
    for prob in (empty, const):
        a, b = prob.copy(), prob.copy()
        with capi.DeviceProblem(a) as dp:
            s, tr = dp.solve(capi.default_options(max_num_iterations=8))
        r, _ = oracle.solve(b, oracle.default_options(max_num_iterations=8))
        assert (s.num_iterations, s.termination_type, s.num_residual_blocks_reduced, s.num_parameters_reduced) == (r.num_iterations, r.termination_type, r.num_residual_blocks_reduced, r.num_parameters_reduced) == (1, 0, 0, 0)
        assert abs(s.initial_cost - r.initial_cost) <= 1e-12 * max(1.0, r.initial_cost) and s.final_cost == s.initial_cost
        assert abs(s.fixed_cost - r.fixed_cost) <= 1e-12 * max(1.0, r.fixed_cost)
        assert np.array_equal(a.poses, prob.poses) and np.array_equal(a.points, prob.points)

    one = BAProblem(poses=p.poses[:1].copy(), points=p.points[:3].copy(), intrinsics=p.intrinsics.copy(),
                    obs_xy=np.array([[600.0, 300.0], [500.0, 310.0], [640.0, 200.0]]), obs_frame=np.zeros(3, dtype=np.int32), obs_point=np.arange(3, dtype=np.int32))
    for k in range(3):
        one.points[k] = one.poses[0, 0, 3:] + np.array([0.1 * k, 0.2, 9.0])
    # (15 unknowns, 6 residuals: the normal equations are singular but for the damping, so rounding is amplified — the step-by-step
    # 1e-9 does not apply; the contract on the final cost does)
    compare_solves(capi, oracle, one, iters=8, final_tol=1e-6, expect_same_path=False)


@pytest.mark.parametrize("hook,message", [("RSBA_TEST_FAIL_PLAN", "RSBA_TEST_FAIL_PLAN"), ("RSBA_TEST_FAIL_DEVICE_PLAN", "device plan")])
def test_a_failed_plan_is_torn_down_and_a_retry_fails_or_succeeds_afresh(capi, tmp_path, hook, message):
    """A symbolic phase that fails half-way (out of memory, an unsupported size; here a switch of the instrumented library: before the lists
    are built, or the device lists' allocations) must not leave a half-built solver behind that the next call takes for a finished one (it
    would launch kernels on null tables): every retry reports the error again, and once the cause is gone the same handle plans and solves
    as a fresh one does."""
    out = run_hook_case("failed_plan:" + hook, tmp_path)
    assert len(out["errors"]) == 3 and all(e is not None and message in e for e in out["errors"]), out["errors"]
    assert out["same"]


def test_device_blocks_are_cached_between_handles_and_given_back(capi):
    """windowedBA builds a handle per frame (VideoSfMHandler.cc:185-214): the device blocks of a destroyed handle are kept by the
    library (rsba_amd/csrc/devmem.hpp) and handed to the next one — same results, no growth from handle to handle — and
    rsba_release_host_scratch() returns them to the driver."""
    import torch
    def in_use():
        torch.cuda.synchronize()
        free_b, total_b = torch.cuda.mem_get_info(0)
        return total_b - free_b
    capi.release_host_scratch()
    base = in_use()
    results, levels = [], []
    for rep in range(4):
        p = small_scene(frames=24, points=1500, seed=9)
        with capi.DeviceProblem(p) as dp:
            s, _ = dp.solve(capi.default_options(max_num_iterations=6))
        results.append((s.final_cost, p.poses.copy(), p.points.copy()))
        levels.append(in_use())
    for r in results[1:]:
        assert r[0] == results[0][0] and np.array_equal(r[1], results[0][1]) and np.array_equal(r[2], results[0][2])
    assert levels[0] > base                                  # the first handle's blocks are still with the library ...
    assert max(levels[1:]) - levels[0] <= 4 << 20           # ... the next handles live in them
    capi.release_host_scratch()
    assert in_use() - base <= 8 << 20                        # ... and they go back on request
    # streams, events and the loop's pinned block are pooled the same way (they were 1.0 of the 1.1 ms of a destroy): a handle
    # built from the pool, after the pool was given back, and two handles alive at once (each its own streams) solve alike
    p = small_scene(frames=24, points=1500, seed=9)
    q = small_scene(frames=24, points=1500, seed=9)
    with capi.DeviceProblem(p) as dp, capi.DeviceProblem(q) as dq:
        s, _ = dp.solve(capi.default_options(max_num_iterations=6))
        t, _ = dq.solve(capi.default_options(max_num_iterations=6))
    assert s.final_cost == t.final_cost == results[0][0] and np.array_equal(p.poses, results[0][1]) and np.array_equal(q.points, results[0][2])


@pytest.mark.parametrize("case", ["c2", "rejections", "failure", "tolerances", "max_iterations", "huber", "priors", "priors_rejections", "c2_priors", "intrinsics", "intrinsics_rejections", "intrinsics_priors",
                                  "free_ratio", "free_ratio_rejections", "free_ratio_acceleration", "c2_free_ratio", "pose_priors", "pose_priors_rejections", "pose_priors_free_ratio",
                                  "spherical", "spherical_pose_priors", "per_frame_intrinsics", "per_frame_intrinsics_rejections", "mixed_intrinsics_priors"])
def test_device_side_trust_region_equals_the_host_form(capi, monkeypatch, case):
    """accept / reject, the radius update and the convergence tests of the LM loop run in a single-thread kernel, the
    iteration's kernels read the radius from HBM and skip themselves where the host form would not have launched them, the host
    enqueues iterations ahead and looks at the state every other iteration.  Same rules in the same order on IEEE operations: the two
    forms must agree bit for bit — every field of every iteration record, the summary, the parameters — whatever happens on the way
    (rejected steps, termination by each tolerance, by the iteration limit), and whatever the look-ahead."""
    from rsba_amd.scene import make_config
    def problem():
        if case in ("c2", "c2_priors", "c2_free_ratio"):
            p = make_config("C2").problem
            if case != "c2":
                p.prior_kind, p.prior_scale, p.inter_frame_ratio = 2, 25.0, 1.2
                p.prior_frames = np.arange(1, p.num_frames, dtype=np.int32)
            if case == "c2_free_ratio":
                p.inter_frame_ratio, p.ratio_free = 1.0, True
            return p, dict(max_num_iterations=12)
        p = small_scene(frames=24, points=1500, seed=13, outlier_ratio=0.1 if case == "huber" else 0.0)
        if case == "huber":
            p.huber_a = 1.5
            return p, dict(max_num_iterations=25)
        if case.startswith("intrinsics"):            # the shared intrinsics as a parameter block (what BASELINE config 5 has): pseudo frames in the reduced system, virtual records
            p.calibrated = False; p.huber_a = 2.0
            p.intrinsics = p.intrinsics * (1.0 + 1e-3 * np.array([[1, -1, 20, -20, 10, 10, -10, 0.5, -0.5]]))
        if case in ("per_frame_intrinsics", "per_frame_intrinsics_rejections", "mixed_intrinsics_priors"):   # per-frame f.cam blocks (CeresHandler.h:256-264,273-280): several 9-blocks — these problems
            p.calibrated = False; p.huber_a = 2.0                  # keep their records; a candidate's go to a second set (device_state.hpp: rec_alt), so that this loop can run them too (round 6)
            rng = np.random.default_rng(3)
            F = p.num_frames
            fi = np.arange(F, dtype=np.int32) if case != "mixed_intrinsics_priors" else np.concatenate([np.zeros(F // 2, dtype=np.int32), np.arange(1, F - F // 2 + 1, dtype=np.int32)])
            ni = int(fi.max()) + 1
            p.intrinsics = np.tile(p.intrinsics[:1], (ni, 1)) * (1.0 + 1e-3 * rng.normal(size=(ni, 9)) * np.array([[1, 1, 20, 20, 10, 10, 10, 0.5, 0.5]]))
            p.frame_intrinsics = fi
        if case.startswith("priors") or case in ("intrinsics_priors", "mixed_intrinsics_priors"):                # motion priors with a known interFrameRatio (CeresHandler.h:147-185): their cost, blocks and model change are part of every decision
            p.prior_kind, p.prior_scale, p.inter_frame_ratio = 1, 1.0 if case == "priors_rejections" else 10.0, 0.8
            p.prior_frames = np.arange(1, p.num_frames, dtype=np.int32)
        if case.startswith("spherical"):     # the SphericalPrior on the first pose of frame 1 of a session that starts at the origin (CeresHandler.h:36-50,127-130): 1e20 on the residual — the
            p.poses[0] = 0.0; p.poses[1] = 0.0; p.poses[1, :, 3:] += 1e-4   # two forms must still agree bit for bit, the collapse of the trust region once the constraint is met included
            p.spherical_pose_block = 2
        if case.startswith("pose_priors") or case == "spherical_pose_priors":   # GoodPosePrior blocks (CeresHandler.h:188-204): a free priorPoses block per pose, eliminated in closed form beside the points
            rng = np.random.default_rng(5)
            p.pose_prior_block = np.arange(2, 2 * p.num_frames, dtype=np.int32)
            p.pose_prior_values = p.poses.reshape(-1, 6)[p.pose_prior_block] + rng.normal(0, 0.01, (len(p.pose_prior_block), 6))
            p.pose_prior_rotation, p.pose_prior_position = 3.0, 5.0
        if case.startswith("free_ratio") or case == "pose_priors_free_ratio":   # the reference's default with motion priors (CeresHandler.h:161,172,175): the interFrameRatio is a free, lower-bounded block — one more
            p.prior_kind = 2 if case == "free_ratio_acceleration" else 1   # unknown of every decision (its step out of the factorisation's second right-hand side, its candidate, its projected gradient)
            p.prior_scale, p.inter_frame_ratio, p.ratio_free = 1.0 if case == "free_ratio_rejections" else 10.0, 1.0, True
            p.prior_frames = np.arange(1, p.num_frames, dtype=np.int32)
        if case in ("rejections", "failure", "priors_rejections", "intrinsics_rejections", "free_ratio_rejections", "pose_priors_rejections", "per_frame_intrinsics_rejections"):      # a start far from the minimum and a huge first radius: Gauss-Newton steps that overshoot (or never recover)
            rng = np.random.default_rng(2)
            sc = 3.0 if case == "failure" else 2.0
            p.points += rng.normal(0, 0.6 * sc, p.points.shape); p.poses[1:, :, 3:] += rng.normal(0, 0.25 * sc, p.poses[1:, :, 3:].shape)
            p.poses[1:, :, :3] += rng.normal(0, 0.05 * sc, p.poses[1:, :, :3].shape)
            return p, dict(max_num_iterations=30, initial_trust_region_radius=1e12)
        if case == "tolerances":
            return p, dict(max_num_iterations=50)
        if case in ("priors", "intrinsics", "intrinsics_priors", "free_ratio", "free_ratio_acceleration", "pose_priors", "pose_priors_free_ratio", "spherical", "spherical_pose_priors", "per_frame_intrinsics", "mixed_intrinsics_priors"):
            return p, dict(max_num_iterations=15)
        return p, dict(max_num_iterations=3)
    out = {}
    for mode, env in (("host", {"RSBA_DEVICE_LM": "0"}), ("device", {}), ("device_ahead_1", {"RSBA_LM_AHEAD": "1"}), ("device_ahead_5", {"RSBA_LM_AHEAD": "5"})):
        for k in ("RSBA_DEVICE_LM", "RSBA_LM_AHEAD"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        p, kw = problem()
        with capi.DeviceProblem(p) as dp:
            s, tr = dp.solve(capi.default_options(**kw))
            st = dp.plan_stats()
        assert (st["device_loop_solves"], st["host_loop_solves"]) == ((0, 1) if mode == "host" else (1, 0)), (mode, st)   # the form that was asked for is the form that ran
        rec = [(t.iteration, t.step_is_valid, t.step_is_successful, t.cost, t.cost_change, t.gradient_max_norm, t.step_norm, t.relative_decrease, t.trust_region_radius, t.model_cost_change) for t in tr]
        out[mode] = (rec, (s.termination_type, s.num_successful_steps, s.num_unsuccessful_steps, s.num_iterations, s.initial_cost, s.final_cost, s.is_solution_usable, float(p.inter_frame_ratio)),
                     p.poses.copy(), p.points.copy(), p.intrinsics.copy(), None if p.pose_prior_values is None else p.pose_prior_values.copy())
    ref = out["host"]
    for mode in ("device", "device_ahead_1", "device_ahead_5"):
        got = out[mode]
        assert got[0] == ref[0], mode
        assert got[1] == ref[1], (mode, got[1], ref[1])
        assert np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3]) and np.array_equal(got[4], ref[4]), mode
        assert (got[5] is None and ref[5] is None) or np.array_equal(got[5], ref[5]), mode   # the solved priorPoses values
    if case in ("rejections", "priors_rejections", "intrinsics_rejections", "free_ratio_rejections", "pose_priors_rejections"):
        assert ref[1][1] >= 5 and ref[1][2] >= 3          # the case does accept and reject steps
    if "free_ratio" in case:
        assert ref[1][7] != 1.0                            # the ratio was solved for
    if case.startswith("pose_priors"):
        p0, _ = problem()
        assert np.max(np.abs(ref[5] - p0.pose_prior_values)) > 1e-5   # ... and the priorPoses blocks moved
    if case == "failure":
        assert ref[1][2] >= 3                              # ... and this one only rejects (invalid or unsuccessful steps to the end)
    if case == "max_iterations":
        assert ref[1][0] == 1 and ref[1][3] == 4      # NO_CONVERGENCE after 3 iterations (+ the record of iteration 0)
    if case == "tolerances":
        assert ref[1][0] == 0


def test_the_large_problem_path_of_the_symbolic_phase_builds_the_same_plan(capi, monkeypatch):
    """Beyond 2 048 tile columns the entry passes of the symbolic phase list the tile pairs that exist before they count (per-thread counters
    as long as the list instead of 4 nt^2 bytes each).  RSBA_PLAN_LISTED_KEYS=1 takes that path at any size: same plan figures, and a
    solve whose every bit is the same."""
    out = {}
    for mode in ("0", "1"):
        monkeypatch.delenv("RSBA_PLAN_LISTED_KEYS", raising=False)
        if mode == "1":
            monkeypatch.setenv("RSBA_PLAN_LISTED_KEYS", "1")
        p = small_scene(frames=60, points=6000, seed=4)     # (4 096 points and more: the passes run on several threads)
        with capi.DeviceProblem(p) as dp:
            s, tr = dp.solve(capi.default_options(max_num_iterations=5))
            st = dp.plan_stats()
        out[mode] = (s.final_cost, [t.cost for t in tr], p.poses.copy(), p.points.copy(), {k: st[k] for k in ("tiles", "factor_tiles", "tasks", "schur_entries", "schur_chunks", "schur_block_products")})
    assert out["0"][4] == out["1"][4]
    assert out["0"][0] == out["1"][0] and out["0"][1] == out["1"][1]
    assert np.array_equal(out["0"][2], out["1"][2]) and np.array_equal(out["0"][3], out["1"][3])


def test_parameter_arrays_assigned_after_create_are_copied_into_the_bound_ones(capi):
    """The handle keeps the HOST ADDRESSES of the parameter arrays it was created with (rsba_solve writes the result back through them,
    as ceres::Solve writes into the caller's blocks).  The binding therefore keeps those arrays alive and copies later assignments
    INTO them: `prob.poses = new; upload_parameters(); solve()` must solve from the new values and leave the result in prob.poses."""
    p = small_scene()
    ref = p.copy()
    x0 = (p.poses.copy(), p.points.copy())
    opts = dict(max_num_iterations=6)
    with capi.DeviceProblem(ref) as dr:
        s_ref, _ = dr.solve(capi.default_options(**opts))
    with capi.DeviceProblem(p) as dp:
        dp.solve(capi.default_options(**opts))                       # moves the bound arrays away from x0
        bound = p.poses
        p.poses = x0[0].astype(np.float64).copy()                    # NEW arrays, as a caller would assign them
        p.points = [list(r) for r in x0[1]]                          # ... even something that is not an array yet
        dp.upload_parameters()
        assert p.poses is bound and np.array_equal(p.poses, x0[0])   # copied into the array the handle knows
        s, _ = dp.solve(capi.default_options(**opts))
        assert s.final_cost == s_ref.final_cost
        assert np.array_equal(p.poses, ref.poses) and np.array_equal(p.points, ref.points)
        p.poses = np.zeros((3, 2, 6))
        with pytest.raises(ValueError):
            dp.upload_parameters()
        p.poses = bound


@pytest.mark.parametrize("case", ["c2", "layers", "intrinsics", "constant_points", "global_shutter", "priors", "blocked_chunks", "blocked_chunks_intrinsics"])
def test_the_symbolic_phase_on_the_device_builds_the_same_plan(capi, monkeypatch, case):
    """The passes of the symbolic phase over observations, points and (point, tile pair) entries run on the device (plan_device.hip:
    stable radix sorts and prefix sums); RSBA_PLAN_DEVICE=0 keeps them on host threads.  Same lists in the same order: the same plan
    figures and a solve whose every bit is the same — with a point seen twice in one frame (a second layer of groups), with the shared
    intrinsics block's virtual slots, with constant points (the reduced program's block count), with one pose per frame."""
    from rsba_amd.scene import make_config
    def problem():
        if case == "c2":
            return make_config("C2").problem
        p = small_scene(frames=40, points=2500, seed=31, rolling=case != "global_shutter")
        if case == "layers":      # some observations twice (another detection of the same point in the same frame): cross-layer entries
            dup = np.arange(0, p.num_observations, 17)
            order = np.argsort(np.concatenate([p.obs_frame, p.obs_frame[dup]]), kind="stable")
            p.obs_xy = np.concatenate([p.obs_xy, p.obs_xy[dup] + 0.25])[order]
            p.obs_point = np.concatenate([p.obs_point, p.obs_point[dup]])[order]
            p.obs_frame = np.concatenate([p.obs_frame, p.obs_frame[dup]])[order]
        if case in ("intrinsics", "blocked_chunks_intrinsics"):
            p.calibrated = False; p.huber_a = 2.0
        if case.startswith("blocked_chunks"):   # the chunk numbering of large problems (blocks of points; here: of 64): the device plan hands it the entry list's segments
            monkeypatch.setenv("RSBA_SCHUR_BLOCK", "64")
        if case == "constant_points":
            p.point_constant = np.zeros(p.num_points, dtype=np.uint8); p.point_constant[::7] = 1
        if case == "priors":
            p.prior_kind, p.prior_scale, p.inter_frame_ratio, p.ratio_free = 1, 10.0, 1.0, True
            p.prior_frames = np.arange(1, p.num_frames, dtype=np.int32)
        return p
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("RSBA_PLAN_DEVICE", mode)
        p = problem()
        with capi.DeviceProblem(p) as dp:
            s, tr = dp.solve(capi.default_options(max_num_iterations=5))
            st = dp.plan_stats()
        keys = ("tiles", "factor_tiles", "levels", "tasks", "schur_entries", "schur_chunks", "schur_block_products", "schur_groups", "schur_group_bytes", "schur_factored_groups", "schur_mfma_issued")
        out[mode] = (s.final_cost, s.num_residual_blocks_reduced, [t.cost for t in tr], p.poses.copy(), p.points.copy(), p.intrinsics.copy(), {k: st[k] for k in keys})
    a, b = out["0"], out["1"]
    assert a[6] == b[6], (a[6], b[6])
    assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2]
    assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5])


@pytest.mark.parametrize("case", ["c2", "small_intrinsics", "few_chunks"])
def test_resident_schur_workgroups_leave_the_same_bits_as_a_workgroup_per_chunk(capi, monkeypatch, case):
    """The Schur kernel's default form keeps 512 workgroups resident; each takes chunk after chunk from per-XCD counters (another XCD's
    when its own eighth of the list is done) with the next chunk's tables loaded under the epilogue (kernels_normal.hip,
    schur_tile_kernel).  RSBA_SCHUR_VARIANT=2 launches a workgroup per chunk, as rounds 2 - 4 did.  Which workgroup forms a chunk's
    partial tile must not show: same partial tiles, same merge order, every bit of the solve the same — with more chunks than resident
    workgroups (C2's 453 < 512 exercises the start-up map only; the blocked numbering of the second case makes thousands of small
    chunks, so that the counters and the stealing run), and with fewer chunks than XCDs."""
    from rsba_amd.scene import make_config
    def problem():
        if case == "c2":
            return make_config("C2").problem
        if case == "few_chunks":
            return small_scene(frames=6, points=300, seed=5, rolling=True)
        p = small_scene(frames=60, points=6000, seed=37, rolling=True)
        p.calibrated = False; p.huber_a = 2.0
        monkeypatch.setenv("RSBA_SCHUR_BLOCK", "16")
        return p
    out = {}
    for variant in ("0", "2"):
        monkeypatch.setenv("RSBA_SCHUR_VARIANT", variant)
        p = problem()
        with capi.DeviceProblem(p) as dp:
            s, tr = dp.solve(capi.default_options(max_num_iterations=6))
            st = dp.plan_stats()
            s2, tr2 = dp.solve(capi.default_options(max_num_iterations=2))   # (a second launch sequence on the same counters)
        out[variant] = (s.final_cost, [t.cost for t in tr], s2.final_cost, p.poses.copy(), p.points.copy(), p.intrinsics.copy(), st["schur_chunks"], st["schur_mfma_issued"])
    a, b = out["0"], out["2"]
    if case == "small_intrinsics":
        assert a[6] > 1024, a[6]
    assert a[6] == b[6] and a[7] == b[7]
    assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2]
    assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5])


@pytest.mark.parametrize("case", ["small", "huber_layers"])
def test_the_fused_projection_and_virtual_record_sweep_leaves_the_same_bits(capi, monkeypatch, case):
    """A problem with ONE shared intrinsics block whose two-pose frames all sit in factored tiles evaluates every observation once for
    the projection pass (the slots' P records) and the virtual-record sweep of the intrinsics pseudo frames together
    (kernels_normal.hip, virtual_project_rc_kernel); RSBA_NO_FUSED_SWEEP=1 keeps project_rc_kernel and virtual_records_rc_kernel
    apart.  Same expressions in the same order: every bit of the solve the same."""
    def problem():
        p = small_scene(frames=48, points=4000, seed=41, rolling=True)
        p.calibrated = False
        if case == "huber_layers":
            p.huber_a = 2.0
            dup = np.arange(0, p.num_observations, 23)   # some points seen twice in a frame: a second layer of groups
            order = np.argsort(np.concatenate([p.obs_frame, p.obs_frame[dup]]), kind="stable")
            p.obs_xy = np.concatenate([p.obs_xy, p.obs_xy[dup] + 0.25])[order]
            p.obs_point = np.concatenate([p.obs_point, p.obs_point[dup]])[order]
            p.obs_frame = np.concatenate([p.obs_frame, p.obs_frame[dup]])[order]
        return p
    out = {}
    for off in ("", "1"):
        if off: monkeypatch.setenv("RSBA_NO_FUSED_SWEEP", off)
        else: monkeypatch.delenv("RSBA_NO_FUSED_SWEEP", raising=False)
        p = problem()
        with capi.DeviceProblem(p) as dp:
            s, tr = dp.solve(capi.default_options(max_num_iterations=6))
        out[off] = (s.final_cost, [t.cost for t in tr], p.poses.copy(), p.points.copy(), p.intrinsics.copy())
    a, b = out[""], out["1"]
    assert a[0] == b[0] and a[1] == b[1]
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
