"""Parity of the HIP residual+Jacobian path (through the C ABI) with the CPU oracle and the committed
golden vectors.  fp64: residuals within 1e-11, Jacobians within 1e-9, both relative to max(1,|value|)
(the analytic Jacobian and the oracle's dual numbers are two exact derivatives of the same expression;
they differ by rounding only)."""
import numpy as np
import pytest

from helpers import batch_cases, load_golden, rel_err

pytestmark = pytest.mark.gpu

R_TOL, J_TOL = 1e-11, 1e-9
J_TOL_TINY_ANGLE = 1e-6   # see tests/test_oracle_golden.py


@pytest.fixture(scope="module")
def capi():
    from rsba_amd import capi
    assert capi.device_count() >= 1
    return capi


def check_against_oracle(capi, oracle, prob, r_tol=R_TOL, j_tol=J_TOL):
    r_ref, J_ref, ok_ref = oracle.evaluate_blocks(prob)
    with capi.DeviceProblem(prob) as dp:
        out = dp.evaluate()
    assert out["num_failed"] == int((~ok_ref).sum())
    good = ok_ref
    assert rel_err(out["residuals"][good], r_ref[good]) <= r_tol
    assert rel_err(out["jacobians"][good], J_ref[good]) <= j_tol
    # failed blocks are reported as zeros
    assert np.all(out["residuals"][~good] == 0) and np.all(out["jacobians"][~good] == 0)
    ok_c, cost_ref, _ = oracle.evaluate(prob, gradient=False)
    if good.all():
        assert abs(out["cost"] - cost_ref) <= 1e-12 * max(1.0, cost_ref)
    return out


def test_golden_vectors(capi):
    cases = load_golden("per_observation.json")
    n = 0
    for prob, idxs in batch_cases(cases):
        with capi.DeviceProblem(prob) as dp:
            out = dp.evaluate()
        assert out["num_failed"] == sum(1 for i in idxs if not cases[i]["ok"])
        for row, i in enumerate(idxs):
            c = cases[i]
            if not c["ok"]:
                assert np.all(out["residuals"][row] == 0) and np.all(out["jacobians"][row] == 0)
                continue
            assert rel_err(out["residuals"][row], c["residual"]) <= R_TOL, c["name"]
            tol = J_TOL_TINY_ANGLE if c["name"].startswith("tiny") else J_TOL
            assert rel_err(out["jacobians"][row], c["jacobian"]) <= tol, (c["name"], np.abs(out["jacobians"][row] - np.array(c["jacobian"])).max())
            n += 1
    assert n >= 40


@pytest.mark.parametrize("rolling", [True, False])
@pytest.mark.parametrize("calibrated", [True, False])
@pytest.mark.parametrize("interp", [True, False])
def test_functor_shapes_vs_oracle(capi, oracle, rolling, calibrated, interp):
    from rsba_amd.scene import make_scene
    sc = make_scene(24, 1200, rolling=rolling, seed=7 + rolling + 2 * calibrated)
    p = sc.problem
    p.calibrated = calibrated
    p.interpolate_rotation = interp
    check_against_oracle(capi, oracle, p)


@pytest.mark.parametrize("shutter", [0, 1, 2])
def test_shutter_modes_with_two_poses(capi, oracle, shutter):
    from rsba_amd.scene import make_scene
    p = make_scene(10, 400, seed=3).problem
    p.shutter = shutter
    p.scanlines = (17, 1203)
    check_against_oracle(capi, oracle, p)


def test_config_c2_full(capi, oracle):
    from rsba_amd.scene import make_config
    p = make_config("C2").problem
    out = check_against_oracle(capi, oracle, p)
    assert out["residuals"].shape[0] > 150_000


def test_config_c1_global_shutter_pinhole(capi, oracle):
    from rsba_amd.scene import make_config
    check_against_oracle(capi, oracle, make_config("C1").problem)


def test_huber_cost(capi, oracle):
    from rsba_amd.scene import make_scene
    p = make_scene(20, 800, outlier_ratio=0.1, seed=5).problem
    p.huber_a = 2.0
    out = check_against_oracle(capi, oracle, p)
    p2 = p.copy(); p2.huber_a = 0.0
    assert out["cost"] < oracle.evaluate(p2, gradient=False)[1]


def test_unsorted_observations_and_per_frame_intrinsics(capi, oracle):
    from rsba_amd.scene import make_scene
    p = make_scene(16, 700, seed=9).problem
    rng = np.random.default_rng(1)
    perm = rng.permutation(p.num_observations)
    p.obs_xy, p.obs_frame, p.obs_point = p.obs_xy[perm].copy(), p.obs_frame[perm].copy(), p.obs_point[perm].copy()
    p.intrinsics = np.tile(p.intrinsics, (p.num_frames, 1)) * (1.0 + 0.01 * rng.normal(size=(p.num_frames, 1)))
    p.frame_intrinsics = rng.permutation(p.num_frames).astype(np.int32)
    p.calibrated = False
    check_against_oracle(capi, oracle, p)


def test_points_behind_camera_fail_like_the_functor(capi, oracle):
    from rsba_amd.scene import make_scene
    p = make_scene(8, 300, seed=2).problem
    p.points[::7, 2] = -3.0     # behind every camera -> functor returns false (mat/cam.h:410-412)
    r_ref, J_ref, ok_ref = oracle.evaluate_blocks(p)
    assert (~ok_ref).sum() > 10
    check_against_oracle(capi, oracle, p)


def test_ragged_and_tiny_inputs(capi, oracle):
    from rsba_amd.problem import BAProblem
    from rsba_amd.scene import make_scene
    # frames without observations, points without observations, N not a multiple of the wave size
    p = make_scene(9, 120, seed=4).problem
    keep = (p.obs_frame != 3) & (p.obs_point % 5 != 0)
    p.obs_xy, p.obs_frame, p.obs_point = p.obs_xy[keep][:257].copy(), p.obs_frame[keep][:257].copy(), p.obs_point[keep][:257].copy()
    check_against_oracle(capi, oracle, p)
    # a single observation; and an empty observation list
    one = BAProblem(poses=p.poses[:1], points=p.points[:1], intrinsics=p.intrinsics, obs_xy=np.array([[600.0, 300.0]]),
                    obs_frame=np.array([0]), obs_point=np.array([0]))
    one.points[0] = one.poses[0, 0, 3:] + np.array([0.1, 0.2, 9.0])
    check_against_oracle(capi, oracle, one)
    empty = BAProblem(poses=p.poses, points=p.points, intrinsics=p.intrinsics, obs_xy=np.zeros((0, 2)),
                      obs_frame=np.zeros(0, dtype=np.int32), obs_point=np.zeros(0, dtype=np.int32))
    with capi.DeviceProblem(empty) as dp:
        out = dp.evaluate()
    assert out["cost"] == 0.0 and out["num_failed"] == 0 and out["residuals"].shape == (0, 2)


def test_many_frames_per_workgroup_falls_back_to_global_poses(capi, oracle):
    """< 16 observations per frame: a 256-lane workgroup spans more camera blocks than LDS staging holds."""
    from rsba_amd.scene import make_scene
    p = make_scene(200, 150, seed=6, track_len=12).problem
    assert p.num_observations / p.num_frames < 16
    check_against_oracle(capi, oracle, p)


def test_invalid_arguments_are_rejected(capi):
    from rsba_amd.scene import make_scene
    p = make_scene(4, 40).problem
    p.obs_point = p.obs_point.copy(); p.obs_point[0] = p.num_points + 3
    with pytest.raises(capi.RsbaError):
        capi.DeviceProblem(p)


def test_config_c4_full_size(capi, oracle):
    """BASELINE.json's headline size (1k cameras / 100k points / ~2M observations): bit-identical
    repeat evaluations, the rolling-shutter block structure J_pose1 = tau/(1-tau) J_pose0, and a full
    comparison with the oracle (a few seconds of CPU)."""
    from rsba_amd.scene import make_config
    p = make_config("C4").problem
    assert p.num_observations > 1_500_000
    with capi.DeviceProblem(p) as dp:
        a = dp.evaluate()
        b = dp.evaluate()
    assert a["cost"] == b["cost"] and np.array_equal(a["jacobians"], b["jacobians"])
    tau = np.clip((p.obs_xy[:, 0] - p.scanlines[0]) / float(p.scanlines[1] - p.scanlines[0]), 0, 1)
    J = a["jacobians"]
    lhs = J[:, :, 6:12] * (1.0 - tau)[:, None, None]
    rhs = J[:, :, 0:6] * tau[:, None, None]
    assert rel_err(lhs, rhs) <= 1e-9
    assert np.allclose(J[:, :, 12:15], -J[:, :, 3:6] - J[:, :, 9:12], rtol=0, atol=1e-9 * np.abs(J).max())
    r_ref, J_ref, ok_ref = oracle.evaluate_blocks(p)
    assert ok_ref.all() and a["num_failed"] == 0
    # conditioning: the 1k-frame path is 800 m long, so one ulp of a world coordinate (1.1e-13 m) is
    # already 1e-11 px through fx/z ~ 100 px/m; both sides round differently (FMA contraction)
    assert rel_err(a["residuals"], r_ref) <= 1e-10 and rel_err(J, J_ref) <= J_TOL
