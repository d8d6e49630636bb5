"""Parity of the batched reprojection / validation filter (struct/VideoSfM.cc:103-169)
with the oracle's scalar restatement, through the C ABI.  Validity flags are exact; reprojected pixels are
fp64 fixed points and must agree to 1e-9 px."""
import numpy as np
import pytest

from rsba_amd.problem import GLOBAL, HORIZONTAL, VERTICAL
from rsba_amd.scene import make_config, make_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from rsba_amd import capi
    return capi


def oracle_validate(oracle, prob, idx, thr, min_dist):
    out = np.zeros(len(idx), dtype=bool)
    for n, i in enumerate(idx):
        f, j = prob.obs_frame[i], prob.obs_point[i]
        cam = prob.intrinsics[0 if prob.frame_intrinsics is None else prob.frame_intrinsics[f]]
        out[n] = oracle.validate_obs(cam, prob.poses[f], prob.shutter, prob.scanlines, prob.points[j], prob.obs_xy[i],
                                     thr, min_dist, prob.interpolate_rotation)
    return out


@pytest.mark.parametrize("shutter", [GLOBAL, HORIZONTAL, VERTICAL])
@pytest.mark.parametrize("interp", [True, False])
def test_validate_matches_oracle(capi, oracle, shutter, interp):
    sc = make_scene(30, 1500, rolling=True, seed=11, outlier_ratio=0.1, noise_px=1.0, rot_noise=0.002, pos_noise=0.01, pt_noise=0.01)
    prob = sc.problem
    prob.shutter, prob.interpolate_rotation = shutter, interp
    if shutter == VERTICAL:
        prob.scanlines = (0, 720)
    thr, min_dist = 25.0, 9.5          # both tests bite: ~half the points are nearer than 9.5 at some frame
    rng = np.random.default_rng(5)
    idx = rng.choice(prob.num_observations, 3000, replace=False)
    with capi.DeviceProblem(prob) as dp:
        got = dp.validate_observations(thr, min_dist)
    ref = oracle_validate(oracle, prob, idx, thr, min_dist)
    assert 0.05 < ref.mean() < 0.95
    # a flag may legitimately differ only where a test sits on its threshold to rounding; none do here
    assert np.array_equal(got[idx], ref)


def test_validate_global_shutter_single_pose_and_unsorted_input(capi, oracle):
    sc = make_scene(12, 800, rolling=False, seed=3, outlier_ratio=0.2, all_visible=True)
    prob = sc.problem
    perm = np.random.default_rng(0).permutation(prob.num_observations)
    prob.obs_xy, prob.obs_frame, prob.obs_point = prob.obs_xy[perm], prob.obs_frame[perm], prob.obs_point[perm]
    with capi.DeviceProblem(prob) as dp:
        got = dp.validate_observations(16.0, 0.0)
    ref = oracle_validate(oracle, prob, np.arange(prob.num_observations)[:2500], 16.0, 0.0)
    assert np.array_equal(got[:2500], ref)


def test_points_behind_the_camera_are_invalid(capi, oracle):
    sc = make_scene(10, 300, rolling=True, seed=2)
    prob = sc.problem
    prob.points[::7, 2] = -5.0
    with capi.DeviceProblem(prob) as dp:
        got = dp.validate_observations(1e30, 0.0)
    behind = (np.arange(prob.num_points) % 7 == 0)[prob.obs_point]
    assert not got[behind].any() and got[~behind].all()


@pytest.mark.parametrize("shutter", [GLOBAL, HORIZONTAL, VERTICAL])
def test_reproject_matches_oracle(capi, oracle, shutter):
    sc = make_scene(25, 1000, rolling=True, seed=17, intra_frame=0.8)
    prob = sc.problem
    prob.shutter = shutter
    if shutter == VERTICAL:
        prob.scanlines = (0, 720)
    prob.points[::50, 2] = -3.0        # some pairs fail in w2i
    rng = np.random.default_rng(9)
    idx = rng.choice(prob.num_observations, 2000, replace=False)
    fr, pt = prob.obs_frame[idx], prob.obs_point[idx]
    with capi.DeviceProblem(prob) as dp:
        xy, ok = dp.reproject(fr, pt)
    nfail = 0
    for n in range(len(idx)):
        ok_ref, xy_ref = oracle.reproject(prob.intrinsics[0], prob.poses[fr[n]], shutter, prob.scanlines, prob.points[pt[n]], 1.0,
                                          prob.interpolate_rotation)
        assert ok[n] == ok_ref
        nfail += not ok_ref
        if ok_ref:
            assert np.abs(xy[n] - xy_ref).max() <= 1e-9 * max(1.0, np.abs(xy_ref).max())
    assert 0 < nfail < len(idx) // 10


def test_reproject_is_a_fixed_point_that_validates(capi):
    """Size-independent property at C2 size: synthesising observations with reproject and validating them
    against the same parameters accepts every one with zero error (what createTracks relies on)."""
    sc = make_config("C2")
    prob = sc.problem
    with capi.DeviceProblem(prob) as dp:
        xy, ok = dp.reproject(prob.obs_frame, prob.obs_point)
    assert ok.all()
    prob.obs_xy = xy
    with capi.DeviceProblem(prob) as dp:
        out = dp.evaluate(jacobians=False)
        valid = dp.validate_observations(1e-4, 0.0)
    assert valid.all()
    # HORIZONTAL shutter: the functor's tau-from-x equals the true tau, so residuals vanish at the fixed point
    # up to the 1e-3 px convergence threshold of the iteration
    assert np.abs(out["residuals"]).max() < 1e-2


def test_reproject_rejects_bad_indices_and_accepts_empty(capi):
    sc = make_scene(5, 50, rolling=True, seed=1)
    with capi.DeviceProblem(sc.problem) as dp:
        xy, ok = dp.reproject([], [])
        assert xy.shape == (0, 2) and ok.shape == (0,)
        with pytest.raises(capi.RsbaError):
            dp.reproject([5], [0])
        with pytest.raises(capi.RsbaError):
            dp.reproject([0], [-1])
