"""Row f4 of SURVEY §8f: the pose covariance blocks VideoSfMHandler::BA prints (VideoSfMHandler.cc:602-621,
ceres::Covariance on (p0,p0), (p0,p1), (p1,p1)) — the device path (one solve per unit vector through the tile
Cholesky of the undamped reduced camera system) against the oracle's dense inverse.  fp64: entries within 1e-8 of
the largest entry of the block (the systems are solved by different factorisations)."""
import numpy as np
import pytest

from rsba_amd.problem import apply_gauge_masks
from rsba_amd.scene import make_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from rsba_amd import capi
    return capi


def gauge_fixed_scene(rolling=True, huber=0.0, frames=14, points=600, seed=61, **kw):
    p = make_scene(frames, points, rolling=rolling, seed=seed, outlier_ratio=0.05 if huber else 0.0, **kw).problem
    p.huber_a = huber
    apply_gauge_masks(p, fix_first_n_cameras=1, fix_scale=False)
    p.pose_fixed_mask[-1, -1] |= 0b111000      # position of the last pose: fixes the scale, J^T J is regular
    return p


def check(capi, oracle, p, frames):
    with capi.DeviceProblem(p) as dp:
        for f in frames:
            got = dp.pose_covariance(f)
            ref, ok = oracle.pose_covariance(p, f)
            assert ok
            scale = np.abs(ref).max()
            assert scale > 0
            assert np.abs(got - ref).max() <= 1e-8 * scale, (f, np.abs(got - ref).max() / scale)
            assert np.allclose(got, got.T, rtol=0, atol=1e-9 * scale)
            # the blocks rsba prints
            pp, pe, ee = got[:6, :6], got[:6, 6:], got[6:, 6:]
            assert pp.shape == (6, 6) and (p.poses_per_frame == 1 or (pe.shape == (6, 6) and ee.shape == (6, 6)))


@pytest.mark.parametrize("rolling", [True, False])
@pytest.mark.parametrize("huber", [0.0, 2.0])
def test_covariance_blocks_match_the_oracle(capi, oracle, rolling, huber):
    p = gauge_fixed_scene(rolling=rolling, huber=huber)
    check(capi, oracle, p, [1, 5, p.num_frames - 2])


def test_fixed_coordinates_have_zero_covariance(capi, oracle):
    p = gauge_fixed_scene()
    with capi.DeviceProblem(p) as dp:
        c0 = dp.pose_covariance(0)                       # constant block (fixFirstNCameras = 1)
        cl = dp.pose_covariance(p.num_frames - 1)        # last frame: position of its second pose fixed
    assert np.all(c0 == 0)
    ref, ok = oracle.pose_covariance(p, p.num_frames - 1)
    assert ok and np.all(cl[9:, :] == 0) and np.all(cl[:, 9:] == 0) and np.abs(cl[:9, :9]).max() > 0
    assert np.abs(cl - ref).max() <= 1e-8 * np.abs(ref).max()


def test_covariance_after_a_solve_and_with_shared_intrinsics(capi, oracle):
    p = gauge_fixed_scene(frames=12, points=500, seed=62)
    p.calibrated = False
    with capi.DeviceProblem(p) as dp:
        dp.solve(capi.default_options(max_num_iterations=10))
        got = dp.pose_covariance(4)
    ref, ok = oracle.pose_covariance(p, 4)               # p holds the adjusted parameters now
    # (intrinsics in the problem make J^T J much worse conditioned: two different factorisations agree to ~3e-7 here)
    assert ok and np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max()


def test_rank_deficient_problem_is_refused(capi, oracle):
    p = make_scene(8, 200, rolling=True, seed=63).problem      # no gauge fixed at all
    with capi.DeviceProblem(p) as dp:
        with pytest.raises(capi.RsbaError):
            dp.pose_covariance(3)


@pytest.mark.parametrize("kind", [1, 2])
def test_covariance_with_the_free_inter_frame_ratio(capi, oracle, kind):
    """The reference's default with motion priors: the interFrameRatio is a free parameter block coupled to every pose, so
    ceres::Covariance inverts J^T J including its column.  The device path adds the bordered correction
    S^-1 + v v^T / (h - b.v); the constant-ratio covariance of the same scene is measurably different."""
    p = gauge_fixed_scene(frames=12, points=500, seed=64)
    p.prior_kind, p.prior_scale, p.inter_frame_ratio, p.ratio_free = kind, 9.0, 1.0, True
    p.prior_frames = np.arange(1, p.num_frames, dtype=np.int32)
    check(capi, oracle, p, [1, 6, p.num_frames - 1])
    q = p.copy(); q.ratio_free = False
    with capi.DeviceProblem(p) as dp, capi.DeviceProblem(q) as dq:
        a, b = dp.pose_covariance(6), dq.pose_covariance(6)
    assert np.abs(a - b).max() > 1e-6 * np.abs(a).max()
