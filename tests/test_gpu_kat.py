"""The reference's own known-answer inputs (src/rsba/test/mat_test.cc:171-214: 11 poses x 18 points x 6 cameras, with the
+-_EPS poses / points and the k1 = +-_EPS cameras) pushed through the HIP path — the global-shutter functor of
rsba_evaluate (residual + observation = w2i), rsba_reproject_frame and rsba_validate_frame — and held against the oracle,
which tests/test_oracle_kat.py pins on the same table.  Pins rows a5-a8 of SURVEY §8 (w2i, w2c, c2i, distort) on the device."""
import numpy as np
import pytest

from helpers import rel_err
from kat_tables import CAMS, POSE_REF, POSES, PTS, deep_cases
from rsba_amd.problem import BAProblem, GLOBAL

pytestmark = pytest.mark.gpu

SCAN = (0, 1280)


@pytest.fixture(scope="module")
def capi():
    from rsba_amd import capi
    assert capi.device_count() >= 1
    return capi


def table():
    """every (pose, cam) pair as one global-shutter frame with its own baked intrinsics, every point seen from every frame"""
    poses = np.array([[p] for p in POSES for _ in CAMS], dtype=np.float64)                       # [66, 1, 6]
    frame_cam = np.array([c for _ in POSES for c in range(len(CAMS))], dtype=np.int32)
    fr, pt = np.meshgrid(np.arange(len(poses)), np.arange(len(PTS)), indexing="ij")
    return poses, frame_cam, fr.ravel().astype(np.int32), pt.ravel().astype(np.int32)


def test_w2i_of_the_reference_table_through_the_functor(capi, oracle):
    poses, frame_cam, fr, pt = table()
    pts, cams = np.array(PTS, dtype=np.float64), np.array(CAMS, dtype=np.float64)
    ok_ref = np.zeros(len(fr), dtype=bool); img_ref = np.zeros((len(fr), 2)); z = np.zeros(len(fr))
    for i, (f, j) in enumerate(zip(fr, pt)):
        ok_ref[i], img_ref[i] = oracle.w2i(cams[frame_cam[f]], poses[f, 0], pts[j])
        z[i] = oracle.w2c(poses[f, 0], pts[j])[2]
    assert np.array_equal(ok_ref, ~(z < 1e-8))                          # mat/cam.h:410-412, oracle side
    assert 100 < (~ok_ref).sum() < len(fr) - 100
    off = np.array([0.25, -0.5])
    obs = np.where(ok_ref[:, None], img_ref + off, 0.0)
    prob = BAProblem(poses=poses, points=pts, intrinsics=cams, obs_xy=obs, obs_frame=fr, obs_point=pt, shutter=GLOBAL, scanlines=SCAN,
                     interpolate_rotation=True, calibrated=True, frame_intrinsics=frame_cam)
    with capi.DeviceProblem(prob) as dp:
        out = dp.evaluate()
    # (i) the functor fails exactly where w2c(...).z < 1e-8, and reports such blocks as zeros
    assert out["num_failed"] == int((~ok_ref).sum())
    assert np.all(out["residuals"][~ok_ref] == 0) and np.all(out["jacobians"][~ok_ref] == 0)
    assert np.all(np.any(out["jacobians"][ok_ref] != 0, axis=(1, 2)))
    # (iii) residual + observation = w2i, equal to the oracle's to 1e-12 (relative to max(1, |value|): pixels reach 1e18 on the
    # z ~ 1e-8 rows); residual blocks: r and the 2 x 9 Jacobian against the oracle's dual numbers
    assert rel_err(out["residuals"][ok_ref] + obs[ok_ref], img_ref[ok_ref]) <= 1e-12
    r_ref, J_ref, ok_blocks = oracle.evaluate_blocks(prob)
    assert np.array_equal(ok_blocks, ok_ref)
    mag = np.maximum(1.0, np.abs(img_ref[ok_ref]))                     # a residual is pixel - observation: its rounding scales with the pixel (1e18 on the z ~ 1e-8 rows)
    assert np.max(np.abs(out["residuals"][ok_ref] - r_ref[ok_ref]) / mag) <= 1e-12
    scale = np.maximum(1.0, np.abs(J_ref[ok_ref]).max(axis=(1, 2), keepdims=True))   # per block: the z ~ 1e-8 rows have entries of 1e26
    assert np.max(np.abs(out["jacobians"][ok_ref] - J_ref[ok_ref]) / scale) <= 1e-9


def test_reproject_and_validate_frame_on_the_reference_table(capi, oracle):
    pts = np.array(PTS, dtype=np.float64)
    nvalid = 0
    for pose in POSES + [POSE_REF]:
        for cam in CAMS:
            xy, ok = capi.reproject_frame(cam, [pose], GLOBAL, SCAN, pts)
            ref = [oracle.w2i(cam, pose, X) for X in pts]
            ok_ref = np.array([r[0] for r in ref]); xy_ref = np.array([r[1] for r in ref])
            assert np.array_equal(ok, ok_ref)
            if ok_ref.any():
                assert rel_err(xy[ok_ref], xy_ref[ok_ref]) <= 1e-12
            # validate(cam, pose, observation, point, 1.0) flag for flag, on exact, slightly off and far-off observations
            for shift in (0.0, 0.6, 0.8, 30.0):
                obs = np.where(ok_ref[:, None], xy_ref + np.array([shift, -shift]), 0.0)
                got = capi.validate_frame(cam, [pose], GLOBAL, SCAN, pts, obs, 1.0, 0.0)
                want = np.array([oracle.validate(cam, pose, o, X, 1.0) for o, X in zip(obs, pts)])
                finite = np.all(np.abs(xy_ref) < 1e12, axis=1)          # (2 * 0.6^2 = 0.72 < 1 passes, 2 * 0.8^2 = 1.28 fails — where pixels are small enough to resolve the shift)
                assert np.array_equal(got[finite], want[finite])
                nvalid += int(got.sum())
    assert nvalid > 500


def test_validate_holds_where_the_reference_test_asserts_it(capi, oracle):
    """mat_test.cc:280-281: validate(cam, pose, img, pt, 1.0) and validate(cam, poseRef, imgRef, pt, 1.0) on every triple that
    reaches those lines."""
    n = 0
    for pose, pt, cam, img, img_ref in deep_cases(oracle):
        assert capi.validate_frame(cam, [pose], GLOBAL, SCAN, [pt], [img], 1.0, 0.0)[0]            # :280
        assert capi.validate_frame(cam, [POSE_REF], GLOBAL, SCAN, [pt], [img_ref], 1.0, 0.0)[0]    # :281
        n += 1
    assert n > 50
