"""RS-PnP RANSAC hypotheses (solveRSpnp.cpp pnpTask / solveRsPnP / project3dPoints), batched on the
device through rsba_pnp_tasks, against the oracle's per-hypothesis restatement (orc_pnp_task: the same RsBA residual
blocks through the LM restatement).  Skip flags and inlier counts are integers (exact, up to observations within a
float ulp of the threshold); refined poses follow the same LM rules from the same start (SURVEY C.6 tolerances)."""
import numpy as np
import pytest

from rsba_amd.problem import GLOBAL, HORIZONTAL, VERTICAL

pytestmark = pytest.mark.gpu

CAM = np.array([800.0, 800.0, -0.05, 0.01, 1e-3, -1e-3, 2e-3, 640.0, 360.0])


@pytest.fixture(scope="module")
def capi():
    from rsba_amd import capi
    return capi


def pnp_scene(oracle, shutter, n=240, outliers=0.25, seed=3):
    """One rolling-shutter frame seeing n points; observations by the reference's own fixed point (reproject)."""
    rng = np.random.default_rng(seed)
    scan = (0, 1280) if shutter != VERTICAL else (0, 720)
    pose0 = np.concatenate([rng.normal(0, 0.05, 3), rng.normal(0, 0.3, 3)])
    pose1 = pose0 + np.concatenate([rng.normal(0, 0.01, 3), [0.35, 0.05, -0.04]])
    poses = np.stack([pose0, pose1])
    X = np.stack([rng.uniform(-5, 5, 4 * n), rng.uniform(-3, 3, 4 * n), rng.uniform(6, 14, 4 * n)], axis=1).astype(np.float32)
    xy, keep = [], []
    for j in range(len(X)):
        ok, p = oracle.reproject(CAM, poses, shutter, scan, X[j].astype(np.float64), 1e12)
        if ok and 0 < p[0] < 1280 and 0 < p[1] < 720:
            keep.append(j); xy.append(p)
        if len(keep) == n:
            break
    X, xy = X[keep], np.array(xy)
    xy += rng.normal(0, 0.4, xy.shape)
    bad = rng.random(len(xy)) < outliers
    xy[bad] += rng.normal(0, 40.0, (bad.sum(), 2))
    init = poses + np.concatenate([rng.normal(0, 0.01, (2, 3)), rng.normal(0, 0.08, (2, 3))], axis=1)
    return dict(scan=scan, poses=poses, X=X, xy=xy.astype(np.float32), init=init, outlier=bad)


def random_subsets(rng, n, H, m):
    return np.stack([rng.choice(n, m, replace=False) for _ in range(H)]).astype(np.int32)


@pytest.mark.parametrize("shutter", [HORIZONTAL, VERTICAL, GLOBAL])
def test_hypotheses_match_oracle(capi, oracle, shutter):
    sc = pnp_scene(oracle, shutter)
    rng = np.random.default_rng(9)
    H, m = 96, 6
    subs = random_subsets(rng, len(sc["X"]), H, m)
    out = capi.pnp_tasks(CAM, shutter, sc["scan"], sc["X"], sc["xy"], subs, sc["init"], max_num_iterations=10, reprojection_error=3.0)
    assert out["status"].min() >= 1
    worst_pose, exact = 0.0, 0
    for h in range(H):
        ref = oracle.pnp_task(CAM, shutter, sc["scan"], sc["X"], sc["xy"], subs[h], sc["init"], 10, 3.0)
        assert ref is not None and ref["usable"] == (out["status"][h] == 1)
        assert abs(out["final_cost"][h] - ref["final_cost"]) <= 1e-7 * max(1.0, ref["final_cost"]), (h, out["final_cost"][h], ref["final_cost"])
        worst_pose = max(worst_pose, float(np.max(np.abs(out["poses"][h] - ref["poses"]))))
        # inlier counts: identical unless some observation sits within rounding of the threshold for the two poses
        assert abs(int(out["num_inliers"][h]) - ref["num_inliers"]) <= 1
        exact += int(out["num_inliers"][h]) == ref["num_inliers"]
    assert worst_pose <= 1e-6, worst_pose
    assert exact >= H - 2
    # all-inlier subsets find the pose: the best hypothesis explains (nearly) all true inliers
    best = int(np.argmax(out["num_inliers"]))
    if shutter != VERTICAL:   # (the functor takes tau from x even for a VERTICAL shutter — reference quirk — so its fits are poor there)
        assert out["num_inliers"][best] >= 0.8 * (~sc["outlier"]).sum()
    mask = capi.pnp_inliers(CAM, shutter, sc["scan"], sc["X"], sc["xy"], out["poses"][best], 3.0)
    assert mask.sum() == out["num_inliers"][best]
    ref = oracle.pnp_task(CAM, shutter, sc["scan"], sc["X"], sc["xy"], subs[best], sc["init"], 10, 3.0)
    assert (mask != ref["mask"]).sum() <= 1


def test_coincident_points_skip_the_hypothesis_and_per_task_initial_poses(capi, oracle):
    sc = pnp_scene(oracle, HORIZONTAL, n=60, outliers=0.0)
    X = sc["X"].copy(); X[7] = X[3]                     # two identical 3-D points
    subs = np.array([[0, 1, 2, 3, 4, 7], [0, 1, 2, 3, 4, 5], [10, 11, 12, 13, 14, 15]], dtype=np.int32)
    inits = np.stack([sc["init"].reshape(12), sc["init"].reshape(12) * 1.0, sc["poses"].reshape(12)])
    out = capi.pnp_tasks(CAM, HORIZONTAL, sc["scan"], X, sc["xy"], subs, inits, reprojection_error=2.0)
    assert list(out["status"]) == [0, 1, 1]
    assert np.all(out["poses"][0] == 0) and out["num_inliers"][0] == 0          # nothing written for a skipped task
    assert oracle.pnp_task(CAM, HORIZONTAL, sc["scan"], X, sc["xy"], subs[0], inits[0], 10, 2.0) is None
    for h in (1, 2):
        ref = oracle.pnp_task(CAM, HORIZONTAL, sc["scan"], X, sc["xy"], subs[h], inits[h], 10, 2.0)
        assert np.max(np.abs(out["poses"][h] - ref["poses"])) <= 1e-6 and out["num_inliers"][h] == ref["num_inliers"]


def test_refinement_on_many_points_and_zero_iterations(capi, oracle):
    """The final solveRsPnP over all inliers (solveRSpnp.cpp:484-506) is the same call with m = their number."""
    sc = pnp_scene(oracle, HORIZONTAL, n=200, outliers=0.0, seed=5)
    subs = np.arange(200, dtype=np.int32)[None, :]
    out = capi.pnp_tasks(CAM, HORIZONTAL, sc["scan"], sc["X"], sc["xy"], subs, sc["init"], max_num_iterations=10, reprojection_error=2.0)
    ref = oracle.pnp_task(CAM, HORIZONTAL, sc["scan"], sc["X"], sc["xy"], subs[0], sc["init"], 10, 2.0)
    assert out["status"][0] == 1 and abs(out["final_cost"][0] - ref["final_cost"]) <= 1e-8 * ref["final_cost"]
    assert np.max(np.abs(out["poses"][0] - ref["poses"])) <= 1e-7
    assert np.max(np.abs(out["poses"][0] - sc["poses"])) <= 0.02 and out["num_inliers"][0] >= 190
    out0 = capi.pnp_tasks(CAM, HORIZONTAL, sc["scan"], sc["X"], sc["xy"], subs, sc["init"], max_num_iterations=0, reprojection_error=2.0)
    assert np.array_equal(out0["poses"][0], sc["init"])


def test_bad_arguments(capi):
    X = np.zeros((8, 3), dtype=np.float32); xy = np.zeros((8, 2), dtype=np.float32)
    with pytest.raises(capi.RsbaError):
        capi.pnp_tasks(CAM, HORIZONTAL, (0, 1280), X, xy, np.array([[0, 1, 2, 3, 4, 8]], dtype=np.int32), np.zeros(12))
    with pytest.raises(capi.RsbaError):
        capi.pnp_tasks(CAM, HORIZONTAL, (5, 5), X, xy, np.array([[0, 1, 2, 3, 4, 5]], dtype=np.int32), np.zeros(12))
