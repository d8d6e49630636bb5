"""RS-PnP RANSAC hypotheses (solveRSpnp.cpp pnpTask / solveRsPnP / project3dPoints), batched on the
device through rsba_pnp_tasks, against the oracle's per-hypothesis restatement (orc_pnp_task: the same RsBA residual
blocks through the LM restatement).  Skip flags and inlier counts are integers (exact, up to observations within a
float ulp of the threshold); refined poses follow the same LM rules from the same start (SURVEY C.6 tolerances)."""
import numpy as np
import pytest

from helpers import load_golden
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


@pytest.mark.parametrize("idx", [0, 1])
def test_refinement_reaches_the_independent_minimum(capi, idx):
    """the committed scipy minima of an independent numpy model (tests/golden/pnp_solves.json)"""
    c = load_golden("pnp_solves.json")[idx]
    X, xy = np.array(c["object_points"], dtype=np.float32), np.array(c["image_points"], dtype=np.float32)
    subs = np.arange(len(X), dtype=np.int32)[None, :]
    out = capi.pnp_tasks(c["cam"], c["shutter"], c["scanlines"], X, xy, subs, c["init_poses"], max_num_iterations=100, reprojection_error=3.0)
    assert out["status"][0] == 1
    # Ceres' default tolerances, as in the reference (function_tolerance 1e-6): within that of the minimum
    assert 0 <= out["final_cost"][0] - c["expected"]["final_cost"] <= 2e-6 * c["expected"]["final_cost"]
    assert np.max(np.abs(out["poses"][0] - np.array(c["expected"]["poses"]))) <= 1e-3
    assert out["num_inliers"][0] >= 0.95 * len(X)


def test_large_batch_counts_agree_with_the_per_pose_inlier_kernel(capi, oracle):
    """16k hypotheses in one call: the counts of the (hypothesis, point) scoring kernel equal the per-pose inlier lists"""
    sc = pnp_scene(oracle, HORIZONTAL, n=400, outliers=0.2, seed=12)
    rng = np.random.default_rng(2)
    H = 16384
    subs = random_subsets(rng, len(sc["X"]), H, 6)
    out = capi.pnp_tasks(CAM, HORIZONTAL, sc["scan"], sc["X"], sc["xy"], subs, sc["init"], reprojection_error=3.0)
    assert out["status"].min() >= 1 and out["num_inliers"].max() >= 0.8 * (~sc["outlier"]).sum()
    again = capi.pnp_tasks(CAM, HORIZONTAL, sc["scan"], sc["X"], sc["xy"], subs, sc["init"], reprojection_error=3.0)
    assert np.array_equal(out["poses"], again["poses"]) and np.array_equal(out["num_inliers"], again["num_inliers"])   # deterministic
    for h in rng.choice(H, 24, replace=False):
        assert capi.pnp_inliers(CAM, HORIZONTAL, sc["scan"], sc["X"], sc["xy"], out["poses"][h], 3.0).sum() == out["num_inliers"][h]


def test_bad_arguments(capi):
    X = np.zeros((8, 3), dtype=np.float32); xy = np.zeros((8, 2), dtype=np.float32)
    with pytest.raises(capi.RsbaError):
        capi.pnp_tasks(CAM, HORIZONTAL, (0, 1280), X, xy, np.array([[0, 1, 2, 3, 4, 8]], dtype=np.int32), np.zeros(12))
    with pytest.raises(capi.RsbaError):
        capi.pnp_tasks(CAM, HORIZONTAL, (5, 5), X, xy, np.array([[0, 1, 2, 3, 4, 5]], dtype=np.int32), np.zeros(12))


# ---- the whole RANSAC flow through the C++ mirror (include/rsba/solve_rs_pnp.hpp) vs a sequential replay through the oracle ----
class CvRng:
    """cv::RNG restated (multiply-with-carry), as in solve_rs_pnp.hpp"""
    def __init__(self, state=0xffffffff):
        self.state = state
    def next(self):
        self.state = ((self.state & 0xffffffff) * 4164903690 + (self.state >> 32)) & 0xffffffffffffffff
        return self.state & 0xffffffff
    def uniform(self, a, b):
        return a if a == b else int(self.next() % (b - a) + a)


def to_pose(oracle, rvec, tvec):
    return np.concatenate([rvec, oracle.angle_axis_rotate(-np.asarray(rvec), -np.asarray(tvec))])


def from_pose(oracle, pose):
    return pose[:3].copy(), -oracle.angle_axis_rotate(pose[:3], pose[3:])


def ransac_replay(oracle, sc, shutter, init, iterations, err, min_inliers, m, state):
    """solveRsPnPRansac (solveRSpnp.cpp:413-524), single-threaded, every solve by the oracle"""
    n = len(sc["X"])
    mask = np.zeros(n, dtype=bool); mask[:m] = True
    gen = CvRng(state)
    best, best_pose = 0, None
    for _ in range(iterations):
        for _ in range(n):
            i1, i2 = gen.uniform(0, n), gen.uniform(0, n)
            mask[i1], mask[i2] = mask[i2], mask[i1]
        r = oracle.pnp_task(CAM, shutter, sc["scan"], sc["X"], sc["xy"], np.flatnonzero(mask), init, 10, err)
        if r is not None and r["num_inliers"] > best:
            best, best_pose, best_mask = r["num_inliers"], r["poses"], r["mask"]
        if best >= min_inliers:
            break
    if best_pose is None or best < m:
        return None
    idx = np.flatnonzero(best_mask)
    r = oracle.pnp_task(CAM, shutter, sc["scan"], sc["X"], sc["xy"], idx, best_pose, 10, err, drop_coincident=False)
    return (r["poses"] if r["usable"] else best_pose), idx


@pytest.mark.parametrize("min_inliers", [100, 10 ** 6])
def test_ransac_host_program_matches_sequential_oracle_replay(oracle, tmp_path, min_inliers):
    import os, struct, subprocess
    import __graft_entry__ as G
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "pnp_ransac")
    if not os.path.exists(exe):
        G.build()
    sc = pnp_scene(oracle, HORIZONTAL, n=220, outliers=0.3, seed=8)
    init = sc["init"]
    (r1, t1), (r2, t2) = from_pose(oracle, init[0]), from_pose(oracle, init[1])
    # the program converts rvec/tvec to poses itself: start the replay from the same converted values
    init_rt = np.stack([to_pose(oracle, r1, t1), to_pose(oracle, r2, t2)])
    iterations, err, m, state = 60, 3.0, 6, 0x1234567
    with open(tmp_path / "p.bin", "wb") as f:
        f.write(struct.pack("<7i", len(sc["X"]), HORIZONTAL, sc["scan"][0], sc["scan"][1], iterations, min(min_inliers, 2 ** 30), m))
        f.write(struct.pack("<f", err)); f.write(struct.pack("<Q", state))
        f.write(CAM.astype("<f8").tobytes())
        for v in (r1, t1, r2, t2):
            f.write(np.asarray(v, dtype="<f8").tobytes())
        f.write(sc["X"].astype("<f4").tobytes()); f.write(sc["xy"].astype("<f4").tobytes())
    r = subprocess.run([exe, str(tmp_path / "p.bin"), str(tmp_path / "o.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = open(tmp_path / "o.bin", "rb").read()
    v = np.frombuffer(raw[:96], dtype="<f8"); cnt = struct.unpack("<i", raw[96:100])[0]
    inl = np.frombuffer(raw[100:100 + 4 * cnt], dtype="<i4")
    ref = ransac_replay(oracle, sc, HORIZONTAL, init_rt, iterations, err, min_inliers, m, state)
    assert ref is not None
    ref_poses, ref_idx = ref
    assert len(np.setxor1d(inl, ref_idx)) <= 1                     # same winner, same inlier list (one borderline point at most)
    got = np.stack([to_pose(oracle, v[0:3], v[3:6]), to_pose(oracle, v[6:9], v[9:12])])
    assert np.max(np.abs(got - ref_poses)) <= 1e-6
    assert cnt >= 0.8 * (~sc["outlier"]).sum() and np.max(np.abs(got - sc["poses"])) <= 0.05


def run_pnp_program(tmp_path, sc, shutter, vecs, iterations, err, min_inliers, m, state):
    import os, struct, subprocess
    import __graft_entry__ as G
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "pnp_ransac")
    if not os.path.exists(exe):
        G.build()
    with open(tmp_path / "p.bin", "wb") as f:
        f.write(struct.pack("<7i", len(sc["X"]), shutter, sc["scan"][0], sc["scan"][1], iterations, min(min_inliers, 2 ** 30), m))
        f.write(struct.pack("<f", err)); f.write(struct.pack("<Q", state))
        f.write(CAM.astype("<f8").tobytes())
        f.write(np.asarray(vecs, dtype="<f8").tobytes())
        f.write(sc["X"].astype("<f4").tobytes()); f.write(sc["xy"].astype("<f4").tobytes())
    r = subprocess.run([exe, str(tmp_path / "p.bin"), str(tmp_path / "o.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = open(tmp_path / "o.bin", "rb").read()
    v = np.frombuffer(raw[:96], dtype="<f8"); cnt = struct.unpack("<i", raw[96:100])[0]
    return v, np.frombuffer(raw[100:100 + 4 * cnt], dtype="<i4")


def test_solve_rs_pnp_from_zero_vectors_uses_the_global_shutter_initialisation(oracle, tmp_path):
    """solveRSpnp.cpp:111-117: all four vectors zero => cv::solvePnP first (here: DLT + device refinement with shutter GLOBAL),
    both poses start from it; the rolling-shutter solve must then land where it lands from a good guess."""
    sc = pnp_scene(oracle, HORIZONTAL, n=200, outliers=0.0, seed=12)
    v0, flag = run_pnp_program(tmp_path, sc, HORIZONTAL, np.zeros(12), 0, 3.0, 0, 6, 1)
    assert len(flag) == 0                                   # usable
    init = sc["init"]
    (r1, t1), (r2, t2) = from_pose(oracle, init[0]), from_pose(oracle, init[1])
    v1, _ = run_pnp_program(tmp_path, sc, HORIZONTAL, np.concatenate([r1, t1, r2, t2]), 0, 3.0, 0, 6, 1)
    got0 = np.stack([to_pose(oracle, v0[0:3], v0[3:6]), to_pose(oracle, v0[6:9], v0[9:12])])
    got1 = np.stack([to_pose(oracle, v1[0:3], v1[3:6]), to_pose(oracle, v1[6:9], v1[9:12])])
    assert np.max(np.abs(got0 - sc["poses"])) <= 0.05 and np.max(np.abs(got1 - sc["poses"])) <= 0.05
    assert np.max(np.abs(got0 - got1)) <= 2e-2              # (ten LM iterations from two different starts: the same basin)


def test_ransac_from_zero_vectors_finds_the_pose_among_outliers(oracle, tmp_path):
    """solveRSpnp.cpp:437-449: all four vectors zero => global-shutter RANSAC first (sampled DLT poses refined and scored on the
    device in one launch), then the rolling-shutter RANSAC from that start."""
    sc = pnp_scene(oracle, HORIZONTAL, n=220, outliers=0.3, seed=8)
    v, inl = run_pnp_program(tmp_path, sc, HORIZONTAL, np.zeros(12), 60, 3.0, 100, 6, 0x1234567)
    got = np.stack([to_pose(oracle, v[0:3], v[3:6]), to_pose(oracle, v[6:9], v[9:12])])
    assert len(inl) >= 0.8 * (~sc["outlier"]).sum()
    assert np.mean(sc["outlier"][inl]) <= 0.05
    assert np.max(np.abs(got - sc["poses"])) <= 0.05
