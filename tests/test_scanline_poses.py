"""Frames with MORE than two poses ("fullDoF": a pose per scan line) — src/rsba/struct/VideoSfM.cc:83-97 getPose,
src/rsba/CeresHandler.h:266-285: an observation of such a frame is a ReprojectionError block over the pose of its rounded,
clamped scan line.  CPU: the oracle's restatement of the rule (known answers read off the reference's lines) and the product's
lowering (rsba_amd/problem.py) against the oracle's Add loop.  GPU: the per-frame filters with such frames, the solve of a
session that mixes one-, two- and many-pose frames against the oracle's trajectory, and the same session through
CeresHandler::Add / BA() of the C++ host side."""
import os
import struct
import subprocess

import numpy as np
import pytest

from rsba_amd.problem import GLOBAL, HORIZONTAL, VERTICAL, lower_scanline_poses, scatter_scanline_poses

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# VideoSfM.cc:85-96 read line by line: line = obs[0] for HORIZONTAL else obs[1]; < 0 -> 0; > size - 1 -> size - 1; round()
KNOWN = [  # (poses, shutter, (x, y), index)
    (5, HORIZONTAL, (-3.0, 2.0), 0), (5, HORIZONTAL, (0.49, 4.0), 0), (5, HORIZONTAL, (0.5, 4.0), 1), (5, HORIZONTAL, (1.5, 0.0), 2),
    (5, HORIZONTAL, (2.5, 0.0), 3), (5, HORIZONTAL, (3.999, 0.0), 4), (5, HORIZONTAL, (4.0, 0.0), 4), (5, HORIZONTAL, (400.0, 1.0), 4),
    (5, VERTICAL, (400.0, 1.2), 1), (5, VERTICAL, (0.0, 3.5), 4), (5, VERTICAL, (1.0, -0.1), 0),
    (5, GLOBAL, (2.0, 3.4), 3),                     # sess.rs == GLOBAL takes the else branch: y
    (3, HORIZONTAL, (0.49999999999999994, 0.0), 0),  # the largest double below one half still rounds down
    (720, VERTICAL, (10.0, 359.5), 360), (720, VERTICAL, (10.0, 719.4), 719), (720, VERTICAL, (10.0, 719.6), 719),
]


def test_the_oracle_picks_the_pose_the_reference_picks(oracle):
    for n, sh, xy, want in KNOWN:
        assert oracle.scanline_pose_index(n, sh, xy) == want, (n, sh, xy)


def random_session(seed, sizes, n_obs=600, n_points=40):
    rng = np.random.default_rng(seed)
    fp = [rng.normal(size=(n, 6)) for n in sizes]
    of = rng.integers(0, len(sizes), n_obs)
    xy = rng.uniform(-2.0, max(sizes) + 1.0, size=(n_obs, 2))
    xy[:100] = np.round(xy[:100] * 2) / 2          # exact halves and integers
    return fp, of, rng.integers(0, n_points, n_obs), xy, rng.normal(size=(n_points, 3))


@pytest.mark.parametrize("shutter", [GLOBAL, HORIZONTAL, VERTICAL])
@pytest.mark.parametrize("sizes", [(2, 1, 7, 5, 2, 3), (1, 9, 4, 1), (6, 6, 6)])
def test_lowering_equals_the_oracles_add_loop(oracle, shutter, sizes):
    fp, of, op, xy, pts = random_session(3, sizes)
    prob, blocks = lower_scanline_poses(fp, of, op, xy, shutter=shutter, points=pts, intrinsics=np.ones((1, 9)))
    want, which = oracle.add_loop_pose_blocks(fp, of, xy, shutter)
    assert [tuple(b) for b in blocks.tolist()] == [(f, -1 if q is None else q) for f, q in want]
    assert np.array_equal(prob.obs_frame, which)
    assert prob.poses_per_frame == (2 if 2 in sizes else 1)
    for d, (f, q) in enumerate(blocks):
        if q < 0:
            assert np.array_equal(prob.poses[d], fp[f])
        else:
            assert np.array_equal(prob.poses[d, 0], fp[f][q]) and (prob.frame_global is None or prob.frame_global[d] == 1)
    # poses nobody picked are not part of the problem
    assert len(blocks) == len({(f, -1 if len(fp[f]) == 2 else (0 if len(fp[f]) == 1 else oracle.scanline_pose_index(len(fp[f]), shutter, o)))
                               for f, o in zip(of, xy)})


# ---- GPU ----------------------------------------------------------------------------------------------------------------------

@pytest.fixture(scope="module")
def capi():
    from rsba_amd import capi
    assert capi.device_count() >= 1
    return capi


LINES = 33   # poses of a many-pose frame: one per column of a 32-pixel-wide image


def coarse_columns(p):
    """The scene seen through a camera whose x axis has 32 pixels instead of 1280 (fx, cx, the observations' x and the scan-line
    range divided by 40): the reference indexes a frame's poses by the pixel coordinate itself, so a pose per column is 33 poses."""
    p.intrinsics[:, 0] /= 40.0; p.intrinsics[:, 7] /= 40.0
    p.obs_xy[:, 0] /= 40.0
    p.scanlines = (0, 32)
    return p


def scanline_session(p, every=3, single_every=5):
    """Per-frame pose arrays: every `every`-th frame carries LINES poses — samples of its own motion between poses[0] and
    poses[1] —, every `single_every`-th a single pose, the others their two."""
    fp = []
    for f in range(p.num_frames):
        if f % every == every - 1:
            t = np.linspace(0.0, 1.0, LINES)[:, None]
            fp.append(p.poses[f, 0] * (1 - t) + p.poses[f, 1] * t)
        elif single_every and f % single_every == single_every - 1:
            fp.append(p.poses[f, :1].copy())
        else:
            fp.append(p.poses[f].copy())
    return fp


@pytest.mark.gpu
@pytest.mark.parametrize("shutter", [GLOBAL, HORIZONTAL, VERTICAL])
def test_frame_filters_with_a_pose_per_scan_line(capi, oracle, shutter):
    """rsba_validate_frame / rsba_reproject_frame on a frame with 33 poses: per item the pose getPose picks (VideoSfM.cc:118-132),
    re-picked in every step of reproject's fixed point — flags and pixels equal the oracle's."""
    from rsba_amd.scene import make_scene
    sc = make_scene(6, 700, rolling=True, seed=23)
    p = sc.problem
    f = 3
    lines = 33
    t = np.linspace(0.0, 1.0, lines)[:, None]
    poses = p.poses[f, 0] * (1 - t) + p.poses[f, 1] * t
    sel = p.obs_frame == f
    pts = p.points[p.obs_point[sel]].copy()
    pts[::17, 2] = -4.0                                      # some items fail in w2i
    cam = p.intrinsics[0].copy()
    cam[[7, 8]] = [16.0, 16.0]; cam[[0, 1]] = [30.0, 60.0]   # a 32 x 32 "image": projections land on scan lines 0 .. 32 and beyond, in x and in y
    xy, ok = capi.reproject_frame(cam, poses, shutter, (0, 32), pts)
    nfail = nclamp = 0
    for n in range(len(pts)):
        ok_ref, xy_ref = oracle.reproject(cam, poses, shutter, (0, 32), pts[n], 1.0)
        assert bool(ok[n]) == ok_ref
        nfail += not ok_ref
        if ok_ref:
            assert np.abs(xy[n] - xy_ref).max() <= 1e-9 * max(1.0, np.abs(xy_ref).max())
            line = xy_ref[0] if shutter == HORIZONTAL else xy_ref[1]
            nclamp += line < 0 or line > lines - 1
    assert 0 < nfail < len(pts) // 4 and nclamp > 0          # both the failing and the clamped branch ran
    good = ok.astype(bool)
    obs = np.where(good[:, None], xy, 5.0) + np.random.default_rng(1).normal(scale=0.7, size=xy.shape)
    got = capi.validate_frame(cam, poses, shutter, (0, 32), pts, obs, 1.0, 0.3)
    want = np.array([oracle.validate_obs(cam, poses, shutter, (0, 32), pts[n], obs[n], 1.0, 0.3) for n in range(len(pts))])
    assert np.array_equal(got.astype(bool), want) and 0 < want.sum() < len(want)


@pytest.mark.gpu
@pytest.mark.parametrize("huber", [0.0, 2.0])
def test_session_with_pose_per_scan_line_frames_solves_to_the_oracles_trajectory(capi, oracle, huber):
    """One-, two- and 33-pose frames in one session: the flat problem of CeresHandler::Add (lowered by rsba_amd/problem.py, block for
    block the oracle's Add loop) solved on the device equals the oracle's LM trajectory to 1e-9; the picked pose blocks move and
    are written back, the others stay untouched."""
    from rsba_amd.scene import make_scene
    p0 = coarse_columns(make_scene(10, 500, rolling=True, seed=5, outlier_ratio=0.05 if huber else 0.0).problem)
    fp = scanline_session(p0)
    prob, blocks = lower_scanline_poses(fp, p0.obs_frame, p0.obs_point, p0.obs_xy, shutter=HORIZONTAL, points=p0.points.copy(), intrinsics=p0.intrinsics.copy(),
                                        scanlines=p0.scanlines, huber_a=huber)
    want, which = oracle.add_loop_pose_blocks(fp, p0.obs_frame, p0.obs_xy, HORIZONTAL)
    assert len(want) == len(blocks) and np.array_equal(which, prob.obs_frame)
    assert prob.num_frames > 40 and prob.frame_global.sum() == prob.num_frames - sum(len(q) == 2 for q in fp)
    prob.pose_fixed_mask[blocks[:, 0] == 0] = 0x3F                    # fixFirstNCameras = 1 (CeresHandler.h:282-285,342-348)
    p_dev, p_cpu = prob.copy(), prob.copy()
    opts = dict(max_num_iterations=8)
    with capi.DeviceProblem(p_dev) as dp:
        s, tr = dp.solve(capi.default_options(**opts))
    s_ref, tr_ref = oracle.solve(p_cpu, oracle.default_options(**opts))
    assert s.num_iterations == s_ref.num_iterations and s.num_residual_blocks_reduced == s_ref.num_residual_blocks_reduced
    assert s.num_parameters_reduced == s_ref.num_parameters_reduced
    for a, b in zip(tr, tr_ref):
        assert a.step_is_successful == b.step_is_successful
        assert abs(a.cost - b.cost) <= 1e-9 * b.cost
    assert abs(s.final_cost - s_ref.final_cost) <= 1e-9 * s_ref.final_cost and s.final_cost < 0.9 * s.initial_cost
    assert np.max(np.abs(p_dev.poses - p_cpu.poses)) <= 1e-7 and np.max(np.abs(p_dev.points - p_cpu.points)) <= 1e-6
    before = [q.copy() for q in fp]
    scatter_scanline_poses(p_dev, blocks, fp)
    picked = {(int(f), int(q)) for f, q in blocks if q >= 0}
    moved = 0
    for f, q in enumerate(fp):
        if len(q) <= 2:
            continue
        for k in range(len(q)):
            if (f, k) in picked:
                moved += not np.array_equal(q[k], before[f][k])
            else:
                assert np.array_equal(q[k], before[f][k])
    assert moved > 10


def write_scanline_scene_file(path, p, max_iter):
    with open(path, "wb") as f:
        f.write(struct.pack("<11i", p.num_frames, 2, p.num_points, int(p.shutter), int(p.scanlines[0]), int(p.scanlines[1]), 1, 1, 1, 0, max_iter))
        f.write(struct.pack("<q", p.num_observations))
        f.write(struct.pack("<dddddd", float(p.huber_a), 0.0, -1.0, 0.0, 0.0, 1.0))
        f.write(p.intrinsics[0].astype("<f8").tobytes()); f.write(p.poses.astype("<f8").tobytes()); f.write(p.points.astype("<f8").tobytes())
        f.write(p.obs_xy.astype("<f8").tobytes()); f.write(p.obs_frame.astype("<i4").tobytes()); f.write(p.obs_point.astype("<i4").tobytes())


@pytest.mark.gpu
def test_ba_of_a_session_with_pose_per_scan_line_frames_through_the_host_side(capi, oracle, tmp_path):
    """The same kind of session through the C++ host side: examples/ba_session gives every third frame 33 poses (samples of its own
    motion), CeresHandler::Add (include/rsba/ceres_handler.hpp) hands getPose's pick to ReprojectionError::Create per observation and
    fixes the picked poses of the first frame; BA() lands where the oracle lands on the flat problem the Python lowering builds."""
    import __graft_entry__ as G
    exe = os.path.join(ROOT, "examples", "ba_session")
    if not os.path.exists(exe):
        G.build()
    from rsba_amd.scene import make_scene
    every = 3
    p = coarse_columns(make_scene(10, 500, rolling=True, seed=5).problem)
    write_scanline_scene_file(tmp_path / "s.bin", p, 8)
    r = subprocess.run([exe, str(tmp_path / "s.bin"), str(tmp_path / "o.bin"), "0", "0", "BA", "-1", "1", "0", str(every), str(LINES)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    raw = np.fromfile(tmp_path / "o.bin", dtype="<f8")
    head, body = raw[:6], raw[6:]
    fp = scanline_session(p, every, 0)     # the session the program built: the same sampling of each third frame's motion
    prob, blocks = lower_scanline_poses(fp, p.obs_frame, p.obs_point, p.obs_xy, shutter=p.shutter, points=p.points.copy(), intrinsics=p.intrinsics.copy(),
                                        scanlines=p.scanlines)
    prob.pose_fixed_mask[blocks[:, 0] == 0] = 0x3F
    s_ref, _ = oracle.solve(prob, oracle.default_options(max_num_iterations=8))
    assert head[5] == 1.0 and int(head[3]) == s_ref.num_residual_blocks_reduced
    assert abs(head[0] - s_ref.initial_cost) <= 1e-12 * s_ref.initial_cost
    assert abs(head[1] - s_ref.final_cost) <= 1e-6 * s_ref.final_cost
    scatter_scanline_poses(prob, blocks, fp)
    got = body[:p.num_frames * LINES * 6].reshape(p.num_frames, LINES, 6)     # [F][max poses][6], short frames repeat their last pose
    for f, q in enumerate(fp):
        assert np.max(np.abs(got[f, :len(q)] - q)) <= 1e-5
