"""The C++ host side: Ceres-shaped facade + CeresHandler / BA() mirror (include/rsba/*.hpp) driving the
C ABI.  CPU: it builds and fails loudly without a device.  GPU: the BA() of a session equals the oracle."""
import os
import subprocess

import numpy as np
import pytest

from helpers import read_result_file, write_scene_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "ba_session")


@pytest.fixture(scope="module")
def exe():
    import __graft_entry__ as G
    if not os.path.exists(EXE):
        G.build()
    return EXE


def small_problem(rolling=True, huber=0.0):
    from rsba_amd.scene import make_scene
    p = make_scene(14, 500, rolling=rolling, seed=41, outlier_ratio=0.05 if huber else 0.0).problem
    p.huber_a = huber
    return p


def test_host_program_fails_loudly_without_a_gpu(exe, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = small_problem()
    write_scene_file(tmp_path / "s.bin", p)
    r = subprocess.run([exe, str(tmp_path / "s.bin"), str(tmp_path / "o.bin")], capture_output=True, text=True)
    assert r.returncode == 1                      # BA() returned !IsSolutionUsable()
    assert "no HIP device" in r.stderr            # and said why; nothing was computed on the CPU
    out = read_result_file(tmp_path / "o.bin", p)
    assert not out["usable"] and np.array_equal(out["poses"], p.poses)


@pytest.mark.gpu
@pytest.mark.parametrize("rolling,huber,fix_scale", [(True, 0.0, False), (False, 0.0, False), (True, 2.0, True)])
def test_ba_of_a_session_matches_the_oracle(exe, oracle, tmp_path, rolling, huber, fix_scale):
    from rsba_amd.problem import apply_gauge_masks
    p = small_problem(rolling, huber)
    write_scene_file(tmp_path / "s.bin", p, fix_first_n=1, fix_scale=fix_scale, max_iter=20)
    r = subprocess.run([exe, str(tmp_path / "s.bin"), str(tmp_path / "o.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "average reprojection error" in r.stdout
    out = read_result_file(tmp_path / "o.bin", p)
    # the same constness rules applied by the Python mirror of CeresHandler::Add
    q = p.copy()
    apply_gauge_masks(q, fix_first_n_cameras=1, fix_scale=fix_scale)
    s_ref, _ = oracle.solve(q, oracle.default_options(max_num_iterations=20))
    assert out["usable"] and out["reduced"] == s_ref.num_residual_blocks_reduced
    assert abs(out["initial_cost"] - s_ref.initial_cost) <= 1e-12 * s_ref.initial_cost
    assert abs(out["final_cost"] - s_ref.final_cost) <= 1e-6 * s_ref.final_cost
    assert np.max(np.abs(out["poses"] - q.poses)) <= 1e-5 and np.max(np.abs(out["points"] - q.points)) <= 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("kind,huber", [(1, 0.0), (2, 2.0)])
def test_ba_with_motion_priors_matches_the_oracle(exe, oracle, tmp_path, kind, huber):
    """CeresHandler::Add's motion priors (CeresHandler.h:147-185) with a known interFrameRatio through the C++ host path."""
    from rsba_amd.problem import apply_gauge_masks
    p = small_problem(True, huber)
    scale = 8.0 if kind == 1 else 20.0
    write_scene_file(tmp_path / "s.bin", p, fix_first_n=1, max_iter=20, const_frame_velocity=scale if kind == 1 else 0.0,
                     const_frame_acceleration=scale if kind == 2 else 0.0, inter_frame_ratio=0.8)
    r = subprocess.run([exe, str(tmp_path / "s.bin"), str(tmp_path / "o.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = read_result_file(tmp_path / "o.bin", p)
    q = p.copy()
    apply_gauge_masks(q, fix_first_n_cameras=1)
    q.prior_kind, q.prior_scale, q.inter_frame_ratio = kind, scale, 0.8
    q.prior_frames = np.arange(1, q.num_frames, dtype=np.int32)
    s_ref, _ = oracle.solve(q, oracle.default_options(max_num_iterations=20))
    assert out["usable"] and out["reduced"] == s_ref.num_residual_blocks_reduced
    assert abs(out["initial_cost"] - s_ref.initial_cost) <= 1e-12 * s_ref.initial_cost
    assert abs(out["final_cost"] - s_ref.final_cost) <= 1e-6 * s_ref.final_cost
    assert np.max(np.abs(out["poses"] - q.poses)) <= 1e-5 and np.max(np.abs(out["points"] - q.points)) <= 1e-4
    # and the priors did something: the plain solve ends elsewhere
    q0 = p.copy(); apply_gauge_masks(q0, fix_first_n_cameras=1)
    s0, _ = oracle.solve(q0, oracle.default_options(max_num_iterations=20))
    assert s0.final_cost < s_ref.final_cost


@pytest.mark.gpu
@pytest.mark.parametrize("free_ratio", [False, True])
def test_windowed_ba_with_motion_priors_links_the_frame_before_the_window(exe, oracle, tmp_path, free_ratio):
    """BA(startFrame > 0) with a motion prior: CeresHandler::Add(startFrame) links frame startFrame to startFrame - 1
    (CeresHandler.h:148-186), whose poses no reprojection block of the window touches — Ceres optimises that prior-only
    block; old tracks are frozen (:288-300)."""
    p = small_problem(True, 0.0)
    s0 = 5
    ratio = 1.0 if free_ratio else 0.8
    write_scene_file(tmp_path / "s.bin", p, fix_first_n=1, max_iter=15, const_frame_velocity=8.0, inter_frame_ratio=ratio)
    r = subprocess.run([exe, str(tmp_path / "s.bin"), str(tmp_path / "o.bin"), str(s0)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr
    out = read_result_file(tmp_path / "o.bin", p)
    # the same program for the oracle: frames s0 - 1 .. F - 1 (the first one without observations), tracks seen before s0 constant
    keep = p.obs_frame >= s0
    q = p.copy()
    q.poses = p.poses[s0 - 1:].copy()
    q.obs_xy, q.obs_frame, q.obs_point = p.obs_xy[keep], p.obs_frame[keep] - (s0 - 1), p.obs_point[keep]
    q.pose_fixed_mask = np.zeros((q.num_frames, 2), dtype=np.uint8)
    pc = np.zeros(p.num_points, dtype=np.uint8); pc[np.unique(p.obs_point[p.obs_frame < s0])] = 1
    q.point_constant = pc
    q.prior_kind, q.prior_scale, q.inter_frame_ratio, q.ratio_free = 1, 8.0, ratio, free_ratio
    q.prior_frames = np.arange(1, q.num_frames, dtype=np.int32)
    first_pose_before = q.poses[0].copy()
    s_ref, _ = oracle.solve(q, oracle.default_options(max_num_iterations=15))
    assert out["usable"] and out["reduced"] == s_ref.num_residual_blocks_reduced
    assert abs(out["initial_cost"] - s_ref.initial_cost) <= 1e-12 * s_ref.initial_cost
    assert abs(out["final_cost"] - s_ref.final_cost) <= 1e-6 * s_ref.final_cost
    assert np.max(np.abs(out["poses"][s0 - 1:] - q.poses)) <= 1e-5
    assert np.array_equal(out["poses"][:s0 - 1], p.poses[:s0 - 1])              # frames before the link are not part of the problem
    assert np.max(np.abs(q.poses[0] - first_pose_before)) > 1e-6                 # the prior-only frame did move


@pytest.mark.gpu
def test_frames_with_their_own_cam_block(exe, oracle, tmp_path):
    """opt.model.calibrated = false with Frame.cam set on every third frame: CeresHandler::Add hands f.cam to the residual
    blocks of those frames and sess.cam to the others (CeresHandler.h:256-264) — four intrinsics parameter blocks here."""
    from rsba_amd.problem import apply_gauge_masks
    p = small_problem(True, 2.0)
    p.calibrated = False
    write_scene_file(tmp_path / "s.bin", p, fix_first_n=1, max_iter=15)
    r = subprocess.run([exe, str(tmp_path / "s.bin"), str(tmp_path / "o.bin"), "0", "3"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr
    out = read_result_file(tmp_path / "o.bin", p)
    q = p.copy()
    apply_gauge_masks(q, fix_first_n_cameras=1)
    own = np.arange(q.num_frames) % 3 == 2
    fi = np.zeros(q.num_frames, dtype=np.int32); fi[own] = 1 + np.arange(own.sum())
    q.frame_intrinsics = fi
    q.intrinsics = np.tile(p.intrinsics[:1], (1 + int(own.sum()), 1))
    s_ref, _ = oracle.solve(q, oracle.default_options(max_num_iterations=15))
    assert out["usable"] and out["reduced"] == s_ref.num_residual_blocks_reduced
    assert abs(out["initial_cost"] - s_ref.initial_cost) <= 1e-12 * s_ref.initial_cost
    assert abs(out["final_cost"] - s_ref.final_cost) <= 1e-6 * s_ref.final_cost
    assert np.max(np.abs(out["poses"] - q.poses)) <= 1e-5 and np.max(np.abs(out["points"] - q.points)) <= 1e-4
    assert len(np.unique(q.intrinsics, axis=0)) == len(q.intrinsics)          # the blocks went their own ways


@pytest.mark.gpu
def test_replaying_a_thrift_session_cache_equals_the_scene_file_path(exe, tmp_path):
    """A Session cache in the reference's on-disk format (Thrift binary in TFileTransport events, written here by the
    independent encoder tests/thrift_encode.py) goes through session_cache.hpp into BA(): same solve as the flat scene file."""
    import thrift_encode as T
    p = small_problem(True, 2.0)
    write_scene_file(tmp_path / "s.bin", p, fix_first_n=1, max_iter=12)
    r = subprocess.run([exe, str(tmp_path / "s.bin"), str(tmp_path / "o.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    (tmp_path / "s.cache").write_bytes(T.file_events(T.session_of_problem(p, descriptors=True), np.random.default_rng(2), max_event=4096))
    r = subprocess.run([exe, "--cache", str(tmp_path / "s.cache"), str(tmp_path / "o2.bin"), "1", "12", "2.0", "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    a, b = read_result_file(tmp_path / "o.bin", p), read_result_file(tmp_path / "o2.bin", p)
    assert a["usable"] and b["usable"] and a["iterations"] == b["iterations"] and a["reduced"] == b["reduced"]
    assert a["initial_cost"] == b["initial_cost"] and a["final_cost"] == b["final_cost"]
    assert np.array_equal(a["poses"], b["poses"]) and np.array_equal(a["points"], b["points"])


@pytest.mark.gpu
def test_match_based_track_lookup_and_pose_initialisation(exe, oracle, tmp_path):
    """CeresHandler::Add's "also add bad reprojections" branch (CeresHandler.h:220-236) and the pose initialisation of a
    frame that arrives without poses (:99-144), replayed from a Session cache: observations without a track take the
    track of their first match whose point validates against the frame; the last frame's poses are extrapolated."""
    import thrift_encode as T
    from rsba_amd.problem import apply_gauge_masks
    p = small_problem(True, 0.0)
    F, thr = p.num_frames, 400.0
    rng = np.random.default_rng(4)
    order = np.argsort(p.obs_frame, kind="stable")
    local = np.zeros(p.num_observations, dtype=np.int64)              # index of each observation inside its frame
    for f in range(F):
        idx = order[p.obs_frame[order] == f]
        local[idx] = np.arange(len(idx))
    by_point = {j: np.flatnonzero(p.obs_point == j) for j in range(p.num_points)}
    detached = set(rng.choice(p.num_observations, p.num_observations // 8, replace=False).tolist())
    # expected: poses of the last frame = linear extrapolation; detached observations resolved by the oracle's validate
    q = p.copy()
    q.poses[F - 1] = q.poses[F - 2] + (q.poses[F - 2] - q.poses[F - 3])
    matches, new_point = {}, q.obs_point.copy()
    keep = np.ones(p.num_observations, dtype=bool)
    for i in sorted(detached):
        f, j = int(p.obs_frame[i]), int(p.obs_point[i])
        others = [k for k in by_point[j] if k != i and k not in detached]
        decoy_pt = int(rng.integers(p.num_points))
        decoys = [k for k in by_point[decoy_pt] if k not in detached][:1] if decoy_pt != j else []
        cands = decoys + others[:1]
        matches[i] = [(int(p.obs_frame[k]), int(local[k]), True) for k in cands]
        found = None
        for k in cands:
            jj = int(p.obs_point[k])
            if oracle.validate_obs(q.intrinsics[0], q.poses[f], q.shutter, q.scanlines, q.points[jj], q.obs_xy[i], thr, 0.0, True):
                found = jj; break
        if found is None:
            keep[i] = False
        else:
            new_point[i] = found
    assert keep.sum() < p.num_observations and (new_point != p.obs_point).sum() >= 0
    obs_bytes = [[] for _ in range(F)]
    for i in order:
        f = int(p.obs_frame[i])
        if i in detached:
            obs_bytes[f].append(T.observation(p.obs_xy[i, 0], p.obs_xy[i, 1], matches=matches[i]))
        else:
            obs_bytes[f].append(T.observation(p.obs_xy[i, 0], p.obs_xy[i, 1], track=int(p.obs_point[i])))
    frames = [T.frame(obs_bytes[f], poses=p.poses[f] if f < F - 1 else None) for f in range(F)]
    tracks = [T.track([(int(p.obs_frame[k]), int(local[k]), True) for k in by_point[j] if k not in detached], pt=p.points[j], valid=True) for j in range(p.num_points)]
    payload = T.session(p.intrinsics[0], frames, tracks, int(p.shutter), list(p.scanlines), 1280, 720)
    (tmp_path / "s.cache").write_bytes(T.file_events(payload, np.random.default_rng(6), max_event=2048))
    r = subprocess.run([exe, "--cache", str(tmp_path / "s.cache"), str(tmp_path / "o.bin"), "1", "15", "0", "1", "0", str(thr)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = read_result_file(tmp_path / "o.bin", p)
    from rsba_amd.problem import BAProblem
    e = BAProblem(poses=q.poses, points=q.points, intrinsics=q.intrinsics, obs_xy=q.obs_xy[keep], obs_frame=q.obs_frame[keep], obs_point=new_point[keep],
                  shutter=q.shutter, scanlines=q.scanlines)
    apply_gauge_masks(e, fix_first_n_cameras=1)
    s_ref, _ = oracle.solve(e, oracle.default_options(max_num_iterations=15))
    assert out["usable"] and out["reduced"] == s_ref.num_residual_blocks_reduced
    assert abs(out["initial_cost"] - s_ref.initial_cost) <= 1e-12 * s_ref.initial_cost
    assert abs(out["final_cost"] - s_ref.final_cost) <= 1e-6 * s_ref.final_cost
    assert np.max(np.abs(out["poses"] - e.poses)) <= 1e-5


@pytest.mark.gpu
def test_free_inter_frame_ratio_is_solved_for(exe, oracle, tmp_path):
    """opt.ceres.interFrameRatio left at 1 — the reference's default: the ratio is a free, lower-bounded parameter block
    (CeresHandler.h:161,172,175); Solve optimises it and CeresHandler::solve prints it (:421-423)."""
    from rsba_amd.problem import apply_gauge_masks
    p = small_problem(True, 0.0)
    write_scene_file(tmp_path / "s.bin", p, fix_first_n=1, max_iter=25, const_frame_velocity=8.0, inter_frame_ratio=1.0)
    r = subprocess.run([exe, str(tmp_path / "s.bin"), str(tmp_path / "o.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = read_result_file(tmp_path / "o.bin", p)
    ratio = float([ln for ln in r.stdout.splitlines() if ln.startswith("interFrameRatio:")][-1].split(":")[1])
    q = p.copy()
    apply_gauge_masks(q, fix_first_n_cameras=1)
    q.prior_kind, q.prior_scale, q.inter_frame_ratio, q.ratio_free = 1, 8.0, 1.0, True
    q.prior_frames = np.arange(1, q.num_frames, dtype=np.int32)
    s_ref, _ = oracle.solve(q, oracle.default_options(max_num_iterations=25))
    assert out["usable"] and out["reduced"] == s_ref.num_residual_blocks_reduced
    assert abs(out["initial_cost"] - s_ref.initial_cost) <= 1e-12 * s_ref.initial_cost
    assert abs(out["final_cost"] - s_ref.final_cost) <= 1e-6 * s_ref.final_cost
    assert abs(ratio - q.inter_frame_ratio) <= 1e-4 and abs(ratio - 1.0) > 1e-3
    assert np.max(np.abs(out["poses"] - q.poses)) <= 1e-4


@pytest.mark.gpu
def test_revalidate_reprojections_drops_what_validate_rejects(exe, oracle, tmp_path):
    """CeresHandler.h:239-243: with revalidateReprojections the observations failing validate() never become
    residual blocks.  The oracle solves the problem with exactly those observations removed."""
    from rsba_amd.problem import apply_gauge_masks
    from rsba_amd.scene import make_scene
    p = make_scene(14, 500, rolling=True, seed=43, outlier_ratio=0.1, rot_noise=0.002, pos_noise=0.01, pt_noise=0.01).problem
    thr = 36.0
    write_scene_file(tmp_path / "s.bin", p, fix_first_n=1, max_iter=15, revalidate=thr)
    r = subprocess.run([exe, str(tmp_path / "s.bin"), str(tmp_path / "o.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = read_result_file(tmp_path / "o.bin", p)
    keep = np.array([oracle.validate_obs(p.intrinsics[0], p.poses[f], p.shutter, p.scanlines, p.points[j], xy, thr, 0.0, True)
                     for f, j, xy in zip(p.obs_frame, p.obs_point, p.obs_xy)])
    assert 0.5 < keep.mean() < 0.97
    q = p.copy()
    q.obs_xy, q.obs_frame, q.obs_point = p.obs_xy[keep], p.obs_frame[keep], p.obs_point[keep]
    apply_gauge_masks(q, fix_first_n_cameras=1, fix_scale=False)
    s_ref, _ = oracle.solve(q, oracle.default_options(max_num_iterations=15))
    assert out["reduced"] == s_ref.num_residual_blocks_reduced
    assert abs(out["initial_cost"] - s_ref.initial_cost) <= 1e-12 * s_ref.initial_cost
    assert abs(out["final_cost"] - s_ref.final_cost) <= 1e-6 * s_ref.final_cost
    # points that lost all their observations keep their input value on both sides
    assert np.max(np.abs(out["poses"] - q.poses)) <= 1e-5 and np.max(np.abs(out["points"] - q.points)) <= 1e-4


@pytest.mark.gpu
def test_calc_covariances_prints_what_the_oracle_computes(exe, oracle, tmp_path):
    """opt.debug.calcCovariances (VideoSfMHandler.cc:599-621): after the solve, ceres::Covariance on (p0,p0), (p0,p1),
    (p1,p1) of every frame.  The facade's blocks of one frame equal the oracle's at the adjusted parameters."""
    from rsba_amd.problem import apply_gauge_masks
    p = small_problem(True, 0.0)
    write_scene_file(tmp_path / "s.bin", p, fix_first_n=1, fix_scale=True, max_iter=15, cov_frame=6)
    r = subprocess.run([exe, str(tmp_path / "s.bin"), str(tmp_path / "o.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "pp:" in r.stdout and "pe:" in r.stdout and "ee:" in r.stdout
    out = read_result_file(tmp_path / "o.bin", p)
    q = p.copy()
    q.poses[...] = out["poses"]; q.points[...] = out["points"]
    apply_gauge_masks(q, fix_first_n_cameras=1, fix_scale=True)
    ref, ok = oracle.pose_covariance(q, 6)
    assert ok and out["covariance"] is not None
    scale = np.abs(ref).max()
    assert np.abs(out["covariance"][0] - ref[:6, :6]).max() <= 1e-7 * scale
    assert np.abs(out["covariance"][1] - ref[:6, 6:]).max() <= 1e-7 * scale
    assert np.abs(out["covariance"][2] - ref[6:, 6:]).max() <= 1e-7 * scale


@pytest.mark.gpu
def test_session_start_and_pose_priors_through_ceres_handler(exe, oracle, tmp_path):
    """CeresHandler::Add on a session that starts without poses (frame 0: zeros, frame 1: zeros + 1e-4 and a SphericalPrior,
    CeresHandler.h:99-144) and whose later frames carry priorPoses (GoodPosePrior, :188-204), replayed from a Session cache."""
    import thrift_encode as T
    p = small_problem(True, 0.0)
    F = p.num_frames
    rng = np.random.default_rng(9)
    order = np.argsort(p.obs_frame, kind="stable")
    local = np.zeros(p.num_observations, dtype=np.int64)
    for f in range(F):
        idx = order[p.obs_frame[order] == f]
        local[idx] = np.arange(len(idx))
    prior = {f: p.poses[f] + rng.normal(0, 0.01, p.poses[f].shape) for f in range(3, F, 2)}
    obs_bytes = [[] for _ in range(F)]
    for i in order:
        obs_bytes[int(p.obs_frame[i])].append(T.observation(p.obs_xy[i, 0], p.obs_xy[i, 1], track=int(p.obs_point[i])))
    frames = [T.frame(obs_bytes[f], poses=p.poses[f] if f >= 2 else None, prior_poses=prior.get(f)) for f in range(F)]
    tracks = [T.track([(int(p.obs_frame[k]), int(local[k]), True) for k in np.flatnonzero(p.obs_point == j)], pt=p.points[j], valid=True) for j in range(p.num_points)]
    (tmp_path / "s.cache").write_bytes(T.file_events(T.session(p.intrinsics[0], frames, tracks, int(p.shutter), list(p.scanlines), 1280, 720), np.random.default_rng(6), max_event=2048))
    r = subprocess.run([exe, "--cache", str(tmp_path / "s.cache"), str(tmp_path / "o.bin"), "1", "12", "0", "1", "1", "16", "3.0", "5.0"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr
    out = read_result_file(tmp_path / "o.bin", p)
    q = p.copy()
    q.poses[0] = 0.0
    q.poses[1] = 0.0; q.poses[1, :, 3:] += 1e-4
    from rsba_amd.problem import apply_gauge_masks
    apply_gauge_masks(q, fix_first_n_cameras=1)
    q.spherical_pose_block = 2
    blocks = [2 * f + i for f in sorted(prior) for i in range(2)]
    q.pose_prior_block = np.array(blocks, dtype=np.int32)
    q.pose_prior_values = np.concatenate([prior[f] for f in sorted(prior)])
    q.pose_prior_rotation, q.pose_prior_position = 3.0, 5.0
    s_ref, tr_ref = oracle.solve(q, oracle.default_options(max_num_iterations=12))
    assert out["usable"] and out["reduced"] == s_ref.num_residual_blocks_reduced
    assert abs(out["initial_cost"] - s_ref.initial_cost) <= 1e-12 * s_ref.initial_cost
    assert abs(1.0 - np.abs(out["poses"][1, 0, 3:]).sum()) <= 4e-16                 # the gauge the SphericalPrior sets
    assert np.max(np.abs(out["poses"][1, 0, 3:] - q.poses[1, 0, 3:])) <= 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("valid_tracks", [60, 150])
def test_full_ba_drops_the_valid_only_filter_below_100_valid_tracks(exe, oracle, tmp_path, valid_tracks):
    """VideoSfMHandler::fullBA (VideoSfMHandler.cc:163-172): with fewer than 100 valid tracks in the session
    useOnlyValidMatches is switched off and every track with a point takes part; with 100 or more only the valid ones do."""
    from rsba_amd.problem import apply_gauge_masks
    p = small_problem(True, 0.0)
    write_scene_file(tmp_path / "s.bin", p, fix_first_n=1, max_iter=12)
    r = subprocess.run([exe, str(tmp_path / "s.bin"), str(tmp_path / "o.bin"), "0", "0", "fullBA", str(valid_tracks)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert ("Not enough valid matches: 60" in r.stdout) == (valid_tracks < 100)
    out = read_result_file(tmp_path / "o.bin", p)
    q = p.copy()
    if valid_tracks >= 100:
        keep = p.obs_point < valid_tracks
        q.obs_xy, q.obs_frame, q.obs_point = p.obs_xy[keep], p.obs_frame[keep], p.obs_point[keep]
    apply_gauge_masks(q, fix_first_n_cameras=1)
    s_ref, _ = oracle.solve(q, oracle.default_options(max_num_iterations=12))
    assert out["usable"] and out["reduced"] == s_ref.num_residual_blocks_reduced
    assert abs(out["initial_cost"] - s_ref.initial_cost) <= 1e-12 * s_ref.initial_cost
    assert abs(out["final_cost"] - s_ref.final_cost) <= 1e-6 * s_ref.final_cost


@pytest.mark.gpu
def test_windowed_ba_switches_fix_scale_off(exe, oracle, tmp_path):
    """VideoSfMHandler::windowedBA (VideoSfMHandler.cc:195): opt.ceres.fixScale = false whatever the session options say —
    BA() on the same window with fixScale keeps the translation of the last frame's last pose, windowedBA() moves it."""
    from rsba_amd.problem import apply_gauge_masks
    p = small_problem(True, 0.0)
    write_scene_file(tmp_path / "s.bin", p, fix_first_n=1, fix_scale=True, max_iter=12)
    res = {}
    for entry in ("BA", "windowedBA"):
        r = subprocess.run([exe, str(tmp_path / "s.bin"), str(tmp_path / f"{entry}.bin"), "0", "0", entry], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        res[entry] = read_result_file(tmp_path / f"{entry}.bin", p)
    assert np.array_equal(res["BA"]["poses"][-1, -1, 3:], p.poses[-1, -1, 3:])
    assert not np.array_equal(res["windowedBA"]["poses"][-1, -1, 3:], p.poses[-1, -1, 3:])
    q = p.copy()
    apply_gauge_masks(q, fix_first_n_cameras=1, fix_scale=False)
    s_ref, _ = oracle.solve(q, oracle.default_options(max_num_iterations=12))
    out = res["windowedBA"]
    assert out["usable"] and abs(out["final_cost"] - s_ref.final_cost) <= 1e-6 * s_ref.final_cost
    assert np.max(np.abs(out["poses"] - q.poses)) <= 1e-5


@pytest.mark.gpu
def test_a_session_that_mixes_two_pose_and_one_pose_frames(exe, oracle, tmp_path):
    """CeresHandler::Add picks the functor per frame by f.poses.size() (CeresHandler.h:245-286): a rolling-shutter session may
    contain one-pose frames, which get the global-shutter functor on their single pose.  Through the facade such a Problem is
    lowered to two pose slots per frame + rsba_set_global_shutter_frames; the oracle solves the same program."""
    from rsba_amd.problem import apply_gauge_masks
    p = small_problem(True, 2.0)
    write_scene_file(tmp_path / "s.bin", p, fix_first_n=1, max_iter=15)
    r = subprocess.run([exe, str(tmp_path / "s.bin"), str(tmp_path / "o.bin"), "0", "0", "BA", "-1", "1", "3"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr
    out = read_result_file(tmp_path / "o.bin", p)
    q = p.copy()
    apply_gauge_masks(q, fix_first_n_cameras=1)
    q.frame_global = (np.arange(q.num_frames) % 3 == 2).astype(np.uint8)
    q.pose_fixed_mask[q.frame_global == 1, 1] = 0x3F
    s_ref, _ = oracle.solve(q, oracle.default_options(max_num_iterations=15))
    assert out["usable"] and out["reduced"] == s_ref.num_residual_blocks_reduced
    assert abs(out["initial_cost"] - s_ref.initial_cost) <= 1e-12 * s_ref.initial_cost
    assert abs(out["final_cost"] - s_ref.final_cost) <= 1e-6 * s_ref.final_cost
    gs = q.frame_global == 1
    assert np.max(np.abs(out["poses"][~gs] - q.poses[~gs])) <= 1e-5 and np.max(np.abs(out["poses"][gs, 0] - q.poses[gs, 0])) <= 1e-5


def test_dlt_of_the_global_shutter_pnp_initialisation_recovers_the_pose(oracle, tmp_path):
    """include/rsba/solve_rs_pnp.hpp: the direct linear transform that stands in for the start of cv::solvePnP (solveRSpnp.cpp:111-117)
    is host glue (undistort + normalise, 12 x 12 eigenproblem, nearest rotation) — checked here without a device: exact data (float32
    inputs, Brown distortion) must give the pose back."""
    import struct
    import __graft_entry__ as G
    exe = os.path.join(ROOT, "examples", "pnp_ransac")
    if not os.path.exists(exe):
        G.build()
    cam = np.array([800.0, 800.0, -0.05, 0.01, 1e-3, -1e-3, 2e-3, 640.0, 360.0])
    rng = np.random.default_rng(4)
    for trial in range(4):
        pose = np.concatenate([rng.normal(0, 0.3, 3), rng.normal(0, 0.5, 3)])
        X = np.stack([rng.uniform(-5, 5, 400), rng.uniform(-3, 3, 400), rng.uniform(6, 14, 400)], axis=1).astype(np.float32)
        xy, keep = [], []
        for j in range(len(X)):
            ok, p = oracle.reproject(cam, np.stack([pose, pose]), 0, (0, 1), X[j].astype(np.float64), 1e12)
            if ok and 0 < p[0] < 1280 and 0 < p[1] < 720:
                keep.append(j); xy.append(p)
        X, xy = X[keep][:60], np.array(xy)[:60].astype(np.float32)
        assert len(X) >= 20
        with open(tmp_path / "p.bin", "wb") as f:
            f.write(struct.pack("<7i", len(X), 0, 0, 1, -1, 0, 6)); f.write(struct.pack("<f", 3.0)); f.write(struct.pack("<Q", 1))
            f.write(cam.astype("<f8").tobytes()); f.write(np.zeros(12, dtype="<f8").tobytes())
            f.write(X.astype("<f4").tobytes()); f.write(xy.astype("<f4").tobytes())
        r = subprocess.run([exe, str(tmp_path / "p.bin"), str(tmp_path / "o.bin")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        raw = open(tmp_path / "o.bin", "rb").read()
        v = np.frombuffer(raw[:96], dtype="<f8"); cnt = struct.unpack("<i", raw[96:100])[0]
        assert cnt == 0
        rvec, tvec = v[0:3], v[3:6]
        centre = -oracle.angle_axis_rotate(-rvec, tvec)          # pose = (rvec, -R^T tvec)
        assert np.max(np.abs(rvec - pose[:3])) <= 2e-3 and np.max(np.abs(centre - pose[3:])) <= 2e-2, (rvec, pose, centre)


def test_dlt_takes_planar_targets_through_the_homography(oracle, tmp_path):
    """Coplanar object points (a checkerboard, a wall) leave the 12-unknown DLT's null space two-dimensional.  cv::solvePnP — what the
    reference calls at solveRSpnp.cpp:111-117 — starts from the plane's homography there (two smallest singular values of the centred
    points' scatter in a ratio below 1e-3); so does the host glue: exactly planar and nearly planar (1e-4 of the extent) targets give
    the pose back, collinear points are declined, a shallow but genuinely three-dimensional cloud takes the 12-unknown DLT."""
    import struct
    import __graft_entry__ as G
    exe = os.path.join(ROOT, "examples", "pnp_ransac")
    if not os.path.exists(exe):
        G.build()
    cam = np.array([800.0, 800.0, -0.05, 0.01, 1e-3, -1e-3, 2e-3, 640.0, 360.0])
    rng = np.random.default_rng(9)
    pose = np.concatenate([rng.normal(0, 0.2, 3), rng.normal(0, 0.3, 3)])

    def run(X):
        X = X.astype(np.float32)
        xy = []
        for j in range(len(X)):
            ok, p = oracle.reproject(cam, np.stack([pose, pose]), 0, (0, 1), X[j].astype(np.float64), 1e12)
            assert ok
            xy.append(p)
        with open(tmp_path / "p.bin", "wb") as f:
            f.write(struct.pack("<7i", len(X), 0, 0, 1, -1, 0, 6)); f.write(struct.pack("<f", 3.0)); f.write(struct.pack("<Q", 1))
            f.write(cam.astype("<f8").tobytes()); f.write(np.zeros(12, dtype="<f8").tobytes())
            f.write(X.astype("<f4").tobytes()); f.write(np.array(xy).astype("<f4").tobytes())
        r = subprocess.run([exe, str(tmp_path / "p.bin"), str(tmp_path / "o.bin")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        raw = open(tmp_path / "o.bin", "rb").read()
        cnt = struct.unpack("<i", raw[96:100])[0]
        return cnt == 0, np.frombuffer(raw[:96], dtype="<f8")     # (accepted?, vectors)

    uv = rng.uniform(-2, 2, (40, 2))
    normal, origin = np.array([0.2, -0.1, 1.0]), np.array([0.0, 0.0, 9.0])
    e1 = np.cross(normal, [1.0, 0, 0]); e1 /= np.linalg.norm(e1); e2 = np.cross(normal, e1); e2 /= np.linalg.norm(e2)
    plane = origin + uv[:, :1] * e1 + uv[:, 1:] * e2
    for X in (plane, plane + 2e-4 * rng.normal(size=(40, 1)) * normal / np.linalg.norm(normal)):   # exactly planar; planar up to 1e-4 of its extent
        ok, v = run(X)
        assert ok
        centre = -oracle.angle_axis_rotate(-v[0:3], v[3:6])
        assert np.max(np.abs(v[0:3] - pose[:3])) <= 5e-3 and np.max(np.abs(centre - pose[3:])) <= 5e-2, (v, pose, centre)
    ok, v = run(plane[:6])                                                          # a minimal RANSAC subset on the plane
    assert ok and np.max(np.abs(v[0:3] - pose[:3])) <= 2e-2
    assert not run(origin + uv[:, :1] * e1 + 1e-3 * uv[:, 1:] * e2)[0]              # (almost) collinear
    ok, v = run(plane + 0.3 * rng.normal(size=(40, 1)) * normal / np.linalg.norm(normal))   # shallow, but three-dimensional
    assert ok
    centre = -oracle.angle_axis_rotate(-v[0:3], v[3:6])
    assert np.max(np.abs(v[0:3] - pose[:3])) <= 5e-3 and np.max(np.abs(centre - pose[3:])) <= 5e-2
