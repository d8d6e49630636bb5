"""Deterministic synthetic rolling-shutter scenes (SURVEY §8d / BASELINE.md §3).

A sideways-looking camera travels along +x past a slab of points, so every point is seen in a
sliding window of about ``track_len`` consecutive frames — the visibility pattern of video SfM,
which is what rsba processes.  Rolling-shutter frames carry two poses (read-out start / end) and
observations are synthesised with the reference's own fixed-point rule on the scan-line time
(src/rsba/struct/VideoSfM.cc:139-155), HORIZONTAL shutter with scanlines {0, width} so that the
functor's "tau from x" quirk (src/rsba/VideoSfmBaRs.h:31) is self-consistent.

Input synthesis only: nothing here is on the measured path.
"""
from __future__ import annotations

import dataclasses

import numpy as np

from .problem import BAProblem, GLOBAL, HORIZONTAL, apply_gauge_masks

SEED = 0x5BA
WIDTH, HEIGHT = 1280, 720
CAM_RS = np.array([800.0, 800.0, -0.05, 0.01, 1e-3, -1e-3, 2e-3, 640.0, 360.0])
CAM_PINHOLE = np.array([800.0, 800.0, 0.0, 0.0, 0.0, 0.0, 0.0, 640.0, 360.0])


def rotate(w: np.ndarray, p: np.ndarray) -> np.ndarray:
    """Rodrigues rotation of points p [n,3] by angle-axis vectors w [n,3] (first-order near zero)."""
    th2 = np.sum(w * w, axis=1)
    small = th2 <= np.finfo(np.float64).eps
    th = np.sqrt(np.where(small, 1.0, th2))
    k = w / th[:, None]
    c, s = np.cos(th)[:, None], np.sin(th)[:, None]
    out = p * c + np.cross(k, p) * s + k * (np.sum(k * p, axis=1)[:, None]) * (1.0 - c)
    out_small = p + np.cross(w, p)
    return np.where(small[:, None], out_small, out)


def project(cam: np.ndarray, pose: np.ndarray, X: np.ndarray):
    """Pinhole + Brown distortion of world points X [n,3] through poses [n,6]; returns (xy, z)."""
    pc = rotate(pose[:, :3], X - pose[:, 3:])
    z = pc[:, 2]
    zs = np.where(np.abs(z) < 1e-12, 1e-12, z)
    x, y = pc[:, 0] / zs, pc[:, 1] / zs
    r2 = x * x + y * y
    d = 1.0 + r2 * (cam[2] + r2 * (cam[3] + r2 * cam[6]))
    xd = d * x + 2.0 * cam[4] * x * y + cam[5] * (r2 + 2.0 * x * x)
    yd = d * y + cam[4] * (r2 + 2.0 * y * y) + 2.0 * cam[5] * x * y
    return np.stack([cam[0] * xd + cam[7], cam[1] * yd + cam[8]], axis=1), z


@dataclasses.dataclass
class Scene:
    problem: BAProblem        # parameters = perturbed initial guess
    true_poses: np.ndarray
    true_points: np.ndarray
    name: str


def _trajectory(s: np.ndarray, step: float) -> np.ndarray:
    """Pose at path parameter s (in frames): centre moves along +x with a gentle weave; the
    world->camera rotation is a smooth wobble that is exactly zero (with zero slope) at s = 0."""
    c = np.stack([step * s, 0.15 * np.sin(0.05 * s), 0.10 * (1.0 - np.cos(0.03 * s))], axis=1)
    r = np.stack([0.05 * (1.0 - np.cos(0.021 * s)), 0.08 * (1.0 - np.cos(0.013 * s)), 0.06 * (1.0 - np.cos(0.017 * s))], axis=1)
    return np.concatenate([r, c], axis=1)


def make_scene(num_frames: int, num_points: int, *, rolling: bool = True, track_len: int = 20,
               seed: int = SEED, noise_px: float = 0.5, outlier_ratio: float = 0.0,
               pinhole: bool = False, rot_noise: float = 0.01, pos_noise: float = 0.05,
               pt_noise: float = 0.05, intra_frame: float = 0.3, name: str = "scene",
               all_visible: bool = False) -> Scene:
    rng = np.random.default_rng(seed)
    cam = (CAM_PINHOLE if pinhole else CAM_RS).copy()
    F, M = num_frames, num_points
    depth_lo, depth_hi = 8.0, 14.0
    # horizontal half field of view at mid depth decides how far the camera moves per frame
    half_w = 0.5 * WIDTH / cam[0] * 0.5 * (depth_lo + depth_hi)
    step = (2.0 * half_w / track_len) if not all_visible else (0.5 * half_w / max(F, 1))
    s0 = np.arange(F, dtype=np.float64)
    P = 2 if rolling else 1
    poses = np.zeros((F, P, 6))
    poses[:, 0] = _trajectory(s0, step)
    if rolling:
        poses[:, 1] = _trajectory(s0 + intra_frame, step)
        poses[0, :, :3] = 0.0          # frame 0: exactly zero rotation -> small-angle branch
    path_len = step * (F - 1)
    if all_visible:
        X = np.stack([rng.uniform(0.25 * path_len, 0.75 * path_len, M) if F > 1 else rng.uniform(-1, 1, M),
                      rng.uniform(-2.0, 2.0, M), rng.uniform(10.0, depth_hi, M)], axis=1)
    else:
        X = np.stack([rng.uniform(-0.5 * half_w, path_len + 0.5 * half_w, M),
                      rng.uniform(-2.6, 2.6, M), rng.uniform(depth_lo, depth_hi, M)], axis=1)

    # candidate (point, frame) pairs: a window of frames around the one facing the point
    W = F if all_visible else min(F, track_len + 8)
    centre = np.clip(np.rint(X[:, 0] / step).astype(np.int64), 0, F - 1)
    first = np.clip(centre - W // 2, 0, max(F - W, 0))
    cand_f = (first[:, None] + np.arange(W)[None, :]).reshape(-1)
    cand_p = np.repeat(np.arange(M), W)
    Xc = X[cand_p]
    scan = (0, WIDTH)
    if rolling:
        # reference rule: start at the principal point, re-interpolate the pose at the projected
        # x (HORIZONTAL: tau = x / width), stop when the projection moves < 1e-3 px
        xy = np.tile(cam[7:9], (len(cand_f), 1))
        for _ in range(50):
            tau = np.clip((xy[:, 0] - scan[0]) / float(scan[1] - scan[0]), 0.0, 1.0)[:, None]
            pose = poses[cand_f, 0] + (poses[cand_f, 1] - poses[cand_f, 0]) * tau
            new_xy, z = project(cam, pose, Xc)
            moved = np.sum((new_xy - xy) ** 2, axis=1)
            xy = new_xy
            if np.all(moved[np.isfinite(moved)] <= 1e-6):
                break
    else:
        xy, z = project(cam, poses[cand_f, 0], Xc)
    ok = (z > 0.5) & (xy[:, 0] >= 0) & (xy[:, 0] < WIDTH) & (xy[:, 1] >= 0) & (xy[:, 1] < HEIGHT) & np.all(np.isfinite(xy), axis=1)
    obs_f, obs_p, obs_xy = cand_f[ok], cand_p[ok], xy[ok]
    obs_xy = obs_xy + rng.normal(0.0, noise_px, obs_xy.shape)
    if outlier_ratio > 0:
        bad = rng.random(len(obs_xy)) < outlier_ratio
        obs_xy[bad] = np.stack([rng.uniform(0, WIDTH, bad.sum()), rng.uniform(0, HEIGHT, bad.sum())], axis=1)
    # drop points seen fewer than twice, renumber, and order observations frame-major (the order in
    # which CeresHandler::Add visits them, src/rsba/VideoSfMHandler.cc:587-590)
    cnt = np.bincount(obs_p, minlength=M)
    keepo = cnt[obs_p] >= 2
    obs_f, obs_p, obs_xy = obs_f[keepo], obs_p[keepo], obs_xy[keepo]
    # Points are numbered in the order the video first sees them, as rsba's createTracks appends tracks while
    # frames arrive (src/rsba/VideoSfMHandler.cc:283-372): the points of one frame then sit next to each other
    # in the point array instead of being scattered over all of it.
    used = np.unique(obs_p)
    first_seen = np.full(M, np.iinfo(np.int64).max, dtype=np.int64)
    np.minimum.at(first_seen, obs_p, obs_f)
    used = used[np.lexsort((X[used, 0], first_seen[used]))]
    remap = np.full(M, -1, dtype=np.int64)
    remap[used] = np.arange(len(used))
    X = X[used]
    obs_p = remap[obs_p]
    order = np.lexsort((obs_p, obs_f))
    obs_f, obs_p, obs_xy = obs_f[order], obs_p[order], obs_xy[order]

    init_poses = poses.copy()
    init_poses[:, :, :3] += rng.normal(0.0, rot_noise, init_poses[:, :, :3].shape)
    init_poses[:, :, 3:] += rng.normal(0.0, pos_noise, init_poses[:, :, 3:].shape)
    init_poses[0] = poses[0]           # frame 0 is the gauge anchor (fixFirstNCameras = 1)
    init_points = X + rng.normal(0.0, pt_noise, X.shape)
    prob = BAProblem(poses=init_poses, points=init_points, intrinsics=cam[None, :].copy(),
                     obs_xy=obs_xy, obs_frame=obs_f.astype(np.int32), obs_point=obs_p.astype(np.int32),
                     shutter=HORIZONTAL if rolling else GLOBAL, scanlines=scan,
                     interpolate_rotation=True, calibrated=True)
    return Scene(problem=prob, true_poses=poses, true_points=X, name=name)


# BASELINE.json configs (frames, points, rolling, extra)
CONFIGS = {
    "C1": dict(num_frames=10, num_points=1000, rolling=False, pinhole=True, all_visible=True),
    "C2": dict(num_frames=100, num_points=10_000, rolling=True),
    "C4": dict(num_frames=1000, num_points=100_000, rolling=True),
    "C5": dict(num_frames=4000, num_points=500_000, rolling=True, outlier_ratio=0.05),
}


def make_config(name: str, *, seed: int = SEED, gauge: bool = True) -> Scene:
    kw = dict(CONFIGS[name])
    sc = make_scene(seed=seed, name=name, **kw)
    if name == "C5":
        sc.problem.huber_a = 2.0
        sc.problem.calibrated = False       # shared intrinsics as a parameter block
    if gauge:
        apply_gauge_masks(sc.problem, fix_first_n_cameras=1, fix_scale=False)
        # translation of the last frame's last pose held fixed (CeresHandler.h:350-360 rule applied
        # to the last frame only, since frame 0 is already constant)
        sc.problem.pose_fixed_mask[-1, -1] |= 0b111000
    return sc
