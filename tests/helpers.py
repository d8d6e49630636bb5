"""Shared helpers for the parity tests."""
import json
import os

import numpy as np

from rsba_amd.problem import BAProblem, GLOBAL

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def problem_from_case(c) -> BAProblem:
    """One-observation problem for a per_observation.json case."""
    poses = [c["pose0"]] if c["pose1"] is None else [c["pose0"], c["pose1"]]
    return BAProblem(poses=np.array([poses]), points=np.array([c["point"]]), intrinsics=np.array([c["cam"]]),
                     obs_xy=np.array([c["obs"]]), obs_frame=np.array([0]), obs_point=np.array([0]),
                     shutter=c["shutter"], scanlines=tuple(c["scanlines"]),
                     interpolate_rotation=c["interpolate_rotation"], calibrated=c["calibrated"])


def batch_cases(cases):
    """Group per-observation cases that share a model into multi-observation problems so a device
    launch sees many of them at once.  Yields (problem, [case indices])."""
    groups = {}
    for idx, c in enumerate(cases):
        key = (c["pose1"] is None, c["shutter"], tuple(c["scanlines"]), c["interpolate_rotation"], c["calibrated"])
        groups.setdefault(key, []).append(idx)
    for key, idxs in groups.items():
        cs = [cases[i] for i in idxs]
        poses = np.array([[c["pose0"]] if c["pose1"] is None else [c["pose0"], c["pose1"]] for c in cs])
        prob = BAProblem(poses=poses, points=np.array([c["point"] for c in cs]), intrinsics=np.array([c["cam"] for c in cs]),
                         obs_xy=np.array([c["obs"] for c in cs]), obs_frame=np.arange(len(cs)), obs_point=np.arange(len(cs)),
                         shutter=key[1], scanlines=key[2], interpolate_rotation=key[3], calibrated=key[4],
                         frame_intrinsics=np.arange(len(cs), dtype=np.int32))
        yield prob, idxs


def problem_from_solve_case(c) -> BAProblem:
    poses = np.array(c["poses"])
    prob = BAProblem(poses=poses, points=np.array(c["points"]), intrinsics=np.array([c["cam"]]),
                     obs_xy=np.array(c["obs_xy"]), obs_frame=np.array(c["obs_frame"]), obs_point=np.array(c["obs_point"]),
                     shutter=c["shutter"], scanlines=tuple(c["scanlines"]), interpolate_rotation=True, calibrated=True,
                     huber_a=c["huber_a"])
    F, P = poses.shape[:2]
    mask = np.zeros((F, P), dtype=np.uint8)
    mask[0, :] = 0x3F
    mask[-1, -1] |= 0b111000
    prob.pose_fixed_mask = mask
    if c.get("prior_kind"):
        prob.prior_kind, prob.prior_scale, prob.inter_frame_ratio = c["prior_kind"], c["prior_scale"], c["inter_frame_ratio"]
        prob.prior_frames = np.array(c["prior_frames"], dtype=np.int32)
        prob.ratio_free = bool(c.get("ratio_free", False))
    return prob


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))) if a.size else 0.0


def write_scene_file(path, prob, *, fix_first_n=1, fix_scale=False, max_iter=20, revalidate=0.0, cov_frame=-1,
                     const_frame_velocity=0.0, const_frame_acceleration=0.0, inter_frame_ratio=1.0):
    """Binary scene file read by examples/ba_session.cpp (layout documented there)."""
    import struct
    with open(path, "wb") as f:
        f.write(struct.pack("<11i", prob.num_frames, prob.poses_per_frame, prob.num_points, int(prob.shutter), int(prob.scanlines[0]),
                            int(prob.scanlines[1]), int(prob.calibrated), int(prob.interpolate_rotation), fix_first_n, int(fix_scale), max_iter))
        f.write(struct.pack("<q", prob.num_observations))
        f.write(struct.pack("<ddd", float(prob.huber_a), float(revalidate), float(cov_frame)))
        f.write(struct.pack("<ddd", float(const_frame_velocity), float(const_frame_acceleration), float(inter_frame_ratio)))
        f.write(prob.intrinsics[0].astype("<f8").tobytes())
        f.write(prob.poses.astype("<f8").tobytes())
        f.write(prob.points.astype("<f8").tobytes())
        f.write(prob.obs_xy.astype("<f8").tobytes())
        f.write(prob.obs_frame.astype("<i4").tobytes())
        f.write(prob.obs_point.astype("<i4").tobytes())


def read_result_file(path, prob):
    raw = np.fromfile(path, dtype="<f8")
    head = raw[:6]
    npose = prob.poses.size
    poses = raw[6:6 + npose].reshape(prob.poses.shape)
    points = raw[6 + npose:6 + npose + prob.points.size].reshape(-1, 3)
    rest = raw[6 + npose + prob.points.size:]
    return dict(initial_cost=head[0], final_cost=head[1], iterations=int(head[2]), reduced=int(head[3]), termination=int(head[4]),
                usable=bool(head[5]), poses=poses, points=points, covariance=rest[:108].reshape(3, 6, 6) if rest.size >= 108 else None)
