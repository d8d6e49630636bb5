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
    return prob


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))) if a.size else 0.0
