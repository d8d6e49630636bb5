"""Flat (SoA) description of one bundle-adjustment problem — the host-side data model.

This is what the reference's ``CeresHandler::Add`` loop (src/rsba/CeresHandler.h:94-390) produces
when it walks a ``Session`` (src/rsba/sfm.thrift:13-74): one residual block per observation over
user-owned parameter arrays, plus constness / subset masks.  The arrays here are handed to the
C-ABI (include/rsba_amd.h) unchanged; nothing in this module computes.
"""
from __future__ import annotations

import dataclasses
from typing import Optional

import numpy as np

# src/rsba/mat/cam.h:37-41
GLOBAL, HORIZONTAL, VERTICAL = 0, 1, 2
# src/rsba/mat/cam.h:23-34
FX, FY, K1, K2, P1, P2, K3, CX, CY = range(9)


@dataclasses.dataclass
class BAProblem:
    """Parameter arrays are modified in place by solves, exactly like ceres::Problem blocks."""

    poses: np.ndarray            # [F, P, 6] float64   (angle-axis world->camera, camera centre)
    points: np.ndarray           # [M, 3]
    intrinsics: np.ndarray       # [NI, 9]  {fx,fy,k1,k2,p1,p2,k3,cx,cy}
    obs_xy: np.ndarray           # [N, 2]
    obs_frame: np.ndarray        # [N] int32
    obs_point: np.ndarray        # [N] int32
    shutter: int = HORIZONTAL    # sess.rs
    scanlines: tuple = (0, 1280)  # sess.scanlines
    interpolate_rotation: bool = True   # opt.model.interpolateRotation
    calibrated: bool = True             # opt.model.calibrated
    frame_intrinsics: Optional[np.ndarray] = None   # [F] int32, None = every frame uses intrinsics[0]
    pose_fixed_mask: Optional[np.ndarray] = None    # [F, P] uint8, bit i = coordinate i fixed
    point_constant: Optional[np.ndarray] = None     # [M] uint8
    intrinsics_constant: Optional[np.ndarray] = None  # [NI] uint8
    huber_a: float = 0.0                # opt.ceres.huberLoss
    # frame-to-frame motion priors with a constant interFrameRatio (CeresHandler.h:147-185)
    prior_kind: int = 0                 # 0 none, 1 constFrameVelocity, 2 constFrameAcceleration
    prior_frames: Optional[np.ndarray] = None   # [NP] int32, frames f >= 1 that carry a prior against frame f - 1
    prior_scale: float = 0.0            # opt.ceres.constFrameVelocity / constFrameAcceleration
    inter_frame_ratio: float = 1.0      # opt.ceres.interFrameRatio
    ratio_free: bool = False            # True: the ratio is a free, lower-bounded parameter block (the option left at 1)
    # per-pose prior blocks (CeresHandler.h:24-73, 127-130, 188-204)
    pose_prior_block: Optional[np.ndarray] = None    # [NG] int32 pose block f * P + q carrying a GoodPosePrior
    pose_prior_values: Optional[np.ndarray] = None   # [NG, 6] the priorPoses blocks (free parameter blocks: solved for, in place)
    pose_prior_rotation: float = 0.0    # opt.ceres.trustPriorCamRotation
    pose_prior_position: float = 0.0    # opt.ceres.trustPriorCamPosition
    spherical_pose_block: int = -1      # pose block carrying the SphericalPrior (frame 1 of a session started at the origin), -1 none
    frame_global: Optional[np.ndarray] = None   # [F] uint8, two-pose sessions only: 1 = the frame has ONE pose in the session (global-shutter functor on poses[f, 0], CeresHandler.h:266-285)

    def __post_init__(self):
        self.poses = np.ascontiguousarray(self.poses, dtype=np.float64)
        assert self.poses.ndim == 3 and self.poses.shape[2] == 6 and self.poses.shape[1] in (1, 2)
        self.points = np.ascontiguousarray(self.points, dtype=np.float64).reshape(-1, 3)
        self.intrinsics = np.ascontiguousarray(self.intrinsics, dtype=np.float64).reshape(-1, 9)
        self.obs_xy = np.ascontiguousarray(self.obs_xy, dtype=np.float64).reshape(-1, 2)
        self.obs_frame = np.ascontiguousarray(self.obs_frame, dtype=np.int32).reshape(-1)
        self.obs_point = np.ascontiguousarray(self.obs_point, dtype=np.int32).reshape(-1)
        assert len(self.obs_frame) == len(self.obs_point) == len(self.obs_xy)
        if self.pose_prior_values is not None:
            self.pose_prior_values = np.ascontiguousarray(self.pose_prior_values, dtype=np.float64).reshape(-1, 6)
        for name, dt in (("frame_intrinsics", np.int32), ("prior_frames", np.int32), ("pose_prior_block", np.int32), ("pose_fixed_mask", np.uint8),
                         ("point_constant", np.uint8), ("intrinsics_constant", np.uint8)):
            v = getattr(self, name)
            if v is not None:
                setattr(self, name, np.ascontiguousarray(v, dtype=dt))
        if self.pose_fixed_mask is not None:
            self.pose_fixed_mask = self.pose_fixed_mask.reshape(self.num_frames, self.poses_per_frame)

    @property
    def num_frames(self) -> int:
        return self.poses.shape[0]

    @property
    def poses_per_frame(self) -> int:
        return self.poses.shape[1]

    @property
    def num_points(self) -> int:
        return self.points.shape[0]

    @property
    def num_intrinsics(self) -> int:
        return self.intrinsics.shape[0]

    @property
    def num_observations(self) -> int:
        return self.obs_xy.shape[0]

    @property
    def jacobian_cols(self) -> int:
        """Columns of one residual block: [cam 9]? [pose0 6] [pose1 6]? [point 3]."""
        return (0 if self.calibrated else 9) + 6 * self.poses_per_frame + 3

    def copy(self) -> "BAProblem":
        kw = {}
        for f in dataclasses.fields(self):
            v = getattr(self, f.name)
            kw[f.name] = v.copy() if isinstance(v, np.ndarray) else v
        return BAProblem(**kw)

    def shard(self, rank: int, world: int, owner: Optional[np.ndarray] = None) -> "BAProblem":
        """Point-partitioned shard for multi-GPU runs: rank r keeps every observation of
        the points it owns — owner[j] == r (rsba_amd.capi.partition_points: the cut along the top separators of
        the reduced system's elimination tree, which lets every rank factor its own part), or j % world == r without
        an owner array; frames / intrinsics are replicated; points keep their global
        numbering so parameter arrays stay comparable across ranks."""
        keep = ((self.obs_point % world) == rank) if owner is None else (np.asarray(owner)[self.obs_point] == rank)
        p = self.copy()
        p.obs_xy = np.ascontiguousarray(self.obs_xy[keep])
        p.obs_frame = np.ascontiguousarray(self.obs_frame[keep])
        p.obs_point = np.ascontiguousarray(self.obs_point[keep])
        return p


def apply_gauge_masks(prob: BAProblem, *, fix_first_n_cameras: int = 0, fix_scale: bool = False,
                      fix_rotation: bool = False, fix_position: bool = False, const3d: bool = False,
                      start_frame: int = 0) -> BAProblem:
    """Constness rules of CeresHandler::Add, applied to a problem whose every frame has observations.

    src/rsba/CeresHandler.h:342-348 first-N cameras constant; :350-360 fixScale -> translation
    {3,4,5} fixed on the first frame's poses[0] and the last frame's poses.back(); :361-371
    fixRotation -> {0,1,2} on every pose; :372-382 fixPosition -> {3,4,5} on every pose (first
    matching rule wins, per frame); :288-300 const3d / window BA freezes points.
    """
    F, P = prob.num_frames, prob.poses_per_frame
    mask = np.zeros((F, P), dtype=np.uint8)
    for f in range(F):
        if f < fix_first_n_cameras:
            mask[f, :] = 0x3F
        elif fix_scale and (f == 0 or f == F - 1):
            if f == 0:
                mask[f, 0] |= 0b111000
            else:
                mask[f, P - 1] |= 0b111000
        elif fix_rotation:
            mask[f, :] |= 0b000111
        elif fix_position:
            mask[f, :] |= 0b111000
    prob.pose_fixed_mask = mask
    pc = np.zeros(prob.num_points, dtype=np.uint8)
    if const3d:
        pc[:] = 1
    if start_frame > 0:
        old = np.unique(prob.obs_point[prob.obs_frame < start_frame])
        pc[old] = 1
    prob.point_constant = pc
    return prob


def lower_scanline_poses(frame_poses, obs_frame, obs_point, obs_xy, *, shutter: int = HORIZONTAL, **problem_kw):
    """Flat problem of a session whose frames carry one, two or MORE poses.

    CeresHandler::Add picks the functor per frame by ``f.poses.size()`` (src/rsba/CeresHandler.h:245-286): two poses ->
    RsBundleAdjustment over both; otherwise ReprojectionError over ONE pose block, ``getPose(sess, f, opt, obs)``
    (src/rsba/struct/VideoSfM.cc:75-99) — for a frame with more than two poses ("fullDoF", a pose per scan line) that is
    ``poses[round(clamp(line, 0, size - 1))]`` with ``line`` = x for a HORIZONTAL shutter, y otherwise.  Every pose block that
    some observation picks becomes a frame of the flat problem (flagged ``frame_global`` when the session also has two-pose
    frames: its second slot is a constant copy); the others never reach the solver, as they never reach Ceres.

    ``frame_poses``: sequence of [P_f, 6] arrays.  Returns ``(problem, blocks)``; ``blocks[d] = (frame, pose index)`` of flat
    frame d, pose index -1 for a two-pose frame.  Flat frames are numbered in the order the observations first use them.
    """
    obs_frame = np.asarray(obs_frame, dtype=np.int64).reshape(-1)
    obs_xy = np.asarray(obs_xy, dtype=np.float64).reshape(-1, 2)
    sizes = np.array([len(p) for p in frame_poses], dtype=np.int64)
    assert sizes.min() >= 1, "empty frame"                               # getPose throws (VideoSfM.cc:77)
    n_of = sizes[obs_frame]
    line = obs_xy[:, 0] if shutter == HORIZONTAL else obs_xy[:, 1]
    line = np.where(line < 0, 0.0, np.where(line > n_of - 1, (n_of - 1).astype(np.float64), line))
    low = np.floor(line)
    pick = (low + (line - low >= 0.5)).astype(np.int64)                  # std::round: halves away from zero (line >= 0 here)
    pick = np.where(n_of == 2, -1, np.where(n_of == 1, 0, pick))
    key = obs_frame * (int(sizes.max()) + 1) + pick + 1
    _, first, inverse = np.unique(key, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")                             # flat frames in the order the observations first use them
    rank = np.empty_like(order); rank[order] = np.arange(len(order))
    flat_of_obs = rank[inverse]
    blocks = np.stack([obs_frame[first[order]], pick[first[order]]], axis=1)
    P = 2 if (sizes == 2).any() else 1
    poses = np.zeros((len(blocks), P, 6))
    frame_global = np.zeros(len(blocks), dtype=np.uint8)
    for d, (f, q) in enumerate(blocks):
        fp = np.asarray(frame_poses[f], dtype=np.float64).reshape(-1, 6)
        if q < 0:
            poses[d] = fp
        else:
            poses[d, :] = fp[q]
            frame_global[d] = 1
    mask = np.zeros((len(blocks), P), dtype=np.uint8)
    if P == 2:
        mask[frame_global == 1, 1] = 0x3F                                # the second slot of a one-pose flat frame is data, not a block
    prob = BAProblem(poses=poses, obs_xy=obs_xy, obs_frame=flat_of_obs.astype(np.int32), obs_point=obs_point, shutter=shutter,
                     frame_global=frame_global if (P == 2 and frame_global.any()) else None, pose_fixed_mask=mask, **problem_kw)
    return prob, blocks


def scatter_scanline_poses(prob: BAProblem, blocks, frame_poses) -> None:
    """The solved pose blocks of ``lower_scanline_poses`` back into the session's per-frame pose arrays (in place)."""
    for d, (f, q) in enumerate(blocks):
        if q < 0:
            frame_poses[f][:] = prob.poses[d]
        else:
            frame_poses[f][q] = prob.poses[d, 0]
