#!/usr/bin/env python3
"""Generates the committed golden vectors in tests/golden/*.json.

The reference (henrique/rsba) cannot be compiled or imported here (C++ on Ceres/Eigen/glog/Thrift/
OpenCV, none installed — SURVEY §8c), and its only test (src/rsba/test/mat_test.cc) holds no vector
for the rolling-shutter functor, any Jacobian, the Huber loss or any solve.  These goldens are
therefore produced by an INDEPENDENT implementation of the path's mathematics:

  * per-observation residuals and Jacobians: mpmath at 50 significant digits, derivatives by
    mpmath's high-order numerical differentiation (error << 1e-20), following SURVEY Appendix A;
  * Huber rho triples: closed form in mpmath;
  * minima of tiny scenes: scipy.optimize.least_squares (trust-region-reflective, unrelated to
    Ceres' LM / Schur code) run to machine-precision tolerances on a numpy residual function
    written here.

Nothing in this script imports the oracle or the product.  Run:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import mpmath as mp
import numpy as np
from scipy.optimize import least_squares

HERE = os.path.dirname(os.path.abspath(__file__))
mp.mp.dps = 50
EPS = mp.mpf(2) ** -52

GLOBAL, HORIZONTAL, VERTICAL = 0, 1, 2


# ------------------------------- mpmath model of one observation -------------------------------
def mp_rotate(w, p):
    th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2]
    cx = [w[1] * p[2] - w[2] * p[1], w[2] * p[0] - w[0] * p[2], w[0] * p[1] - w[1] * p[0]]
    if th2 > EPS:
        th = mp.sqrt(th2)
        a, b, c = mp.sin(th) / th, (1 - mp.cos(th)) / th2, mp.cos(th)
        wp = w[0] * p[0] + w[1] * p[1] + w[2] * p[2]
        return [p[i] * c + a * cx[i] + b * wp * w[i] for i in range(3)]
    return [p[i] + cx[i] for i in range(3)]


def mp_residual(cam, pose0, pose1, X, obs, shutter, scan, interp_rot, tau_from_x_always=True):
    """Returns None when the point is not in front of the camera (z < 1e-8)."""
    if pose1 is None or shutter == GLOBAL:
        pose = list(pose0)
    else:
        # reference quirk: the functor feeds (x, x) to interpolate_rs, so VERTICAL also reads x
        coord = obs[0] if (tau_from_x_always or shutter != VERTICAL) else obs[1]
        tau = (coord - scan[0]) / mp.mpf(scan[1] - scan[0])
        tau = min(max(tau, mp.mpf(0)), mp.mpf(1))
        rot = [pose0[i] + (pose1[i] - pose0[i]) * tau for i in range(3)] if interp_rot else list(pose0[:3])
        pose = rot + [pose0[i] + (pose1[i] - pose0[i]) * tau for i in range(3, 6)]
    pc = mp_rotate(pose[:3], [X[i] - pose[3 + i] for i in range(3)])
    if pc[2] < mp.mpf("1e-8"):
        return None
    x, y = pc[0] / pc[2], pc[1] / pc[2]
    fx, fy, k1, k2, p1, p2, k3, cx, cy = cam
    r2 = x * x + y * y
    d = 1 + r2 * (k1 + r2 * (k2 + r2 * k3))
    xd = d * x + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = d * y + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return [fx * xd + cx - obs[0], fy * yd + cy - obs[1]]


def mp_case(name, cam, pose0, pose1, X, obs, shutter, scan, interp_rot, calibrated):
    cam_m = [mp.mpf(v) for v in cam]
    p0 = [mp.mpf(v) for v in pose0]
    p1 = None if pose1 is None else [mp.mpf(v) for v in pose1]
    Xm = [mp.mpf(v) for v in X]
    om = [mp.mpf(v) for v in obs]
    res = mp_residual(cam_m, p0, p1, Xm, om, shutter, scan, interp_rot)
    out = dict(name=name, cam=list(map(float, cam)), pose0=list(map(float, pose0)),
               pose1=None if pose1 is None else list(map(float, pose1)), point=list(map(float, X)),
               obs=list(map(float, obs)), shutter=shutter, scanlines=list(scan),
               interpolate_rotation=bool(interp_rot), calibrated=bool(calibrated))
    if res is None:
        out.update(ok=False, residual=None, jacobian=None)
        return out
    # parameter vector in block order [cam]? pose0 [pose1]? point
    blocks = ([] if calibrated else [("cam", 9)]) + [("pose0", 6)] + ([("pose1", 6)] if p1 is not None else []) + [("point", 3)]
    x0 = ([] if calibrated else cam_m) + p0 + ([] if p1 is None else p1) + Xm

    def f(k, *xs):
        xs = list(xs)
        i = 0
        c = cam_m
        if not calibrated:
            c = xs[:9]; i = 9
        a = xs[i:i + 6]; i += 6
        b = None
        if p1 is not None:
            b = xs[i:i + 6]; i += 6
        Xv = xs[i:i + 3]
        r = mp_residual(c, a, b, Xv, om, shutter, scan, interp_rot)
        return r[k]

    n = len(x0)
    J = [[0.0] * n for _ in range(2)]
    for k in range(2):
        for j in range(n):
            order = tuple(1 if t == j else 0 for t in range(n))
            J[k][j] = float(mp.diff(lambda *xs: f(k, *xs), tuple(x0), order))
    out.update(ok=True, residual=[float(res[0]), float(res[1])], jacobian=J, blocks=blocks)
    return out


def per_observation_cases():
    rng = np.random.default_rng(20260929)
    cam = [800.0, 800.0, -0.05, 0.01, 1e-3, -1e-3, 2e-3, 640.0, 360.0]
    cam2 = [860.0, 845.0, 0.001, -0.002, 5e-4, 3e-4, 1e-4, 100.0, 200.0]
    scan = (0, 1280)
    cases = []

    def rnd_pose(scale_r=0.3):
        return list(rng.normal(0, scale_r, 3)) + list(rng.normal(0, 0.5, 3))

    def target(pose0, pose1, tau_x, depth=9.0):
        # a world point that lands near image column tau_x*1280 for the tau-interpolated pose (pinhole guess)
        return [float(pose0[3] + (tau_x * 1280 - 640) / 800 * depth), float(pose0[4] + rng.uniform(-2, 2)), float(pose0[5] + depth)]

    i = 0
    for calibrated in (True, False):
        for interp in (True, False):
            for shutter in (HORIZONTAL, VERTICAL, GLOBAL):
                for _ in range(2):
                    p0 = rnd_pose(0.15); p1 = [p0[k] + rng.normal(0, 0.02) for k in range(6)]
                    X = target(p0, p1, rng.uniform(0.1, 0.9))
                    obs = [float(rng.uniform(50, 1230)), float(rng.uniform(30, 690))]
                    cases.append(mp_case(f"rs{i}", cam if i % 2 == 0 else cam2, p0, p1, X, obs, shutter, scan, interp, calibrated)); i += 1
    # tau clamped below 0 and above 1
    for calibrated in (True, False):
        for ox in (-25.5, 1300.25):
            p0 = rnd_pose(0.2); p1 = [p0[k] + rng.normal(0, 0.03) for k in range(6)]
            cases.append(mp_case(f"clamp{i}", cam, p0, p1, target(p0, p1, 0.5), [ox, 300.0], HORIZONTAL, scan, True, calibrated)); i += 1
    # non-zero first scan-line, reversed read-out direction
    p0 = rnd_pose(0.2); p1 = [p0[k] + rng.normal(0, 0.03) for k in range(6)]
    cases.append(mp_case(f"scan{i}", cam, p0, p1, target(p0, p1, 0.5), [500.0, 100.0], HORIZONTAL, (40, 1240), True, True)); i += 1
    cases.append(mp_case(f"scan{i}", cam, p0, p1, target(p0, p1, 0.5), [500.0, 100.0], HORIZONTAL, (1280, 0), True, True)); i += 1
    # exactly-zero rotation (small-angle branch of AngleAxisRotatePoint), both poses / one pose
    z = [0.0, 0.0, 0.0]
    for calibrated in (True, False):
        cases.append(mp_case(f"zero{i}", cam, z + [0.1, -0.2, 0.3], z + [0.15, -0.2, 0.31], [0.7, -0.4, 8.0], [700.0, 320.0], HORIZONTAL, scan, True, calibrated)); i += 1
        cases.append(mp_case(f"zero{i}", cam, z + [0.1, -0.2, 0.3], [1e-3, 2e-3, -1e-3, 0.15, -0.2, 0.31], [0.7, -0.4, 8.0], [0.0, 320.0], HORIZONTAL, scan, True, calibrated)); i += 1
    # tiny-but-not-zero rotation: |w|^2 just above / below DBL_EPSILON
    for mag in (1.0e-8, 2.0e-8):
        w = [mag, 0.0, 0.0]
        cases.append(mp_case(f"tiny{i}", cam, w + [0.0, 0.0, 0.0], w + [0.01, 0.0, 0.0], [0.5, 0.25, 6.0], [640.0, 360.0], HORIZONTAL, scan, True, True)); i += 1
    # large rotations
    for _ in range(3):
        p0 = list(rng.normal(0, 1.2, 3)) + [0.0, 0.0, 0.0]
        # put the point in front of the camera: X = R^T (0,0,7) + small offset
        Rt = np.array([[float(v) for v in mp_rotate([mp.mpf(-p0[0]), mp.mpf(-p0[1]), mp.mpf(-p0[2])], [mp.mpf(a), mp.mpf(b), mp.mpf(c)])] for a, b, c in ((1, 0, 0), (0, 1, 0), (0, 0, 1))]).T
        X = list(Rt @ np.array([0.6, -0.3, 7.0]))
        p1 = [p0[k] + rng.normal(0, 0.01) for k in range(6)]
        cases.append(mp_case(f"big{i}", cam, p0, p1, X, [400.0, 200.0], HORIZONTAL, scan, True, True)); i += 1
    # global-shutter single-pose functor (ReprojectionError), calibrated and with cam block
    for calibrated in (True, False):
        for _ in range(3):
            p0 = rnd_pose(0.3)
            cases.append(mp_case(f"gs{i}", cam2 if calibrated else cam, p0, None, target(p0, p0, rng.uniform(0.2, 0.8)), [float(rng.uniform(100, 1100)), float(rng.uniform(100, 600))], GLOBAL, scan, True, calibrated)); i += 1
    # zero-rotation GS (frame 0 of a fresh session: CeresHandler.h:132-140)
    cases.append(mp_case(f"gs{i}", cam, [0.0] * 6, None, [1.0, -1.0, 10.0], [720.0, 280.0], GLOBAL, scan, True, True)); i += 1
    # behind the camera / z just below and above the 1e-8 gate -> evaluation fails / succeeds
    cases.append(mp_case(f"behind{i}", cam, [0.0] * 6, [0.0] * 6, [0.1, 0.1, -5.0], [640.0, 360.0], HORIZONTAL, scan, True, True)); i += 1
    cases.append(mp_case(f"behind{i}", cam, [0.0] * 6, None, [0.0, 0.0, 0.5e-8], [640.0, 360.0], GLOBAL, scan, True, True)); i += 1
    cases.append(mp_case(f"behind{i}", cam, [0.0] * 6, None, [1e-9, -1e-9, 2e-8], [640.0, 360.0], GLOBAL, scan, True, True)); i += 1
    return cases


def huber_cases():
    out = []
    for a in (0.5, 2.0, 5.0):
        for s in (0.0, 1e-6, 0.2, a * a, a * a * (1 + 1e-12), 7.5, 144.0, 1e6):
            am, sm = mp.mpf(a), mp.mpf(s)
            if sm <= am * am:
                rho = [sm, mp.mpf(1), mp.mpf(0)]
            else:
                r = mp.sqrt(sm)
                rho = [2 * am * r - am * am, am / r, -am / (2 * r * sm)]
            out.append(dict(a=a, s=s, rho=[float(v) for v in rho]))
    return out


# ------------------------------- numpy model of a tiny scene ------------------------------------
def np_rotate(w, p):
    th2 = np.sum(w * w, axis=1)
    small = th2 <= np.finfo(np.float64).eps
    th = np.sqrt(np.where(small, 1.0, th2))
    k = w / th[:, None]
    c, s = np.cos(th)[:, None], np.sin(th)[:, None]
    big = p * c + np.cross(k, p) * s + k * np.sum(k * p, axis=1)[:, None] * (1 - c)
    return np.where(small[:, None], p + np.cross(w, p), big)


def np_residuals(cam, poses, points, obs_xy, obs_f, obs_p, shutter, scan, interp):
    P = poses.shape[1]
    if P == 2 and shutter != GLOBAL:
        tau = np.clip((obs_xy[:, 0] - scan[0]) / float(scan[1] - scan[0]), 0.0, 1.0)[:, None]
        p0, p1 = poses[obs_f, 0], poses[obs_f, 1]
        rot = p0[:, :3] + (p1[:, :3] - p0[:, :3]) * tau if interp else p0[:, :3]
        ctr = p0[:, 3:] + (p1[:, 3:] - p0[:, 3:]) * tau
    else:
        rot, ctr = poses[obs_f, 0, :3], poses[obs_f, 0, 3:]
    pc = np_rotate(rot, points[obs_p] - ctr)
    x, y = pc[:, 0] / pc[:, 2], pc[:, 1] / pc[:, 2]
    r2 = x * x + y * y
    d = 1 + r2 * (cam[2] + r2 * (cam[3] + r2 * cam[6]))
    xd = d * x + 2 * cam[4] * x * y + cam[5] * (r2 + 2 * x * x)
    yd = d * y + cam[4] * (r2 + 2 * y * y) + 2 * cam[5] * x * y
    return np.stack([cam[0] * xd + cam[7] - obs_xy[:, 0], cam[1] * yd + cam[8] - obs_xy[:, 1]], axis=1)


def tiny_scene(seed, F, M, rolling, huber_a, outliers):
    rng = np.random.default_rng(seed)
    cam = np.array([800.0, 800.0, -0.05, 0.01, 1e-3, -1e-3, 2e-3, 640.0, 360.0])
    scan = (0, 1280)
    P = 2 if rolling else 1
    poses = np.zeros((F, P, 6))
    for f in range(F):
        base = np.concatenate([rng.normal(0, 0.03, 3), [1.2 * f, rng.normal(0, 0.1), rng.normal(0, 0.1)]])
        poses[f, 0] = base
        if rolling:
            poses[f, 1] = base + np.concatenate([rng.normal(0, 0.004, 3), [0.25, 0.0, 0.0] + rng.normal(0, 0.01, 3)])
    poses[0, :, :3] = 0.0
    X = np.stack([rng.uniform(-2, 1.2 * F + 1, M), rng.uniform(-2, 2, M), rng.uniform(7, 12, M)], axis=1)
    of, op = np.meshgrid(np.arange(F), np.arange(M), indexing="ij")
    of, op = of.reshape(-1), op.reshape(-1)
    xy = np.tile(cam[7:9], (len(of), 1))
    for _ in range(60):   # fixed point on the scan-line time: projection = residual + "observation"
        xy = np_residuals(cam, poses, X, xy, of, op, HORIZONTAL if rolling else GLOBAL, scan, True) + xy
    ok = (xy[:, 0] > 0) & (xy[:, 0] < 1280) & (xy[:, 1] > 0) & (xy[:, 1] < 720)
    of, op, xy = of[ok], op[ok], xy[ok]
    xy = xy + rng.normal(0, 0.5, xy.shape)
    if outliers:
        bad = rng.random(len(xy)) < 0.08
        xy[bad] += rng.normal(0, 25.0, (bad.sum(), 2))
    cnt = np.bincount(op, minlength=M)
    keep = cnt[op] >= 3
    of, op, xy = of[keep], op[keep], xy[keep]
    used = np.unique(op); remap = -np.ones(M, dtype=int); remap[used] = np.arange(len(used))
    X = X[used]; op = remap[op]
    init_poses = poses + np.concatenate([rng.normal(0, 0.004, poses.shape[:2] + (3,)), rng.normal(0, 0.03, poses.shape[:2] + (3,))], axis=2)
    init_poses[0] = poses[0]
    init_X = X + rng.normal(0, 0.04, X.shape)
    return dict(cam=cam, scan=scan, shutter=HORIZONTAL if rolling else GLOBAL, poses=init_poses, points=init_X,
                obs_xy=xy, obs_frame=of.astype(np.int32), obs_point=op.astype(np.int32), huber_a=huber_a, rolling=rolling)


def np_prior_residuals(poses, kind, scale, ratio, frames):
    """Frame-to-frame motion priors (video_bundler_rs_inter.h:55-173) from their physical meaning:
    the frame's first pose against the previous frame's last pose extrapolated over the inter-frame gap
    (ratio = gap / exposure), the frame's last pose against its first pose extrapolated over the exposure —
    with the previous velocity (kind 1) or with the mean of the previous and the current velocity (kind 2)."""
    out = []
    w = scale * np.array([0.01] * 3 + [1.0] * 3)
    for f in frames:
        start, end, pstart, pend = poses[f, 0], poses[f, 1], poses[f - 1, 0], poses[f - 1, 1]
        if kind == 1:
            r1 = start - (pend + ratio * (pend - pstart))
            r2 = end - (start + (start - pend) / ratio)
        else:
            gap_prev, gap_now = ratio * (pend - pstart), start - pend          # displacement over the gap at either velocity
            r1 = start - (pend + 0.5 * (gap_prev + gap_now))
            exp_prev, exp_now = (start - pend) / ratio, end - start            # displacement over the exposure
            r2 = end - (start + 0.5 * (exp_prev + exp_now))
        out.append(np.concatenate([w * r1, w * r2]))
    return np.array(out)


def minimise_free_ratio(sc):
    """As minimise(), with interFrameRatio as one more unknown bounded below (0 / DBL_EPSILON), scipy's bounded TRF."""
    F, P = sc["poses"].shape[:2]
    free = np.ones((F, P, 6), dtype=bool)
    free[0] = False
    free[-1, -1, 3:] = False
    a = sc["huber_a"]
    kind, scale, ratio0, frames = sc["prior"]
    n = int(free.sum())

    def fun(x):
        poses = sc["poses"].copy(); poses[free] = x[:n]
        pts = x[n:-1].reshape(-1, 3)
        r = np_residuals(sc["cam"], poses, pts, sc["obs_xy"], sc["obs_frame"], sc["obs_point"], sc["shutter"], sc["scan"], True)
        pr = np_prior_residuals(poses, kind, scale, x[-1], frames)
        if a > 0:
            for blk in (r, pr):
                s = np.sum(blk * blk, axis=1)
                rho = np.where(s <= a * a, s, 2 * a * np.sqrt(np.maximum(s, 1e-300)) - a * a)
                blk *= np.sqrt(rho / np.maximum(s, 1e-300))[:, None]
        return np.concatenate([r.reshape(-1), pr.reshape(-1)])

    x0 = np.concatenate([sc["poses"][free], sc["points"].reshape(-1), [ratio0]])
    lo = np.full(len(x0), -np.inf); lo[-1] = 0.0 if kind == 1 else np.finfo(np.float64).eps
    sol = least_squares(fun, x0, bounds=(lo, np.inf), method="trf", x_scale="jac", ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=600)
    sol = least_squares(fun, sol.x, bounds=(lo, np.inf), method="trf", x_scale="jac", ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=600)
    poses = sc["poses"].copy(); poses[free] = sol.x[:n]
    return dict(initial_cost=float(0.5 * np.sum(fun(x0) ** 2)), final_cost=float(sol.cost), grad_inf=float(np.max(np.abs(sol.grad))),
                poses=poses.tolist(), points=sol.x[n:-1].reshape(-1, 3).tolist(), ratio=float(sol.x[-1]), nfev=int(sol.nfev))


def free_ratio_cases():
    out = []
    for name, seed, F, M, a, outl, kind, scale, ratio in (("rs_velocity_free_ratio", 41, 6, 60, 0.0, False, 1, 6.0, 1.0),
                                                         ("rs_acceleration_free_ratio", 42, 7, 70, 0.0, False, 2, 25.0, 1.0)):
        sc = tiny_scene(seed, F, M, True, a, outl)
        frames = list(range(1, F))
        sc["prior"] = (kind, scale, ratio, frames)
        res = minimise_free_ratio(sc)
        out.append(dict(name=name, rolling=True, huber_a=a, cam=sc["cam"].tolist(), scanlines=list(sc["scan"]), shutter=sc["shutter"],
                        poses=sc["poses"].tolist(), points=sc["points"].tolist(), obs_xy=sc["obs_xy"].tolist(),
                        obs_frame=sc["obs_frame"].tolist(), obs_point=sc["obs_point"].tolist(),
                        prior_kind=kind, prior_scale=scale, inter_frame_ratio=ratio, ratio_free=True, prior_frames=frames, expected=res))
        print(name, "cost", res["initial_cost"], "->", res["final_cost"], "ratio", res["ratio"], "|g|inf", res["grad_inf"], file=sys.stderr)
    return out


def minimise(sc):
    """Gauge: frame 0 constant, translation of the last frame's last pose fixed.  Robust cost enters as
    r~ = r * sqrt(rho(s)/s) per 2-D block, so that 1/2 |r~|^2 = 1/2 rho(s) (Ceres' block-wise loss)."""
    F, P = sc["poses"].shape[:2]
    free = np.ones((F, P, 6), dtype=bool)
    free[0] = False
    free[-1, -1, 3:] = False
    a = sc["huber_a"]

    def unpack(x):
        poses = sc["poses"].copy()
        n = free.sum()
        poses[free] = x[:n]
        return poses, x[n:].reshape(-1, 3)

    def fun(x):
        poses, pts = unpack(x)
        r = np_residuals(sc["cam"], poses, pts, sc["obs_xy"], sc["obs_frame"], sc["obs_point"], sc["shutter"], sc["scan"], True)
        if a > 0:
            s = np.sum(r * r, axis=1)
            rho = np.where(s <= a * a, s, 2 * a * np.sqrt(np.maximum(s, 1e-300)) - a * a)
            r = r * np.sqrt(rho / np.maximum(s, 1e-300))[:, None]
        r = r.reshape(-1)
        if sc.get("prior"):
            pr = np_prior_residuals(poses, *sc["prior"])
            if a > 0:   # the shared loss function acts on the 12-D block as a whole
                s = np.sum(pr * pr, axis=1)
                rho = np.where(s <= a * a, s, 2 * a * np.sqrt(np.maximum(s, 1e-300)) - a * a)
                pr = pr * np.sqrt(rho / np.maximum(s, 1e-300))[:, None]
            r = np.concatenate([r, pr.reshape(-1)])
        return r

    x0 = np.concatenate([sc["poses"][free], sc["points"].reshape(-1)])
    sol = least_squares(fun, x0, method="trf", x_scale="jac", ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=400)
    sol = least_squares(fun, sol.x, method="trf", x_scale="jac", ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=400)
    poses, pts = unpack(sol.x)
    return dict(initial_cost=float(0.5 * np.sum(fun(x0) ** 2)), final_cost=float(sol.cost), grad_inf=float(np.max(np.abs(sol.grad))),
                poses=poses.tolist(), points=pts.tolist(), nfev=int(sol.nfev))


def solve_cases():
    out = []
    for name, seed, F, M, rolling, a, outl in (("rs_plain", 11, 6, 60, True, 0.0, False), ("gs_plain", 12, 5, 50, False, 0.0, False),
                                               ("rs_huber", 13, 6, 70, True, 2.0, True)):
        sc = tiny_scene(seed, F, M, rolling, a, outl)
        res = minimise(sc)
        out.append(dict(name=name, rolling=rolling, huber_a=a, cam=sc["cam"].tolist(), scanlines=list(sc["scan"]), shutter=sc["shutter"],
                        poses=sc["poses"].tolist(), points=sc["points"].tolist(), obs_xy=sc["obs_xy"].tolist(),
                        obs_frame=sc["obs_frame"].tolist(), obs_point=sc["obs_point"].tolist(), expected=res))
        print(name, "obs", len(sc["obs_frame"]), "cost", res["initial_cost"], "->", res["final_cost"], "|g|inf", res["grad_inf"], file=sys.stderr)
    return out


def prior_solve_cases():
    out = []
    for name, seed, F, M, a, outl, kind, scale, ratio in (("rs_velocity", 21, 6, 60, 0.0, False, 1, 6.0, 0.8),
                                                         ("rs_acceleration_huber", 22, 7, 70, 2.0, True, 2, 25.0, 1.25)):
        sc = tiny_scene(seed, F, M, True, a, outl)
        frames = list(range(1, F))
        sc["prior"] = (kind, scale, ratio, frames)
        res = minimise(sc)
        res["prior_residuals_at_start"] = np_prior_residuals(sc["poses"], kind, scale, ratio, frames).tolist()
        out.append(dict(name=name, rolling=True, huber_a=a, cam=sc["cam"].tolist(), scanlines=list(sc["scan"]), shutter=sc["shutter"],
                        poses=sc["poses"].tolist(), points=sc["points"].tolist(), obs_xy=sc["obs_xy"].tolist(),
                        obs_frame=sc["obs_frame"].tolist(), obs_point=sc["obs_point"].tolist(),
                        prior_kind=kind, prior_scale=scale, inter_frame_ratio=ratio, prior_frames=frames, expected=res))
        print(name, "obs", len(sc["obs_frame"]), "cost", res["initial_cost"], "->", res["final_cost"], "|g|inf", res["grad_inf"], file=sys.stderr)
    return out


def pnp_cases():
    """RS-PnP refinement (solveRSpnp.cpp:100-192): one rolling-shutter frame, float points and observations
    as constants, the two poses free — minimised by scipy from the same start the solvers get."""
    out = []
    cam = np.array([800.0, 800.0, -0.05, 0.01, 1e-3, -1e-3, 2e-3, 640.0, 360.0])
    for name, seed, m in (("pnp_24", 31, 24), ("pnp_60", 32, 60)):
        rng = np.random.default_rng(seed)
        scan = (0, 1280)
        pose0 = np.concatenate([rng.normal(0, 0.05, 3), rng.normal(0, 0.3, 3)])
        poses = np.stack([pose0, pose0 + np.concatenate([rng.normal(0, 0.01, 3), [0.35, 0.05, -0.04]])])[None]
        X = np.stack([rng.uniform(-4, 4, m), rng.uniform(-2.5, 2.5, m), rng.uniform(7, 14, m)], axis=1).astype(np.float32).astype(np.float64)
        of, op = np.zeros(m, dtype=int), np.arange(m)
        xy = np.tile(cam[7:9], (m, 1))
        for _ in range(60):
            xy = np_residuals(cam, poses, X, xy, of, op, HORIZONTAL, scan, True) + xy
        xy = (xy + rng.normal(0, 0.5, xy.shape)).astype(np.float32).astype(np.float64)
        init = poses + np.concatenate([rng.normal(0, 0.01, (1, 2, 3)), rng.normal(0, 0.08, (1, 2, 3))], axis=2)

        def fun(x):
            return np_residuals(cam, x.reshape(1, 2, 6), X, xy, of, op, HORIZONTAL, scan, True).reshape(-1)

        sol = least_squares(fun, init.reshape(-1), method="trf", x_scale="jac", ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=400)
        sol = least_squares(fun, sol.x, method="trf", x_scale="jac", ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=400)
        out.append(dict(name=name, cam=cam.tolist(), scanlines=list(scan), shutter=HORIZONTAL, object_points=X.tolist(), image_points=xy.tolist(),
                        init_poses=init[0].tolist(), expected=dict(initial_cost=float(0.5 * np.sum(fun(init.reshape(-1)) ** 2)), final_cost=float(sol.cost),
                                                                   poses=sol.x.reshape(2, 6).tolist(), grad_inf=float(np.max(np.abs(sol.grad))))))
        print(name, "cost", out[-1]["expected"]["initial_cost"], "->", sol.cost, "|g|inf", np.max(np.abs(sol.grad)), file=sys.stderr)
    return out


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "free_ratio":   # only the free interFrameRatio cases
        with open(os.path.join(HERE, "free_ratio_solves.json"), "w") as f:
            json.dump(free_ratio_cases(), f)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "pnp":       # only the RS-PnP cases
        with open(os.path.join(HERE, "pnp_solves.json"), "w") as f:
            json.dump(pnp_cases(), f)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "priors":   # only the motion-prior cases (added after the others were committed)
        with open(os.path.join(HERE, "prior_solves.json"), "w") as f:
            json.dump(prior_solve_cases(), f)
        return
    with open(os.path.join(HERE, "per_observation.json"), "w") as f:
        json.dump(per_observation_cases(), f, indent=0)
    with open(os.path.join(HERE, "huber.json"), "w") as f:
        json.dump(huber_cases(), f, indent=0)
    with open(os.path.join(HERE, "tiny_solves.json"), "w") as f:
        json.dump(solve_cases(), f)


if __name__ == "__main__":
    main()
