"""Stress test of the persistent task-DAG Cholesky: random scenes of random sizes, each solved with the DAG driver
and with the per-level schedule; every result must agree to the bit.  The verification of the DAG result (solver.hip) is on
as in production: a solve it rejects is repeated on the level schedule and counted as a fallback — both counts should be 0.  usage: python tools/dag_stress.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rsba_amd import capi
from rsba_amd.problem import apply_gauge_masks
from rsba_amd.scene import make_scene

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
runs = bad = 0
fallbacks = 0
while time.time() < t_end:
    frames = int(rng.integers(6, 400)); points = int(rng.integers(50, 40 * frames))
    rolling = bool(rng.integers(0, 2)); shared = bool(rng.integers(0, 4) == 0); huber = float(rng.choice([0.0, 2.0]))
    seed = int(rng.integers(0, 1 << 30))
    tl = int(rng.integers(6, 30))
    prior = int(rng.integers(0, 3)) if rolling and rng.integers(0, 3) == 0 else 0   # motion priors on a third of the rolling scenes
    free = bool(prior and rng.integers(0, 2))                                       # half of those with the free interFrameRatio
    own_cams = bool(rng.integers(0, 3) == 0)
    pose_priors = bool(rng.integers(0, 5) == 0)

    def solve(mode):
        os.environ["RSBA_CHOL_LEVELS"] = mode
        p = make_scene(frames, points, rolling=rolling, seed=seed, outlier_ratio=0.03 if huber else 0.0, track_len=tl).problem
        p.huber_a = huber
        p.calibrated = not shared
        apply_gauge_masks(p, fix_first_n_cameras=1)
        p.pose_fixed_mask[-1, -1] |= 0b111000
        if prior:
            p.prior_kind, p.prior_scale, p.inter_frame_ratio = prior, 10.0, 0.8
            p.prior_frames = np.arange(1, frames, dtype=np.int32)
            if free:
                p.inter_frame_ratio, p.ratio_free = 1.0, True
        if own_cams and shared:                                                       # per-frame intrinsics blocks on every third frame
            fi = np.zeros(frames, dtype=np.int32); own = np.arange(frames) % 3 == 2; fi[own] = 1 + np.arange(own.sum())
            p.frame_intrinsics = fi; p.intrinsics = np.tile(p.intrinsics[:1], (1 + int(own.sum()), 1))
        if pose_priors:
            blocks = np.arange(2 if rolling else 1, p.num_frames * p.poses_per_frame, 3, dtype=np.int32)
            p.pose_prior_block = blocks
            p.pose_prior_values = p.poses.reshape(-1, 6)[blocks] + np.random.default_rng(seed).normal(0, 0.01, (len(blocks), 6))
            p.pose_prior_rotation, p.pose_prior_position = 3.0, 5.0
        with capi.DeviceProblem(p) as dp:
            s, _ = dp.solve(capi.default_options(max_num_iterations=5))
        global fallbacks
        fallbacks += s.num_dag_fallbacks
        return (s.final_cost, p.poses.copy(), p.points.copy(), p.intrinsics.copy(), np.array([p.inter_frame_ratio]))

    def same(a, b):
        return a[0] == b[0] and all(np.array_equal(x, y) for x, y in zip(a[1:], b[1:]))

    res = [solve("0"), solve("1")]
    runs += 1
    if runs % 200 == 0:
        print(f"progress: {runs} problems, {bad} mismatches, {fallbacks} fallbacks", flush=True)   # survives a cut-off run
    if not same(res[0], res[1]):
        bad += 1
        # which side moved?  solve each driver twice more
        again = {m: [solve(m), solve(m)] for m in ("0", "1")}
        print(f"MISMATCH frames={frames} points={points} rolling={rolling} shared={shared} huber={huber} seed={seed} track_len={tl} prior={prior} free={free}: "
              f"dag {res[0][0]!r} levels {res[1][0]!r}; max pose diff {np.abs(res[0][1] - res[1][1]).max():.3e}; "
              f"dag repeats equal first dag run: {[same(res[0], r) for r in again['0']]}, equal levels: {[same(res[1], r) for r in again['0']]}; "
              f"levels repeats equal first levels run: {[same(res[1], r) for r in again['1']]}", flush=True)
print(f"dag_stress: {runs} random problems, {bad} mismatches, {fallbacks} verification fallbacks (RSBA_CHOL_VERIFY={os.environ.get('RSBA_CHOL_VERIFY', '1')})")
sys.exit(1 if bad else 0)
