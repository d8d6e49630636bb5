"""Stress test of this round's reorganised passes: random scenes of random sizes, each solved with the default forms — resident Schur
workgroups that take chunk after chunk, the fused projection / virtual-record sweep — and with the forms they replaced
(RSBA_SCHUR_VARIANT=2: a workgroup per chunk; RSBA_NO_FUSED_SWEEP=1: two passes); every result must agree to the bit.
usage: python tools/schur_stress.py [seconds] [seed]"""
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
while time.time() < t_end:
    frames = int(rng.integers(6, 600)); points = int(rng.integers(50, 60 * frames))
    rolling = bool(rng.integers(0, 4) != 0); shared = bool(rng.integers(0, 2)); huber = float(rng.choice([0.0, 2.0]))
    seed = int(rng.integers(0, 1 << 30)); tl = int(rng.integers(6, 30))
    block = int(rng.choice([0, 0, 16, 64, 512]))     # chunk numbering by blocks of points: thousands of small chunks

    def solve(old_forms):
        for k, v in (("RSBA_SCHUR_VARIANT", "2"), ("RSBA_NO_FUSED_SWEEP", "1")):
            if old_forms: os.environ[k] = v
            else: os.environ.pop(k, None)
        if block: os.environ["RSBA_SCHUR_BLOCK"] = str(block)
        else: os.environ.pop("RSBA_SCHUR_BLOCK", None)
        p = make_scene(frames, points, rolling=rolling, seed=seed, outlier_ratio=0.03 if huber else 0.0, track_len=tl).problem
        p.huber_a = huber; p.calibrated = not shared
        apply_gauge_masks(p, fix_first_n_cameras=1)
        p.pose_fixed_mask[-1, -1] |= 0b111000
        with capi.DeviceProblem(p) as dp:
            s, tr = dp.solve(capi.default_options(max_num_iterations=4))
            st = dp.plan_stats()
        return (s.final_cost, [t.cost for t in tr], p.poses.copy(), p.points.copy(), p.intrinsics.copy()), st["schur_chunks"]

    (a, chunks), (b, _) = solve(False), solve(True)
    runs += 1
    same = a[0] == b[0] and a[1] == b[1] and all(np.array_equal(x, y) for x, y in zip(a[2:], b[2:]))
    if not same:
        bad += 1
        print(f"MISMATCH: frames {frames} points {points} rolling {rolling} shared {shared} huber {huber} seed {seed} tl {tl} block {block} chunks {chunks}", flush=True)
    if runs % 100 == 0: print(f"progress: {runs} problems, {bad} mismatches", flush=True)
print(f"schur_stress: {runs} random problems, {bad} mismatches")
