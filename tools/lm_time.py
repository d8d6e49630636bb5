"""LM iteration wall time of one configuration with the DAG Cholesky and with the per-level schedule (same task
bodies), and the difference of the two solves.  usage: python tools/lm_time.py [C4] [iters] [priors]
("per_frame_intrinsics": every frame its own intrinsics block; "priors": constant-velocity motion priors between all consecutive frames, scale 10, interFrameRatio 0.8;
 "free_ratio": the same with the ratio a free, lower-bounded parameter starting at 1; "-" for none of them)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rsba_amd import capi
from rsba_amd.scene import make_config

name = sys.argv[1] if len(sys.argv) > 1 else "C4"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
res = {}
for mode in (("dag",) if "dagonly" in sys.argv else ("dag", "levels")):   # ("dagonly" as a last argument: skip the level schedule and the comparison)
    os.environ["RSBA_CHOL_LEVELS"] = "1" if mode == "levels" else "0"
    prob = make_config(name).problem
    if len(sys.argv) > 3 and sys.argv[3] in ("priors", "free_ratio"):
        prob.prior_kind, prob.prior_scale, prob.inter_frame_ratio = 1, 10.0, 0.8
        prob.prior_frames = np.arange(1, prob.num_frames, dtype=np.int32)
        if sys.argv[3] == "free_ratio":   # the reference's default: the ratio starts at 1 and is solved for
            prob.inter_frame_ratio, prob.ratio_free = 1.0, True
    if len(sys.argv) > 3 and sys.argv[3] == "per_frame_intrinsics":   # every frame its own f.cam block (CeresHandler.h:256-264,273-280): the problems that keep records
        prob.calibrated = False; prob.huber_a = 2.0
        rng = np.random.default_rng(3)
        prob.intrinsics = np.tile(prob.intrinsics[:1], (prob.num_frames, 1)) * (1.0 + 1e-3 * rng.normal(size=(prob.num_frames, 9)) * np.array([[1, 1, 20, 20, 10, 10, 10, 0.5, 0.5]]))
        prob.frame_intrinsics = np.arange(prob.num_frames, dtype=np.int32)
    p0, x0 = prob.poses.copy(), prob.points.copy()
    i0 = prob.intrinsics.copy()
    with capi.DeviceProblem(prob) as dp:
        opt = capi.default_options(max_num_iterations=iters, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
        for rep in range(2):
            prob.poses[...] = p0; prob.points[...] = x0; prob.intrinsics[...] = i0
            dp.upload_parameters()
            t0 = time.perf_counter()
            summ, trace = dp.solve(opt)
            dt = time.perf_counter() - t0
        n = max(summ.num_iterations - 1, 1)   # LM iterations = steps taken (num_iterations also counts iteration 0, the initial evaluation): the divisor bench.py uses
        # the steady-state cost of one more iteration: a solve of twice as many against this one (what DESIGN.md / README lead with)
        prob.poses[...] = p0; prob.points[...] = x0; prob.intrinsics[...] = i0
        dp.upload_parameters()
        opt2 = capi.default_options(max_num_iterations=2 * iters, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
        summ2, _ = dp.solve(opt2)
        extra = summ2.num_iterations - summ.num_iterations
        marginal = (summ2.total_time_s - summ.total_time_s) / extra * 1e3 if extra > 0 else float("nan")
        prob.poses[...] = p0; prob.points[...] = x0; prob.intrinsics[...] = i0
        dp.upload_parameters()
        summ, trace = dp.solve(opt)
        print(f"{name} {mode}: {summ.num_iterations - 1} LM iterations, marginal {marginal:.3f} ms per iteration (solve of {2 * iters} - solve of {iters}, per extra iteration); "
              f"whole solve / {n} iterations {1e3 * dt / n:.3f} ms (wall {dt * 1e3:.1f} ms: incl. iteration 0 and the write-back), "
              f"cost {summ.initial_cost:.6e} -> {summ.final_cost:.9e}" + (f", interFrameRatio {prob.inter_frame_ratio:.9f}" if prob.prior_kind else ""), flush=True)
        res[mode] = (summ.final_cost, prob.poses.copy(), prob.points.copy())
if "levels" in res: print("final cost rel diff", abs(res["dag"][0] - res["levels"][0]) / res["levels"][0],
      "max pose diff", np.abs(res["dag"][1] - res["levels"][1]).max(), "max point diff", np.abs(res["dag"][2] - res["levels"][2]).max())
