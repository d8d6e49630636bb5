#!/usr/bin/env python3
"""Generates tests/golden/c4_trajectory.json and c5_trajectory.json: the CPU oracle's LM trajectories on the
full-size BASELINE configurations C4 (1000 frames / 100k points / ~2.04M observations, calibrated) and C5
(4000 frames / 500k points / ~10.3M observations, Huber loss + shared intrinsics as a parameter block).

The oracle's reduced camera system lives in envelope storage (oracle/rsba_oracle.cpp: EnvMatrix), which is what makes
these sizes tractable on a CPU: C4 takes about a minute per LM iteration on 8 cores, C5 several.  The -m gpu tests
(tests/test_gpu_fullsize.py) rebuild the same scenes from the same seeds (rsba_amd/scene.py is deterministic) and compare
the device solver with what is stored here: per-iteration cost / step records, the summary, and a fixed sample of the
solved parameters.

A name with the suffix "free" (C4free, C5free) is the same scene with NOTHING held fixed — the reference's default options
(SfmOptions.h:66-70: fixFirstNCameras = 0, fixScale / fixRotation / fixPosition off, so CeresHandler.h:342-382 marks no block
constant): the reduced camera system is rank deficient by the seven gauge freedoms and only the LM damping makes it definite
("a second run with nothing fixed (reference default) for parity of the rank-deficient case").  Stored as
c4_free_gauge_trajectory.json / c5_free_gauge_trajectory.json: the step-by-step trajectory and a shorter long run.

This is a checker-vs-product comparison at the headline size; it does not pin the oracle itself (see
rsba_oracle_math.hpp: "parity unpinned").  Run (build container, CPU only):
    python tests/golden/make_trajectories.py C4 [C5] [C4free] [C5free]
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from rsba_amd.scene import make_config  # noqa: E402

# (LM iterations of the step-by-step trajectory, iterations of the long run)
PLAN = {"C4": (6, 40), "C5": (4, 12), "C4free": (6, 16), "C5free": (4, 6)}
POSE_STRIDE = {"C4": 37, "C5": 149}
POINT_STRIDE = {"C4": 2003, "C5": 10007}


def record(t):
    return dict(iteration=t.iteration, step_is_valid=t.step_is_valid, step_is_successful=t.step_is_successful, cost=t.cost, cost_change=t.cost_change,
                gradient_max_norm=t.gradient_max_norm, step_norm=t.step_norm, relative_decrease=t.relative_decrease,
                trust_region_radius=t.trust_region_radius, model_cost_change=t.model_cost_change)


def summary(s):
    return {k: getattr(s, k) for k in ("termination_type", "num_successful_steps", "num_unsuccessful_steps", "num_iterations", "num_residual_blocks",
                                       "num_residual_blocks_reduced", "num_parameters_reduced", "initial_cost", "final_cost", "fixed_cost")}


def run(name):
    iters, iters_min = PLAN[name]
    free = name.endswith("free")
    name = name[:-4] if free else name
    sc = make_config(name, gauge=not free)
    p = sc.problem.copy()
    out = dict(config=name, gauge="nothing fixed (SfmOptions.h:66-70 defaults)" if free else "frame 0 constant, translation of the last pose fixed", num_frames=p.num_frames, num_points=p.num_points, num_observations=p.num_observations,
               obs_checksum=float(np.sum(p.obs_xy)), pose_checksum=float(np.sum(p.poses)), point_checksum=float(np.sum(p.points)),
               pose_stride=POSE_STRIDE[name], point_stride=POINT_STRIDE[name])
    ok, cost, g = O.evaluate(p, gradient=True)
    assert ok
    out["evaluate"] = dict(cost=cost, gradient_pose_sample=g["poses"][::POSE_STRIDE[name]].ravel().tolist(),
                           gradient_point_sample=g["points"][::POINT_STRIDE[name]].ravel().tolist(),
                           gradient_intrinsics=g["intrinsics"].ravel().tolist(), gradient_abs_sum=float(np.abs(g["poses"]).sum() + np.abs(g["points"]).sum()))
    t0 = time.time()
    s, tr = O.solve(p, O.default_options(max_num_iterations=iters, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0))
    out["trajectory"] = dict(options=dict(max_num_iterations=iters, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0),
                             iterations=[record(t) for t in tr], summary=summary(s), wall_s=time.time() - t0,
                             pose_sample=p.poses[::POSE_STRIDE[name]].ravel().tolist(), point_sample=p.points[::POINT_STRIDE[name]].ravel().tolist(),
                             intrinsics=p.intrinsics.ravel().tolist())
    print(name, "trajectory", [t.cost for t in tr], f"{time.time() - t0:.0f} s", flush=True)
    if iters_min:
        # the long run: Ceres' default tolerances, as BA() / CeresHandler::solve leave them (these scenes do not converge
        # inside the cap: the cost still falls by ~4e-4 per iteration, so no termination test is borderline)
        q = sc.problem.copy()
        t0 = time.time()
        s, tr = O.solve(q, O.default_options(max_num_iterations=iters_min))
        out["long"] = dict(options=dict(max_num_iterations=iters_min), summary=summary(s),
                           costs=[t.cost for t in tr], successful=[t.step_is_successful for t in tr], wall_s=time.time() - t0,
                           pose_sample=q.poses[::POSE_STRIDE[name]].ravel().tolist(), point_sample=q.points[::POINT_STRIDE[name]].ravel().tolist(),
                           intrinsics=q.intrinsics.ravel().tolist())
        print(name, "long", s.final_cost, s.num_iterations, f"{time.time() - t0:.0f} s", flush=True)
    with open(os.path.join(HERE, f"{name.lower()}{'_free_gauge' if free else ''}_trajectory.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    for name in (sys.argv[1:] or ["C4"]):
        run(name)
