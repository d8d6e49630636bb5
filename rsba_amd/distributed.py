"""Multi-GPU driver for the LM solve: one process per GPU, torch.distributed over RCCL.

Observations are partitioned by point (BAProblem.shard); cameras are replicated.  [single-GPU timing
helper only for now — the exchange step lands with the sharded solve]
"""
from __future__ import annotations

import time


def solve_timed(dp, prob, world: int, iters: int):
    """Run `iters` LM iterations of the device solver on this rank's problem and report wall time per
    iteration (the metric's second half).  Parameters are restored afterwards."""
    from . import capi
    saved = (prob.poses.copy(), prob.points.copy())
    t0 = time.perf_counter()
    s, trace = dp.solve(capi.default_options(max_num_iterations=iters, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0))
    wall = time.perf_counter() - t0
    prob.poses[:], prob.points[:] = saved
    dp.upload_parameters()
    n_it = max(1, s.num_iterations - 1)
    return {"iterations": n_it, "ms_per_lm_iteration": s.total_time_s / n_it * 1e3, "wall_s": wall,
            "initial_cost": s.initial_cost, "final_cost": s.final_cost,
            "residual_jacobian_s": s.residual_jacobian_time_s, "linear_solver_s": s.linear_solver_time_s,
            "n_gpus": world, "note": "first solve includes the one-off symbolic phase"}
