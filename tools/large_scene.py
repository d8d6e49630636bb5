"""A scene beyond the BASELINE configurations: does the plan and the solve hold up at twice C5?  usage: python tools/large_scene.py [frames] [points] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rsba_amd import capi
from rsba_amd.problem import apply_gauge_masks
from rsba_amd.scene import make_scene
F = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 6
t0 = time.perf_counter()
p = make_scene(F, M, seed=11).problem
apply_gauge_masks(p, fix_first_n_cameras=1)
p.pose_fixed_mask[-1, -1] |= 0b111000
print(f"scene: {p.num_frames} frames, {p.num_points} points, {p.num_observations} observations ({time.perf_counter() - t0:.1f} s to generate)", flush=True)
t0 = time.perf_counter()
with capi.DeviceProblem(p) as dp:
    t1 = time.perf_counter()
    st = dp.plan_stats()
    t2 = time.perf_counter()
    ms = dp.time_evaluate(True, warmup=5, iters=20)
    s, tr = dp.solve(capi.default_options(max_num_iterations=iters, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0))
    t3 = time.perf_counter()
    free_b, total_b = torch.cuda.mem_get_info(0)
    print(f"rsba_create {1e3 * (t1 - t0):.0f} ms, symbolic phase {1e3 * (t2 - t1):.0f} ms; evaluation {ms:.3f} ms ({p.num_observations / ms / 1e6:.2f} G observations / s); "
          f"{s.num_iterations - 1} LM iterations {1e3 * s.total_time_s / max(1, s.num_iterations - 1):.2f} ms each, cost {s.initial_cost:.6e} -> {s.final_cost:.6e}, "
          f"{s.num_dag_fallbacks} fallbacks; device memory in use {(total_b - free_b) / 1e9:.1f} GB", flush=True)
    print({k: st[k] for k in ("tiles", "factor_tiles", "levels", "tasks", "schur_entries", "schur_chunks")})
