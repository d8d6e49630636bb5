import os, sys, time
sys.path.insert(0, "/root/repo")
from rsba_amd import capi
from rsba_amd.scene import make_config
prob = make_config("C4").problem
p0, x0 = prob.poses.copy(), prob.points.copy()
with capi.DeviceProblem(prob) as dp:
    for it in (12, 0, 0, 1, 1, 2, 6, 12, 12, 24):
        prob.poses[...] = p0; prob.points[...] = x0
        dp.upload_parameters()
        t0 = time.perf_counter()
        s, _ = dp.solve(capi.default_options(max_num_iterations=it, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0))
        dt = time.perf_counter() - t0
        print(f"max_iter {it}: {s.num_iterations} iterations, wall {dt*1e3:.3f} ms, total_time_s {s.total_time_s*1e3:.3f} ms, jac {s.residual_jacobian_time_s*1e3:.3f} lin {s.linear_solver_time_s*1e3:.3f}")
