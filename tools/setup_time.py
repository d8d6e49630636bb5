"""What a fresh handle costs before its first LM iteration — the per-call cost of windowedBA, which builds a new problem per frame
(VideoSfMClient.cc:231-251 -> VideoSfMHandler.cc:185-214): rsba_create (upload, index check, sort), the symbolic phase + allocations
of the first solve, and the steady-state iteration for comparison.  usage: python tools/setup_time.py [C4] (RSBA_DEBUG_PLAN=1: host phases)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rsba_amd import capi
from rsba_amd.scene import make_config
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
prob = make_config(name).problem
p0, x0 = prob.poses.copy(), prob.points.copy()
with capi.DeviceProblem(prob) as warm:      # the process's own first-touch costs (HIP context, kernel images) are not the handle's
    warm.solve(capi.default_options(max_num_iterations=2))
for rep in range(3):
    prob.poses[...] = p0; prob.points[...] = x0
    t0 = time.perf_counter()
    dp = capi.DeviceProblem(prob)
    t1 = time.perf_counter()
    st = dp.plan_stats()                     # runs the symbolic phase
    t2 = time.perf_counter()
    s, _ = dp.solve(capi.default_options(max_num_iterations=12, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0))
    t3 = time.perf_counter()
    prob.poses[...] = p0; prob.points[...] = x0
    dp.upload_parameters()
    s2, _ = dp.solve(capi.default_options(max_num_iterations=12, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0))
    t4 = time.perf_counter()
    dp.close()
    t5 = time.perf_counter()
    print(f"{name} fresh handle {rep}: rsba_create {1e3 * (t1 - t0):.1f} ms, symbolic phase + allocations {1e3 * (t2 - t1):.1f} ms, first solve ({s.num_iterations - 1} iterations) "
          f"{1e3 * (t3 - t2):.1f} ms, second solve {1e3 * (t4 - t3):.1f} ms, destroy {1e3 * (t5 - t4):.1f} ms -> create + plan + first solve {1e3 * (t3 - t0):.1f} ms", flush=True)
