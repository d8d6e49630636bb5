"""Per-phase device time of the LM iteration (HIP events recorded by the solver on its own stream, rsba_solver_options::profile_phases)
for one configuration.  usage: python tools/phase_time.py [C4] [iters]   (environment knobs of the solver apply)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rsba_amd import capi
from rsba_amd.scene import make_config
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
prob = make_config(name).problem
p0, x0, c0 = prob.poses.copy(), prob.points.copy(), prob.intrinsics.copy()
with capi.DeviceProblem(prob) as dp:
    for rep in range(2):
        prob.poses[...] = p0; prob.points[...] = x0; prob.intrinsics[...] = c0
        dp.upload_parameters()
        s, _ = dp.solve(capi.default_options(max_num_iterations=iters, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0, profile_phases=1))
    t, st = dp.phase_times(), dp.plan_stats()
n = max(1, s.num_iterations - 1)
tot = sum(ms for ms, _ in t.values())
print(f"{name}: {s.num_iterations} iterations, final cost {s.final_cost:.9e}, sum of phases {tot / n:.3f} ms/iteration; chunks {st['schur_chunks']}, "
      f"MFMAs/entry {st['schur_mfma_issued'] / max(1, st['schur_launches']) / max(1, st['schur_entries']):.2f}")
print("  " + "  ".join(f"{k} {ms / n:.3f}" for k, (ms, c) in t.items() if c))
