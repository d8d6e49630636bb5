"""Several independent bundle adjustments on one GPU at once: K host threads, each with its own handle (own streams), solving the same
100-camera window over and over — what a host that serves several video streams would do.  An LM iteration of that size is a latency
chain that keeps a few per cent of the chip busy, so the solves should overlap.  usage: python tools/concurrent_handles.py [C2] [repeats]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rsba_amd import capi
from rsba_amd.scene import make_config
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
base = make_config(name).problem
opt = capi.default_options(max_num_iterations=12, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
with capi.DeviceProblem(base.copy()) as warm:
    warm.solve(opt)

def worker(k, out, fresh):
    p = base.copy(); p0, x0 = p.poses.copy(), p.points.copy()
    dp = None if fresh else capi.DeviceProblem(p)
    t0 = time.perf_counter()
    for _ in range(reps):
        p.poses[...] = p0; p.points[...] = x0
        if fresh:
            with capi.DeviceProblem(p) as d: s, _t = d.solve(opt)
        else:
            dp.upload_parameters(); s, _t = dp.solve(opt)
    out[k] = (time.perf_counter() - t0, s.final_cost)
    if dp: dp.close()

for fresh in (False, True):
    for K in (1, 2, 4, 8):
        out = [None] * K
        th = [threading.Thread(target=worker, args=(k, out, fresh)) for k in range(K)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        wall = time.perf_counter() - t0
        assert len({o[1] for o in out}) == 1
        print(f"{name} {'fresh handle per solve' if fresh else 'one handle per thread   '} K={K}: {K * reps / wall:7.1f} solves/s ({1e3 * wall / reps:6.2f} ms per round of {K}; 12 iterations each), final cost {out[0][1]:.9e}", flush=True)
