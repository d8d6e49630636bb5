"""Steady-state timing of the evaluation kernel (diagnostic, GPU box)"""
import sys
sys.path.insert(0, ".")
from rsba_amd import capi
from rsba_amd.scene import make_config
p = make_config("C4").problem
with capi.DeviceProblem(p) as dp:
    for w, n in ((50, 200), (500, 500)):
        print(f"jac warmup {w:5d} iters {n:5d}: {dp.time_evaluate(True, w, n)*1e3:7.1f} us   res-only {dp.time_evaluate(False, w, n)*1e3:7.1f} us", flush=True)
