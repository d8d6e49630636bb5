"""Throughput of the f2 filter kernels (validate / reproject) at a BASELINE config; run under rocprofv3 --kernel-trace
--stats for the kernel times.  usage: python tools/filter_time.py [C4]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rsba_amd import capi
from rsba_amd.scene import make_config
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
p = make_config(name).problem
with capi.DeviceProblem(p) as dp:
    for _ in range(5):
        v = dp.validate_observations(16.0, 0.0)
    t = time.perf_counter()
    for _ in range(10):
        v = dp.validate_observations(16.0, 0.0)
    tv = (time.perf_counter() - t) / 10
    for _ in range(5):
        xy, ok = dp.reproject(p.obs_frame, p.obs_point)
    t = time.perf_counter()
    for _ in range(10):
        xy, ok = dp.reproject(p.obs_frame, p.obs_point)
    tr = (time.perf_counter() - t) / 10
print(f"{name}: validate {p.num_observations} observations: {tv * 1e3:.2f} ms per call incl. D2H ({p.num_observations / tv:.3g} obs/s), valid {v.mean():.3f}; "
      f"reproject: {tr * 1e3:.2f} ms per call incl. H2D + D2H ({p.num_observations / tr:.3g} pairs/s), ok {ok.mean():.3f}")
