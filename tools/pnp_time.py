"""Throughput of the batched RS-PnP hypotheses (rsba_pnp_tasks) and the oracle's per-hypothesis time beside it.
usage: python tools/pnp_time.py [hypotheses=4096] [points=1000] [subset=6]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from rsba_amd import capi
from oracle import oracle as O
from test_gpu_pnp import CAM, pnp_scene, random_subsets

H = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
m = int(sys.argv[3]) if len(sys.argv) > 3 else 6
sc = pnp_scene(O, 1, n=n)
subs = random_subsets(np.random.default_rng(1), len(sc["X"]), H, m)
for rep in range(3):
    t0 = time.perf_counter()
    out = capi.pnp_tasks(CAM, 1, sc["scan"], sc["X"], sc["xy"], subs, sc["init"], 10, 3.0)
    dt = time.perf_counter() - t0
print(f"device: {H} hypotheses x {m} points, {len(sc['X'])} points scored each: {dt * 1e3:.2f} ms per call incl. staging ({H / dt:.3g} hypotheses/s); "
      f"best {out['num_inliers'].max()} inliers of {(~sc['outlier']).sum()} true")
k = min(H, 200)
t0 = time.perf_counter()
for h in range(k):
    O.pnp_task(CAM, 1, sc["scan"], sc["X"], sc["xy"], subs[h], sc["init"], 10, 3.0)
dt = (time.perf_counter() - t0) / k
print(f"oracle (1 thread): {dt * 1e6:.0f} us per hypothesis ({1 / dt:.3g} hypotheses/s)")
