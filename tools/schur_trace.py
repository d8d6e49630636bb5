"""Timeline of the Schur kernel's chunks (RSBA_SCHUR_TRACE): per chunk the time to stage its tables, the MFMA loop and the epilogue.
usage: python tools/schur_trace.py [C4]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
path = "/tmp/schur_trace.bin"
os.environ["RSBA_SCHUR_TRACE"] = path
from rsba_amd import capi
from rsba_amd.scene import make_config
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
prob = make_config(name).problem
with capi.DeviceProblem(prob) as dp:
    dp.solve(capi.default_options(max_num_iterations=3))
tr = np.fromfile(path, dtype=np.int64).reshape(-1, 8)
t0 = tr[:, 2].min()
us = (tr[:, 2:6] - t0) * 0.01
n = tr[:, 6]
print(f"{len(tr)} chunks, span {us[:, 3].max():.1f} us; entries per chunk: mean {n.mean():.0f}, p10 {np.percentile(n, 10):.0f}, p50 {np.median(n):.0f}, max {n.max()}")
pre, loop, epi = us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2]
print(f"tables: mean {pre.mean():.2f} us (p90 {np.percentile(pre, 90):.2f}); loop: mean {loop.mean():.2f} us; epilogue: mean {epi.mean():.2f} us (p90 {np.percentile(epi, 90):.2f})")
print(f"sum over chunks: tables {pre.sum():.0f}, loop {loop.sum():.0f}, epilogue {epi.sum():.0f} us -> per slot of 512: {(pre.sum() + loop.sum() + epi.sum()) / 512:.1f} us")
print(f"loop time per entry: {1e3 * loop.sum() / n.sum():.1f} ns; per group of four entries of a wave: {1e3 * loop.sum() / (n.sum() / 16):.0f} ns")
for lo, hi in ((0, 64), (64, 128), (128, 256), (256, 384), (384, 513)):
    m = (n >= lo) & (n < hi)
    if m.any(): print(f"  chunks with {lo:3d}..{hi - 1:3d} entries: {m.sum():5d}, loop {loop[m].mean():6.2f} us, {1e3 * loop[m].sum() / n[m].sum():6.1f} ns per entry")
starts = np.sort(us[:, 0]); ends = np.sort(us[:, 3])
grid = np.linspace(0, us[:, 3].max(), 21)
print("resident chunks over time:", [int((starts <= g).sum() - (ends <= g).sum()) for g in grid])
print(f"shader clock inside the loops: {tr[:, 7].sum() / (loop.sum() * 1e-6) * 1e-6:.0f} MHz (s_memtime cycles / wall time of the loops)")
hw = tr[:, 1]
print("distinct (se, cu, simd ...) ids seen:", len(np.unique(hw >> 8)))
