"""Timeline of the DAG Cholesky's tasks (RSBA_CHOL_TRACE): per kind, how long tasks waited for inputs and ran,
and the end-to-end span.  usage: python tools/chol_trace.py [C4]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
path = "/tmp/chol_trace.bin"
os.environ["RSBA_CHOL_TRACE"] = path
from rsba_amd import capi
from rsba_amd.scene import make_config
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
prob = make_config(name).problem
with capi.DeviceProblem(prob) as dp:
    dp.solve(capi.default_options(max_num_iterations=4))
raw = open(path, "rb").read()
n = np.frombuffer(raw[:4], dtype=np.int32)[0]
tasks = np.frombuffer(raw[4:4 + 8 * n], dtype=np.int32).reshape(n, 2)
tr = np.frombuffer(raw[4 + 8 * n:], dtype=np.int64).reshape(n, 8).astype(np.float64)
t0 = tr[:, 1].min()
us = (tr[:, 1:] - t0) * 0.01
us = np.concatenate([us[:, :2], us[:, 6:7], us[:, 2:6]], axis=1)   # claimed, ready, done, stamps 3..6
print(f"{n} tasks, span {us[:, 2].max():.1f} us, workgroups {int(tr[:, 0].max()) + 1}")
names = ["update", "diag", "sub", "back"]
for k in range(4):
    m = tasks[:, 0] == k
    if not m.any(): continue
    wait, run = us[m, 1] - us[m, 0], us[m, 2] - us[m, 1]
    print(f"{names[k]:7s} n={m.sum():5d}  claimed->last late input: mean {wait.mean():8.1f} max {wait.max():8.1f}   after it: mean {run.mean():6.2f} p50 {np.median(run):6.2f} max {run.max():6.2f}   first start {us[m,0].min():8.1f} last end {us[m,2].max():8.1f}")
    st = us[m][:, 3:7] - us[m][:, 1:2]
    print("        stamps relative to the last late input (mean us):", np.round(np.nanmean(np.where(tr[m][:, 3:7] > 0, st, np.nan), axis=0), 2))
m = tasks[:, 0] == 0
left = tr[m, 6].astype(np.int64)
after = (us[m, 3] - us[m, 1])
for k in range(0, 8):
    q = left == k
    if q.any(): print(f"update tasks with {k} contributors left at the last late input: n={q.sum()}, accumulate end after it: mean {after[q].mean():.2f} us")
m = (tasks[:, 0] == 0) & (tr[:, 2] == tr[:, 1])
print('update tasks that never waited:', m.sum())
t5 = (tr[m, 5] - tr[m, 1]) * 0.01; t6 = (tr[m, 6] - tr[m, 1]) * 0.01; t3 = (tr[m, 3] - tr[m, 1]) * 0.01; t7 = (tr[m, 7] - tr[m, 1]) * 0.01
print("update tasks, since claim (median us): first group ready", np.median(t5), " last group's loads landed", np.median(t6), " accumulate end", np.median(t3), " done", np.median(t7))
# critical chain: walk the diag tasks in order of completion
d = np.where(tasks[:, 0] == 1)[0]
order = d[np.argsort(us[d, 2])]
ends = us[order, 2]
print("diag completions (us), every 10th:", np.round(ends[::max(1, len(ends) // 25)], 1))
bk = np.where(tasks[:, 0] == 3)[0]
print("back completions (us), every 10th:", np.round(np.sort(us[bk, 2])[::10], 1))
print("back: claimed->late input / late input->done for the last 12 finishing:", np.round(us[bk[np.argsort(us[bk, 2])][-12:], 1] - us[bk[np.argsort(us[bk, 2])][-12:], 0], 1), np.round(us[bk[np.argsort(us[bk, 2])][-12:], 2] - us[bk[np.argsort(us[bk, 2])][-12:], 1], 1))
np.save("/tmp/chol_trace.npy", np.concatenate([tasks, us], axis=1))
# the DIAG tasks in order of completion: gap to the previous completion and the task's own stamps relative to its completion
print("last 40 DIAG tasks by completion: done(us)  gap  | relative to done: claimed  late-input  accumulate-end(3)  W-in-registers(5)  factor-start(4)  factor-end(6) | workgroup")
for k in order[-40:]:
    prev = ends[np.searchsorted(ends, us[k, 2]) - 1] if us[k, 2] > ends[0] else 0.0
    rel = [us[k, 0] - us[k, 2], us[k, 1] - us[k, 2]] + [((tr[k, j] - tr[k, 1]) * 0.01 + us[k, 0] - us[k, 2]) if tr[k, j] > 0 else float('nan') for j in (3, 5, 4, 6)]
    print(f"  {us[k, 2]:8.1f} {us[k, 2] - prev:6.1f} | " + " ".join(f"{x:8.1f}" for x in rel) + f" | {int(tr[k, 0])}")
# the forward phase level by level: the DIAG tasks of a level sit next to each other in the task list
runs, start = [], None
for i in range(n + 1):
    isd = i < n and tasks[i, 0] == 1
    if isd and start is None: start = i
    if not isd and start is not None: runs.append((start, i)); start = None
print("level: DIAG tasks | first / last completion (us) | pace of the slowest chain (last completion - previous level's) | mean wait claimed->late input | mean late input->done")
prev = 0.0
for l, (a, b) in enumerate(runs):
    done = us[a:b, 2]
    sub = np.where((tasks[:, 0] == 2))[0]
    print(f"  {l:3d}: {b - a:3d} | {done.min():7.1f} {done.max():7.1f} | {done.max() - prev:6.1f} | {np.mean(us[a:b, 1] - us[a:b, 0]):6.1f} | {np.mean(us[a:b, 2] - us[a:b, 1]):6.1f}")
    prev = done.max()
# how busy the workgroups are: tasks running (claimed .. done) at sample times
ts = np.linspace(0, us[:, 2].max(), 40)
busy = [(int(((us[:, 0] <= t) & (us[:, 2] > t)).sum()), int(((us[:, 1] <= t) & (us[:, 2] > t)).sum())) for t in ts]
print("claimed / working (past their last late input) tasks at 40 sample times:", " ".join(f"{a}/{b}" for a, b in busy))
# the slowest DIAG task of each of the first levels, absolute times: where do the leaf levels lose their time?
print("slowest DIAG task of a level (absolute us): level | claimed  late-input  accumulate-end  W-in-registers  factor-start  factor-end  done | prev level's slowest done")
prev_done = 0.0
for li, (a, b) in enumerate(runs[:24]):
    idx = np.arange(a, b)
    k = idx[np.argmax(us[idx, 2])]
    ab = [us[k, 0], us[k, 1]] + [((tr[k, j] - tr[k, 1]) * 0.01 + us[k, 0]) if tr[k, j] > 0 else float('nan') for j in (3, 5, 4, 6)] + [us[k, 2]]
    print(f"  {li:3d} | " + " ".join(f"{x:8.1f}" for x in ab) + f" | {prev_done:8.1f}   (W-in-registers - prev done {ab[3] - prev_done:5.1f}; done - W-in-registers {ab[6] - ab[3]:5.1f})")
    prev_done = us[k, 2]
