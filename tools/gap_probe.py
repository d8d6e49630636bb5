"""How much of an LM iteration's wall time is the device idle between kernels?  Run under the kernel tracer:
   rocprofv3 --kernel-trace --output-format csv -d OUT -o t -- python tools/gap_probe.py C2 24
   python tools/gap_probe.py --read OUT/t_kernel_trace.csv 24
The solve runs twice (the second is the one looked at); the reader takes the window between the first and the last chol_dag_kernel
of the last 24 iterations and reports the union of all kernels' busy intervals in it."""
import csv, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "--read":
    rows = list(csv.DictReader(open(sys.argv[2])))
    n = int(sys.argv[3])
    ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
    dag = [i for i, k in enumerate(ks) if "chol_dag_kernel" in k[2]]
    a, b = dag[-n], dag[-1]
    t0, t1 = ks[a][0], ks[b][0]          # n - 1 whole iteration periods
    busy, cur = 0, t0
    per = {}
    for s, e, name in ks[a:b]:
        s, e = max(s, cur), min(e, t1)
        if e > s: busy += e - s; cur = e
    for s, e, name in ks[a:b]:
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("::")[-1][:40]
        per[short] = per.get(short, 0) + (e - s)
    it = n - 1
    print(f"{it} iteration periods: {(t1 - t0) / it / 1e3:.1f} us each, device busy {busy / it / 1e3:.1f} us ({busy / (t1 - t0):.3f}), idle {(t1 - t0 - busy) / it / 1e3:.1f} us; kernels per iteration {(b - a) / it:.1f}")
    for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:14]:
        print(f"   {k:42s} {v / it / 1e3:8.1f} us / iteration")
    if len(sys.argv) > 4:   # the kernels of one iteration period in launch order: start (us since the period began), duration, gap to the previous end
        c = dag[-2]; prev_end = None
        first = dag[-3]
        for s, e, name in ks[first:c + 1]:
            short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("::")[-1][:48]
            print(f"   {(s - ks[first][0]) / 1e3:8.1f}  {(e - s) / 1e3:7.1f}  gap {((s - prev_end) / 1e3 if prev_end else 0):6.1f}  {short}")
            prev_end = max(prev_end or 0, e)
    sys.exit(0)

import time
from rsba_amd import capi
from rsba_amd.scene import make_config
name, iters = sys.argv[1], int(sys.argv[2])
prob = make_config(name).problem
p0, x0 = prob.poses.copy(), prob.points.copy()
with capi.DeviceProblem(prob) as dp:
    opt = capi.default_options(max_num_iterations=iters, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
    for rep in range(2):
        prob.poses[...] = p0; prob.points[...] = x0
        dp.upload_parameters()
        t0 = time.perf_counter(); summ, _ = dp.solve(opt); dt = time.perf_counter() - t0
    print(f"{name}: {summ.num_iterations} iterations, {1e3 * dt / max(1, summ.num_iterations):.3f} ms/iteration")
