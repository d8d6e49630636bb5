"""What the persistent Cholesky kernel reads coherently (from the memory side), by kind, per launch — the counters of the INSTRUMENTED library
(rsba_amd/_lib/librsba_amd_hooks.so: cholesky.hip, CHOL_COUNT), to set beside the kernel's fabric reads from a rocprofv3 --pmc pass
(TCC_EA0_RDREQ: profiles/r06/pmc_l2_*.csv).  usage: RSBA_AMD_LIB=rsba_amd/_lib/librsba_amd_hooks.so python tools/chol_poll_split.py [C4] [iters]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rsba_amd import capi
from rsba_amd.scene import make_config
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 8
assert capi.LIB_PATH.endswith("_hooks.so"), "needs the instrumented library: RSBA_AMD_LIB=rsba_amd/_lib/librsba_amd_hooks.so"
L = capi.lib()
prob = make_config(name).problem
with capi.DeviceProblem(prob) as dp:
    opt = capi.default_options(max_num_iterations=iters, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
    p0, x0 = prob.poses.copy(), prob.points.copy()
    dp.solve(opt)
    prob.poses[...] = p0; prob.points[...] = x0
    dp.upload_parameters()
    out = (C.c_ulonglong * 8)()
    assert L.rsba_debug_chol_coherent(out, 1) == 0
    s, _ = dp.solve(opt)
    assert L.rsba_debug_chol_coherent(out, 0) == 0
    st = dp.plan_stats()
n = max(1, s.num_iterations - 1)
looks, empty, frags, wrows, wretry = (out[k] / n for k in range(5))
fb = st["factor_tiles"] * 48 * 48 * 8
print(f"{name}: {n} factorisations; factor {fb / 1e6:.1f} MB ({st['factor_tiles']} tiles), {st['tasks']} tasks")
print(f"  looks at a watched cell          {looks:12.0f} per launch = {looks * 64 / 1e6:8.2f} MB  (of them {empty:.0f} found it empty: the polling proper, {empty * 64 / 1e6:.2f} MB)")
print(f"  operand fragments read again     {frags:12.0f} per launch = {frags * 9 * 512 / 1e6:8.2f} MB  (a wave's 9 x 512 B of a tile that came in incomplete through the caches)")
print(f"  rows of W read by SUB / DIAG     {wrows:12.0f} loads      = {wrows * 512 / 1e6:8.2f} MB  (of them {wretry * 512 / 1e6:.2f} MB in reads that came back incomplete and were repeated)")
print(f"  coherent reads in all            {(looks * 64 + frags * 9 * 512 + wrows * 512) / 1e6:8.2f} MB per launch = {(looks * 64 + frags * 9 * 512 + wrows * 512) / fb:.2f} x the factor")
