"""Single-process cases that need a fault injected: run by tests/test_gpu_solve.py in a process of their own with RSBA_AMD_LIB pointing at the
INSTRUMENTED library (rsba_amd/_lib/librsba_amd_hooks.so, -DRSBA_TEST_HOOKS) — the release library does not read these switches.
   python tests/hook_worker.py <case> <out.json>"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def small_scene(frames=12, points=500):
    from rsba_amd.problem import apply_gauge_masks
    from rsba_amd.scene import make_scene
    p = make_scene(frames, points, seed=11).problem
    apply_gauge_masks(p, fix_first_n_cameras=1)
    p.pose_fixed_mask[-1, -1] |= 0b111000
    return p


def corrupt_dag(capi):
    from rsba_amd.scene import make_config
    out = {}
    for mode in ("corrupt", "levels", "plain"):
        os.environ.pop("RSBA_CHOL_TEST_CORRUPT", None)
        if mode == "corrupt":
            os.environ["RSBA_CHOL_TEST_CORRUPT"] = "1"
        p = make_config("C2").problem
        with capi.DeviceProblem(p) as dp:
            s, _ = dp.solve(capi.default_options(max_num_iterations=6, level_scheduled_cholesky=int(mode == "levels")))
        out[mode] = dict(final_cost=s.final_cost, iters=s.num_iterations, fallbacks=s.num_dag_fallbacks, poses=p.poses.tobytes().hex(), points=p.points.tobytes().hex())
    return out


def failed_plan(capi, hook):
    p = small_scene()
    q = p.copy()
    with capi.DeviceProblem(q) as dp:
        s_ref, _ = dp.solve(capi.default_options(max_num_iterations=5))
    errors = []
    with capi.DeviceProblem(p) as dp:
        os.environ[hook] = "1"
        for call in (lambda: dp.solve(capi.default_options(max_num_iterations=5)),) * 2 + (dp.plan_stats,):
            try:
                call(); errors.append(None)
            except capi.RsbaError as e:
                errors.append(str(e))
        os.environ.pop(hook)
        s, _ = dp.solve(capi.default_options(max_num_iterations=5))
    return dict(errors=errors, same=bool(s.final_cost == s_ref.final_cost and s.num_iterations == s_ref.num_iterations and np.array_equal(p.poses, q.poses) and np.array_equal(p.points, q.points)))


def main():
    case, out_path = sys.argv[1], sys.argv[2]
    from rsba_amd import capi
    if case == "corrupt_dag":
        out = corrupt_dag(capi)
    elif case.startswith("failed_plan:"):
        out = failed_plan(capi, case.split(":")[1])
    else:
        raise SystemExit(f"unknown case {case}")
    out["library"] = capi.LIB_PATH
    with open(out_path, "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
