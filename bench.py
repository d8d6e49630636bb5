#!/usr/bin/env python3
"""Headline benchmark: residual+Jacobian evaluations per second and LM-iteration wall time on the 1k-camera /
100k-point rolling-shutter scene (BASELINE.json metric, config C4; --config C5 = the 4k-camera Huber + shared
intrinsics scene), one process per GPU.

A step = one pass of the hot path over the scene's observations: the fused residual + analytic Jacobian kernel
(results materialised in HBM, as ceres' CostFunction::Evaluate materialises them), inputs resident in HBM before the
timed region.  With N > 1 the SAME scene is sharded by point (BAProblem.shard: cameras replicated, every observation
of a point on one rank): strong scaling, `value` = the scene's observations / the slowest rank's time.  The evaluation
needs no collective; the LM iteration (the metric's second half, reported under "lm") all-reduces the per-camera
blocks and the packed reduced camera system over RCCL, called by the library itself on its stream
(rsba_set_exchange_rccl).  "roofline" prices the evaluation kernel against HBM, "roofline_lm" every phase of an LM
iteration (HIP events recorded by the solver, rsba_get_phase_times).  At N = 1 the CPU oracle is timed beside it
("cpu_baseline": evaluation at 1 and at the best of all / half / quarter of the host's threads, and its LM iteration) and
the rows next to the path add bounded wall-clock figures under "next_rows".

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_F64_PEAK_TFLOPS = 78.6  # v_mfma_f64_16x16x4_f64: 64 cycles per 2048 flop (tools/mfma_f64_check.hip) x 4 SIMDs x 256 CUs x 2.4 GHz


def provenance(summary: dict, path: str) -> dict:
    """Which code a committed PMC summary was measured on, against the code that is running: the summaries carry the stamp of the
    device sources (tools/source_stamp.py) and, when they were copied into profiles/, the commit; a summary without a stamp, or
    with another one, is marked stale — the figure is then last round's, not this build's."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from source_stamp import source_sha16
    now = source_sha16()
    then = summary.get("source_sha16")
    return {"file": path, "measured_on_commit": summary.get("commit"), "measured_on_source_sha16": then, "running_source_sha16": now, "stale": then != now}


def algorithmic_bytes(prob) -> float:
    """N*(16 obs + 8 idx + 16 r + 16 K) + F*P*48 + M*24 + 72 (M = the points the observations refer to)."""
    import numpy as np
    n, k = prob.num_observations, prob.jacobian_cols
    m = len(np.unique(prob.obs_point)) if n else 0
    return n * (16 + 8 + 16 + 16 * k) + prob.num_frames * prob.poses_per_frame * 48 + m * 24 + 72


def host_cpu():
    """CPU model / sockets / physical cores / threads of the box, from lscpu (north_star: core count stated)."""
    info = {"threads": os.cpu_count()}
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        kv = {l.split(":", 1)[0].strip(): l.split(":", 1)[1].strip() for l in txt.splitlines() if ":" in l}
        info["model"] = kv.get("Model name")
        info["sockets"] = int(kv.get("Socket(s)", "0") or 0)
        info["physical_cores"] = info["sockets"] * int(kv.get("Core(s) per socket", "0") or 0)
    except Exception as e:  # noqa: BLE001
        info["error"] = repr(e)
    return info


def cpu_baseline(prob, budget_s: float = 8.0, lm_iters: int = 3):
    """Oracle timed the way Ceres runs rsba's functors (checker, never the product path).  Ceres would use
    hardware_concurrency() threads (CeresHandler.h:408-415); on a 2-socket host fewer threads can be
    faster for this memory-light loop, so a short calibration picks the best of {all, half, quarter}.  Also one thread,
    and the oracle's LM iteration (Schur complement in envelope storage + Cholesky) on the same scene."""
    from oracle import oracle as O
    build = os.path.basename(O.select_build("host"))   # the AVX-512 build of the same source where the host has it
    ncpu = os.cpu_count() or 1
    best = None
    for threads in sorted({ncpu, max(1, ncpu // 2), max(1, ncpu // 4)}, reverse=True):
        ev = O.CeresStyleEvaluator(prob, threads)
        ev.run()
        t0 = time.perf_counter(); ev.run(); dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, threads)
    first, threads = best
    ev = O.CeresStyleEvaluator(prob, threads)
    ev.run()
    reps = max(1, min(200, int(budget_s / max(first, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(reps):
        ev.run()
    dt = (time.perf_counter() - t0) / reps
    ev1 = O.CeresStyleEvaluator(prob, 1)
    ev1.run()
    t0 = time.perf_counter(); ev1.run(); dt1 = time.perf_counter() - t0
    out = {"value": prob.num_observations / dt, "unit": "obs evals/s", "cores": threads, "kind": "port",
           "sample": f"{reps} x full residual+Jacobian evaluation of the {prob.num_observations}-observation scene, "
                     f"Dual<{prob.jacobian_cols}> autodiff, one cost object per observation, OpenMP {threads} threads",
           "ms_per_eval": dt * 1e3, "threads_1": {"value": prob.num_observations / dt1, "ms_per_eval": dt1 * 1e3},
           "host": host_cpu(), "build": build + (" (-march=x86-64-v4, AVX-512)" if "v4" in build else " (-march=x86-64-v3, AVX2 + FMA)")}
    try:   # the LM iteration of the same CPU path, bounded: lm_iters iterations of the whole scene
        O.lib().orc_set_num_threads(min(ncpu, 64))
        q = prob.copy()
        t0 = time.perf_counter()
        s, tr = O.solve(q, O.default_options(max_num_iterations=lm_iters, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0))
        wall = time.perf_counter() - t0
        out["lm"] = {"ms_per_lm_iteration": wall / max(1, s.num_iterations - 1) * 1e3, "iterations": s.num_iterations - 1, "threads": min(ncpu, 64),
                     "initial_cost": s.initial_cost, "cost_after": [t.cost for t in tr],
                     "sample": f"{lm_iters} LM iterations of the same scene: Dual<{prob.jacobian_cols}> Jacobian, Schur complement (envelope storage), "
                               "Cholesky, back-substitution — the Ceres-1.9 rules restated (oracle/rsba_oracle.cpp)"}
    except Exception as e:  # noqa: BLE001
        out["lm"] = {"error": repr(e)}
    return out


def roofline_lm(prob, dp, iters, capi):
    """Per-phase device time of the LM iteration (HIP events on the solver's stream) against the bound of each phase.
    Bytes / flops are ALGORITHMIC: what the phase must move or compute, not what it happened to."""
    saved = (prob.poses.copy(), prob.points.copy(), prob.intrinsics.copy())
    s, _ = dp.solve(capi.default_options(max_num_iterations=iters, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0, profile_phases=1))
    prob.poses[:], prob.points[:], prob.intrinsics[:] = saved
    dp.upload_parameters()
    times, st = dp.phase_times(), dp.plan_stats()
    mfma_per_launch = st["schur_mfma_issued"] / max(1, st["schur_launches"])   # measured by the kernel itself: all-zero 16 x 16 operand blocks are not issued
    n, k, m = prob.num_observations, prob.jacobian_cols, prob.num_points
    cd = 6 * prob.poses_per_frame
    rec = 8 * (2 + 2 * k)                       # the point-major record of one observation
    prec = 8 * cd * 3                           # its P record
    pgroups = st.get("schur_group_bytes") or st["schur_groups"] * 3 * 48 * 8   # the P records as the Schur kernel reads them: one block per (point, frame tile) — [3][48] doubles, or the 640-B factored form of a two-pose frame tile
    recompute = prob.calibrated or prob.num_intrinsics == 1
    if recompute:   # the point-side passes recompute the records from the observations (24 B each, slot order) instead of streaming 256-B copies
        obs = n * 24 + prob.num_frames * prob.poses_per_frame * 96 + m * 48
        work_point = {"eval_lm": ("hbm", obs + n * 32), "point_blocks": ("hbm", obs + m * 72), "project": ("hbm", obs + n * 4 + m * 48 + pgroups),
                      "back_substitute": ("hbm", obs + m * (48 + 24 + 72 + 24))}
    else:
        work_point = {"eval_lm": ("hbm", n * (24 + rec + 32) + prob.num_frames * prob.poses_per_frame * 48 + m * 24), "point_blocks": ("hbm", n * 64 + m * 72),
                      "project": ("hbm", n * rec + pgroups), "back_substitute": ("hbm", n * rec + m * 48)}
    work = {   # phase -> (bound, algorithmic bytes or flops per call)
        **work_point,
        "schur": ("mfma", mfma_per_launch * 2048),
        "cholesky": ("mfma", st["cholesky_flops"]),
        "eval_trial": ("hbm", n * 24 + prob.num_frames * prob.poses_per_frame * 48 + m * 24),
    }
    rows = []
    for name, (ms, calls) in times.items():
        if calls == 0:
            continue
        row = {"phase": name, "ms_per_call": ms / calls, "calls": calls, "ms_per_lm_iteration": ms / max(1, s.num_iterations - 1)}
        if name in work:
            bound, amount = work[name]
            per_s = amount / (ms / calls * 1e-3)
            if bound == "hbm":
                row.update(bound="hbm", algorithmic_bytes=amount, achieved=per_s / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", frac=per_s / 1e9 / HBM_PEAK_GBS)
            else:
                row.update(bound="mfma", algorithmic_flops=amount, achieved=per_s / 1e12, peak=MFMA_F64_PEAK_TFLOPS, unit="TFLOP/s",
                           frac=per_s / 1e12 / MFMA_F64_PEAK_TFLOPS)
        rows.append(row)
    notes = {"schur": f"issued fp64 MFMA flops, counted by the kernel: {mfma_per_launch / max(1, st['schur_entries']):.2f} MFMAs per entry (6.75 without the zero-block skip); "
                      f"structurally non-zero block products: {st['schur_block_products']} x {2 * cd * cd * 3} flop = {st['schur_block_products'] * 2 * cd * cd * 3 / 1e9:.2f} Gflop useful",
             "cholesky": f"latency-bound dependency chain: {st['levels']} elimination levels, {st['tasks']} tile tasks, {st['factor_tiles']} factor tiles",
             "eval_trial": "residual only: bound by the fp64 projection math, not by HBM"}
    if recompute:
        for ph in ("eval_lm", "point_blocks", "project", "back_substitute"):
            notes[ph] = "records recomputed from the observations (rsba_amd/csrc/lm_record.hpp): ~0.6 kflop of fp64 per observation instead of a 256-B record from HBM — bound by the fp64 vector unit; bytes are what the pass still has to move"
        notes["project"] += "; writes the P records in the (point, tile) group layout"

    for r in rows:
        if r["phase"] in notes:
            r["note"] = notes[r["phase"]]
    # matrix-pipe utilisation by counter (north_star: "MFMA utilisation against gfx950 peak"): SQ_VALU_MFMA_BUSY_CYCLES of the phase's
    # kernel over the SIMD cycles of its dispatch, from the committed PMC pass (a separate rocprofv3 --pmc run; tools/profile_round.sh)
    cfg = "C5" if prob.num_frames >= 4000 else ("C4" if prob.num_frames >= 1000 else "C2")
    for pr in sorted((p for p in os.listdir(os.path.join(ROOT, "profiles")) if os.path.exists(os.path.join(ROOT, "profiles", p, "pmc_mfma_summary.json"))), reverse=True):
        with open(os.path.join(ROOT, "profiles", pr, "pmc_mfma_summary.json")) as fh:
            whole = json.load(fh)
        pm = whole.get(cfg, {})
        prov = provenance(whole, f"profiles/{pr}/pmc_mfma_summary.json")
        for r in rows:
            k = {"schur": "schur_tile_kernel", "cholesky": "chol_dag_kernel", "eval_lm": "eval_kernel"}.get(r["phase"])
            hit = next((v for name, v in pm.items() if k and k in name and (r["phase"] != "eval_lm" or ", 2>" in name)), None)
            if hit:
                r["mfma_busy_frac"] = hit["mfma_busy_frac"]
                r["mfma_busy_source"] = f"profiles/{pr}/pmc_mfma_summary.json (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs))"
                r["mfma_busy_provenance"] = prov
        break
    # the passes that recompute the observation model are bound by the fp64 VECTOR unit, not by HBM: price them with the fp64 operations the
    # counters saw (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 per dispatch, a separate rocprofv3 --pmc pass: tools/profile_round.sh,
    # tools/pmc_valu_summary.py) against the vector unit's fp64 peak (= the MFMA peak: 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz)
    valu_kernels = {"eval_trial": ("eval_kernel", ", 0>"), "eval_lm": ("eval_kernel", ", 2>"), "point_blocks": ("point_blocks_rc_kernel", ""), "project": ("project_rc_kernel", ""),
                    "back_substitute": ("point_step_rc_kernel", "")}
    for pr in sorted((p for p in os.listdir(os.path.join(ROOT, "profiles")) if os.path.exists(os.path.join(ROOT, "profiles", p, "pmc_valu_summary.json"))), reverse=True):
        with open(os.path.join(ROOT, "profiles", pr, "pmc_valu_summary.json")) as fh:
            whole = json.load(fh)
        pv = whole.get(cfg, {})
        prov = provenance(whole, f"profiles/{pr}/pmc_valu_summary.json")
        for r in rows:
            if r["phase"] not in valu_kernels or not recompute:
                continue
            sub, tail = valu_kernels[r["phase"]]
            hits = [v for name, v in pv.items() if sub in name and (not tail or tail in name)]
            if not hits and r["phase"] == "eval_trial":   # (candidates are evaluated in LM mode straight away: the phase's kernel is the LM-mode one)
                hits = [v for name, v in pv.items() if sub in name and ", 2>" in name]
            hit = max(hits, key=lambda v: v.get("f64_flops", 0.0), default=None)
            if not hit or not hit.get("f64_flops"):
                continue
            flops = hit["f64_flops"]
            per_s = flops / (r["ms_per_call"] * 1e-3)
            r.update(hbm_frac=r.get("frac"), hbm_algorithmic_bytes=r.get("algorithmic_bytes"), bound="valu_f64", algorithmic_flops=flops, flops_per_observation=flops / max(1, n),
                     achieved=per_s / 1e12, peak=MFMA_F64_PEAK_TFLOPS, unit="TFLOP/s", frac=per_s / 1e12 / MFMA_F64_PEAK_TFLOPS, valu_busy_frac=hit["valu_busy_frac"],
                     valu_source=f"profiles/{pr}/pmc_valu_summary.json (fp64 wave instructions x 64 lanes, FMA = 2 flop: counted, exec mask not known to the counter; valu_busy_frac = 4 x SQ_ACTIVE_INST_VALU / SIMD cycles)",
                     valu_provenance=prov)
            r.pop("algorithmic_bytes", None)
        break
    return {"phases": rows, "plan": st, "sum_ms_per_lm_iteration": sum(r["ms_per_lm_iteration"] for r in rows),
            "note": "phases are timed with HIP events around each step of the loop's HOST form (rsba_solve with profile_phases: the host decides, a pair of "
                    "events per step, ~10 us each); lm.ms_per_lm_iteration is the unprofiled solve — on one GPU the device-side loop with fewer launches — "
                    "and is therefore below this sum"}


def shard_eval(full, world: int, device, capi, steps: int = 20):
    """The metric's kernel on ONE rank's shard of an N-rank run, each shard alone on this GPU (rsba_partition_points, BAProblem.shard —
    what bench.py --gpus N gives rank r).  Two figures per shard: the HIP-event kernel time of back-to-back launches, and the STEP CADENCE
    — the wall clock around `steps` rsba_evaluate_device calls + a device sync, i.e. exactly the timed region of this script on that
    rank (launch path, dependent-kernel boundary and the closing sync included; median of 15 repetitions).  The prediction for N GPUs is
    quoted from the cadence of the slowest shard: the evaluation has no collective, so that is what the driver's N-GPU line measures,
    stragglers and the other ranks' hosts aside."""
    import statistics
    import torch
    owner, _ = capi.partition_points(full, world)
    rows = []
    for r in range(world):
        sh = full.shard(r, world, owner)
        with capi.DeviceProblem(sh, device=device) as d:
            ms = d.time_evaluate(True, warmup=100, iters=200)
            walls = []
            for _rep in range(15):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    d.evaluate_device(True)
                torch.cuda.synchronize()
                walls.append((time.perf_counter() - t0) / steps * 1e3)
        ab = algorithmic_bytes(sh)
        rows.append({"rank": r, "observations": int(sh.num_observations), "kernel_ms": ms, "step_ms_wall": statistics.median(walls), "step_ms_wall_min": min(walls), "step_ms_wall_max": max(walls),
                     "algorithmic_bytes": ab, "frac_of_hbm_peak": ab / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS})
    worst = max(x["kernel_ms"] for x in rows)
    worst_step = max(x["step_ms_wall"] for x in rows)
    return {"world": world, "steps_per_repetition": steps, "shards": rows, "slowest_shard_kernel_ms": worst, "slowest_shard_step_ms_wall": worst_step,
            "predicted_value": full.num_observations / (worst_step * 1e-3), "predicted_value_from_kernel_time": full.num_observations / (worst * 1e-3), "unit": "obs evals/s",
            "note": "predicted_value = scene observations / the slowest shard's STEP CADENCE (wall clock around K back-to-back rsba_evaluate_device calls + device sync on this GPU: "
                    "what each rank's timed region is made of); predicted_value_from_kernel_time uses the HIP-event kernel time alone and is the upper bound"}


def next_rows(prob, dp, device):
    """Wall-clock figures of the rows next to the hot path, on rank 0 at N = 1 only: bounded (about a second
    in total) and never fatal — they ride in the JSON line under "next_rows", they are not the metric."""
    import numpy as np
    from rsba_amd import capi
    from rsba_amd.problem import BAProblem
    out = {}

    def timed(fn, reps=3):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = fn()
        return (time.perf_counter() - t0) / reps, r

    try:   # BASELINE configs 2-3 (100 frames, 10k points): evaluation kernel, LM iteration, and the whole BA() call of a fresh handle
        from rsba_amd.scene import make_config
        c2 = make_config("C2").problem
        p0, x0 = c2.poses.copy(), c2.points.copy()
        opt20 = capi.default_options(max_num_iterations=20)     # what VideoSfMHandler::BA asks for (VideoSfMHandler.cc:582)
        # a fresh handle per call is windowedBA's pattern; the symbolic phase is host work on a shared, busy host (one sample in five
        # is off by milliseconds), so: three fresh handles, the fastest is quoted, all are listed
        runs = []
        for _rep in range(3):
            c2.poses[:], c2.points[:] = p0, x0
            t0 = time.perf_counter()
            d2 = capi.DeviceProblem(c2, device=device)
            t_create = time.perf_counter() - t0
            t0 = time.perf_counter()
            s_first, _ = d2.solve(opt20)
            t_first = time.perf_counter() - t0
            t0 = time.perf_counter()
            if _rep < 2:
                d2.close()
            t_destroy = time.perf_counter() - t0
            runs.append((t_create + t_first, t_create, t_first, t_destroy))
        t_total, t_create, t_first, _ = min(runs)
        eval_ms = d2.time_evaluate(True, warmup=20, iters=50)
        c2.poses[:], c2.points[:] = p0, x0
        d2.upload_parameters()
        fixed = capi.default_options(max_num_iterations=12, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0)
        d2.solve(fixed)
        c2.poses[:], c2.points[:] = p0, x0
        d2.upload_parameters()
        t0 = time.perf_counter()
        s2, _ = d2.solve(fixed)
        t_steady = time.perf_counter() - t0
        d2.close()
        out["c2"] = {"observations": int(c2.num_observations), "eval_kernel_ms": eval_ms, "obs_evals_per_s": c2.num_observations / (eval_ms * 1e-3),
                     "ms_per_lm_iteration": t_steady / max(1, s2.num_iterations - 1) * 1e3,
                     "end_to_end_ba": {"rsba_create_ms": t_create * 1e3, "first_solve_ms": t_first * 1e3, "iterations": int(s_first.num_iterations - 1),
                                       "total_ms": (t_create + t_first) * 1e3, "final_cost": s_first.final_cost,
                                       "all_runs_ms": [{"create": r[1] * 1e3, "plan_and_solve": r[2] * 1e3, "destroy": r[3] * 1e3} for r in runs],
                                       "note": "fresh handle: upload + index check, symbolic phase, allocations, up to 20 LM iterations with Ceres' default tolerances, parameters back on the host"}}
    except Exception as e:  # noqa: BLE001
        out["c2"] = {"error": repr(e)}
    try:   # f2: validate every observation of the workload (kernel + the [N] flag copy back)
        dt, flags = timed(lambda: dp.validate_observations(16.0, 0.0))
        out["f2_validate"] = {"ms_per_call": dt * 1e3, "observations": int(prob.num_observations), "valid": int(flags.sum())}
    except Exception as e:  # noqa: BLE001
        out["f2_validate"] = {"error": repr(e)}
    try:   # f1: LM iterations with a constant-velocity motion prior on every frame
        q = prob.copy()
        q.prior_kind, q.prior_scale, q.inter_frame_ratio = 1, 10.0, 0.8
        q.prior_frames = np.arange(1, q.num_frames, dtype=np.int32)
        with capi.DeviceProblem(q, device=device) as dq:
            opt = capi.default_options(max_num_iterations=8, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0)
            p0, x0 = q.poses.copy(), q.points.copy()
            dq.solve(opt)
            q.poses[:], q.points[:] = p0, x0
            dq.upload_parameters()
            s, _ = dq.solve(opt)
        out["f1_lm_with_motion_priors"] = {"ms_per_lm_iteration": s.total_time_s / max(1, s.num_iterations - 1) * 1e3, "prior_blocks": int(q.num_frames - 1),
                                           "initial_cost": s.initial_cost, "final_cost": s.final_cost}
        q.poses[:], q.points[:] = p0, x0
        q.inter_frame_ratio, q.ratio_free = 1.0, True            # the reference's default: the ratio is solved for
        with capi.DeviceProblem(q, device=device) as dq:
            s1, _ = dq.solve(opt)
            first_ratio = q.inter_frame_ratio
            q.poses[:], q.points[:] = p0, x0
            dq.upload_parameters()
            s, _ = dq.solve(opt)                                  # steady state (the ratio continues from its solved value)
        out["f1_lm_with_free_inter_frame_ratio"] = {"ms_per_lm_iteration": s.total_time_s / max(1, s.num_iterations - 1) * 1e3, "solved_ratio": first_ratio,
                                                    "final_cost": s1.final_cost}
    except Exception as e:  # noqa: BLE001
        out["f1_lm_with_motion_priors"] = {"error": repr(e)}
    try:   # f3: RS-PnP RANSAC hypotheses; the scene's observations come from the device's own reproject (f2)
        rng = np.random.default_rng(3)
        n, H, m = 1000, 16384, 6
        cam = prob.intrinsics[0]
        pose0 = np.concatenate([rng.normal(0, 0.05, 3), rng.normal(0, 0.3, 3)])
        poses = np.stack([pose0, pose0 + np.concatenate([rng.normal(0, 0.01, 3), [0.35, 0.05, -0.04]])])
        X = np.stack([rng.uniform(-4, 4, n), rng.uniform(-2.5, 2.5, n), rng.uniform(7, 14, n)], axis=1).astype(np.float32)
        one = BAProblem(poses=poses[None], points=X.astype(np.float64), intrinsics=cam[None], obs_xy=np.zeros((1, 2)), obs_frame=np.zeros(1), obs_point=np.zeros(1),
                        shutter=1, scanlines=(0, 1280))
        with capi.DeviceProblem(one, device=device) as d1:
            xy, ok = d1.reproject(np.zeros(n, dtype=np.int32), np.arange(n, dtype=np.int32))
        X, xy = X[ok], (xy[ok] + rng.normal(0, 0.4, (int(ok.sum()), 2))).astype(np.float32)
        bad = rng.random(len(xy)) < 0.25
        xy[bad] += rng.normal(0, 40.0, (int(bad.sum()), 2)).astype(np.float32)
        subs = np.stack([rng.choice(len(X), m, replace=False) for _ in range(H)]).astype(np.int32)
        init = poses + np.concatenate([rng.normal(0, 0.01, (2, 3)), rng.normal(0, 0.08, (2, 3))], axis=1)
        dt, r = timed(lambda: capi.pnp_tasks(cam, 1, (0, 1280), X, xy, subs, init, 10, 3.0, device=device))
        out["f3_pnp_hypotheses"] = {"ms_per_call": dt * 1e3, "hypotheses": H, "subset": m, "points_scored": int(len(X)), "hypotheses_per_s": H / dt,
                                    "best_inliers": int(r["num_inliers"].max()), "true_inliers": int((~bad).sum()),
                                    "roofline": {"bound": "latency", "waves_per_cu": H / 64 / 256,
                                                 "note": "a hypothesis per lane: 16 384 hypotheses are ONE wave per CU — one SIMD of four, running ten dependent LM iterations of "
                                                         "fp64 code each; neither HBM nor the matrix pipe is near a limit (0.31 ms kernel by rocprofv3, DESIGN.md §6); more hypotheses per call fill the idle SIMDs"}}
    except Exception as e:  # noqa: BLE001
        out["f3_pnp_hypotheses"] = {"error": repr(e)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C4", choices=["C2", "C4", "C5"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-lm", action="store_true")
    ap.add_argument("--lm-iters", type=int, default=12)
    ap.add_argument("--no-next-rows", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # test hook (one-GPU boxes): RSBA_BENCH_TEST_ONE_GPU=1 runs every rank on device 0 over gloo with the callback exchange
    # (RCCL refuses two ranks on one device), which exercises the whole multi-rank flow of this script without several GPUs
    one_gpu = os.environ.get("RSBA_BENCH_TEST_ONE_GPU") == "1"
    # ... and RSBA_BENCH_NATIVE=1 with RSBA_RCCL_LIB=tools/libmock_rccl.so (a stream-ordered stand-in for librccl, test infrastructure) sends the
    # ranks through the branch the driver's node takes — attach_rccl -> rsba_rccl_comm_create -> rsba_set_exchange_rccl, the warm handle on
    # the same communicator, the LM leg under its watchdog — instead of the gloo callback
    native_on_one_gpu = one_gpu and os.environ.get("RSBA_BENCH_NATIVE") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from rsba_amd import capi
    from rsba_amd.scene import SEED, make_config

    full = make_config(args.config, seed=SEED).problem       # the same scene on every rank
    # this rank's observations: by point, cameras replicated — the points cut along the top separators of the reduced system's
    # elimination tree (rsba_partition_points: every rank then factors its own part of S, only the separators' tiles travel);
    # RSBA_BENCH_PARTITION=modulo keeps the round-robin partition of the replicated factorisation for A/B runs
    partition = "single GPU"
    if world > 1:
        owner = None
        if os.environ.get("RSBA_BENCH_PARTITION", "separators") != "modulo":
            try:
                owner, ntop = capi.partition_points(full, world)
                partition = f"by point along the top separators of the elimination tree ({ntop} shared tile columns), cameras replicated"
            except capi.RsbaError as e:
                partition = f"by point (round robin: {e}), cameras replicated"
        else:
            partition = "by point (round robin), cameras replicated"
        prob = full.shard(rank, world, owner)
    else:
        prob = full
    dp = capi.DeviceProblem(prob, device=local_rank)
    comm = None
    serialize = False
    transport = "none (single GPU)"
    transport_error = None
    if world > 1 and not args.no_lm:
        from rsba_amd.distributed import attach, attach_rccl
        if one_gpu and not native_on_one_gpu:
            serialize = os.environ.get("RSBA_BENCH_SERIALIZE", "1") == "1"   # ranks sharing the GPU take turns: per-rank device times as on a node
            attach(dp, serialize=serialize)
            transport = "callback: torch.distributed gloo staged through the host (test hook" + ("; ranks take turns on the shared GPU)" if serialize else ")")
        else:
            # the library's own communicator beside torch's: if ANY rank cannot set it up, every rank leaves the LM leg out (agreed through
            # torch's process group) and says why — the evaluation, which needs no collective, is still timed and reported
            try:
                comm = attach_rccl(dp, local_rank); transport = "native: ncclAllReduce (RCCL over xGMI) issued by librsba_amd on its stream"
                if native_on_one_gpu:
                    transport = "native: ncclAllReduce issued by librsba_amd on its stream, served by " + os.path.basename(os.environ.get("RSBA_RCCL_LIB", "?")) + " (test hook: all ranks on one GPU)"
            except Exception as e:  # noqa: BLE001 - reported in the JSON line
                transport_error = repr(e)
            ok = torch.tensor([0.0 if transport_error else 1.0], device="cpu" if one_gpu else "cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok.item()) == 0.0:
                transport_error = transport_error or "another rank could not set up the library's RCCL communicator"
                args.no_lm = True

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed: let the GPU clocks settle (a cold chip runs the first few milliseconds ~5 % slower), then
    # the W warm-up steps the contract asks for
    for _ in range(200):
        dp.evaluate_device(True)
    for _ in range(args.warmup):
        dp.evaluate_device(True)
    # The timed region: every rank starts behind a barrier + device sync and stops at ITS OWN device sync; the region's length is the
    # MAX over the ranks (the all-reduce below).  A closing collective barrier inside the region would add 50 - 150 us of rendezvous to
    # what is 20 x 15 us of kernels at N = 8 and buys nothing: the maximum is taken afterwards anyway.  The evaluation has no
    # collective on its data path, so a rank's K steps end when its own stream is idle.
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dp.evaluate_device(True)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    cdev = "cpu" if one_gpu else "cuda"
    n_obs = torch.tensor([float(prob.num_observations)], device=cdev, dtype=torch.float64)
    t_max = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(n_obs, op=dist.ReduceOp.SUM)
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    elapsed = float(t_max.item())
    total_obs = float(n_obs.item())
    assert int(total_obs) == full.num_observations, (total_obs, full.num_observations)

    # dominant kernel, HIP events on the stream it is launched on (rank 0's shard)
    kernel_ms = dp.time_evaluate(True, warmup=50, iters=max(50, args.steps))
    abytes = algorithmic_bytes(prob)
    achieved = abytes / (kernel_ms * 1e-3) / 1e9
    # HBM traffic of the same kernel on the same workload from the committed PMC passes (separate
    # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this command; tools/profile_round.sh)
    traffic, traffic_src, traffic_prov = None, None, None
    if world == 1:
        key = "hbm_bytes_per_launch" if args.config == "C4" else f"hbm_bytes_per_launch_{args.config}"
        prof = sorted(p for p in os.listdir(os.path.join(ROOT, "profiles")) if os.path.exists(os.path.join(ROOT, "profiles", p, "pmc_summary.json")))
        for pr in reversed(prof):
            with open(os.path.join(ROOT, "profiles", pr, "pmc_summary.json")) as fh:
                pm = json.load(fh)
            if pm.get(key):
                traffic, traffic_src = pm[key], f"profiles/{pr}/pmc_summary.json (2*FETCH_SIZE + WRITE_SIZE, calibrated)"
                traffic_prov = provenance(pm, f"profiles/{pr}/pmc_summary.json")
                break

    lm, lm_roof = None, None
    lm_hung = False
    if not args.no_lm:
        # The LM solve is the one part of this script with a collective on the data path (three all-reduces per
        # iteration).  It runs under a watchdog so that a wedged exchange can never take the bench line with it.
        import threading
        box = {}

        def run_lm():
            try:
                torch.cuda.set_device(local_rank)          # the current device is per thread
                from rsba_amd.distributed import solve_timed
                # first_solve_wall_s is what a fresh HANDLE costs — symbolic phase + allocations + the solve — in a process that has
                # solved before (windowedBA builds a handle per frame, VideoSfMHandler.cc:185-214): another handle on the same scene
                # takes the process's own first-time costs (kernel images, first streams, the plan's host scratch being mapped)
                t_cold = time.perf_counter()
                with capi.DeviceProblem(prob.copy(), device=local_rank) as warm:
                    if world > 1:   # (the same exchange as the timed handle; every rank does this, so the collectives pair up)
                        if comm is not None:
                            warm.set_exchange_rccl(comm, rank, world); warm.sync_block_structure()
                        else:
                            from rsba_amd.distributed import attach
                            attach(warm, serialize=serialize)
                    warm.solve(capi.default_options(max_num_iterations=1))
                    t_cold = time.perf_counter() - t_cold
                box["lm"] = solve_timed(dp, prob, world, args.lm_iters)
                box["lm"]["first_handle_of_the_process_s"] = t_cold     # create + symbolic phase + ONE iteration, incl. kernel images, first streams, the plan's host scratch
                box["lm"]["first_solve_wall_s_note"] = "a fresh handle in a process that has solved before (what windowedBA pays per call): symbolic phase + allocations + the solve"
                if world == 1:
                    # the reference's DEFAULT gauge (SfmOptions.h:66-70: nothing fixed; CeresHandler.h:342-382 marks no block constant): the
                    # reduced system is rank deficient up to the LM damping — same scene, same iterations, no coordinate held
                    try:
                        q = prob.copy(); q.pose_fixed_mask = None
                        x0 = (q.poses.copy(), q.points.copy(), q.intrinsics.copy())
                        opt = capi.default_options(max_num_iterations=args.lm_iters, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0)
                        with capi.DeviceProblem(q, device=local_rank) as dq:
                            dq.solve(opt)
                            q.poses[:], q.points[:], q.intrinsics[:] = x0
                            dq.upload_parameters()
                            sf, _ = dq.solve(opt)
                        box["lm"]["lm_free_gauge"] = {"ms_per_lm_iteration": sf.total_time_s / max(1, sf.num_iterations - 1) * 1e3, "iterations": int(sf.num_iterations - 1),
                                                      "initial_cost": sf.initial_cost, "final_cost": sf.final_cost, "num_parameters_reduced": int(sf.num_parameters_reduced),
                                                      "dag_fallbacks": int(sf.num_dag_fallbacks), "successful_steps": int(sf.num_successful_steps)}
                    except Exception as e:  # noqa: BLE001
                        box["lm"]["lm_free_gauge"] = {"error": repr(e)}
                before = dp.exchange_stats() if world > 1 else None
                box["roof"] = roofline_lm(prob, dp, args.lm_iters, capi)
                if world > 1:   # calls / bytes of THAT solve alone (the library counts since the handle was created); its ms are the solve's own
                    after = dp.exchange_stats()
                    for k, v in after["collectives"].items():
                        v["calls"] -= before["collectives"][k]["calls"]; v["bytes"] -= before["collectives"][k]["bytes"]
                    box["collectives"] = after
            except Exception as e:  # noqa: BLE001 - reported in the JSON line
                box.setdefault("lm", {"error": repr(e)})
                box.setdefault("roof", {"error": repr(e)})

        th = threading.Thread(target=run_lm, daemon=True)
        th.start()
        th.join(timeout=float(os.environ.get("RSBA_BENCH_LM_TIMEOUT_S", "180")))
        lm_hung = th.is_alive()
        lm = {"error": "LM solve did not finish inside the watchdog window"} if lm_hung else box.get("lm")
        lm_roof = None if lm_hung else box.get("roof")
        if isinstance(lm, dict) and "error" not in lm:
            lm["exchange"] = transport
            lm["observations_total"] = int(total_obs)
        if world > 1 and not lm_hung:
            # a first multi-rank run should be diagnosable from its line alone: every rank's device time per phase (HIP events of its
            # own solve), what each kind of collective carried, and the communicator as the library's RCCL sees it
            mine = {"rank": rank, "observations": int(prob.num_observations)}
            if isinstance(lm, dict) and "error" not in lm:   # every rank must have seen the same sums: a transport that skipped a collective shows here
                mine["initial_cost"], mine["final_cost"] = lm.get("initial_cost"), lm.get("final_cost")
            roof = box.get("roof")
            if isinstance(roof, dict) and "phases" in roof:
                mine["phase_ms_per_lm_iteration"] = {r["phase"]: r["ms_per_lm_iteration"] for r in roof["phases"]}
                mine["device_ms_per_lm_iteration_without_exchange"] = sum(r["ms_per_lm_iteration"] for r in roof["phases"] if r["phase"] != "exchange")
                mine["plan"] = {k: roof["plan"][k] for k in ("sharded_factorisation", "exchange_doubles", "separator_tiles", "separator_factor_tiles", "local_tasks", "separator_tasks",
                                                                 "local_levels", "separator_levels", "levels", "tasks", "schur_chunks")}
            if isinstance(box.get("collectives"), dict):
                n_it = max(1, (lm or {}).get("iterations", 1)) if isinstance(lm, dict) else 1
                mine["collectives_of_the_profiled_solve"] = box["collectives"]["collectives"]
            if comm is not None:
                mine["rccl"] = capi.rccl_describe(comm)
            rows = [None] * world
            dist.all_gather_object(rows, mine)
            if rank == 0 and isinstance(lm, dict):
                lm["per_rank"] = rows
                lm["ranks_agree"] = all(isinstance(r, dict) and r.get("final_cost") == rows[0].get("final_cost") and r.get("initial_cost") == rows[0].get("initial_cost") for r in rows)
                if serialize:
                    lm["note_one_gpu_hook"] = ("all ranks share ONE GPU and take turns between collectives: per_rank phase times are each rank's own device time "
                                               "(what its GPU would be busy for on a node); ms_per_lm_iteration and the collectives' ms include the waiting for the other ranks' turns")

    out = None
    if rank == 0:
        kname = "rsba::eval_kernel<%s,%d,1>" % ("true" if prob.calibrated else "false", prob.poses_per_frame)
        out = {
            "metric": "residual+Jacobian evals/sec", "value": total_obs * args.steps / elapsed, "unit": "obs evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "untimed_clock_ramp_steps": 200, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.config}: rolling-shutter scene, {full.num_frames} frames x {full.poses_per_frame} poses, "
                                   f"{full.num_points} points, {full.num_observations} observations, HORIZONTAL shutter, "
                                   + ("calibrated" if full.calibrated else f"shared intrinsics as a parameter block, Huber({full.huber_a:g})"),
                       "observations_total": int(total_obs), "observations_this_rank": int(prob.num_observations), "jacobian_cols": prob.jacobian_cols,
                       "partition": partition},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "traffic_provenance": traffic_prov, "kernel": kname, "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": abytes, "bytes_per_observation": abytes / max(1, prob.num_observations),
                         "rank": 0},
            "roofline_lm": lm_roof,
        }
        try:
            free_b, total_b = torch.cuda.mem_get_info(local_rank)
            out["device_memory"] = {"in_use_bytes": int(total_b - free_b), "total_bytes": int(total_b),
                                    "note": "hipMemGetInfo at the end of the run: the problem's handle (observations, parameters, evaluation outputs, plan, both sets of the Cholesky's write-once cells) plus the runtime's own"}
        except Exception as e:  # noqa: BLE001
            out["device_memory"] = {"error": repr(e)}
        if not args.no_next_rows and world == 1 and not lm_hung and args.config == "C4":
            out["next_rows"] = next_rows(prob, dp, local_rank)
        if not args.no_next_rows and world == 1 and not lm_hung and args.config in ("C4", "C5"):
            try:   # what each rank of the 8-GPU run will execute, measured one shard at a time on this GPU
                out["shard_eval_n8"] = shard_eval(full, 8, local_rank, capi, steps=max(1, args.steps))
                out["shard_eval_n8"]["predicted_speedup_over_this_run"] = out["shard_eval_n8"]["predicted_value"] / out["value"]
            except Exception as e:  # noqa: BLE001
                out["shard_eval_n8"] = {"error": repr(e)}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(prob, lm_iters=3 if prob.num_observations < 5_000_000 else 1)
        # the LM half of the metric goes LAST (a record that keeps only the tail of this line keeps the scalars): the full block, then
        # its scalars once more as the very last key
        out["lm"] = lm
        if transport_error:
            out["lm"] = {"error": "LM leg left out: the library's RCCL communicator could not be set up on every rank", "transport_error": transport_error}
        if isinstance(out["lm"], dict):
            keep = ("marginal_ms_per_lm_iteration", "ms_per_lm_iteration", "iterations", "initial_cost", "final_cost", "first_solve_wall_s", "n_gpus", "exchange", "error",
                    "lm_free_gauge", "predicted_value_n8")
            out["lm_headline"] = {k: out["lm"][k] for k in keep if k in out["lm"]}
            if isinstance(lm_roof, dict) and "sum_ms_per_lm_iteration" in lm_roof:
                out["lm_headline"]["host_form_phase_sum_ms"] = lm_roof["sum_ms_per_lm_iteration"]
            out["lm_headline"]["workload"] = args.config
            out["lm_headline"]["definition"] = ("marginal_ms_per_lm_iteration = (solve of 2k iterations - solve of k) / k extra iterations: what one more LM iteration costs (quoted first everywhere); "
                                                "ms_per_lm_iteration = whole k-iteration solve / k (also carries iteration 0 — the initial evaluation, two linearisations, the Jacobi scales — and the write-back)")
            if isinstance(out.get("shard_eval_n8"), dict) and "predicted_value" in out["shard_eval_n8"]:
                out["lm_headline"]["predicted_value_n8"] = out["shard_eval_n8"]["predicted_value"]
                out["lm_headline"]["predicted_speedup_n8"] = out["shard_eval_n8"]["predicted_speedup_over_this_run"]
    if lm_hung:                      # do not touch the device or the process group again: report and leave
        if rank == 0:
            print(json.dumps(out), flush=True)
        os._exit(0)
    dp.close()
    if comm is not None:
        capi.rccl_comm_destroy(comm)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
