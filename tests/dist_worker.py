"""Worker for tests/test_distributed.py, launched by torch.distributed.run with 2 ranks.
   python -m torch.distributed.run --nproc-per-node 2 ... tests/dist_worker.py <mode> <outdir>
mode "cpu": host-side exchange logic over gloo, partial blocks from the oracle (no GPU needed)
mode "gpu": the sharded on-device LM solve, both ranks on GPU 0, exchange staged through gloo ("gpu_priors": with motion priors,
"gpu_free_ratio": with a free interFrameRatio, "gpu_pose_priors": with GoodPosePrior blocks)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def scene():
    from rsba_amd.problem import apply_gauge_masks
    from rsba_amd.scene import make_scene
    p = make_scene(16, 700, seed=77).problem
    apply_gauge_masks(p, fix_first_n_cameras=1)
    p.pose_fixed_mask[-1, -1] |= 0b111000
    return p


def nd_mode(mode, outdir, rank, world, dist, torch):
    """"nd:<config>:<iterations>[:huber]": a BASELINE configuration sharded along the top separators of the reduced system's elimination tree
    (rsba_partition_points) — every rank factors its own part, the separators' tiles are what travels — against the single-GPU solve."""
    from rsba_amd import capi
    from rsba_amd.distributed import attach
    from rsba_amd.scene import make_config
    _, cfg, iters = mode.split(":")[:3]
    flags = mode.split(":")[3:]
    if cfg.startswith("S"):   # "S<frames>": a scene of that many rolling-shutter frames, 70 points per frame
        from rsba_amd.problem import apply_gauge_masks
        from rsba_amd.scene import make_scene
        full = make_scene(int(cfg[1:]), 70 * int(cfg[1:]), seed=3).problem
        apply_gauge_masks(full, fix_first_n_cameras=1)
        full.pose_fixed_mask[-1, -1] |= 0b111000
    else:
        full = make_config(cfg).problem
    if "intr" in flags:      # shared intrinsics as a parameter block + Huber, as BASELINE config 5 has them: a dense border of the reduced system
        full.calibrated = False; full.huber_a = 2.0
        full.intrinsics = full.intrinsics * (1.0 + 1e-3 * np.array([[1, -1, 20, -20, 10, 10, -10, 0.5, -0.5]]))
    if "perframe" in flags:  # a 9-block of intrinsics PER FRAME (f.cam, CeresHandler.h:256-264,273-280): since round 6 in the sharded plan (a block's pseudo frames sit in its frame's part or in a separator) and in the device-side loop (a candidate's records in a second set)
        full.calibrated = False; full.huber_a = 2.0
        rng = np.random.default_rng(3)
        full.intrinsics = np.tile(full.intrinsics[:1], (full.num_frames, 1)) * (1.0 + 1e-3 * rng.normal(size=(full.num_frames, 9)) * np.array([[1, 1, 20, 20, 10, 10, 10, 0.5, 0.5]]))
        full.frame_intrinsics = np.arange(full.num_frames, dtype=np.int32)
        if "mixedintr" in flags:   # ... and the frames WITHOUT f.cam fall back on the session's block (sess.cam): every third frame keeps its own, the rest share block 0 — a dense border again, beside the blocks inside the parts
            own = np.arange(full.num_frames) % 3 == 1
            full.frame_intrinsics = np.where(own, np.cumsum(own), 0).astype(np.int32)
            full.intrinsics = np.ascontiguousarray(full.intrinsics[: int(own.sum()) + 1])
    if "priors" in flags:    # a motion prior between every two consecutive frames (CeresHandler.h:147-185), known interFrameRatio: each rank contributes the priors of its part
        full.prior_kind, full.prior_scale, full.inter_frame_ratio = 2, 25.0, 1.2
        full.prior_frames = np.arange(1, full.num_frames, dtype=np.int32)
        if "freeratio" in flags:   # ... the reference's default: the ratio is a free, lower-bounded block (CeresHandler.h:161,172,175) — its column's forward solve runs part by part
            full.prior_kind, full.inter_frame_ratio, full.ratio_free = 1, 1.0, True
    if "posepriors" in flags:   # GoodPosePrior blocks (CeresHandler.h:188-204): every rank passes the same blocks; their terms go to the rank whose part holds the pose
        rng = np.random.default_rng(5)
        full.pose_prior_block = np.arange(2, 2 * full.num_frames, dtype=np.int32)
        full.pose_prior_values = full.poses.reshape(-1, 6)[full.pose_prior_block] + rng.normal(0, 0.01, (len(full.pose_prior_block), 6))
        full.pose_prior_rotation, full.pose_prior_position = 3.0, 5.0
    if "spherical" in flags:   # the SphericalPrior of a session that starts at the origin (CeresHandler.h:36-50,127-130): ONE 2-residual block on the first pose of frame 1 — its terms go to the rank whose part holds that pose
        full.poses[0] = 0.0; full.poses[1] = 0.0; full.poses[1, :, 3:] += 1e-4
        full.spherical_pose_block = 2
    if "hostrank" in flags and rank == 1:   # ONE rank cannot run the loop without the host (test hook of the library; in the field: a rank without observations, or with phase timers on): ALL ranks must then take the host form — their collectives pair up or the solve hangs
        os.environ["RSBA_DEVICE_LM_OFF_ON_THIS_RANK"] = "1"
    if "corrupt" in flags:   # the first persistent-driver solve loses an entry of its result (test hook of the library): every rank must notice through exchange (3)
        os.environ["RSBA_CHOL_TEST_CORRUPT"] = "1"
    if "emptyrank" in flags:   # the last rank owns no point at all (a host that has fewer pieces of work than ranks): its kernels have nothing to do, its exchanges still pair up
        owner, ntop = capi.partition_points(full, world - 1)
    else:
        owner, ntop = capi.partition_points(full, world)
    shard = full.shard(rank, world, owner)
    torch.cuda.set_device(0)
    opts = dict(max_num_iterations=int(iters), function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0)
    dp = capi.DeviceProblem(shard, device=0)
    comm, transport = None, "callback (gloo, blocking)"
    if "mock" in flags:      # the library's NATIVE exchange (ncclAllReduce on its own stream, nothing of the host inside an iteration) over tools/libmock_rccl.so:
        from rsba_amd.distributed import attach_rccl   # a stream-ordered stand-in for RCCL that lets the ranks share one GPU (RSBA_RCCL_LIB, set by the test)
        comm = attach_rccl(dp, 0)
        d = capi.rccl_describe(comm)
        transport = f"native, stream-ordered: nccl version {d['rccl_version']}, {d['comm_ranks']} ranks"
    else:
        attach(dp)
    s, tr = dp.solve(capi.default_options(**opts))
    st = dp.plan_stats()
    xs = dp.exchange_stats()
    dp.close()
    if comm is not None:
        capi.rccl_comm_destroy(comm)
    out = {"rank": rank, "transport": transport, "collective_calls": {k: v["calls"] for k, v in xs["collectives"].items()}, "world": world, "n_full": int(full.num_observations), "n_shard": int(shard.num_observations), "top_tile_columns": ntop,
           "final_cost": s.final_cost, "initial_cost": s.initial_cost, "iters": s.num_iterations, "dag_fallbacks": s.num_dag_fallbacks,
           "reduced": s.num_residual_blocks_reduced, "params": s.num_parameters_reduced, "costs": [t.cost for t in tr], "plan": st,
           "poses_sum": float(np.abs(shard.poses).sum()), "points_sum": float(np.abs(shard.points).sum()), "ratio": float(shard.inter_frame_ratio),
           "prior_values_sum": None if shard.pose_prior_values is None else float(np.abs(shard.pose_prior_values).sum())}
    os.environ.pop("RSBA_CHOL_TEST_CORRUPT", None)
    os.environ.pop("RSBA_DEVICE_LM_OFF_ON_THIS_RANK", None)
    if rank == 0:
        ref = full.copy()
        with capi.DeviceProblem(ref) as d1:
            s1, tr1 = d1.solve(capi.default_options(**dict(opts, level_scheduled_cholesky=int("corrupt" in flags))))
            st1 = d1.plan_stats()
        out.update(ref_final=s1.final_cost, ref_initial=s1.initial_cost, ref_iters=s1.num_iterations, ref_reduced=s1.num_residual_blocks_reduced, ref_params=s1.num_parameters_reduced,
                   pose_err=float(np.abs(ref.poses - shard.poses).max()), point_err=float(np.abs(ref.points - shard.points).max()),
                   traj_err=float(max(abs(a.cost - b.cost) / b.cost for a, b in zip(tr, tr1))), ref_plan=st1, ref_ratio=float(ref.inter_frame_ratio),
                   prior_err=None if ref.pose_prior_values is None else float(np.abs(ref.pose_prior_values - shard.pose_prior_values).max()),
                   decisions_equal=bool(all(a.step_is_successful == b.step_is_successful for a, b in zip(tr, tr1))))
    with open(os.path.join(outdir, f"rank{rank}.json"), "w") as f:
        json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


def planfail_mode(mode, outdir, rank, world, dist, torch):
    """"planfail:<hook>": the symbolic phase fails on RANK 1 ONLY (a test hook of the instrumented library: RSBA_TEST_FAIL_PLAN before the lists are
    built, RSBA_TEST_FAIL_DEVICE_PLAN = the device lists' allocations).  The ranks vote in the middle of the plan: rank 1 must still enter
    the vote, every rank must come back with an error (nobody waits for ever), and a retry without the fault must solve as a fresh handle does."""
    from rsba_amd import capi
    from rsba_amd.distributed import attach
    from rsba_amd.scene import make_config
    hook = mode.split(":")[1]
    full = make_config("C2").problem
    owner, _ = capi.partition_points(full, world)
    shard = full.shard(rank, world, owner)
    torch.cuda.set_device(0)
    dp = capi.DeviceProblem(shard, device=0)
    attach(dp)
    opts = capi.default_options(max_num_iterations=4)
    if rank == 1:
        os.environ[hook] = "1"
    err = None
    try:
        dp.solve(opts)
    except capi.RsbaError as e:
        err = str(e)
    os.environ.pop(hook, None)
    dist.barrier()
    s, _ = dp.solve(opts)
    st = dp.plan_stats()
    dp.close()
    out = {"rank": rank, "error": err, "final_cost": s.final_cost, "iters": s.num_iterations, "sharded": st["sharded_factorisation"]}
    if rank == 0:
        ref = full.copy()
        with capi.DeviceProblem(ref) as d1:
            s1, _ = d1.solve(opts)
        out["ref_final"] = s1.final_cost
    with open(os.path.join(outdir, f"rank{rank}.json"), "w") as f:
        json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


def mock_timeout_mode(outdir, rank, world, dist, torch):
    """Two ranks over tools/libmock_rccl.so: one all-reduce both enter (checked), then one that ONLY RANK 0 enters — its device-side wait
    must give up after RSBA_MOCK_RCCL_TIMEOUT_S and the communicator must say so when it is destroyed."""
    import ctypes as C
    import time
    class UniqueId(C.Structure):   # ncclUniqueId travels BY VALUE (a 128-byte struct)
        _fields_ = [("internal", C.c_char * 128)]
    lib = C.CDLL(os.environ["RSBA_RCCL_LIB"])
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.ncclCommDestroy.argtypes = [C.c_void_p]
    torch.cuda.set_device(0)
    uid = UniqueId()
    if rank == 0:
        assert lib.ncclGetUniqueId(C.byref(uid)) == 0
    box = [bytes(uid) if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    uid = UniqueId.from_buffer_copy(box[0])
    comm = C.c_void_p()
    assert lib.ncclCommInitRank(C.byref(comm), world, uid, rank) == 0
    x = torch.full((100000,), float(rank + 1), dtype=torch.float64, device="cuda")
    assert lib.ncclAllReduce(x.data_ptr(), x.data_ptr(), x.numel(), 8, 0, comm, None) == 0
    torch.cuda.synchronize()
    out = {"rank": rank, "sum_ok": bool((x == float(sum(range(1, world + 1)))).all().item())}
    y = torch.full((1000,), float(rank), dtype=torch.float64, device="cuda")
    assert lib.ncclAllReduce(y.data_ptr(), y.data_ptr(), y.numel(), 8, 2, comm, None) == 0     # max
    torch.cuda.synchronize()
    out["max_ok"] = bool((y == float(world - 1)).all().item())
    t0 = time.time()
    if rank == 0:
        assert lib.ncclAllReduce(x.data_ptr(), x.data_ptr(), x.numel(), 8, 0, comm, None) == 0   # enqueued: returns at once
        out["enqueue_s"] = time.time() - t0
        torch.cuda.synchronize()                                                                  # ... and gives up on the device
        out["gave_up_after_s"] = time.time() - t0
    dist.barrier()
    out["destroy"] = int(lib.ncclCommDestroy(comm))
    with open(os.path.join(outdir, f"rank{rank}.json"), "w") as f:
        json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


def main():
    mode, outdir = sys.argv[1], sys.argv[2]
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    if mode.startswith("nd:"):
        return nd_mode(mode, outdir, rank, world, dist, torch)
    if mode == "mock_timeout":
        return mock_timeout_mode(outdir, rank, world, dist, torch)
    if mode.startswith("planfail:"):
        return planfail_mode(mode, outdir, rank, world, dist, torch)
    full = scene()
    if mode in ("gpu_priors", "gpu_free_ratio"):   # motion priors are replicated terms: every rank lists them, rank 0 contributes them
        full.prior_kind, full.prior_scale, full.inter_frame_ratio = 2, 25.0, 1.2
        full.prior_frames = np.arange(1, full.num_frames, dtype=np.int32)
        if mode == "gpu_free_ratio":   # the free ratio's scalar LM state is replicated on every rank
            full.prior_kind, full.inter_frame_ratio, full.ratio_free = 1, 1.0, True
    if mode == "gpu_pose_priors":   # GoodPosePrior blocks (CeresHandler.h:188-204): every rank passes the same blocks and must leave with the same solved priorPoses
        rng = np.random.default_rng(5)
        full.pose_prior_block = np.arange(2, 2 * full.num_frames, dtype=np.int32)
        full.pose_prior_values = full.poses.reshape(-1, 6)[full.pose_prior_block] + rng.normal(0, 0.01, (len(full.pose_prior_block), 6))
        full.pose_prior_rotation, full.pose_prior_position = 3.0, 5.0
    shard = full.shard(rank, world)
    out = {"rank": rank, "world": world, "n_full": full.num_observations, "n_shard": shard.num_observations}
    if mode == "cpu":
        from oracle import oracle as O
        from rsba_amd.distributed import union_structure
        # every observation on exactly one rank; a point's observations never split
        n = torch.tensor([shard.num_observations]); dist.all_reduce(n); out["n_sum"] = int(n.item())
        owners = np.zeros(full.num_points, dtype=np.int64); owners[np.unique(shard.obs_point)] = 1
        t = torch.from_numpy(owners); dist.all_reduce(t); out["max_owners_per_point"] = int(t.max().item())
        # exchange (1): partial camera blocks add up to the full ones; point blocks are complete on the owner
        U, gc, V, gp = O.normal_equations(shard)
        Uf, gcf, Vf, gpf = O.normal_equations(full)
        tU, tg = torch.from_numpy(U.copy()), torch.from_numpy(gc.copy())
        dist.all_reduce(tU); dist.all_reduce(tg)
        out["U_err"] = float(np.abs(tU.numpy() - Uf).max() / np.abs(Uf).max())
        out["gc_err"] = float(np.abs(tg.numpy() - gcf).max() / np.abs(gcf).max())
        own = (np.arange(full.num_points) % world) == rank
        out["V_err"] = float(np.abs(V[own] - Vf[own]).max() / np.abs(Vf).max())
        out["V_foreign"] = float(np.abs(V[~own]).max())
        # cost is additive over shards
        c = torch.tensor([O.evaluate(shard, gradient=False)[1]], dtype=torch.float64); dist.all_reduce(c)
        out["cost_err"] = abs(float(c.item()) - O.evaluate(full, gradient=False)[1]) / O.evaluate(full, gradient=False)[1]
        # structure union == structure of the whole problem
        def structure(p):
            F = p.num_frames; m = np.zeros((F, F), dtype=np.uint8); cnt = np.bincount(p.obs_frame, minlength=F).astype(np.int64)
            for j in np.unique(p.obs_point):
                fr = p.obs_frame[p.obs_point == j]
                a, b = np.meshgrid(fr, fr, indexing="ij"); m[np.maximum(a, b), np.minimum(a, b)] = 1
            return m, cnt
        m, cnt = structure(shard)
        mu, cu = union_structure(m, cnt)
        mf, cf = structure(full)
        out["mask_equal"] = bool(np.array_equal(mu, mf)); out["count_equal"] = bool(np.array_equal(cu, cf))
    else:
        from rsba_amd import capi
        from rsba_amd.distributed import attach, gather_points
        torch.cuda.set_device(0)
        opts = dict(max_num_iterations=15)
        dp = capi.DeviceProblem(shard, device=0)
        attach(dp)
        s, tr = dp.solve(capi.default_options(**opts))
        dp.close()
        # rsba_solve merged the points over the ranks on the device: the host-side merge must be a no-op
        merged = shard.points.copy()
        gather_points(shard)
        out["native_merge_equals_host_merge"] = bool(np.array_equal(merged, shard.points))
        out.update(ratio=shard.inter_frame_ratio, final_cost=s.final_cost, initial_cost=s.initial_cost, iters=s.num_iterations, reduced=s.num_residual_blocks_reduced,
                   params=s.num_parameters_reduced, term=s.termination_type)
        if mode == "gpu_pose_priors": out["prior_values"] = shard.pose_prior_values.ravel().tolist()
        if rank == 0:
            ref = full.copy()
            with capi.DeviceProblem(ref) as d1:
                s1, tr1 = d1.solve(capi.default_options(**opts))
            out.update(ref_ratio=ref.inter_frame_ratio, ref_final=s1.final_cost, ref_initial=s1.initial_cost, ref_iters=s1.num_iterations, ref_reduced=s1.num_residual_blocks_reduced,
                       ref_params=s1.num_parameters_reduced, pose_err=float(np.abs(ref.poses - shard.poses).max()),
                       point_err=float(np.abs(ref.points - shard.points).max()),
                       traj_err=float(max(abs(a.cost - b.cost) / b.cost for a, b in zip(tr, tr1))))
            if mode == "gpu_pose_priors":
                out.update(prior_err=float(np.abs(ref.pose_prior_values - shard.pose_prior_values).max()),
                           prior_moved=float(np.abs(ref.pose_prior_values - full.pose_prior_values).max()))
    with open(os.path.join(outdir, f"rank{rank}.json"), "w") as f:
        json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
