"""The N > 1 path: point-partitioned shards + all-reduce exchange, world_size 2 (and 4).
CPU (gloo): sharding and the exchange payloads, with per-shard blocks from the oracle.
GPU (-m gpu): the full sharded on-device LM solve equals the single-GPU solve (two ranks share GPU 0 and
the exchange is staged through gloo — the RCCL transport is the same callback with backend "nccl")."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


MOCK_RCCL = os.path.join(ROOT, "tools", "libmock_rccl.so")   # tools/mock_rccl.hip, built by __graft_entry__.build(): test infrastructure
HOOKS_LIB = os.path.join(ROOT, "rsba_amd", "_lib", "librsba_amd_hooks.so")   # the instrumented build (-DRSBA_TEST_HOOKS): the only one that reads the fault-injection switches


def run_two_ranks(mode, outdir, world=2, timeout=900, env_extra=None):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    if any(flag in mode.split(":") for flag in ("hostrank", "corrupt")) or mode.startswith("planfail:"):
        env["RSBA_AMD_LIB"] = HOOKS_LIB   # modes that inject a fault run on the instrumented library
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py"), mode, str(outdir)],
                       capture_output=True, text=True, env=env, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return [json.load(open(os.path.join(outdir, f"rank{k}.json"))) for k in range(world)]


@pytest.mark.parametrize("world", [2, 4])
def test_sharding_and_exchange_payloads_gloo(tmp_path, oracle, world):
    res = run_two_ranks("cpu", tmp_path, world)
    for o in res:
        assert o["world"] == world
        assert o["n_sum"] == o["n_full"] and o["max_owners_per_point"] == 1
        assert o["U_err"] <= 1e-13 and o["gc_err"] <= 1e-13 and o["cost_err"] <= 1e-13
        assert o["V_err"] <= 1e-15 and o["V_foreign"] == 0.0
        assert o["mask_equal"] and o["count_equal"]
    assert sum(o["n_shard"] for o in res) == res[0]["n_full"] and all(o["n_shard"] > 0 for o in res)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["gpu", "gpu_priors", "gpu_free_ratio"])
def test_sharded_solve_equals_single_gpu_solve(tmp_path, mode):
    res = run_two_ranks(mode, tmp_path)
    a, b = res
    assert a["final_cost"] == b["final_cost"] and a["iters"] == b["iters"]          # ranks decide identically
    assert a["native_merge_equals_host_merge"] and b["native_merge_equals_host_merge"]
    assert a["iters"] == a["ref_iters"] and a["reduced"] == a["ref_reduced"] and a["params"] == a["ref_params"]
    assert abs(a["initial_cost"] - a["ref_initial"]) <= 1e-12 * a["ref_initial"]
    assert a["traj_err"] <= 1e-9
    assert abs(a["final_cost"] - a["ref_final"]) <= 1e-9 * a["ref_final"]
    assert a["pose_err"] <= 1e-7 and a["point_err"] <= 1e-6
    assert a["ratio"] == b["ratio"] and abs(a["ratio"] - a["ref_ratio"]) <= 1e-8
    assert (mode == "gpu_free_ratio") == (a["ratio"] not in (1.0, 1.2))


@pytest.mark.gpu
def test_sharded_solve_on_four_ranks_equals_single_gpu_solve(tmp_path):
    """world_size 4 (four ranks share GPU 0, the exchange staged through gloo): a quarter of the points per rank, three
    all-reduces per LM iteration over four contributors — the ranks decide identically and land on the single-GPU solve."""
    res = run_two_ranks("gpu", tmp_path, 4)
    a = res[0]
    for o in res[1:]:
        assert o["final_cost"] == a["final_cost"] and o["iters"] == a["iters"] and o["ratio"] == a["ratio"]
    assert all(o["native_merge_equals_host_merge"] for o in res)
    assert a["iters"] == a["ref_iters"] and a["reduced"] == a["ref_reduced"] and a["params"] == a["ref_params"]
    assert a["traj_err"] <= 1e-9 and abs(a["final_cost"] - a["ref_final"]) <= 1e-9 * a["ref_final"]
    assert a["pose_err"] <= 1e-7 and a["point_err"] <= 1e-6


@pytest.mark.gpu
def test_sharded_solve_with_pose_priors_leaves_every_rank_with_the_solved_prior_poses(tmp_path):
    """GoodPosePrior blocks in a sharded solve: the priorPoses blocks are free parameter blocks (CeresHandler.h:188-204), their
    replicated normal-equation terms come from rank 0 — but EVERY rank steps them and writes the solved values back."""
    a, b = run_two_ranks("gpu_pose_priors", tmp_path)
    assert a["final_cost"] == b["final_cost"] and a["iters"] == b["iters"]
    assert a["prior_values"] == b["prior_values"]                       # rank 1 used to return the values it was given
    assert a["iters"] == a["ref_iters"] and a["params"] == a["ref_params"] and a["reduced"] == a["ref_reduced"]
    assert a["traj_err"] <= 1e-9 and abs(a["final_cost"] - a["ref_final"]) <= 1e-9 * a["ref_final"]
    assert a["prior_err"] <= 1e-7 and a["pose_err"] <= 1e-7 and a["prior_moved"] > 1e-4


@pytest.mark.gpu
def test_native_rccl_transport_one_rank():
    """rsba_set_exchange_rccl: ncclCommInitRank + ncclAllReduce issued by the library on the solver's stream.  One rank is
    all a one-GPU box can hold (RCCL refuses two ranks on one device); the all-reduces still run — as identities — so the
    solve must equal the plain single-GPU solve to the bit."""
    import numpy as np
    from rsba_amd import capi
    from rsba_amd.distributed import attach_rccl
    from rsba_amd.problem import apply_gauge_masks
    from rsba_amd.scene import make_scene
    p = make_scene(24, 1500, seed=5).problem
    apply_gauge_masks(p, fix_first_n_cameras=1)
    p.pose_fixed_mask[-1, -1] |= 0b111000
    q = p.copy()
    with capi.DeviceProblem(p) as dp:
        s, _ = dp.solve(capi.default_options(max_num_iterations=8))
    dq = capi.DeviceProblem(q)
    comm = attach_rccl(dq, 0)
    s2, _ = dq.solve(capi.default_options(max_num_iterations=8, profile_phases=1))
    t = dq.phase_times()
    dq.close()
    capi.rccl_comm_destroy(comm)
    assert s2.final_cost == s.final_cost and s2.num_iterations == s.num_iterations
    assert np.array_equal(p.poses, q.poses) and np.array_equal(p.points, q.points)
    assert t["exchange"][1] > 0 and t["cholesky"][1] == s.num_iterations - 1


def run_bench(nproc, config, extra_env=None, lm_iters=6, timeout=1200):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RSBA_BENCH_TEST_ONE_GPU="1", **(extra_env or {}))
    args = ["--gpus", str(nproc), "--steps", "5", "--warmup", "1", "--config", config, "--no-cpu-baseline", "--no-next-rows", "--lm-iters", str(lm_iters)]
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args if nproc == 1 else \
          [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.gpu
@pytest.mark.parametrize("world,config", [(2, "C2"), (5, "C4"), (8, "C4")])
def test_bench_takes_its_native_rccl_branch_with_several_ranks(world, config):
    """The branch of bench.py the driver's multi-GPU command takes — attach_rccl -> rsba_rccl_comm_create -> rsba_set_exchange_rccl, the warm
    handle of the LM leg on the SAME communicator, the LM solve on the watchdog thread with comm set — run here with 2 and 8 ranks on one
    GPU: RSBA_BENCH_NATIVE=1 + RSBA_RCCL_LIB = the stream-ordered stand-in for librccl (tools/mock_rccl.hip; RCCL itself refuses two
    ranks on one device).  Every rank must report the communicator as the library sees it, the sharded plan, and the one-GPU cost."""
    one = run_bench(1, config)
    many = run_bench(world, config, {"RSBA_BENCH_NATIVE": "1", "RSBA_RCCL_LIB": MOCK_RCCL})
    assert many["n_gpus"] == world and many["scaling"] == "strong"
    assert many["config"]["observations_total"] == one["config"]["observations_total"]
    lm = many["lm"]
    assert "error" not in lm, lm
    assert "native" in lm["exchange"] and "libmock_rccl.so" in lm["exchange"]
    assert len(lm["per_rank"]) == world and lm["ranks_agree"]
    for r, row in enumerate(lm["per_rank"]):
        assert row["rank"] == r and row["rccl"]["comm_ranks"] == world and row["rccl"]["comm_rank"] == r
        assert row["plan"]["sharded_factorisation"] == 1
        assert row["collectives_of_the_profiled_solve"]["(2) reduced system"]["calls"] > 0
    assert abs(lm["initial_cost"] - one["lm"]["initial_cost"]) <= 1e-12 * one["lm"]["initial_cost"]
    assert abs(lm["final_cost"] - one["lm"]["final_cost"]) <= 1e-9 * one["lm"]["final_cost"]
    assert many["lm_headline"]["ms_per_lm_iteration"] > 0


@pytest.mark.gpu
def test_bench_strong_scaling_two_ranks_on_one_gpu(tmp_path):
    """bench.py --gpus 2 through its one-GPU hook: the SAME C2 scene sharded by point over two ranks (gloo callback
    exchange), observations_total unchanged, the sharded LM reaching the single-GPU cost."""
    def run(nproc):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RSBA_BENCH_TEST_ONE_GPU="1")
        args = ["--gpus", str(nproc), "--steps", "5", "--warmup", "1", "--config", "C2", "--no-cpu-baseline", "--no-next-rows", "--lm-iters", "6"]
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args if nproc == 1 else \
              [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.join(ROOT, "bench.py")] + args
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    one, two = run(1), run(2)
    assert one["scaling"] == two["scaling"] == "strong"
    assert one["config"]["observations_total"] == two["config"]["observations_total"]
    assert two["n_gpus"] == 2 and two["config"]["observations_this_rank"] < one["config"]["observations_this_rank"]
    assert abs(two["lm"]["final_cost"] - one["lm"]["final_cost"]) <= 1e-9 * one["lm"]["final_cost"]
    assert abs(two["lm"]["initial_cost"] - one["lm"]["initial_cost"]) <= 1e-12 * one["lm"]["initial_cost"]
    assert {r["phase"] for r in one["roofline_lm"]["phases"]} >= {"eval_lm", "schur", "cholesky", "project"}


@pytest.mark.gpu
def test_a_failed_exchange_says_what_failed():
    """A transport that fails must not leave the caller with a bare status: the error names the collective (size, operation, rank) and
    keeps whatever the transport itself reported (the RCCL transport: the ncclResult string)."""
    import ctypes as C
    from rsba_amd import capi
    from rsba_amd.distributed import ALLREDUCE_FN
    from rsba_amd.scene import make_scene
    p = make_scene(12, 400, seed=3).problem
    dp = capi.DeviceProblem(p)
    cb = ALLREDUCE_FN(lambda ctx, ptr, count, op, stream: 1)
    capi._check(capi.lib().rsba_set_exchange(dp._h, cb, None, C.c_int32(0), C.c_int32(1)))
    with pytest.raises(capi.RsbaError) as err:
        dp.solve(capi.default_options(max_num_iterations=3))
    dp.close()
    msg = str(err.value)
    assert "all-reduce" in msg and "doubles" in msg and "rank 0 of 1" in msg, msg


# ---- the sharded factorisation: points partitioned along the top separators of the reduced system's elimination tree ----

@pytest.mark.parametrize("config,worlds", [("C2", (2,)), ("C4", (2, 4, 8))])
def test_partition_points_cuts_along_separators(config, worlds):
    """rsba_partition_points (host only): every point on one rank by construction; the observations are balanced; the tile columns
    (4 rolling-shutter frames each) that more than one rank's points are seen in are exactly the separators' — every other column of
    the reduced camera system is complete on one rank, which is what lets that rank factor it alone."""
    from rsba_amd import capi
    from rsba_amd.scene import make_config
    p = make_config(config).problem
    FT = 4
    nt = (p.num_frames + FT - 1) // FT
    for world in worlds:
        owner, ntop = capi.partition_points(p, world)
        assert owner.shape == (p.num_points,) and owner.min() >= 0 and owner.max() == world - 1
        load = np.bincount(owner[p.obs_point], minlength=world)
        assert load.min() > 0 and load.max() <= (1.02 if nt >= 30 * world else 1.35) * load.mean()   # (cut tile by tile: equal to a per cent when a rank has tiles to choose from)
        touched = np.zeros((world, nt), dtype=bool)
        touched[owner[p.obs_point], p.obs_frame // FT] = True
        shared = int((touched.sum(0) > 1).sum())
        assert 0 < shared <= ntop and ntop <= {2: 0.35, 4: 0.2, 8: 0.2}[world] * nt + 8
        # shards built from it: disjoint, complete
        n = sum(p.shard(r, world, owner).num_observations for r in range(world))
        assert n == p.num_observations
    one, ntop = capi.partition_points(p, 1)
    assert not one.any() and ntop == 0


@pytest.mark.parametrize("kind", ["loop_closure", "shuffled_point_numbers"])
def test_partition_points_on_graphs_that_are_not_a_plain_band(kind):
    """The cut is a frontier of a linear arrangement of the tile graph, whatever the graph: a video that revisits a place (5 % of the points
    seen again 200 frames later: long-range edges) is still cut validly — the tile columns that more than one rank's points are seen in are
    the separators' — and the numbering of the points (here: shuffled, so that a point's number says nothing about where it is seen) does
    not enter the cut at all."""
    from rsba_amd import capi
    from rsba_amd.scene import make_scene
    rng = np.random.default_rng(0)
    p = make_scene(400, 20000, seed=5).problem
    FT, M = 4, p.num_points
    nt = (p.num_frames + FT - 1) // FT
    base = {w: capi.partition_points(p, w) for w in (2, 4)}
    if kind == "loop_closure":
        first = np.full(M, 10**9); np.minimum.at(first, p.obs_point, p.obs_frame)
        again = rng.choice(M, M // 20, replace=False)
        ef = np.concatenate([(first[j] + 200 + np.arange(6)) % p.num_frames for j in again]); ep = np.repeat(again, 6)
        order = np.argsort(np.concatenate([p.obs_frame, ef]), kind="stable")
        p.obs_frame = np.concatenate([p.obs_frame, ef])[order]; p.obs_point = np.concatenate([p.obs_point, ep])[order]
        p.obs_xy = np.concatenate([p.obs_xy, np.zeros((len(ef), 2))])[order]
    else:
        perm = rng.permutation(M)
        p.obs_point = perm[p.obs_point]      # (an int64 array: the binding coerces what it hands to the C side)
    for world in (2, 4):
        owner, ntop = capi.partition_points(p, world)
        load = np.bincount(owner[p.obs_point], minlength=world)
        touched = np.zeros((world, nt), dtype=bool)
        touched[owner[p.obs_point], p.obs_frame // FT] = True
        assert load.min() > 0 and load.max() <= 1.2 * load.mean()
        assert 0 < int((touched.sum(0) > 1).sum()) <= ntop <= (0.6 if kind == "loop_closure" else 0.4) * nt   # (long-range edges double a separator: both places it touches)
        if kind == "shuffled_point_numbers":   # same cut, the owners follow the points
            assert ntop == base[world][1]


def test_partition_points_with_an_intrinsics_block_per_frame():
    """A 9-block of intrinsics per frame (CeresHandler.h:256-264,273-280) adds pseudo frames to the reduced system — one per block behind the
    real frames, in tiles of their own: the cut takes them along (a block's tile is adjacent to the tiles of the frames seen through it, so it
    lands in their part or in a separator) — every tile, real or pseudo, that more than one rank's points reach is a separator tile."""
    from rsba_amd import capi
    from rsba_amd.scene import make_scene
    p = make_scene(300, 21000, seed=3).problem
    p.calibrated = False
    p.intrinsics = np.tile(p.intrinsics[:1], (p.num_frames, 1))
    p.frame_intrinsics = np.arange(p.num_frames, dtype=np.int32)
    FT, FR = 4, p.num_frames
    nt = (2 * FR + FT - 1) // FT          # two-pose frames: 12 unknowns per frame, one pseudo frame per intrinsics block
    for world in (2, 4):
        owner, ntop = capi.partition_points(p, world)
        load = np.bincount(owner[p.obs_point], minlength=world)
        touched = np.zeros((world, nt), dtype=bool)
        touched[owner[p.obs_point], p.obs_frame // FT] = True
        touched[owner[p.obs_point], (FR + p.frame_intrinsics[p.obs_frame]) // FT] = True
        assert load.min() > 0 and load.max() <= 1.25 * load.mean()
        assert 0 < int((touched.sum(0) > 1).sum()) <= ntop <= 0.4 * nt


def test_partition_points_refuses_what_cannot_be_cut():
    from rsba_amd import capi
    from rsba_amd.scene import make_scene
    p = make_scene(8, 200, seed=3).problem        # two tile columns: nothing to cut
    with pytest.raises(capi.RsbaError, match="cannot be cut"):
        capi.partition_points(p, 2)


def check_nd(res, world, sharded=True, fallbacks=0):
    a = res[0]
    for o in res[1:]:
        assert o["final_cost"] == a["final_cost"] and o["iters"] == a["iters"] and o["costs"] == a["costs"]        # ranks decide identically
        assert o["poses_sum"] == a["poses_sum"] and o["points_sum"] == a["points_sum"]                               # ... and leave with the same parameters
    assert all(o["dag_fallbacks"] == fallbacks for o in res)
    assert sum(o["n_shard"] for o in res) == a["n_full"]
    assert all(o["plan"]["sharded_factorisation"] == int(sharded) for o in res)
    assert a["iters"] == a["ref_iters"] and a["reduced"] == a["ref_reduced"] and a["params"] == a["ref_params"] and a["decisions_equal"]
    assert abs(a["initial_cost"] - a["ref_initial"]) <= 1e-12 * a["ref_initial"]
    assert a["traj_err"] <= 1e-9 and abs(a["final_cost"] - a["ref_final"]) <= 1e-9 * a["ref_final"]
    assert a["pose_err"] <= 1e-7 and a["point_err"] <= 1e-6
    return a


@pytest.mark.gpu
def test_sharded_factorisation_two_ranks_c2(tmp_path):
    """BASELINE configs 2-3 on two ranks (one GPU, gloo): each rank factors its own part of the reduced system, the separators' tiles
    are all-reduced between the two launches of the factorisation — the LM trajectory is the single-GPU one to 1e-9."""
    res = run_two_ranks("nd:C2:8", tmp_path, 2)
    a = check_nd(res, 2)
    assert a["plan"]["exchange_doubles"] < a["ref_plan"]["exchange_doubles"]


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_factorisation_at_c4_size(tmp_path, world):
    """The headline configuration (1000 frames, 100k points, 2.04 M observations) on 2, 4 and 8 ranks sharing one GPU: the sharded
    factorisation reproduces the single-GPU LM trajectory to 1e-9, and exchange (2) carries the separators only."""
    res = run_two_ranks("nd:C4:4", tmp_path, world)
    a = check_nd(res, world)
    full = a["ref_plan"]["exchange_doubles"]                 # the replicated factorisation's payload: every non-zero tile of S | rhs
    assert a["plan"]["exchange_doubles"] <= {2: 0.12, 4: 0.25, 8: 0.6}[world] * full
    assert a["plan"]["separator_tiles"] == a["top_tile_columns"]


@pytest.mark.gpu
def test_sharded_factorisation_at_c5_size(tmp_path):
    """BASELINE config 5 as it is quoted — 4000 frames, 500k points, 10.3 M observations, Huber loss, the shared intrinsics block as a
    parameter block — on EIGHT ranks (sharing one GPU; gloo callback transport): every rank factors its own part of the reduced system,
    the 9-wide intrinsics border is one more separator all ranks share, and the LM trajectory is the single-GPU one to 1e-9 (which
    tests/test_gpu_fullsize.py holds against the oracle's)."""
    res = run_two_ranks("nd:C5:3", tmp_path, 8, timeout=2400)
    a = check_nd(res, 8)
    full = a["ref_plan"]["exchange_doubles"]
    assert a["plan"]["exchange_doubles"] <= 0.15 * full       # (12.2 MB of separators against 156 MB of tiles)
    assert a["plan"]["separator_tiles"] >= a["top_tile_columns"]   # (+ the intrinsics pseudo tile)


@pytest.mark.gpu
def test_sharded_factorisation_with_shared_intrinsics_block(tmp_path):
    """Shared intrinsics as a parameter block + Huber (what BASELINE config 5 adds): the block's 9 unknowns are a dense border of the
    reduced system — one more separator every rank shares — and its rows are complete where their frames' parts are."""
    res = run_two_ranks("nd:C2:8:intr", tmp_path, 2)
    check_nd(res, 2)


@pytest.mark.gpu
def test_a_suspect_sharded_solve_sends_every_rank_back_to_the_replicated_factorisation(tmp_path):
    """RSBA_CHOL_TEST_CORRUPT makes the first persistent-driver solve of every rank lose an entry of its result.  The residual check
    (each rank: the rows of its own part) raises the flag, exchange (3) carries it to all ranks, and ALL of them repeat the iteration —
    and finish the problem — with the replicated factorisation on the level schedule: one fallback each, and the trajectory of the
    single-GPU level-scheduled solve."""
    res = run_two_ranks("nd:C2:6:corrupt", tmp_path, 2)
    check_nd(res, 2, fallbacks=1)


@pytest.mark.gpu
@pytest.mark.parametrize("hook,message", [("RSBA_TEST_FAIL_PLAN", "RSBA_TEST_FAIL_PLAN"), ("RSBA_TEST_FAIL_DEVICE_PLAN", "device plan")])
def test_a_plan_that_fails_on_one_rank_fails_on_every_rank(tmp_path, hook, message):
    """The ranks vote in the middle of the symbolic phase (one all-reduce: do everyone's points respect the cut?).  A rank whose plan fails
    before that — an unsupported size, or the device lists' allocations running out of memory — must still enter the vote and
    say so: rank 1 reports its own error, rank 0 that another rank failed, nobody waits; without the fault the same handles solve."""
    a, b = run_two_ranks("planfail:" + hook, tmp_path, 2, timeout=300)
    assert b["error"] is not None and message in b["error"], b["error"]
    assert a["error"] is not None and "another rank" in a["error"], a["error"]
    assert a["final_cost"] == b["final_cost"] and a["sharded"] == b["sharded"] == 1
    assert abs(a["final_cost"] - a["ref_final"]) <= 1e-9 * a["ref_final"]


@pytest.mark.gpu
def test_sharded_factorisation_on_three_ranks(tmp_path):
    """A rank count that is not a power of two: the top of the dissection is cut 1 + 2 (tile_order.hpp: the separator sits where the
    weight per rank balances), three parts, two levels of separators — the same trajectory as one GPU."""
    res = run_two_ranks("nd:S300:6", tmp_path, 3)
    check_nd(res, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,world", [("nd:C2:8:priors", 2), ("nd:S300:6:priors", 3), ("nd:S300:6:priors:intr", 4),
                                        ("nd:C2:8:priors:freeratio", 2), ("nd:S300:6:priors:freeratio", 3), ("nd:S300:6:priors:intr:freeratio", 4), ("nd:C2:6:priors:freeratio:hostrank", 2),
                                        ("nd:C2:8:posepriors", 2), ("nd:S300:6:posepriors:priors:freeratio", 3), ("nd:C2:3:spherical", 2), ("nd:S300:3:spherical:priors", 4),
                                        ("nd:S300:5:posepriors:perframe:spherical", 3)])
def test_sharded_factorisation_with_motion_priors(tmp_path, mode, world):
    """A motion prior between every two consecutive frames with a known interFrameRatio (the reference's usual video configuration,
    CeresHandler.h:147-185).  The prior between frames f and f - 1 goes to the rank whose part holds either frame (rank 0 when both
    sit in separators), so a part's columns are still complete on its rank; every rank adds its own priors' cost, blocks and model
    change.  Same trajectory as one GPU, where all priors are on the one rank."""
    res = run_two_ranks(mode, tmp_path, world)
    a = check_nd(res, world)
    if "freeratio" in mode:   # the free ratio (the reference's default): one more unknown, solved for alike on every rank — out of the second right-hand side that
        assert all(o["ratio"] == a["ratio"] for o in res) and a["ratio"] != 1.0   # rides through both launches of the sharded factorisation (FWD2 / FWD2P / ETA tasks)
        assert abs(a["ratio"] - a["ref_ratio"]) <= 1e-8 * abs(a["ref_ratio"])
    if "spherical" in mode:   # the SphericalPrior keeps the sharded plan and the device-side loop (round 6): three iterations, while its 1e20-weighted residual is still above rounding
        assert a["initial_cost"] > 1e38
    if "posepriors" in mode:  # GoodPosePrior blocks: every rank leaves with the same solved priorPoses values, the single-GPU ones
        assert all(o["prior_values_sum"] == a["prior_values_sum"] for o in res) and a["prior_err"] <= 1e-7
    if "hostrank" not in mode:   # every one of these runs the loop whose decisions are taken on the device (GoodPosePrior blocks on several ranks: since round 6 —
        assert all(o["plan"]["device_loop_solves"] >= 1 and o["plan"]["host_loop_solves"] == 0 for o in res)   # their blocks' gradient maximum is the lead rank's, one MAX exchange of its own)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,world", [("nd:S300:5:perframe", 2), ("nd:S300:5:perframe", 4), ("nd:S300:5:perframe:priors", 3), ("nd:S300:5:perframe:mixedintr", 2),
                                        ("nd:S300:5:perframe:mixedintr:priors:freeratio", 4), ("nd:C4:3:perframe", 4)])   # (the last: 1000 blocks at the headline size — exchange (2) 12 MB instead of 131)
def test_per_frame_intrinsics_blocks_on_several_ranks(tmp_path, mode, world):
    """Several intrinsics blocks (a 9-block per frame, CeresHandler.h:256-264,273-280; "mixedintr": two frames in three fall back on the session's
    block) in BOTH fast paths on several ranks (round 6): the SHARDED factorisation — a block's pseudo frames sit in a tile adjacent to the
    tiles of the frames seen through it, so the dissection puts it in their part or in a separator, and only the separators' tiles travel — and
    the loop whose decisions are taken on the device (a candidate's records go to a second set), its exchanges enqueued between its kernels:
    the single-GPU trajectory to 1e-9, every rank with the same parameters."""
    res = run_two_ranks(mode, tmp_path, world)
    a = check_nd(res, world)
    assert a["plan"]["exchange_doubles"] < a["ref_plan"]["exchange_doubles"]
    assert all(o["plan"]["device_loop_solves"] >= 1 for o in res)


@pytest.mark.gpu
def test_per_frame_intrinsics_blocks_with_the_replicated_factorisation(tmp_path, monkeypatch):
    """... and the same problem with the sharded plan switched off (RSBA_SHARDED=0: what any by-point partition that does not follow the
    separators gets): every structurally non-zero tile is all-reduced, every rank factors the whole system — the same trajectory."""
    monkeypatch.setenv("RSBA_SHARDED", "0")
    res = run_two_ranks("nd:S300:5:perframe", tmp_path, 2)
    check_nd(res, 2, sharded=False)


@pytest.mark.gpu
def test_every_rank_takes_the_same_form_of_the_trust_region_loop(tmp_path):
    """On several ranks the loop whose decisions are taken on the device (the exchanges of an iteration enqueued between its kernels,
    no host wait: solver.hip) is the default where it applies — the sharded tests above run it.  A rank that cannot run it says so in the
    problem-size exchange and ALL ranks take the host form: here rank 1 of 2 is made to (test hook), and the solve still pairs its
    collectives up and reproduces the single-GPU trajectory."""
    res = run_two_ranks("nd:C2:6:hostrank", tmp_path, 2)
    check_nd(res, 2)


@pytest.mark.gpu
def test_a_rank_without_observations(tmp_path):
    """Three ranks, the points cut for two: rank 2 owns nothing.  Its evaluation, its blocks and its share of every sum are empty, it
    still takes part in every exchange, says in the problem-size exchange that it needs the host form of the loop (all ranks follow),
    and the plan falls back to the replicated factorisation (the points do not respect a three-way cut).  Same trajectory as one GPU."""
    res = run_two_ranks("nd:C2:6:emptyrank", tmp_path, 3)
    a = check_nd(res, 3, sharded=False)
    assert sorted(o["n_shard"] for o in res)[0] == 0


# ---- the same over a STREAM-ORDERED transport: the library's native exchange (ncclAllReduce enqueued on the solver's stream, no host code
# inside an iteration) against tools/libmock_rccl.so, which sums through hipIpc-shared staging blocks with device-side waits — what RCCL
# does between GPUs, between ranks that share this one.  The loop whose decisions are taken on the device then runs as it will on a node:
# the host polls a pinned stamp while kernels and collectives of the iteration are in flight.

@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 5, 8])   # (5 and 8: round 6 — the stand-in's waits used to spin in every workgroup of its kernels, and with five or eight ranks on one GPU they starved each other's publish kernels until the bounded waits gave up: tools/mock_rccl.hip, gate_kernel)
def test_sharded_factorisation_at_c4_size_over_a_stream_ordered_transport(tmp_path, world):
    assert os.path.exists(MOCK_RCCL), "tools/libmock_rccl.so is built by __graft_entry__.build()"
    res = run_two_ranks("nd:C4:4:mock", tmp_path, world, env_extra={"RSBA_RCCL_LIB": MOCK_RCCL})
    a = check_nd(res, world)
    assert all("version 99900" in o["transport"] and f"{world} ranks" in o["transport"] for o in res)
    it = a["iters"] - 1
    # per LM iteration: (1) camera blocks, (2') separators, (3) step scalars, (4) camera step — issued by the library itself
    assert all(o["collective_calls"]["(2) reduced system"] == it and o["collective_calls"]["(4) camera step"] == it for o in res)
    assert all(o["collective_calls"]["(3) step scalars"] == it for o in res)
    # ... and the blocking gloo callback gives the same trajectory (two ranks: a + b is the same sum whoever adds — bit for bit; four: the
    # mock adds in rank order, gloo in its own — the last bits of a sum of four may differ)
    out2 = tmp_path / "gloo"; out2.mkdir()
    ref = run_two_ranks("nd:C4:4", out2, world)
    if world == 2:
        assert ref[0]["costs"] == a["costs"] and ref[0]["poses_sum"] == a["poses_sum"] and ref[0]["points_sum"] == a["points_sum"]
    else:
        assert max(abs(x - y) / y for x, y in zip(a["costs"], ref[0]["costs"])) <= 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["nd:S300:6:priors:intr:mock", "nd:S300:6:priors:freeratio:mock"])
def test_stream_ordered_transport_with_motion_priors_and_shared_intrinsics(tmp_path, mode):
    res = run_two_ranks(mode, tmp_path, 3, env_extra={"RSBA_RCCL_LIB": MOCK_RCCL})
    check_nd(res, 3)


@pytest.mark.gpu
def test_a_rank_that_never_arrives_is_reported_not_waited_for_forever(tmp_path):
    """The mock transport's device-side waits are bounded: a collective only ONE rank enters gives up after the timeout and the
    communicator reports it when it is destroyed — a wedged exchange must never hang the GPU.  (Also the transport's own arithmetic:
    a sum and a max over two ranks.)"""
    a, b = run_two_ranks("mock_timeout", tmp_path, 2, env_extra={"RSBA_RCCL_LIB": MOCK_RCCL, "RSBA_MOCK_RCCL_TIMEOUT_S": "0.5"})
    assert a["sum_ok"] and b["sum_ok"] and a["max_ok"] and b["max_ok"]
    assert a["enqueue_s"] < 0.25 and 0.4 <= a["gave_up_after_s"] < 10.0
    assert a["destroy"] == 2 and b["destroy"] == 0      # ncclSystemError on the rank that waited; nothing to report on the one that never called
