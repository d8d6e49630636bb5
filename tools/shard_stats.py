"""Plan figures of the sharded factorisation on N ranks sharing GPU 0 (gloo hook): what travels, how long the two dependency chains are.
usage: python -m torch.distributed.run --nproc-per-node N ... tools/shard_stats.py C4"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from rsba_amd import capi
from rsba_amd.distributed import attach
from rsba_amd.scene import make_config
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
full = make_config(cfg).problem
owner, ntop = capi.partition_points(full, world)
shard = full.shard(rank, world, owner)
torch.cuda.set_device(0)
with capi.DeviceProblem(shard, device=0) as dp:
    attach(dp)
    st = dp.plan_stats()
keys = ("tiles", "levels", "tasks", "factor_tiles", "sharded_factorisation", "exchange_doubles", "separator_tiles", "separator_factor_tiles", "local_tasks", "separator_tasks", "local_levels", "separator_levels", "schur_entries", "schur_chunks")
rows = [None] * world
dist.all_gather_object(rows, {k: st[k] for k in keys} | {"observations": int(shard.num_observations)})
if rank == 0:
    print(json.dumps({"config": cfg, "world": world, "ranks": rows}))
dist.barrier(); dist.destroy_process_group()
