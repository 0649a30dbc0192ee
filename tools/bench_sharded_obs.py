#!/usr/bin/env python3
"""obs_vector over sharded houses (message records + halo + ext kernel) against the unsharded kernels, 1 env x 1,000,000 houses."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mdr_amd
from mdr_amd.sharding import LocalShardGroup


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


cfg = mdr_amd.default_config()
cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = 1_000_000
cfg["default_env_prop"]["power_grid_prop"]["base_power_mode"] = "constant"
whole = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=1, seed=1)
whole.reset(episode=0)
out = {"unsharded_rows_us": round(timeit(lambda: whole.obs_vector("rows")), 1), "unsharded_planes_us": round(timeit(lambda: whole.obs_vector("planes")), 1)}
del whole
torch.cuda.empty_cache()
group = LocalShardGroup(cfg, nb_envs=1, nb_shards=8, devices=("cuda:0",), seed=1)
group.reset(episode=0)
shard = group.shards[3]
out["one_shard_125k_messages_us"] = round(timeit(lambda: shard._obs_messages()), 1)
padded = [e._obs_messages() for e in group.shards]
gathered = torch.stack(padded)
out["one_shard_125k_ext_rows_us"] = round(timeit(lambda: shard._obs_from_gathered("rows", gathered)), 1)
out["one_shard_125k_ext_planes_us"] = round(timeit(lambda: shard._obs_from_gathered("planes", gathered)), 1)
out["eight_shards_in_one_process_rows_us"] = round(timeit(lambda: group.obs_vector("rows"), iters=5, warm=1), 1)
print(json.dumps(out))
