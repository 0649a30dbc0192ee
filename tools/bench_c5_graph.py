#!/usr/bin/env python3
"""Sharded (C5) step on ONE rank through RCCL: eager (host-issued begin / all-gather / end) against the captured form, for the
per-rank shares of 1, 2, 4, 8 ranks.  One JSON line per share."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import bench
import mdr_amd

os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29631"), RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
K = 640
for share in (1_000_000, 500_000, 250_000, 125_000):
    cfg = bench.c3_config(mdr_amd)
    cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = share
    row = {"houses_on_rank": share}
    for name, graph, unroll in (("eager", False, 0), ("graph_unroll1", True, 1), ("graph_unroll16", True, 16), ("graph_unroll32", True, 32)):
        env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=1, device=dev, seed=2024, house_shard=(0, share), exchange_always=True,
                                               table_steps=64, graph_mode=graph)
        if unroll:
            env.SHARD_GRAPH_UNROLLS = (unroll,) if unroll > 1 else ()
        env.reset(episode=0)
        env.rollout(128)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        env.rollout(K)
        torch.cuda.synchronize()
        row[name + "_us"] = round((time.perf_counter() - t0) / K * 1e6, 2)
        row.setdefault("checksum", float(env.t["Ta"].double().sum()))
        assert row["checksum"] == float(env.t["Ta"].double().sum())
        del env
    print(json.dumps(row), flush=True)
dist.destroy_process_group()
