#!/usr/bin/env python3
"""Persistent rollout (mdr_env_rollout_persistent) on ONE rank for the per-rank shares of 1, 2, 4, 8 ranks of BASELINE config 5
(1 env x 1,000,000 houses): us per step with and without the accumulators, beside the split path's one-launch-per-step rollout
of the same env.  One JSON line per share."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import mdr_amd

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
K = int(os.environ.get("K", "1280"))
shares = [int(s) for s in os.environ.get("SHARES", "1000000,500000,250000,125000").split(",")]
for share in shares:
    cfg = bench.c3_config(mdr_amd)
    cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = share
    if os.environ.get("PENALTY"):
        cfg["default_env_prop"]["reward_prop"]["temp_penalty_mode"] = os.environ["PENALTY"]
    row = {"houses_on_rank": share, "steps": K}
    for name, fn in (("split_rollout", lambda e: e.rollout(K)),
                     ("persistent", lambda e: e.rollout_persistent(K, check=False)),
                     ("persistent_no_acc", lambda e: e.rollout_persistent(K, accumulate=False, check=False))):
        env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=1, device=dev, seed=2024, table_steps=64)
        env.reset(episode=0)
        fn(env)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        fn(env)
        ev1.record()
        torch.cuda.synchronize()
        row[name + "_us"] = round((time.perf_counter() - t0) / K * 1e6, 3)
        row[name + "_event_us"] = round(ev0.elapsed_time(ev1) / K * 1e3, 3)
        assert env.persist_status() == 0
        row.setdefault("checksum", float(env.t["Ta"].double().sum()))
        assert row["checksum"] == float(env.t["Ta"].double().sum()), name
        del env
    print(json.dumps(row), flush=True)
