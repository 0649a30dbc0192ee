#!/usr/bin/env python3
"""Unsharded envs on the split path (more than 4096 houses): microseconds per step of env.rollout(), HIP events."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import mdr_amd

for E, N in ((1, 1_000_000), (1, 125_000), (4, 250_000), (64, 8192), (16, 65536), (1, 1_000_001)):
    cfg = bench.c3_config(mdr_amd)
    cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = N
    env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=2024, table_steps=64)
    env.reset(episode=0)
    env.rollout(70)
    best = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        env.rollout(640)
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / 640 * 1e3)
    print(json.dumps({"envs": E, "houses": N, "us_per_step": round(min(best), 2), "GBps_107B": round(E * N * 107 / min(best) / 1e3, 1),
                      "checksum_Ta": float(env.t["Ta"].double().sum())}), flush=True)
    del env
    torch.cuda.empty_cache()
