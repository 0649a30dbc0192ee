#!/usr/bin/env python3
"""GreedyMyopic action kernel alone (mdr_env_greedy_myopic_actions; agents/greedy_myopic_controller.py:6-50): microseconds per call at
the C3 batch and at the reference's own cluster sizes, state after 64 closed-loop steps.  One JSON line per shape.
    python tools/bench_greedy.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
import mdr_amd  # noqa: E402


def main():
    for E, N in ((4096, 1024), (8192, 512), (16384, 256), (41943, 100), (83886, 50), (209715, 20), (419430, 10)):
        cfg = bench.c3_config(mdr_amd)
        cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = N
        env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=2024, table_steps=64)
        env.reset(episode=0)
        for _ in range(64):
            env.step_greedy_myopic()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for _ in range(reps):
            env.greedy_myopic_actions()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        print(json.dumps({"shape": "%dx%d" % (E, N), "kernel": "mdr_env_greedy_myopic_actions", "us_per_call": round(us, 2),
                          "houses_per_s": round(E * N / us * 1e6), "fraction_on": round(env.t["actions"].float().mean().item(), 4)}), flush=True)
        del env
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
