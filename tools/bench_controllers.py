#!/usr/bin/env python3
"""Closed loops under the reference's rule-based agents (agents/bangbang_controllers.py, agents/greedy_myopic_controller.py; main-deploy.py
drives them one env at a time): microseconds per step of rollout.deploy_controller at the C3 batch and at the reference's own cluster
sizes.  One JSON line per (shape, controller).   python tools/bench_controllers.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
import mdr_amd  # noqa: E402
from mdr_amd.rollout import deploy_controller  # noqa: E402


def main():
    for E, N in ((4096, 1024), (83886, 50), (209715, 20)):
        for kind, steps in (("bangbang", 256), ("deadband", 256), ("always_on", 256), ("greedy_myopic", 32)):
            cfg = bench.c3_config(mdr_amd)
            cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = N
            cfg["default_house_prop"]["deadband"] = 1.0
            env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=2024, table_steps=64)
            env.reset(episode=0)
            deploy_controller(env, kind, 8)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            deploy_controller(env, kind, steps)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / steps * 1e3
            print(json.dumps({"shape": "%dx%d" % (E, N), "controller": kind, "steps": steps, "us_per_step": round(us, 2),
                              "house_steps_per_s": round(E * N / us * 1e6)}), flush=True)
            del env
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
