#!/usr/bin/env python3
"""rollout.deploy_policy (main-deploy.py:99-152 with a learned agent) per step: the network handed over (observation and policy in one
kernel) against a FusedActor on observation rows.  One JSON line per shape.   python tools/bench_deploy_policy.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
import mdr_amd  # noqa: E402
from mdr_amd.policy import FusedActor  # noqa: E402
from mdr_amd.rollout import ActorMLP, deploy_policy  # noqa: E402


def main():
    for E, N in ((4096, 1024), (83886, 50), (209715, 20)):
        cfg = bench.c3_config(mdr_amd)
        cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = N
        env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=2024, table_steps=64)
        env.reset(episode=0)
        torch.manual_seed(0)
        actor = ActorMLP(env.obs_vector_length()).cuda()
        row = {"shape": "%dx%d" % (E, N)}
        for key, pol, kw in (("rows_fp32_us", FusedActor.from_module(actor), {}), ("one_kernel_fp32_us", actor, {}),
                             ("rows_bf16x3_us", FusedActor.from_module(actor, layout=2), {}), ("one_kernel_bf16x3_us", actor, {"policy_precision": "bf16x3"})):
            deploy_policy(env, pol, 4, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            deploy_policy(env, pol, 20, **kw)
            e1.record()
            torch.cuda.synchronize()
            row[key] = round(e0.elapsed_time(e1) / 20 * 1e3, 1)
        print(json.dumps(row), flush=True)
        del env
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
