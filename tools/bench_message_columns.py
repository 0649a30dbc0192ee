#!/usr/bin/env python3
"""Policy-in-the-loop rollout step with the optional MESSAGE columns on (config.py message_properties thermal / hvac, utils.py:858-866:
8 / 7 / 11 floats per sender instead of 4 - F = 91 / 81 / 121 with ten senders): these shapes are beyond the one-kernel observe -> act
forms (F <= 64) and take observation rows + actor.  One JSON line per shape.   python tools/bench_message_columns.py [--shape 4096x1024]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bench_observe_ext import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="4096x1024")
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    import mdr_amd
    from mdr_amd.rollout import ActorMLP, collect_ppo_rollout
    E, N = (int(x) for x in args.shape.split("x"))
    for name, flags in (("default", ()), ("thermal", ("thermal",)), ("hvac", ("hvac",)), ("thermal + hvac", ("thermal", "hvac"))):
        cfg = mdr_amd.default_config()
        env_prop = cfg["default_env_prop"]
        env_prop["cluster_prop"]["nb_agents"] = N
        env_prop["power_grid_prop"]["base_power_mode"] = "constant"
        for f in flags:
            env_prop["message_properties"][f] = True
        env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=1)
        env.reset(episode=0)
        env.rollout(5)
        torch.manual_seed(0)
        actor = ActorMLP(env.obs_vector_length()).to("cuda:0")
        row = {"shape": args.shape, "message_columns": name, "F": env.obs_vector_length()}
        for prec in ("fp32", "bf16x3"):
            try:
                collect_ppo_rollout(env, actor, 2, store_states=False, policy_precision=prec)
                us = timeit(lambda: collect_ppo_rollout(env, actor, args.steps, store_states=False, policy_precision=prec)) / args.steps
                row["rollout_step_%s_us" % prec] = round(us, 1)
            except Exception as err:  # noqa: BLE001
                row["rollout_step_%s_error" % prec] = str(err)[:160]
        rows = timeit(lambda: env.obs_vector("rows"))
        row["obs_rows_us"] = round(rows, 1)
        print(json.dumps(row), flush=True)
        del env
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
