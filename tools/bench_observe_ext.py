#!/usr/bin/env python3
"""Policy-in-the-loop rollout step (observe -> act kernel + env step, states not kept) for observation shapes beyond the reference's
default: optional state columns (F = 58: thermal + hvac, utils.py:774-830), link defects (10 %, env 988-1002), fewer neighbours -
against the default 51-feature shape, and against the two-kernel path (observation rows + actor) the same shapes took before.
One JSON line per shape.   python tools/bench_observe_ext.py [--shape 4096x1024] [--steps 20]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def timeit(fn, reps=3):
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1)
        best = t if best is None else min(best, t)
    return best * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="4096x1024")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--cases", default="", help="comma-separated case indices (default: all); the default shape (0) is the reference time")
    ap.add_argument("--precisions", default="fp32,bf16x3")
    ap.add_argument("--observe-only", action="store_true", help="skip the rows + actor comparison (profiling runs)")
    args = ap.parse_args()
    import mdr_amd
    from mdr_amd.rollout import ActorMLP, collect_ppo_rollout
    E, N = (int(x) for x in args.shape.split("x"))
    cases = (("default F=51", (), 10, 0.0), ("thermal + hvac F=58", ("thermal", "hvac"), 10, 0.0), ("10 % link defects F=51", (), 10, 0.1),
             ("every state column F=63", ("hour", "day", "solar_gain", "thermal", "hvac"), 10, 0.0), ("6 neighbours F=35", (), 6, 0.0),
             ("thermal + hvac, 10 % defects F=58", ("thermal", "hvac"), 10, 0.1),
             # senders by a link table / re-drawn every step: gathered from the message records (mdr_env_actor_sample_links)
             ("closed_groups F=51", (), 10, 0.0, "closed_groups"), ("random_fixed F=51", (), 10, 0.0, "random_fixed"),
             ("random_sample F=51", (), 10, 0.0, "random_sample"))
    base = {}
    if args.cases:
        cases = tuple(cases[int(i)] for i in args.cases.split(","))
    for name, flags, c, defects, *mode in cases:
        cfg = mdr_amd.default_config()
        env_prop = cfg["default_env_prop"]
        env_prop["cluster_prop"]["nb_agents"] = N
        env_prop["cluster_prop"]["agents_comm_mode"] = mode[0] if mode else "neighbours"
        env_prop["cluster_prop"]["nb_agents_comm"] = c
        env_prop["cluster_prop"]["comm_defect_prob"] = defects
        env_prop["power_grid_prop"]["base_power_mode"] = "constant"
        cfg["default_house_prop"]["solar_gain_bool"] = True
        for f in flags:
            env_prop["state_properties"][f] = True
        env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=1)
        env.reset(episode=0)
        env.rollout(5)
        torch.manual_seed(0)
        actor = ActorMLP(env.obs_vector_length()).to("cuda:0")
        row = {"shape": args.shape, "observation": name, "F": env.obs_vector_length()}
        for prec in args.precisions.split(","):
            for key, observe in ((("observe_act", True),) if args.observe_only else (("observe_act", True), ("rows_then_actor", False))):
                collect_ppo_rollout(env, actor, 3, store_states=False, policy_precision=prec, observe_act=observe)
                us = timeit(lambda: collect_ppo_rollout(env, actor, args.steps, store_states=False, policy_precision=prec, observe_act=observe)) / args.steps
                row["%s_%s_us" % (key, prec)] = round(us, 1)
            if name.startswith("default"):
                base[prec] = row["observe_act_%s_us" % prec]
            if prec in base:
                row["vs_default_%s" % prec] = round(row["observe_act_%s_us" % prec] / base[prec], 3)
        print(json.dumps(row), flush=True)
        del env
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
