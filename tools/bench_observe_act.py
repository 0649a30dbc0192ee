#!/usr/bin/env python3
"""Observe -> act (one kernel) against observation rows + actor (two kernels): per-stage and per-rollout-step times, HIP events.

    python tools/bench_observe_act.py [--shapes 4096x1024,32768x64] [--steps 20]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def timeit(fn, iters, warm=3):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="4096x1024,32768x64,1024x64")
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    import mdr_amd
    from mdr_amd.policy import BF16X3, FEATURES_OBSERVE, FRAG16, FRAG16T, FusedActor
    from mdr_amd.rollout import ActorMLP, collect_ppo_rollout
    for shape in args.shapes.split(","):
        E, N = (int(x) for x in shape.split("x"))
        cfg = mdr_amd.default_config()
        cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = N
        cfg["default_env_prop"]["power_grid_prop"]["base_power_mode"] = "constant"
        env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=1)
        env.reset(episode=0)
        env.rollout(5)
        torch.manual_seed(0)
        actor = ActorMLP(51).to("cuda:0")
        out = {"shape": shape, "agents": E * N}
        rows = env.obs_vector("rows").view(E * N, 51)
        out["obs_rows_us"] = round(timeit(lambda: env.obs_vector("rows"), args.steps), 1)
        for name, layout in (("fp32_16x16only", FRAG16), ("fp32", FRAG16T), ("bf16x3", BF16X3)):
            by_rows = FusedActor.from_module(actor, layout=layout)
            by_state = FusedActor.from_module(actor, layout=layout, feature_order=FEATURES_OBSERVE)
            out["actor_on_rows_%s_us" % name] = round(timeit(lambda: by_rows.sample(rows, 1, 2), args.steps), 1)
            out["observe_act_%s_us" % name] = round(timeit(lambda: by_state.sample_env(env, 1, 2), args.steps), 1)
            kept = torch.empty((E * N, 51), device="cuda:0")
            out["observe_act_store_rows_%s_us" % name] = round(timeit(lambda: by_state.sample_env(env, 1, 2, rows_out=kept), args.steps), 1)
            del kept
            if layout == FRAG16:      # the rollout picks FRAG16T for the reference's [100, 100] by itself: kernel comparison only
                continue
            for key, observe, keep, planes in (("rows", False, False, True), ("observe_act", True, False, False), ("observe_act_with_planes", True, False, True),
                                               ("rows_states_kept", False, True, True), ("observe_act_states_kept", True, True, False)):
                collect_ppo_rollout(env, actor, 3, store_states=keep, policy_precision=name, observe_act=observe, obs_planes=planes)
                us = [timeit(lambda: collect_ppo_rollout(env, actor, args.steps, store_states=keep, policy_precision=name, observe_act=observe,
                                                         obs_planes=planes), 1, warm=0) / args.steps
                      for _ in range(3)]
                out["rollout_step_%s_%s_us" % (key, name)] = round(min(us), 1)
                torch.cuda.empty_cache()
        print(json.dumps(out), flush=True)
        del env
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
