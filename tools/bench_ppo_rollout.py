#!/usr/bin/env python3
"""Where a policy-in-the-loop rollout step spends its time: obs rows -> actor MLP -> sample -> env step.

    python tools/bench_ppo_rollout.py [--shapes 1024x50,32768x50,4096x1024] [--steps 20]

Prints one JSON line per shape with the per-step time of each stage (HIP events on torch's current stream)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="1024x50,32768x50,4096x1024")
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    import mdr_amd
    from mdr_amd.rollout import ActorMLP, collect_ppo_rollout
    for shape in args.shapes.split(","):
        E, N = (int(x) for x in shape.split("x"))
        cfg = mdr_amd.default_config()
        cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = N
        cfg["default_env_prop"]["power_grid_prop"]["base_power_mode"] = "constant"
        env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=1)
        env.reset(episode=0)
        F_len = env.obs_vector_length()
        torch.manual_seed(0)
        actor = ActorMLP(F_len).to("cuda:0")
        stages = {"obs_vector": lambda: env.obs_vector("rows")}
        obs = env.obs_vector("rows").view(E * N, F_len)
        probs = actor(obs)
        a = torch.multinomial(probs, 1).squeeze(1)
        act_u8 = a.to(torch.uint8).view(E, N)
        stages["actor_mlp"] = lambda: actor(obs)
        from mdr_amd.policy import FusedActor
        fused = FusedActor.from_module(actor)
        stages["fused_actor_sample"] = lambda: fused.sample(obs, 1, 2)
        planes = env.obs_vector("planes")
        stages["obs_vector_planes"] = lambda: env.obs_vector("planes")
        stages["fused_actor_sample_planes"] = lambda: fused.sample(planes, 1, 2)
        fusedbf = FusedActor.from_module(actor, layout=2)
        stages["fused_actor_sample_bf16x3"] = lambda: fusedbf.sample(obs, 1, 2)
        fused32 = FusedActor.from_module(actor, layout=0)
        stages["fused_actor_sample_frag32"] = lambda: fused32.sample(obs, 1, 2)
        stages["sample"] = lambda: (torch.multinomial(probs, 1).squeeze(1), probs.gather(1, a[:, None]))
        stages["to_uint8"] = lambda: a.to(torch.uint8)
        stages["env_step"] = lambda: env.step(act_u8)
        out = {"shape": shape, "agents": E * N, "F": F_len}
        with torch.no_grad():
            for name, fn in stages.items():
                for _ in range(3):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(args.steps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                out[name + "_us"] = round(e0.elapsed_time(e1) / args.steps * 1e3, 1)
            for key, use_fused, prec in (("torch", False, "fp32"), ("fused", True, "fp32"), ("fused_bf16x3", True, "bf16x3")):
                for _ in range(2):
                    collect_ppo_rollout(env, actor, 4, store_states=False, fused=use_fused, policy_precision=prec)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                collect_ppo_rollout(env, actor, args.steps, store_states=False, fused=use_fused, policy_precision=prec)
                e1.record()
                torch.cuda.synchronize()
                out["rollout_step_us_" + key] = round(e0.elapsed_time(e1) / args.steps * 1e3, 1)
                out["agent_steps_per_s_" + key] = round(E * N / (out["rollout_step_us_" + key] * 1e-6))
            flops = 2.0 * (F_len * 100 + 100 * 100 + 100 * 2) * E * N
            out["fused_actor_TFLOPs"] = round(flops / (out["fused_actor_sample_us"] * 1e-6) / 1e12, 1)
            out["fused_actor_bf16x3_TFLOPs_fp32_equivalent"] = round(flops / (out["fused_actor_sample_bf16x3_us"] * 1e-6) / 1e12, 1)
        print(json.dumps(out), flush=True)
        del env, actor, obs, probs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
