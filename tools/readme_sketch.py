#!/usr/bin/env python3
"""Runs the usage sketch of README.md (kept in sync by hand) - a quick end-to-end sanity run on a GPU box."""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import mdr_amd  # noqa: E402

warnings.simplefilter("ignore")
cfg = mdr_amd.default_config()
cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = 20
env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=100_000, device="cuda:0", seed=1)
obs7 = env.reset()
state = env.obs_vector("rows")
actions = (torch.rand(env.nb_envs, env.nb_houses, device="cuda") < 0.5).to(torch.uint8)
obs7, reward, done, info = env.step(actions)
metrics = env.rollout_fused(1000, power_trace=True)
single = mdr_amd.MADemandResponseEnv(cfg)
o = single.reset()
o, r, d, i = single.step({k: True for k in o})
torch.cuda.synchronize()
print("ok", tuple(obs7.shape), tuple(state.shape), tuple(reward.shape), tuple(info["cluster_hvac_power"].shape),
      sorted(metrics), len(o), round(r[0], 4), float(env.t["base_power"][0]))

from mdr_amd.rollout import ActorMLP, collect_ppo_rollout, deploy_policy  # noqa: E402
from mdr_amd.policy import FusedActor  # noqa: E402
actor = ActorMLP(env.obs_vector_length()).cuda()
batch = collect_ppo_rollout(env, actor, nb_steps=8, store_states=False)
dm = deploy_policy(env, FusedActor.from_module(actor), nb_steps=10)
torch.cuda.synchronize()
print("policy ok", sorted(batch), tuple(batch["action"].shape), sorted(dm), float(dm["reward_sum"].mean()))
