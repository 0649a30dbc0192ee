"""Training metrics on tensors: the running sums of the reference's ``metrics.Metrics`` (metrics.py:13-56) for every env of a batch.

The reference calls ``metrics.update(k, next_obs_dict, rewards_dict, env)`` once per agent and step (train_ppo.py:100-101) and
``metrics.log(t, time_steps_train_log)`` every logging period; each update adds the agent's share (1 / nb_agents, or
1 / nb_agents^2 for the signal terms).  Here one ``update`` per step covers all agents of all envs: per-env reductions of
tensors the step kernel has already produced (house temperature, target, reward, regulation signal, cluster power)."""
from __future__ import annotations

from typing import Dict

import torch

_KEYS = ("cumul_avg_reward", "cumul_temp_offset", "cumul_temp_error", "cumul_signal_offset", "cumul_signal_error",
         "cumul_next_signal_offset", "cumul_next_signal_error")


class BatchedMetrics:
    def __init__(self, env):
        if env.sharded:
            raise ValueError("BatchedMetrics reduces over the houses of an env: use it on unsharded houses")
        self.nb_envs, self.nb_agents, self.device = env.nb_envs, env.nb_agents, env.device
        self.reset()

    def reset(self) -> None:
        """metrics.py:49-56"""
        for k in _KEYS:
            setattr(self, k, torch.zeros(self.nb_envs, dtype=torch.float64, device=self.device))

    def update(self, env, reward: torch.Tensor) -> None:
        """All agents' ``Metrics.update`` calls of one step (metrics.py:23-30): call it right after ``env.step``."""
        n = float(self.nb_agents)
        d = env.house_temp() - env.target_temp()                               # [E, N] float64
        self.cumul_temp_offset += d.sum(dim=1) / n
        self.cumul_temp_error += d.abs().sum(dim=1) / n
        self.cumul_avg_reward += reward.double().sum(dim=1) / n
        s = (env.reg_signal() - env.t["P"]) / n                                # sum over agents of (S - P) / nb_agents^2
        self.cumul_next_signal_offset += s
        self.cumul_next_signal_error += s.abs()
        self.cumul_signal_offset += s                                          # the reference adds the same two terms twice (28-29)
        self.cumul_signal_error += s.abs()

    def log(self, t: int, time_steps_train_log: int) -> Dict[str, object]:
        """metrics.py:32-47: the same keys; every value is a float64 tensor [nb_envs] (one log line per env)."""
        k = float(time_steps_train_log)
        return {"Mean train return": self.cumul_avg_reward / k,
                "Mean temperature offset": self.cumul_temp_offset / k,
                "Mean temperature error": self.cumul_temp_error / k,
                "Mean next signal offset": self.cumul_next_signal_offset / k,
                "Mean next signal error": self.cumul_next_signal_error / k,
                "Mean signal error": self.cumul_signal_error / k,
                "Mean signal offset": self.cumul_signal_offset / k,
                "Training steps": t}
