"""Single-environment adapter with the reference's gym-style dict surface.

``MADemandResponseEnv(config, test=False)`` / ``reset() -> obs_dict`` /
``step(action_dict) -> (obs_dict, rewards_dict, dones_dict, info_dict)`` with the same keys, key order and
value types as env/MA_DemandResponse.py:73-210 and :904-1003 of the reference, so that train_*.py loops,
``utils.normStateDict`` and the controllers in agents/ consume it unchanged.  The arithmetic is the batched
HIP path with E = 1; this class only converts between dicts and device tensors (one device->host copy per
step), which is why large-scale users should talk to ``BatchedDemandResponseEnv`` directly.

Seeding: the reference's env draws from Python's global ``random`` stream (main.py:33 seeds it).  Here the
episode seed is drawn from that same stream (``random.getrandbits``) unless ``seed=`` is given, so
``random.seed(env_seed)`` keeps making runs reproducible; the device then expands it with Philox.
"""
from __future__ import annotations

import copy
import datetime as _dt
import random
import warnings
from typing import Dict, List

import numpy as np
import torch

from .batched_env import BatchedDemandResponseEnv
from .comm import build_comm_links, nb_comm
from .config import from_epoch_seconds


class _HvacView:
    def __init__(self, env, i):
        self._env, self._i = env, i

    @property
    def turned_on(self):
        return bool(self._env._host["on"][self._i])

    @property
    def lockout(self):
        return bool(self._env._host["lock"][self._i])

    @property
    def seconds_since_off(self):
        return int(self._env._host["sso"][self._i])

    @property
    def lockout_duration(self):
        return int(self._env._static["lockout"][self._i])

    @property
    def cooling_capacity(self):
        return float(self._env._static["capacity"][self._i])

    @property
    def COP(self):
        return float(self._env._static["COP"][self._i])

    @property
    def latent_cooling_fraction(self):
        return float(self._env._static["latent"][self._i])

    @property
    def max_consumption(self):
        return self.cooling_capacity / self.COP

    def power_consumption(self):
        return self.max_consumption if self.turned_on else 0


class _HouseView:
    """Read-only stand-in for SingleHouse: the attributes callers and tests look at."""

    def __init__(self, env, i):
        self._env, self.id = env, i
        self.hvac = _HvacView(env, i)

    current_temp = property(lambda self: float(self._env._host["Ta"][self.id]))
    current_mass_temp = property(lambda self: float(self._env._host["Tm"][self.id]))
    target_temp = property(lambda self: float(self._env._static["target"][self.id]))
    deadband = property(lambda self: float(self._env._static["deadband"][self.id]))
    Ua = property(lambda self: float(self._env._static["Ua"][self.id]))
    Cm = property(lambda self: float(self._env._static["Cm"][self.id]))
    Ca = property(lambda self: float(self._env._static["Ca"][self.id]))
    Hm = property(lambda self: float(self._env._static["Hm"][self.id]))
    current_solar_gain = property(lambda self: float(self._env._host["solar"]))


class _ClusterView:
    def __init__(self, env):
        self._env = env
        self.houses = {i: _HouseView(env, i) for i in env.agent_ids}
        self.agent_communicators = env._links
        self.nb_agents = env.nb_agents

    current_OD_temp = property(lambda self: float(self._env._host["od"]))
    cluster_hvac_power = property(lambda self: float(self._env._host["P"]))
    max_power = property(lambda self: float(self._env._host["max_power"]))


class _GridView:
    def __init__(self, env):
        self._env = env
        self.cumulated_abs_noise = 0
        self.nb_steps = 0

    current_signal = property(lambda self: float(self._env._host["S"]))
    max_power = property(lambda self: float(self._env._host["max_power"]))
    artificial_ratio = property(lambda self: float(self._env._host["ratio"]))
    base_power = property(lambda self: (self._env._batched.t["base_power"][0].item() if self._env._batched.spec.base_power_mode == 1
                                        else self._env._batched.spec.avg_power_per_hvac * self._env.nb_agents))


class MADemandResponseEnv:
    """Multi agent demand response environment (drop-in for env/MA_DemandResponse.py:37)."""

    def __init__(self, config, test=False, device=None, seed=None, table_steps=64, interp_grid=None):
        self.test = test
        self.config = config
        self.default_env_prop = config["default_env_prop"]
        self.default_house_prop = config["default_house_prop"]
        self.default_hvac_prop = config["default_hvac_prop"]
        self.noise_house_prop = config["noise_house_prop_test" if test else "noise_house_prop"]
        self.noise_hvac_prop = config["noise_hvac_prop_test" if test else "noise_hvac_prop"]
        self._fixed_seed = seed
        self._batched = BatchedDemandResponseEnv(config, nb_envs=1, device=device, seed=0, test=test,
                                                 table_steps=table_steps, interp_grid=interp_grid)
        self.nb_agents = self._batched.nb_agents
        self.agent_ids = list(range(self.nb_agents))
        self.time_step = _dt.timedelta(seconds=self._batched.spec.time_step)
        self._episode = 0
        self.build_environment()

    # ------------------------------------------------------------------ episode
    def build_environment(self):
        """env 98-133: new houses, new start date, new grid signal."""
        seed = self._fixed_seed if self._fixed_seed is not None else random.getrandbits(63)
        self._batched.reset(seed=seed, episode=self._episode)
        self._episode += 1
        self._episode_started()

    def load_episode(self, params, od_table=None, seed=0):
        """Replay hook (no counterpart in the reference): start an episode from given raw per-house / per-env
        parameters and, optionally, a recorded outdoor-temperature sequence instead of sampling them.
        See BatchedDemandResponseEnv.load_episode for the array names.  Returns the reset observation dict."""
        self._batched.load_episode(params, od_table=od_table, seed=seed, episode=0)
        self._episode_started()
        return self._make_obs_dict()

    def _episode_started(self):
        b = self._batched
        st = {k: b.t[k][0].cpu().numpy() for k in ("Ua", "Cm", "Ca", "Hm", "capacity", "COP", "latent", "deadband")}
        st["lockout"] = b.t["lockout"][0].cpu().numpy()
        st["target"] = b.target_temp()[0].cpu().numpy()
        self._static = st
        self.start_datetime = from_epoch_seconds(int(b.t["t0"][0].item()))
        self.datetime = self.start_datetime
        self._links = self._build_agent_comm_links()
        self.env_properties = copy.deepcopy(self.default_env_prop)
        self.env_properties.update(agent_ids=self.agent_ids, nb_hvac=self.nb_agents, start_datetime=self.start_datetime)
        self._pull()
        self.cluster = _ClusterView(self)
        self.power_grid = _GridView(self)

    def reset(self):
        """env 135-172."""
        self.build_environment()
        return self._make_obs_dict()

    def step(self, action_dict):
        """env 174-210."""
        cmd = np.zeros((1, self.nb_agents), dtype=np.uint8)
        for i in self.agent_ids:
            if i in action_dict:
                cmd[0, i] = 1 if action_dict[i] else 0
            else:  # env 1026-1032
                warnings.warn("HVAC in house {} did not receive any command.".format(i))
        self.datetime += self.time_step
        self._batched.step(torch.from_numpy(cmd).to(self._batched.device))
        self._pull()
        reward = self._batched.t["reward"][0].cpu().numpy()
        obs_dict = self._make_obs_dict()
        rewards_dict = {i: float(reward[i]) for i in self.agent_ids}
        dones_dict = {i: False for i in self.agent_ids}   # env 375-390
        info_dict = {"cluster_hvac_power": float(self._host["P"])}
        if "perlin" in self._batched.spec.signal_mode_name:
            self.power_grid.nb_steps += 1
        return obs_dict, rewards_dict, dones_dict, info_dict

    # ------------------------------------------------------------------ device -> host
    def _pull(self):
        b = self._batched
        flags = b.t["flags"][0].cpu().numpy()
        self._host = {
            "Ta": b.house_temp()[0].cpu().numpy(), "Tm": b.house_mass_temp()[0].cpu().numpy(),
            "sso": b.t["sso"][0].cpu().numpy(), "on": (flags & 1).astype(bool), "lock": (flags & 2).astype(bool),
            "od": b.od_temp()[0].item(), "S": b.reg_signal()[0].item(),
            # SingleHouse.current_solar_gain is 0 until the first update_temperature (env 573)
            "solar": b.solar_gain()[0].item() if b.steps_taken > 0 else 0,
            "P": b.t["P"][0].item(), "max_power": b.t["max_power"][0].item(), "ratio": b.t["ratio"][0].item(),
        }

    # ------------------------------------------------------------------ communication links (env 806-902)
    def _build_agent_comm_links(self) -> Dict[int, List[int]]:
        links = build_comm_links(self.default_env_prop["cluster_prop"])
        return {} if links is None else links

    def _neighbours(self, i):
        cp = self.default_env_prop["cluster_prop"]
        if cp["agents_comm_mode"] == "random_sample":
            return random.sample([j for j in self.agent_ids if j != i], k=nb_comm(cp))
        return self._links[i]

    def _message(self, j, empty):
        """SingleHouse.message (env 624-662)."""
        mp = self.default_env_prop["message_properties"]
        h, s = self._host, self._static
        if empty:
            vals = [0, 0, 0, 0, 0]
        else:
            pmax = float(s["capacity"][j]) / float(s["COP"][j])
            vals = [float(h["Ta"][j]) - float(s["target"][j]), int(h["sso"][j]),
                    pmax if h["on"][j] else 0, pmax, int(s["lockout"][j])]
        m = dict(zip(("current_temp_diff_to_target", "hvac_seconds_since_off", "hvac_curr_consumption",
                      "hvac_max_consumption", "hvac_lockout_duration"), vals))
        if mp["thermal"]:
            for k in ("Ua", "Cm", "Ca", "Hm"):
                m["house_" + k] = 0 if empty else float(s[k][j])
        if mp["hvac"]:
            m["hvac_COP"] = 0 if empty else float(s["COP"][j])
            m["hvac_cooling_capacity"] = 0 if empty else float(s["capacity"][j])
            m["hvac_latent_cooling_fraction"] = 0 if empty else float(s["latent"][j])
        return m

    # ------------------------------------------------------------------ observation dict (env 904-1003, 212-232)
    def _make_obs_dict(self):
        h, s = self._host, self._static
        defect = self.default_env_prop["cluster_prop"]["comm_defect_prob"]
        obs = {}
        for i in self.agent_ids:
            d = {
                "OD_temp": h["od"],
                "datetime": self.datetime,
                "house_temp": float(h["Ta"][i]),
                "house_mass_temp": float(h["Tm"][i]),
                "hvac_turned_on": bool(h["on"][i]),
                "hvac_seconds_since_off": int(h["sso"][i]),
                "hvac_lockout": bool(h["lock"][i]),
                "house_target_temp": float(s["target"][i]),
                "house_deadband": float(s["deadband"][i]),
                "house_Ua": float(s["Ua"][i]),
                "house_Cm": float(s["Cm"][i]),
                "house_Ca": float(s["Ca"][i]),
                "house_Hm": float(s["Hm"][i]),
                "house_solar_gain": h["solar"],
                "hvac_COP": float(s["COP"][i]),
                "hvac_cooling_capacity": float(s["capacity"][i]),
                "hvac_latent_cooling_fraction": float(s["latent"][i]),
                "hvac_lockout_duration": int(s["lockout"][i]),
            }
            d["message"] = [self._message(j, not (np.random.rand() > defect)) for j in self._neighbours(i)]
            d["reg_signal"] = h["S"]
            d["cluster_hvac_power"] = h["P"]
            obs[i] = d
        return obs

    # ------------------------------------------------------------------ copy.deepcopy(env) (utils.py:890-1008)
    def __deepcopy__(self, memo):
        other = MADemandResponseEnv.__new__(MADemandResponseEnv)
        for k, v in self.__dict__.items():
            if k in ("cluster", "power_grid"):
                continue
            setattr(other, k, copy.deepcopy(v, memo))
        other.cluster = _ClusterView(other)
        other.power_grid = _GridView(other)
        other.power_grid.nb_steps = self.power_grid.nb_steps
        return other
