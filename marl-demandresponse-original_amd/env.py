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
from .comm import nb_comm
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
        # PowerGrid.step of the episode's first time index already ran inside build_environment (env 133): in the perlin
        # families it has added its |base * amplitude * perlin| and counted one step (env 1301-1302)
        self.cumulated_abs_noise = 0
        self.nb_steps = 0
        if "perlin" in env._batched.spec.signal_mode_name:
            self.cumulated_abs_noise = env._abs_noise_now()
            self.nb_steps = 1

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
        obs_dict = self._make_obs_dict()
        rewards_dict = dict(zip(self.agent_ids, self._host["reward"].tolist()))
        dones_dict = dict.fromkeys(self.agent_ids, False)   # env 375-390
        info_dict = {"cluster_hvac_power": float(self._host["P"])}
        if "perlin" in self._batched.spec.signal_mode_name:      # env 1301-1302
            self.power_grid.cumulated_abs_noise += self._abs_noise_now()
            self.power_grid.nb_steps += 1
        return obs_dict, rewards_dict, dones_dict, info_dict

    def _abs_noise_now(self) -> float:
        """|base_power * amplitude * perlin| of the current time index (what PowerGrid.step adds to cumulated_abs_noise)."""
        return float(self._host["abs_noise"])

    def norm_states(self) -> np.ndarray:
        """``utils.normStateDict(obs_dict[i], config)`` (utils.py:740-880) for every agent at once, float32 [nb_agents, F],
        computed on the device from the same state the last ``reset`` / ``step`` returned (``mdr_env_obs_vector``).  A trainer
        that calls ``normStateDict`` per agent and step (train_ppo.py:69-72, 87-98: ~30 us per call) can index this instead.
        Link defects and ``random_sample`` senders are the device's Philox draws of this step - the very draws the dict's
        ``message`` lists are built from (``mdr_env_comm_draws``), so both views of a step agree in every mode."""
        return self._batched.obs_vector("rows")[0].cpu().numpy()

    # ------------------------------------------------------------------ device -> host
    def _pull(self):
        """Everything the dicts need, packed on the device into one float64 vector and fetched with ONE copy + sync
        (a dozen separate .cpu() / .item() calls cost ~20 us each and dominated the step at the reference's own sizes)."""
        b = self._batched
        n = self.nb_agents
        k = b.steps_taken
        pack = b.pack_env(0)
        flags = pack[3 * n:4 * n].astype(np.uint8)
        od, S, solar, P, max_power, ratio, abs_noise = pack[5 * n:].tolist()
        self._host = {
            "Ta": pack[0:n], "Tm": pack[n:2 * n], "sso": pack[2 * n:3 * n].astype(np.int64),
            "on": (flags & 1).astype(bool), "lock": (flags & 2).astype(bool), "reward": pack[4 * n:5 * n].astype(np.float32),
            "od": od, "S": S,
            # SingleHouse.current_solar_gain is 0 until the first update_temperature (env 573)
            "solar": solar if k > 0 else 0,
            "P": P, "max_power": max_power, "ratio": ratio, "abs_noise": abs_noise,
        }

    # ------------------------------------------------------------------ communication links (env 806-902)
    def _build_agent_comm_links(self) -> Dict[int, List[int]]:
        """ClusterHouses.agent_communicators: one table per episode, the one the batched env's flat vector gathers through
        ('random_fixed' is re-drawn at every reset from (seed, episode); 'random_sample' has no table: env 846-847)."""
        table = self._batched.comm_links_array()
        if table is None:
            return {}
        rows = table.tolist()
        return {i: rows[i] for i in self.agent_ids}

    _MSG_KEYS = ("current_temp_diff_to_target", "hvac_seconds_since_off", "hvac_curr_consumption",
                 "hvac_max_consumption", "hvac_lockout_duration")

    def _message_keys(self):
        mp = self.default_env_prop["message_properties"]
        keys = list(self._MSG_KEYS)
        if mp["thermal"]:
            keys += ["house_Ua", "house_Cm", "house_Ca", "house_Hm"]
        if mp["hvac"]:
            keys += ["hvac_COP", "hvac_cooling_capacity", "hvac_latent_cooling_fraction"]
        return keys

    def _static_lists(self):
        """Python-float copies of the per-episode constants (bulk .tolist() once per episode, not per step and element)."""
        st = getattr(self, "_static_py", None)
        if st is None or st[0] is not self._static:
            s = self._static
            py = {k: s[k].astype(np.float64).tolist() for k in ("target", "deadband", "Ua", "Cm", "Ca", "Hm", "COP", "capacity", "latent")}
            py["lockout"] = [int(v) for v in s["lockout"].tolist()]
            py["pmax"] = [c / p for c, p in zip(py["capacity"], py["COP"])]
            st = self._static_py = (self._static, py)
        return st[1]

    # ------------------------------------------------------------------ observation dict (env 904-1003, 212-232)
    def _make_obs_dict(self):
        h, py = self._host, self._static_lists()
        cp = self.default_env_prop["cluster_prop"]
        mp = self.default_env_prop["message_properties"]
        defect = cp["comm_defect_prob"]
        Ta, Tm, sso = h["Ta"].tolist(), h["Tm"].tolist(), h["sso"].tolist()
        on, lock = h["on"].tolist(), h["lock"].tolist()
        mkeys = self._message_keys()
        # SingleHouse.message (env 624-662) of every house once; the receivers get their own copy of it below
        msg_vals = []
        for j in self.agent_ids:
            v = [Ta[j] - py["target"][j], sso[j], py["pmax"][j] if on[j] else 0, py["pmax"][j], py["lockout"][j]]
            if mp["thermal"]:
                v += [py["Ua"][j], py["Cm"][j], py["Ca"][j], py["Hm"][j]]
            if mp["hvac"]:
                v += [py["COP"][j], py["capacity"][j], py["latent"][j]]
            msg_vals.append(v)
        empty = dict.fromkeys(mkeys, 0)
        random_links = cp["agents_comm_mode"] == "random_sample"
        od, dtm, solar, S, P = h["od"], self.datetime, h["solar"], h["S"], h["P"]
        # the random part of the gather (env 976-1002) comes from the device: this step's `random.sample` senders and
        # `np.random.rand() > comm_defect_prob` outcomes are the ones norm_states() / obs_vector use
        drawn_senders = drawn_keep = None
        if (random_links or defect > 0) and nb_comm(cp) > 0 and cp["agents_comm_mode"] != "no_message":
            s_dev, k_dev = self._batched.comm_draws()
            drawn_senders, drawn_keep = s_dev[0].cpu().tolist(), k_dev[0].cpu().tolist()
        obs = {}
        for i in self.agent_ids:
            senders = drawn_senders[i] if random_links else self._links[i]
            keep = drawn_keep[i] if drawn_keep is not None else (True,) * len(senders)
            obs[i] = {
                "OD_temp": od,
                "datetime": dtm,
                "house_temp": Ta[i],
                "house_mass_temp": Tm[i],
                "hvac_turned_on": on[i],
                "hvac_seconds_since_off": sso[i],
                "hvac_lockout": lock[i],
                "house_target_temp": py["target"][i],
                "house_deadband": py["deadband"][i],
                "house_Ua": py["Ua"][i],
                "house_Cm": py["Cm"][i],
                "house_Ca": py["Ca"][i],
                "house_Hm": py["Hm"][i],
                "house_solar_gain": solar,
                "hvac_COP": py["COP"][i],
                "hvac_cooling_capacity": py["capacity"][i],
                "hvac_latent_cooling_fraction": py["latent"][i],
                "hvac_lockout_duration": py["lockout"][i],
                "message": [dict(zip(mkeys, msg_vals[j])) if ok else dict(empty) for j, ok in zip(senders, keep)],
                "reg_signal": S,
                "cluster_hvac_power": P,
            }
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
        other.power_grid.cumulated_abs_noise = self.power_grid.cumulated_abs_noise
        return other
