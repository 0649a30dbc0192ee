#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE in the build container.

    cd /root/repo && python tests/golden/make_golden.py

The reference (/root/reference, read-only) is imported through oracle/ref_harness.py; nothing of it
is copied.  Each fixture is data only: the effective env config (JSON), the per-house / per-env
parameters the reference sampled at reset, the action sequence that was fed, the outdoor temperature
the reference drew each step, and the reference's outputs after every step.

Scenario list follows SURVEY.md section 8c (S1..S6) plus a Perlin-wiring and a phase-offset case.
"""
from __future__ import annotations

import copy
import json
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")

from oracle import ref_harness  # noqa: E402
from oracle import mdr_oracle as mo  # noqa: E402

ENV_KEYS = ("default_house_prop", "noise_house_prop", "noise_house_prop_test", "default_hvac_prop",
            "noise_hvac_prop", "noise_hvac_prop_test", "default_env_prop")


def patch(cfg, dotted, value):
    node = cfg
    parts = dotted.split(".")
    for p in parts[:-1]:
        node = node[p]
    node[parts[-1]] = value


def jsonable(obj):
    if isinstance(obj, dict):
        return {str(k): jsonable(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [jsonable(v) for v in obj]
    if isinstance(obj, (np.integer,)):
        return int(obj)
    if isinstance(obj, (np.floating,)):
        return float(obj)
    return obj


def epoch(d):
    return mo.to_epoch_seconds(d)


def capture_state(env):
    hs = [env.cluster.houses[i] for i in env.agent_ids]
    return dict(
        Ta=np.array([h.current_temp for h in hs], dtype=np.float64),
        Tm=np.array([h.current_mass_temp for h in hs], dtype=np.float64),
        on=np.array([bool(h.hvac.turned_on) for h in hs], dtype=np.uint8),
        lock=np.array([bool(h.hvac.lockout) for h in hs], dtype=np.uint8),
        sso=np.array([h.hvac.seconds_since_off for h in hs], dtype=np.int32),
    )


def capture_params(env):
    hs = [env.cluster.houses[i] for i in env.agent_ids]
    g = lambda fn, dt=np.float64: np.array([fn(h) for h in hs], dtype=dt)
    return dict(
        p_Ta=g(lambda h: h.current_temp), p_Tm=g(lambda h: h.current_mass_temp),
        p_target=g(lambda h: h.target_temp), p_deadband=g(lambda h: h.deadband),
        p_Ua=g(lambda h: h.Ua), p_Cm=g(lambda h: h.Cm), p_Ca=g(lambda h: h.Ca), p_Hm=g(lambda h: h.Hm),
        p_capacity=g(lambda h: h.hvac.cooling_capacity), p_COP=g(lambda h: h.hvac.COP),
        p_latent=g(lambda h: h.hvac.latent_cooling_fraction),
        p_lockout=g(lambda h: h.hvac.lockout_duration, np.int64),
        p_t0=np.int64(epoch(env.start_datetime)), p_phase=np.float64(env.cluster.phase),
        p_ratio=np.float64(env.power_grid.artificial_ratio), p_max_power=np.float64(env.power_grid.max_power),
    )


INTERP_DIR = "/tmp/mdr_golden_interp"


def make_small_grid(seed):
    """A small synthetic 10-D base-power grid in the reference's file formats (mergedGridSearchResultFinal.npy is a
    missing blob): the reference's own PowerInterpolator then reads it through the config paths."""
    import csv
    os.makedirs(INTERP_DIR, exist_ok=True)
    axes = {"Ua_ratio": [0.9, 1, 1.1], "Cm_ratio": [0.9, 1.1], "Ca_ratio": [0.95, 1.05], "Hm_ratio": [0.9, 1.1],
            "air_temp": [-4, -1, 0.3, 4], "mass_temp": [-4, 0, 4], "OD_temp": [1, 9, 15],
            "HVAC_power": [10000, 15000], "hour": [0.0, 27000.0, 46800.0, 86399.0],
            "date": [0, 171, 364]}
    rng = np.random.default_rng(seed)
    dims = [len(v) for v in axes.values()]
    # smooth-ish positive field (whole watts) so that interpolation weights matter
    values = np.round(3000.0 + 2500.0 * np.sin(np.arange(int(np.prod(dims))) * 0.37) + rng.uniform(0, 400, int(np.prod(dims))))
    np.savez_compressed(os.path.join(OUT, "interp_grid_small.npz"), values=values.astype(np.int16), axes=np.array(json.dumps(axes)))
    np.save(os.path.join(INTERP_DIR, "grid.npy"), values)
    with open(os.path.join(INTERP_DIR, "params.json"), "w") as f:
        json.dump(axes, f)
    with open(os.path.join(INTERP_DIR, "keys.csv"), "w") as f:
        csv.writer(f).writerow(list(axes.keys()))
    ip = PG + "base_power_parameters.interpolation."
    patches = {PG + "base_power_mode": "interpolation", ip + "path_datafile": os.path.join(INTERP_DIR, "grid.npy"),
               ip + "path_parameter_dict": os.path.join(INTERP_DIR, "params.json"),
               ip + "path_dict_keys": os.path.join(INTERP_DIR, "keys.csv")}
    return patches, dict(interp_grid_file=np.array("interp_grid_small.npz"))


def run_scenario(name, patches, seed, T, policy, perlin=False, norm_steps=(0, 1, -1), extra=None, save=True):
    ref = ref_harness.load_reference()
    cfg = copy.deepcopy({k: ref["config_dict"][k] for k in ENV_KEYS})
    patch(cfg, "default_env_prop.power_grid_prop.base_power_mode", "constant")  # the interpolation grid is a missing blob
    for k, v in patches.items():
        patch(cfg, k, v)
    N = cfg["default_env_prop"]["cluster_prop"]["nb_agents"]
    if perlin:
        # the lattice the oracle itself defines for (seed, env 0, episode 0)
        k0, k1 = mo.seed_key(seed)
        ref_harness.set_perlin_gradient(
            lambda l: 2.0 * mo.u01(mo.philox4x32_10(0, np.asarray(l, dtype=np.int64) & 0xFFFFFFFF, 0, mo.TAG_PERLIN, k0, k1)[0]) - 1.0)
    else:
        ref_harness.set_perlin_gradient(None)
    random.seed(seed)
    np.random.seed(seed)
    # the random part of the message gather (env 976-1002): which senders `random.sample` picked for each house
    # ('random_sample' mode) and which links `np.random.rand() > comm_defect_prob` dropped, recorded at the norm steps so
    # that the oracle's norm_state can be pinned on the reference's vectors with the reference's own draws
    cl = cfg["default_env_prop"]["cluster_prop"]
    record_comm = cl["agents_comm_mode"] == "random_sample" or cl["comm_defect_prob"] > 0
    sampled = []
    real_sample = random.sample
    if cl["agents_comm_mode"] == "random_sample":
        def recording_sample(population, k):
            picked = real_sample(population, k=k)
            sampled.append(list(picked))
            return picked
        random.sample = recording_sample
    msg_keep, msg_senders = [], []

    def capture_comm(obs):
        if not record_comm:
            return
        c = len(obs[0]["message"])
        msg_keep.append(np.array([[m["hvac_max_consumption"] != 0 for m in obs[i]["message"]] for i in range(N)], dtype=np.uint8).reshape(N, c))
        if cl["agents_comm_mode"] == "random_sample":
            assert len(sampled) == N, len(sampled)       # one random.sample call per house, in house order (env 976-983)
            msg_senders.append(np.array(sampled, dtype=np.int32).reshape(N, c))
        else:
            comm_now = env.cluster.agent_communicators
            msg_senders.append(np.array([comm_now[i] for i in range(N)], dtype=np.int32).reshape(N, c))

    env = ref["MADemandResponseEnv"](cfg)
    del sampled[:]
    obs = env.reset()
    rec = capture_params(env)
    od = [env.cluster.current_OD_temp]
    S = [float(env.power_grid.current_signal)]
    base_power = [float(env.power_grid.base_power)]
    abs_noise = [float(env.power_grid.cumulated_abs_noise)]
    act_rng = np.random.default_rng(seed + 1000)
    actors = {i: ref["BangBangController"]({"id": i}, cfg) for i in range(N)}
    norm = []
    want_norm = set(s if s >= 0 else T + 1 + s for s in norm_steps)
    if 0 in want_norm:
        norm.append(np.array([ref["utils"].normStateDict(obs[i], cfg) for i in range(N)]))
        capture_comm(obs)
    keys = ("Ta", "Tm", "on", "lock", "sso")
    out = {k: [] for k in keys}
    out.update(P=[], reward=[], solar=[], actions=[])
    for t in range(T):
        if policy == "bangbang":
            a = ref["utils"].get_actions(actors, obs)
        elif policy in ("deadband", "basic", "always_on"):      # the reference's other rule-based controllers, acting on its own obs
            cls = {"deadband": "DeadbandBangBangController", "basic": "BasicController", "always_on": "AlwaysOnController"}[policy]
            others = {i: ref[cls]({"id": i}, cfg) for i in range(N)}
            # (AlwaysOnController.act returns True whatever it is handed; the other two index obs by their id themselves)
            a = {i: others[i].act(obs) for i in range(N)}
        elif policy == "greedy_myopic":      # the centralised baseline: one object per house sharing a module-level memory
            if t == 0:
                greedy = {i: ref["GreedyMyopic"]({"id": i}, cfg) for i in range(N)}
            a = {i: bool(greedy[i].act(obs)) for i in range(N)}
        elif policy == "on":
            a = {i: True for i in range(N)}
        elif policy == "off":
            a = {i: False for i in range(N)}
        elif policy == "onoff":      # long on / long off blocks: walks through every lockout edge
            a = {i: bool(((t + 3 * i) // 7) % 2) for i in range(N)}
        elif policy.startswith("random"):
            p = float(policy.split(":")[1]) if ":" in policy else 0.5
            r = act_rng.random(N)
            a = {i: int(r[i] < p) for i in range(N)}     # ints, as Categorical.sample().item() gives
        elif policy == "mixed":      # bang-bang with 20 % random flips
            bb = ref["utils"].get_actions(actors, obs)
            r = act_rng.random(N)
            a = {i: (not bb[i]) if r[i] < 0.2 else bb[i] for i in range(N)}
        else:
            raise ValueError(policy)
        out["actions"].append(np.array([bool(a[i]) for i in range(N)], dtype=np.uint8))
        del sampled[:]
        obs, rew, done, info = env.step(a)
        st = capture_state(env)
        for k in keys:
            out[k].append(st[k])
        out["P"].append(float(info["cluster_hvac_power"]))
        out["reward"].append(np.array([rew[i] for i in range(N)], dtype=np.float64))
        out["solar"].append(float(env.cluster.houses[0].current_solar_gain))
        od.append(env.cluster.current_OD_temp)
        S.append(float(env.power_grid.current_signal))
        base_power.append(float(env.power_grid.base_power))
        abs_noise.append(float(env.power_grid.cumulated_abs_noise))
        assert not any(done.values())
        if (t + 1) in want_norm:
            norm.append(np.array([ref["utils"].normStateDict(obs[i], cfg) for i in range(N)]))
            capture_comm(obs)
    random.sample = real_sample
    meta = dict(name=name, seed=seed, T=T, N=N, policy=policy, perlin_standin=bool(perlin),
                norm_steps=sorted(want_norm), config=jsonable(cfg),
                generated_by="tests/golden/make_golden.py from /root/reference (snapshot 2025-03-14)")
    arrays = dict(rec)
    arrays.update({k: np.array(v) for k, v in out.items()})
    arrays["od"] = np.array(od, dtype=np.float64)
    arrays["S"] = np.array(S, dtype=np.float64)
    arrays["norm_state"] = np.array(norm, dtype=np.float64)
    comm = env.cluster.agent_communicators
    if comm and all(len(comm[i]) == len(comm[0]) for i in range(N)):
        arrays["links"] = np.array([comm[i] for i in range(N)], dtype=np.int32).reshape(N, -1)
    if perlin:
        arrays["cumulated_abs_noise"] = np.array(abs_noise, dtype=np.float64)   # PowerGrid.cumulated_abs_noise after reset and every step
        arrays["grid_nb_steps"] = np.int64(env.power_grid.nb_steps)
    if record_comm:
        arrays["msg_keep"] = np.array(msg_keep, dtype=np.uint8)            # [norm step][N][c] 1 = delivered
        arrays["msg_senders"] = np.array(msg_senders, dtype=np.int32)      # [norm step][N][c] sender house ids
    if extra:
        arrays.update(extra)
        arrays["base_power"] = np.array(base_power, dtype=np.float64)
    arrays["meta"] = np.array(json.dumps(meta))
    if not save:          # tests/test_oracle_vs_reference_live.py: compare on the spot, write nothing
        return arrays
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-28s N=%-3d T=%-5d  sum(reward)=%.9f  %6.1f KB" % (
        name, N, T, float(np.sum(arrays["reward"])), os.path.getsize(path) / 1024))
    return arrays


def montecarlo_points():
    """monteCarlo.py:133-201 run through the reference env for a handful of grid points: the bang-bang 'stabilised
    average power' that fills mergedGridSearchResultFinal.npy (the missing blob).  The config edits restate
    eval_parameters_bangbang_average_consumption; the env and the controller are the reference's own."""
    import datetime as dt
    ref = ref_harness.load_reference()
    with open(os.path.join(ref_harness.REFERENCE_ROOT, "monteCarlo", "interp_parameters_dict.json")) as f:
        axes = json.load(f)
    with open(os.path.join(OUT, "reference_interp_axes.json"), "w") as f:
        json.dump(axes, f)
    keys = list(axes.keys())
    rng = np.random.default_rng(123)
    points, results = [], []
    NB_STEPS, NB_AVG = 75, 10          # monteCarlo.py:23-24
    for n in range(48):
        idx = [int(rng.integers(len(axes[k]))) for k in keys]
        p = {k: axes[k][i] for k, i in zip(keys, idx)}
        cfg = copy.deepcopy({k: ref["config_dict"][k] for k in ENV_KEYS})
        date = dt.date(2021, 1, 1) + dt.timedelta(days=p["date"])
        hour = p["hour"]
        cfg["noise_house_prop"]["noise_mode"] = "no_noise"
        cfg["noise_hvac_prop"]["noise_mode"] = "no_noise"
        cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = 1
        cfg["default_hvac_prop"]["cooling_capacity"] = p["HVAC_power"]
        cfg["default_env_prop"]["start_datetime_mode"] = "fixed"
        cfg["default_hvac_prop"]["lockout_duration"] = 1
        cfg["default_env_prop"]["start_datetime"] = str(dt.datetime(date.year, date.month, date.day, int(hour // 3600),
                                                                    int(hour % 3600 // 60), int(hour % 60)))
        for k in ("Ua", "Cm", "Ca", "Hm"):
            cfg["default_house_prop"][k] *= p[k + "_ratio"]
        tgt = cfg["default_house_prop"]["target_temp"]
        cfg["default_house_prop"]["init_air_temp"] = tgt + p["air_temp"]
        cfg["default_house_prop"]["init_mass_temp"] = tgt + p["mass_temp"]
        cfg["default_env_prop"]["cluster_prop"]["temp_mode"] = "constant"
        cfg["default_env_prop"]["cluster_prop"]["temp_parameters"]["constant"]["day_temp"] = tgt + p["OD_temp"]
        cfg["default_env_prop"]["cluster_prop"]["temp_parameters"]["constant"]["night_temp"] = tgt + p["OD_temp"]
        cfg["default_env_prop"]["power_grid_prop"]["base_power_mode"] = "constant"
        ref_harness.set_perlin_gradient(None)
        env = ref["MADemandResponseEnv"](cfg)
        actor = {0: ref["BangBangController"]({"id": 0}, cfg)}
        obs = env.reset()
        actions = ref["utils"].get_actions(actor, obs)
        total, avg = 0.0, 0.0
        for i in range(NB_STEPS):
            obs, _, _, info = env.step(actions)
            total += info["cluster_hvac_power"]
            if i >= NB_STEPS - NB_AVG:
                avg += total / ((i + 1) * NB_AVG)
            actions = ref["utils"].get_actions(actor, obs)
        points.append(idx)
        results.append(avg)
    np.savez_compressed(os.path.join(OUT, "montecarlo_points.npz"), index=np.array(points, dtype=np.int32),
                        hvac_average_power=np.array(results, dtype=np.float64))
    print("montecarlo_points: %d grid points, mean %.1f W" % (len(points), float(np.mean(results))))


CL = "default_env_prop.cluster_prop."
PG = "default_env_prop.power_grid_prop."
RW = "default_env_prop.reward_prop."


def main():
    ref = ref_harness.load_reference()
    # config snapshot (values only) so the product's own default_config() can be checked on any box
    with open(os.path.join(OUT, "reference_env_config.json"), "w") as f:
        json.dump(jsonable({k: ref["config_dict"][k] for k in ENV_KEYS}), f, indent=1, sort_keys=True)

    montecarlo_points()
    # S1: BASELINE config 1 - 1 env x 10 houses, default noise, random start, bang-bang
    a = run_scenario("s1_c1_sinusoidals", {CL + "nb_agents": 10, PG + "signal_mode": "sinusoidals"}, 1, 1000, "bangbang")
    # anchor digits recorded in SURVEY.md section 8c
    assert abs(float(a["reward"].sum()) - (-22544.136763550)) < 1e-6, float(a["reward"].sum())
    run_scenario("s1_c1_flat", {CL + "nb_agents": 10, PG + "signal_mode": "flat"}, 2, 300, "bangbang")
    run_scenario("s1_c1_regular_steps", {CL + "nb_agents": 10, PG + "signal_mode": "regular_steps"}, 3, 300, "bangbang")
    # S2: BASELINE config 2's shape - fixed OD temp, no noise, uniform houses, random actions
    run_scenario("s2_c2_uniform", {
        CL + "nb_agents": 50, CL + "temp_mode": "constant", "noise_house_prop.noise_mode": "no_noise",
        "default_env_prop.start_datetime_mode": "fixed", "default_env_prop.start_datetime": "2021-06-15 12:00:00",
        "default_house_prop.solar_gain_bool": False, PG + "signal_mode": "flat"}, 1234, 500, "random:0.5")
    # S3: BASELINE config 3's shape - heterogeneous houses and HVACs, noisy OD temp, solar edge at 07:30
    run_scenario("s3_c3_heterogeneous", {
        CL + "nb_agents": 64, "noise_house_prop.noise_mode": "house_big_noise", "noise_hvac_prop.noise_mode": "big_noise",
        "default_hvac_prop.lockout_noise": 20, "default_house_prop.deadband": 1,
        "default_env_prop.start_datetime_mode": "fixed", "default_env_prop.start_datetime": "2021-07-14 07:10:00",
        PG + "signal_mode": "sinusoidals"}, 7, 600, "mixed")
    run_scenario("s3_big_noise_random_start", {
        CL + "nb_agents": 24, "noise_house_prop.noise_mode": "big_noise", "noise_hvac_prop.noise_mode": "small_noise",
        PG + "signal_mode": "regular_steps", PG + "artificial_signal_ratio_range": 3}, 11, 400, "mixed")
    # S4: every temperature-penalty mode
    for mode in ("individual_L2", "common_L2", "common_max", "mixture"):
        p = {CL + "nb_agents": 12, RW + "temp_penalty_mode": mode, "default_house_prop.deadband": 1,
             "noise_house_prop.noise_mode": "big_noise", PG + "signal_mode": "sinusoidals",
             RW + "alpha_temp": 0.7, RW + "alpha_sig": 1.3}
        if mode == "mixture":
            p[RW + "temp_penalty_parameters.mixture"] = {"alpha_ind_L2": 1, "alpha_common_L2": 2, "alpha_common_max": 0.5}
        run_scenario("s4_penalty_" + mode, p, 21, 60, "mixed")
    # S5: year-end rollover
    run_scenario("s5_year_rollover", {
        CL + "nb_agents": 12, "noise_house_prop.noise_mode": "house_big_noise",
        "default_env_prop.start_datetime_mode": "fixed", "default_env_prop.start_datetime": "2021-12-31 23:50:00",
        PG + "signal_mode": "flat"}, 5, 300, "mixed")
    # leap day (2024-02-28 23:55 -> 02-29): calendar arithmetic feeds the solar polynomial's y term
    run_scenario("s5_leap_day_noon", {
        CL + "nb_agents": 8, "default_env_prop.start_datetime_mode": "fixed",
        "default_env_prop.start_datetime": "2024-02-29 11:58:00", PG + "signal_mode": "flat"}, 6, 120, "bangbang")
    # S6: lockout edges
    for pol in ("on", "off", "onoff"):
        run_scenario("s6_lockout_" + pol, {CL + "nb_agents": 8, "default_hvac_prop.lockout_noise": 10,
                                           "default_hvac_prop.lockout_duration": 30, PG + "signal_mode": "flat"}, 31, 100, pol)
    # time step other than 4 s and a lockout that is not a multiple of it
    run_scenario("s6_dt7_lockout45", {CL + "nb_agents": 8, "default_env_prop.time_step": 7,
                                      "default_hvac_prop.lockout_duration": 45, PG + "signal_mode": "sinusoidals"}, 32, 150, "onoff")
    # S7: Perlin wiring with this build's lattice as the stand-in for the absent third-party package
    run_scenario("s7_perlin_wiring", {CL + "nb_agents": 16, PG + "signal_mode": "perlin",
                                      PG + "artificial_signal_ratio_range": 3}, 41, 400, "bangbang", perlin=True)
    run_scenario("s7_fastpp_perlin_wiring", {CL + "nb_agents": 16, PG + "signal_mode": "fast++_perlin"}, 42, 200, "mixed", perlin=True)
    # S8: random phase offset of the outdoor sinusoid
    run_scenario("s8_phase_offset", {CL + "nb_agents": 8, CL + "temp_mode": "shifting_sinusoidal_heatwave",
                                     PG + "signal_mode": "flat"}, 51, 200, "bangbang")
    # S10: every optional normStateDict / message column (state_properties, message_properties all True)
    ST, MS = "default_env_prop.state_properties.", "default_env_prop.message_properties."
    allflags = {ST + k: True for k in ("hour", "day", "solar_gain", "thermal", "hvac")}
    allflags.update({MS + "thermal": True, MS + "hvac": True})
    run_scenario("s10_obs_all_columns", dict(allflags, **{
        CL + "nb_agents": 12, "noise_house_prop.noise_mode": "big_noise", "noise_hvac_prop.noise_mode": "big_noise",
        "default_env_prop.time_step": 60, "default_env_prop.start_datetime_mode": "fixed",
        "default_env_prop.start_datetime": "2021-03-10 10:45:00", PG + "signal_mode": "sinusoidals"}),
        71, 90, "mixed", norm_steps=(0, 1, 20, 50, -1))
    run_scenario("s10_obs_thermal_only", {
        ST + "thermal": True, MS + "hvac": True, CL + "nb_agents": 9, CL + "nb_agents_comm": 4,
        "noise_house_prop.noise_mode": "house_big_noise", PG + "signal_mode": "flat"}, 72, 40, "mixed", norm_steps=(0, 7, -1))
    # S11: the other static communication topologies (env 806-902)
    run_scenario("s11_comm_closed_groups", {CL + "nb_agents": 14, CL + "nb_agents_comm": 3, CL + "agents_comm_mode": "closed_groups",
                                            PG + "signal_mode": "flat"}, 81, 30, "mixed", norm_steps=(0, 5, -1))
    run_scenario("s11_comm_neighbours_2D", {CL + "nb_agents": 25, CL + "agents_comm_mode": "neighbours_2D",
                                            PG + "signal_mode": "flat"}, 82, 30, "mixed", norm_steps=(0, 5, -1))
    run_scenario("s11_comm_random_fixed", {CL + "nb_agents": 10, CL + "nb_agents_comm": 4, CL + "agents_comm_mode": "random_fixed",
                                           PG + "signal_mode": "flat"}, 83, 30, "mixed", norm_steps=(0, 5, -1))
    run_scenario("s11_comm_no_message", {CL + "nb_agents": 6, CL + "agents_comm_mode": "no_message",
                                         PG + "signal_mode": "flat"}, 84, 30, "mixed", norm_steps=(0, 5, -1))
    # S13: the random part of the message gather (env 976-1002): link defects and per-step `random.sample` senders, with the
    # reference's own draws recorded next to its normStateDict vectors
    run_scenario("s13_comm_defects_neighbours", dict(allflags, **{
        CL + "nb_agents": 14, CL + "comm_defect_prob": 0.3, "noise_house_prop.noise_mode": "big_noise",
        "noise_hvac_prop.noise_mode": "big_noise", PG + "signal_mode": "flat"}), 85, 30, "mixed", norm_steps=(0, 1, 5, -1))
    run_scenario("s13_comm_defects_closed_groups", {
        CL + "nb_agents": 12, CL + "nb_agents_comm": 3, CL + "agents_comm_mode": "closed_groups", CL + "comm_defect_prob": 0.5,
        PG + "signal_mode": "flat"}, 86, 20, "mixed", norm_steps=(0, 3, -1))
    run_scenario("s13_comm_random_sample", {
        CL + "nb_agents": 13, CL + "nb_agents_comm": 5, CL + "agents_comm_mode": "random_sample",
        "noise_house_prop.noise_mode": "big_noise", MS + "hvac": True, PG + "signal_mode": "flat"}, 87, 20, "mixed", norm_steps=(0, 1, 4, -1))
    run_scenario("s13_comm_random_sample_defects", {
        CL + "nb_agents": 9, CL + "nb_agents_comm": 8, CL + "agents_comm_mode": "random_sample", CL + "comm_defect_prob": 0.1,
        "noise_hvac_prop.noise_mode": "big_noise", PG + "signal_mode": "flat"}, 88, 20, "mixed", norm_steps=(0, 2, -1))
    # S12: base_power_mode "interpolation" (the reference's DEFAULT, env 1195-1255) on a small synthetic grid
    gp, gx = make_small_grid(5)
    run_scenario("s12_interp_default_like", dict(gp, **{
        CL + "nb_agents": 12, "noise_house_prop.noise_mode": "small_noise", "noise_hvac_prop.noise_mode": "big_noise",
        PG + "signal_mode": "sinusoidals"}), 91, 200, "mixed", extra=gx)
    run_scenario("s12_interp_no_solar_dt7", dict(gp, **{
        CL + "nb_agents": 9, "noise_house_prop.noise_mode": "house_small_noise", "default_house_prop.solar_gain_bool": False,
        "default_env_prop.time_step": 7, PG + "signal_mode": "flat"}), 92, 120, "mixed", extra=gx)
    run_scenario("s12_interp_perlin_noon", dict(gp, **{
        CL + "nb_agents": 16, "noise_house_prop.noise_mode": "big_noise", "default_env_prop.start_datetime_mode": "fixed",
        "default_env_prop.start_datetime": "2021-06-20 07:20:00", PG + "signal_mode": "perlin"}), 93, 180, "bangbang",
        perlin=True, extra=gx)
    # N == 1 (config.py's literal default nb_agents) and no neighbours
    run_scenario("s9_single_house", {CL + "nb_agents": 1, PG + "signal_mode": "sinusoidals"}, 61, 200, "bangbang")
    main_controllers()


def main_controllers():
    """Only the S14 fixtures (python tests/golden/make_golden.py s14): added in round 3, the others stay as they were generated."""
    # S14: the other controllers of agents/bangbang_controllers.py in the loop (deadbands of 1-2 degrees so that the hold band is visited)
    run_scenario("s14_controller_deadband", {CL + "nb_agents": 12, PG + "signal_mode": "sinusoidals", "default_house_prop.deadband": 1.0,
                                             "noise_house_prop.noise_mode": "big_noise", "default_hvac_prop.lockout_noise": 10}, 71, 600, "deadband")
    run_scenario("s14_controller_basic", {CL + "nb_agents": 6, PG + "signal_mode": "flat", "default_house_prop.deadband": 2.0}, 72, 400, "basic")
    run_scenario("s14_controller_always_on", {CL + "nb_agents": 5, PG + "signal_mode": "flat"}, 73, 150, "always_on")
    run_scenario("s14_controller_greedy_myopic", {CL + "nb_agents": 14, PG + "signal_mode": "sinusoidals", "noise_house_prop.noise_mode": "big_noise",
                                                  "noise_hvac_prop.noise_mode": "big_noise", "default_hvac_prop.lockout_noise": 10}, 74, 400, "greedy_myopic")


if __name__ == "__main__":
    if sys.argv[1:] == ["s14"]:
        main_controllers()
    else:
        main()
