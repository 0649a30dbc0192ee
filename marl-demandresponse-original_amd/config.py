"""Environment configuration: the reference's ``config.py`` schema, restated, and its flattening.

``default_config()`` returns a nested dict with the same keys and default values as the
environment-related part of the reference's ``config_dict`` (config.py:12-422 - ``default_house_prop``,
``noise_house_prop[_test]``, ``default_hvac_prop``, ``noise_hvac_prop[_test]``, ``default_env_prop``), so a
caller can pass either the reference's own ``config_dict`` or this one.  Agent hyper-parameters
(``PPO_prop`` ...) are not part of the step path and are not restated.
tests/test_config.py checks it key by key against a snapshot of the reference's values.

``flatten_config()`` turns such a dict into the flat ``EnvSpec`` the C ABI consumes (include/mdr.h,
``mdr_config_t``), performing the reference's mode validation (ValueError) before any kernel runs.
"""
from __future__ import annotations

import copy
import datetime as _dt
from dataclasses import dataclass, field
from typing import List, Optional

_EPOCH = _dt.datetime(1970, 1, 1)

SIGNAL_FLAT, SIGNAL_SINUSOIDALS, SIGNAL_REGULAR_STEPS, SIGNAL_PERLIN = range(4)
PENALTY_MODES = {"individual_L2": 0, "common_L2": 1, "common_max": 2, "mixture": 3}


def _house_noise(std_start, std_target, low, high):
    return {"std_start_temp": std_start, "std_target_temp": std_target,
            "factor_thermo_low": low, "factor_thermo_high": high}


def _weather(day, night, std=0, shifting=False):
    return {"day_temp": day, "night_temp": night, "temp_std": std, "random_phase_offset": shifting}


def _perlin(amplitude=0.9, period=400):
    return {"amplitude_ratios": amplitude, "nb_octaves": 5, "octaves_step": 5, "period": period}


def default_config() -> dict:
    house_noise_train = {
        "no_noise": _house_noise(0, 0, 1, 1),
        "dwarf_noise": _house_noise(0.05, 0.05, 1, 1),
        "house_small_noise": _house_noise(0, 0, 0.9, 1.1),
        "house_medium_noise": _house_noise(0, 0, 0.8, 1.2),
        "house_big_noise": _house_noise(0, 0, 0.5, 1.5),
        "small_noise": _house_noise(3, 1, 0.9, 1.1),
        "big_noise": _house_noise(5, 2, 0.8, 1.2),
        "small_start_temp": _house_noise(3, 0, 1, 1),
        "big_start_temp": _house_noise(5, 0, 1, 1),
    }
    house_noise_test = {k: copy.deepcopy(house_noise_train[k]) for k in
                        ("no_noise", "dwarf_noise", "small_noise", "big_noise", "small_start_temp", "big_start_temp")}
    hvac_noise_train = {
        "no_noise": {"cooling_capacity_list": {10000: [10000], 15000: [15000]}},
        "small_noise": {"cooling_capacity_list": {10000: [9000, 10000, 11000], 15000: [12500, 15000, 17500]}},
        "big_noise": {"cooling_capacity_list": {10000: [7500, 9000, 10000, 11000, 12500],
                                                15000: [10000, 12500, 15000, 17500, 20000]}},
    }

    def _hvac_test(std, cop_lo, cop_hi, cap_lo, cap_hi):
        return {"std_latent_cooling_fraction": std, "factor_COP_low": cop_lo, "factor_COP_high": cop_hi,
                "factor_cooling_capacity_low": cap_lo, "factor_cooling_capacity_high": cap_hi}

    hvac_noise_test = {
        "no_noise": _hvac_test(0, 1, 1, 1, 1),
        "small_noise": _hvac_test(0.05, 0.95, 1.05, 0.9, 1.1),
        "big_noise": _hvac_test(0.1, 0.85, 1.15, 0.6666667, 1.3333333333),
    }
    weather = {
        "constant": _weather(26.5, 26.5),
        "sinusoidal": _weather(30, 23),
        "sinusoidal_hot": _weather(30, 28),
        "sinusoidal_heatwave": _weather(34, 28),
        "sinusoidal_hot_heatwave": _weather(38, 32),
        "sinusoidal_cold_heatwave": _weather(30, 24),
        "sinusoidal_cold": _weather(24, 22),
        "noisy_sinusoidal": _weather(30, 23, 0.5),
        "noisy_sinusoidal_hot": _weather(30, 28, 0.5),
        "noisy_sinusoidal_heatwave": _weather(34, 28, 0.5),
        "noisier_sinusoidal_heatwave": _weather(34, 28, 2),
        "noisy_sinusoidal_cold": _weather(24, 22, 0.5),
        "shifting_sinusoidal": _weather(30, 23, 0, True),
        "shifting_sinusoidal_heatwave": _weather(34, 28, 0, True),
    }
    signals = {
        "flat": {},
        "sinusoidals": {"periods": [400, 1200], "amplitude_ratios": [0.1, 0.3]},
        "regular_steps": {"amplitude_per_hvac": 6000, "period": 300},
        "perlin": _perlin(),
        "amplitude+_perlin": _perlin(0.9 * 1.1),
        "amplitude++_perlin": _perlin(0.9 * 1.3),
        "fast+_perlin": _perlin(period=300),
        "fast++_perlin": _perlin(period=200),
    }
    return {
        "default_house_prop": {
            "id": 1, "init_air_temp": 20, "init_mass_temp": 20, "target_temp": 20, "deadband": 0,
            "Ua": 2.18e02, "Cm": 3.45e06, "Ca": 9.08e05, "Hm": 2.84e03,
            "window_area": 7.175, "shading_coeff": 0.67, "solar_gain_bool": True,
        },
        "noise_house_prop": {"noise_mode": "big_start_temp", "noise_parameters": house_noise_train},
        "noise_house_prop_test": {"noise_mode": "small_start_temp", "noise_parameters": house_noise_test},
        "default_hvac_prop": {
            "id": 1, "COP": 2.5, "cooling_capacity": 15000, "latent_cooling_fraction": 0.35,
            "lockout_duration": 40, "lockout_noise": 0,
        },
        "noise_hvac_prop": {"noise_mode": "no_noise", "noise_parameters": hvac_noise_train},
        "noise_hvac_prop_test": {"noise_mode": "no_noise", "noise_parameters": hvac_noise_test},
        "default_env_prop": {
            "start_datetime": "2021-01-01 00:00:00",
            "start_datetime_mode": "random",
            "time_step": 4,
            "cluster_prop": {
                "temp_mode": "noisy_sinusoidal_heatwave",
                "temp_parameters": weather,
                "nb_agents": 1,
                "nb_agents_comm": 10,
                "agents_comm_mode": "neighbours",
                "comm_defect_prob": 0,
                "agents_comm_parameters": {"neighbours_2D": {"row_size": 5, "distance_comm": 2}},
            },
            "state_properties": {"hour": False, "day": False, "solar_gain": False, "thermal": False, "hvac": False},
            "message_properties": {"thermal": False, "hvac": False},
            "power_grid_prop": {
                "base_power_mode": "interpolation",
                "base_power_parameters": {
                    "constant": {"avg_power_per_hvac": 4200, "init_signal_per_hvac": 910},
                    "interpolation": {
                        "path_datafile": "./monteCarlo/mergedGridSearchResultFinal.npy",
                        "path_parameter_dict": "./monteCarlo/interp_parameters_dict.json",
                        "path_dict_keys": "./monteCarlo/interp_dict_keys.csv",
                        "interp_update_period": 300,
                        "interp_nb_agents": 100,
                    },
                },
                "artificial_signal_ratio_range": 1,
                "artificial_ratio": 1.0,
                "signal_mode": "perlin",
                "signal_parameters": signals,
            },
            "reward_prop": {
                "alpha_temp": 1, "alpha_sig": 1, "norm_reg_sig": 7500,
                "temp_penalty_mode": "individual_L2",
                "temp_penalty_parameters": {
                    "individual_L2": {}, "common_L2": {}, "common_max_error": {},
                    "mixture": {"alpha_ind_L2": 1, "alpha_common_L2": 1, "alpha_common_max": 0},
                },
                "sig_penalty_mode": "common_L2",
            },
        },
    }


def _deadband_l2(target, deadband, value):
    """utils.deadbandL2 (utils.py:1266-1274), scalars; used for the two reward norms only."""
    if target + deadband / 2 < value:
        return (value - (target + deadband / 2)) ** 2
    if target - deadband / 2 > value:
        return ((target - deadband / 2) - value) ** 2
    return 0.0


@dataclass
class EnvSpec:
    """Flat view of the config entries the step path consumes; field names follow include/mdr.h."""
    nb_houses_total: int
    time_step: int
    temp_ref: float
    init_air_temp: float
    init_mass_temp: float
    target_temp: float
    deadband: float
    Ua: float
    Cm: float
    Ca: float
    Hm: float
    window_area: float
    shading_coeff: float
    solar_gain: bool
    lockout_duration: int
    lockout_noise: int
    COP: float
    cooling_capacity: float
    latent_cooling_fraction: float
    std_start_temp: float
    std_target_temp: float
    factor_thermo_low: float
    factor_thermo_high: float
    capacity_list: List[float]
    start_random: bool
    start_epoch: int
    day_temp: float
    night_temp: float
    temp_std: float
    random_phase_offset: bool
    signal_mode: int
    signal_mode_name: str
    avg_power_per_hvac: float
    sin_periods: List[float]
    sin_amplitude_ratios: List[float]
    steps_amplitude_per_hvac: float
    steps_period: float
    perlin_amplitude: float
    perlin_nb_octaves: int
    perlin_octaves_step: float
    perlin_period: float
    artificial_ratio: float
    artificial_signal_ratio_range: float
    alpha_temp: float
    alpha_sig: float
    norm_temp_penalty: float
    norm_sig_penalty: float
    penalty_mode: int
    mix_ind_L2: float
    mix_common_L2: float
    mix_common_max: float
    norm_reg_sig: float
    obs_power_norm: float
    nb_agents_comm: int
    agents_comm_mode: str
    comm_defect_prob: float
    base_power_mode: int = 0           # 0 constant, 1 interpolation (env 1248-1255)
    interp_update_period: int = 300
    interp_nb_agents: int = 100
    interp_paths: dict = field(default_factory=dict)
    state_properties: dict = field(default_factory=dict)
    message_properties: dict = field(default_factory=dict)


def to_epoch_seconds(d: _dt.datetime) -> int:
    return int((d - _EPOCH).total_seconds())


def from_epoch_seconds(t: int) -> _dt.datetime:
    return _EPOCH + _dt.timedelta(seconds=int(t))


def flatten_config(config: dict, test: bool = False) -> EnvSpec:
    """Mirrors what MADemandResponseEnv.__init__/build_environment read (env/MA_DemandResponse.py:84-133)."""
    env = config["default_env_prop"]
    house = config["default_house_prop"]
    hvac = config["default_hvac_prop"]
    noise_house = config["noise_house_prop_test" if test else "noise_house_prop"]
    noise_hvac = config["noise_hvac_prop_test" if test else "noise_hvac_prop"]
    cluster = env["cluster_prop"]
    grid = env["power_grid_prop"]
    reward = env["reward_prop"]

    house_noise = noise_house["noise_parameters"][noise_house["noise_mode"]]
    hvac_noise = noise_hvac["noise_parameters"][noise_hvac["noise_mode"]]
    # the reference's test=True path dies here too (utils.py:674: no "cooling_capacity_list" in the *_test table)
    capacities = hvac_noise["cooling_capacity_list"][hvac["cooling_capacity"]]
    weather = cluster["temp_parameters"][cluster["temp_mode"]]

    if env["start_datetime_mode"] not in ("random", "fixed"):
        raise ValueError("start_datetime_mode in default_env_prop in config.py must be random or fixed. "
                         "Current value: {}.".format(env["start_datetime_mode"]))
    base_mode = grid["base_power_mode"]
    if base_mode not in ("constant", "interpolation"):
        raise ValueError("The base_power_mode parameter in the config file can only be 'constant' or 'interpolation'. "
                         "It is currently: {}".format(base_mode))
    mode = grid["signal_mode"]
    if mode == "flat":
        signal = SIGNAL_FLAT
    elif mode == "sinusoidals":
        signal = SIGNAL_SINUSOIDALS
    elif mode == "regular_steps":
        signal = SIGNAL_REGULAR_STEPS
    elif "perlin" in mode:
        signal = SIGNAL_PERLIN
    else:
        raise ValueError("Invalid power grid signal mode: {}. Change value in the config file.".format(mode))
    sp = grid["signal_parameters"][mode]
    periods = list(sp.get("periods", [])) if signal == SIGNAL_SINUSOIDALS else []
    ratios = list(sp.get("amplitude_ratios", [])) if signal == SIGNAL_SINUSOIDALS else []
    if len(periods) != len(ratios):
        raise ValueError("Power grid signal parameters: periods and amplitude_ratios lists should have the same "
                         "length. len(periods): {}, leng(amplitude_ratios): {}.".format(len(periods), len(ratios)))
    if reward["temp_penalty_mode"] not in PENALTY_MODES:
        raise ValueError("Unknown temperature penalty mode: {}".format(reward["temp_penalty_mode"]))
    if reward["sig_penalty_mode"] != "common_L2":
        raise ValueError("Unknown signal penalty mode: {}".format(reward["sig_penalty_mode"]))
    known_comm = ("neighbours", "closed_groups", "random_sample", "random_fixed", "neighbours_2D", "no_message")
    if cluster["agents_comm_mode"] not in known_comm:
        raise ValueError("Cluster property: unknown agents_comm_mode '{}'.".format(cluster["agents_comm_mode"]))
    mix = reward["temp_penalty_parameters"].get("mixture", {})
    start = _dt.datetime.strptime(env["start_datetime"], "%Y-%m-%d %H:%M:%S")
    perlin = sp if signal == SIGNAL_PERLIN else {}
    steps = sp if signal == SIGNAL_REGULAR_STEPS else {}
    target = float(house["target_temp"])
    norm_reg = float(reward["norm_reg_sig"])
    return EnvSpec(
        nb_houses_total=int(cluster["nb_agents"]), time_step=int(env["time_step"]), temp_ref=target,
        init_air_temp=float(house["init_air_temp"]), init_mass_temp=float(house["init_mass_temp"]),
        target_temp=target, deadband=float(house["deadband"]),
        Ua=float(house["Ua"]), Cm=float(house["Cm"]), Ca=float(house["Ca"]), Hm=float(house["Hm"]),
        window_area=float(house["window_area"]), shading_coeff=float(house["shading_coeff"]),
        solar_gain=bool(house["solar_gain_bool"]),
        lockout_duration=int(hvac["lockout_duration"]), lockout_noise=int(hvac["lockout_noise"]),
        COP=float(hvac["COP"]), cooling_capacity=float(hvac["cooling_capacity"]),
        latent_cooling_fraction=float(hvac["latent_cooling_fraction"]),
        std_start_temp=float(house_noise["std_start_temp"]), std_target_temp=float(house_noise["std_target_temp"]),
        factor_thermo_low=float(house_noise["factor_thermo_low"]), factor_thermo_high=float(house_noise["factor_thermo_high"]),
        capacity_list=[float(c) for c in capacities],
        start_random=env["start_datetime_mode"] == "random", start_epoch=to_epoch_seconds(start),
        day_temp=float(weather["day_temp"]), night_temp=float(weather["night_temp"]), temp_std=float(weather["temp_std"]),
        random_phase_offset=bool(weather["random_phase_offset"]),
        signal_mode=signal, signal_mode_name=mode,
        avg_power_per_hvac=float(grid["base_power_parameters"]["constant"]["avg_power_per_hvac"]),
        sin_periods=[float(p) for p in periods], sin_amplitude_ratios=[float(r) for r in ratios],
        steps_amplitude_per_hvac=float(steps.get("amplitude_per_hvac", 0.0)), steps_period=float(steps.get("period", 0.0)),
        perlin_amplitude=float(perlin.get("amplitude_ratios", 0.0)), perlin_nb_octaves=int(perlin.get("nb_octaves", 0)),
        perlin_octaves_step=float(perlin.get("octaves_step", 0.0)), perlin_period=float(perlin.get("period", 0.0)),
        artificial_ratio=float(grid["artificial_ratio"]),
        artificial_signal_ratio_range=float(grid["artificial_signal_ratio_range"]),
        alpha_temp=float(reward["alpha_temp"]), alpha_sig=float(reward["alpha_sig"]),
        norm_temp_penalty=_deadband_l2(target, 0, target + 1),                # env 346-350
        norm_sig_penalty=_deadband_l2(norm_reg, 0, 0.75 * norm_reg),          # env 352-356
        penalty_mode=PENALTY_MODES[reward["temp_penalty_mode"]],
        mix_ind_L2=float(mix.get("alpha_ind_L2", 1)), mix_common_L2=float(mix.get("alpha_common_L2", 1)),
        mix_common_max=float(mix.get("alpha_common_max", 0)),
        norm_reg_sig=norm_reg, obs_power_norm=norm_reg * int(cluster["nb_agents"]),   # utils.py:832-841
        nb_agents_comm=int(cluster["nb_agents_comm"]), agents_comm_mode=cluster["agents_comm_mode"],
        comm_defect_prob=float(cluster["comm_defect_prob"]),
        base_power_mode=1 if base_mode == "interpolation" else 0,
        interp_update_period=int(grid["base_power_parameters"].get("interpolation", {}).get("interp_update_period", 300)),
        interp_nb_agents=int(grid["base_power_parameters"].get("interpolation", {}).get("interp_nb_agents", 100)),
        interp_paths={k: v for k, v in grid["base_power_parameters"].get("interpolation", {}).items() if k.startswith("path_")},
        state_properties=dict(env["state_properties"]), message_properties=dict(env["message_properties"]),
    )


# The grid axes monteCarlo/monteCarlo.py:77-115 generates (= monteCarlo/interp_parameters_dict.json of the reference):
# hours in seconds, dates in days after 1 January 2021.  tests/test_config.py checks them against a snapshot.
DEFAULT_INTERP_AXES = {
    "Ua_ratio": [0.9, 1, 1.1], "Cm_ratio": [0.9, 1, 1.1], "Ca_ratio": [0.9, 1, 1.1], "Hm_ratio": [0.9, 1, 1.1],
    "air_temp": [-4, -2, -1, -0.3, 0, 0.3, 1, 2, 4], "mass_temp": [-4, -2, 0, 2, 4],
    "OD_temp": [1, 3, 5, 7, 9, 11, 13, 15], "HVAC_power": [10000, 15000],
    "hour": [h * 3600 for h in (0.0, 3.0, 6.0, 7.0, 7.5, 11.0, 13.0, 16.0, 17.0, 17.5, 21.0, 24 - 1.0 / 3600)],
    "date": [0, 79, 171, 263, 354, 364],
}

INTERP_KEYS = ("Ua_ratio", "Cm_ratio", "Ca_ratio", "Hm_ratio", "air_temp", "mass_temp", "OD_temp", "HVAC_power", "hour", "date")


class InterpolationGridMissing(FileNotFoundError):
    pass


def load_interp_grid(paths: dict):
    """Read the base-power grid the way PowerGrid.__init__ / PowerInterpolator.__init__ do (env 1130-1150;
    monteCarlo/interpolation.py:21-47): a flat .npy, the JSON of axis values and the CSV with the axis order, at the
    paths given in config (relative to the current directory, like the reference).  Returns (values, axes dict)."""
    import csv
    import json
    import os

    import numpy as np
    data, pdict, keys = paths.get("path_datafile"), paths.get("path_parameter_dict"), paths.get("path_dict_keys")
    for p in (data, pdict, keys):
        if not p or not os.path.isfile(p):
            raise InterpolationGridMissing(
                "base_power_mode='interpolation' needs the bang-bang average-power grid (%r). The reference does not ship "
                "monteCarlo/mergedGridSearchResultFinal.npy; regenerate it on the GPU with "
                "`python tools/regenerate_interp_grid.py`, pass interp_grid=(values, axes), or set base_power_mode='constant'." % p)
    with open(pdict) as f:
        axes = json.load(f)
    with open(keys) as f:
        order = list(csv.reader(f))[0]
    if tuple(order) != INTERP_KEYS:
        raise ValueError("interpolation grid axes must be %s, got %s" % (", ".join(INTERP_KEYS), order))
    values = np.load(data, allow_pickle=False)
    return values, {k: axes[k] for k in order}
