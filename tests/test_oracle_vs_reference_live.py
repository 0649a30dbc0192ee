"""Live pin of the oracle: random configurations run through THE REFERENCE ITSELF (imported from /root/reference
with the inert stand-ins of oracle/ref_harness.py) and replayed by the oracle, fp64 against fp64.

Runs only where the reference is mounted (the build container); skipped on the GPU box.  The committed fixtures in
tests/golden/ are 29 hand-picked scenarios; this widens the pin to seeded random corners of the config space (time
step, start date, signal family, penalty mode / weights, noise modes, lockout noise, deadband, solar gain, comm
topology, optional observation columns) without storing more data."""
import importlib.util
import json
import os

import numpy as np
import pytest

from oracle import mdr_oracle as mo
from oracle import ref_harness
from tests import golden_util as gu

pytestmark = pytest.mark.skipif(not ref_harness.available(), reason="the reference is not mounted here")
SCALE = max(1, int(os.environ.get("MDR_FUZZ_SCALE", "1")))   # one-off campaigns: k times as many seeded cases

ENV = "default_env_prop."
PG = ENV + "power_grid_prop."


def _make_golden():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(gu.GOLDEN_DIR, "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _patches(idx):
    rng = np.random.default_rng(31000 + idx)
    N = int(rng.choice([1, 2, 3, 5, 8, 12, 17, 24]))
    signal = str(rng.choice(["flat", "sinusoidals", "regular_steps", "perlin", "amplitude++_perlin", "fast+_perlin"]))
    comm = str(rng.choice(["neighbours", "closed_groups", "random_fixed", "no_message"]))
    nb_comm = int(rng.integers(0, 8))
    if comm == "closed_groups":      # keep to group sizes the reference itself survives (see tests/test_gpu_fuzz.py)
        nb_comm = min(nb_comm, N - 1)
        if N % (nb_comm + 1) == nb_comm and nb_comm > 0:
            comm = "neighbours"
    lock = int(rng.choice([8, 40, 45, 120]))
    p = {
        ENV + "cluster_prop.nb_agents": N,
        ENV + "cluster_prop.nb_agents_comm": nb_comm,
        ENV + "cluster_prop.agents_comm_mode": comm,
        ENV + "cluster_prop.temp_mode": str(rng.choice(["noisy_sinusoidal", "noisy_sinusoidal_hot", "noisy_sinusoidal_heatwave", "constant"])),
        ENV + "time_step": int(rng.choice([1, 4, 4, 7, 30, 60])),
        ENV + "start_datetime_mode": str(rng.choice(["random", "fixed"])),
        ENV + "start_datetime": str(rng.choice(["2021-01-01 00:00:00", "2021-12-31 23:58:00", "2024-02-28 23:59:00", "2021-06-21 07:29:00",
                                                "2021-09-10 17:29:30"])),
        PG + "signal_mode": signal,
        PG + "artificial_signal_ratio_range": float(rng.choice([1, 1, 2, 3])),
        PG + "base_power_parameters.constant.avg_power_per_hvac": float(rng.choice([4200, 1000, 5900])),
        ENV + "reward_prop.temp_penalty_mode": str(rng.choice(["individual_L2", "common_L2", "common_max", "mixture"])),
        ENV + "reward_prop.alpha_temp": float(rng.uniform(0, 2)),
        ENV + "reward_prop.alpha_sig": float(rng.uniform(0, 2)),
        ENV + "reward_prop.temp_penalty_parameters.mixture": {"alpha_ind_L2": float(rng.uniform(0.1, 2)), "alpha_common_L2": float(rng.uniform(0, 2)),
                                                              "alpha_common_max": float(rng.uniform(0, 2))},
        "noise_house_prop.noise_mode": str(rng.choice(["no_noise", "small_noise", "big_noise", "small_start_temp", "big_start_temp"])),
        "noise_hvac_prop.noise_mode": str(rng.choice(["no_noise", "small_noise", "big_noise"])),
        "default_hvac_prop.lockout_duration": lock,
        "default_hvac_prop.lockout_noise": int(rng.integers(0, min(20, lock))),
        "default_house_prop.deadband": float(rng.choice([0, 0, 0.5, 2])),
        "default_house_prop.solar_gain_bool": bool(rng.integers(0, 2)),
        "default_house_prop.target_temp": float(rng.choice([20, 20, 21.5])),
    }
    for k in ("hour", "day", "solar_gain", "thermal", "hvac"):
        p[ENV + "state_properties." + k] = bool(rng.integers(0, 2))
    for k in ("thermal", "hvac"):
        p[ENV + "message_properties." + k] = bool(rng.integers(0, 2))
    policy = str(rng.choice(["bangbang", "onoff", "random:0.5", "random:0.2", "mixed"]))
    return p, "perlin" in signal, policy, int(rng.integers(1, 10 ** 6))


@pytest.mark.parametrize("idx", range(120 * SCALE))
def test_oracle_tracks_the_live_reference(idx):
    mg = _make_golden()
    patches, perlin, policy, seed = _patches(idx)
    known = mg.ref_harness.load_reference()["config_dict"]
    for dotted in list(patches):          # only keys the reference's config really has (modes differ between snapshots)
        node = known
        for part in dotted.split(".")[:-1]:
            node = node[part]
        assert dotted.split(".")[-1] in node, dotted
    T = 30
    a = mg.run_scenario("live_%d" % idx, patches, seed, T, policy, perlin=perlin, norm_steps=(0, 7, -1), save=False)
    meta = json.loads(str(a["meta"]))
    cfg = gu._intkeys(meta["config"])
    N = meta["N"]
    env = mo.OracleEnv(cfg, nb_envs=1)
    env.seed, env.episode = seed, 0
    params = {k: a["p_" + k][None, :] for k in ("Ta", "Tm", "target", "deadband", "Ua", "Cm", "Ca", "Hm", "capacity", "COP", "latent", "lockout")}
    params.update(t0=np.array([a["p_t0"]], dtype=np.int64), phase=np.array([a["p_phase"]]), ratio=np.array([a["p_ratio"]]))
    env.load_episode(params, od_table=a["od"][:, None])
    links = a["links"].astype(np.int64) if "links" in a else None
    assert env.max_power[0] == pytest.approx(float(a["p_max_power"]), rel=1e-14)
    np.testing.assert_allclose(env.S[0], a["S"][0], rtol=1e-11, atol=1e-8)
    k = 0
    np.testing.assert_allclose(env.norm_state(cfg, links)[0], a["norm_state"][k], rtol=1e-10, atol=1e-11)
    for t in range(T):
        r = env.step(a["actions"][t][None, :])
        np.testing.assert_array_equal(env.on[0].astype(np.uint8), a["on"][t], err_msg="on @%d" % t)
        np.testing.assert_array_equal(env.lock[0].astype(np.uint8), a["lock"][t], err_msg="lock @%d" % t)
        np.testing.assert_array_equal(env.sso[0], a["sso"][t], err_msg="sso @%d" % t)
        assert env.P[0] == a["P"][t]
        np.testing.assert_allclose(env.solar[0], a["solar"][t], rtol=1e-12, atol=1e-10)
        np.testing.assert_allclose(env.Ta[0], a["Ta"][t], rtol=1e-10)
        np.testing.assert_allclose(env.Tm[0], a["Tm"][t], rtol=1e-10)
        np.testing.assert_allclose(r[0], a["reward"][t], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(env.S[0], a["S"][t + 1], rtol=1e-11, atol=1e-8)
        if (t + 1) in meta["norm_steps"]:
            k += 1
            np.testing.assert_allclose(env.norm_state(cfg, links)[0], a["norm_state"][k], rtol=1e-10, atol=1e-11)
    assert N == env.N


@pytest.mark.parametrize("idx", range(40 * SCALE))
def test_oracle_tracks_the_live_reference_in_interpolation_mode(idx, tmp_path):
    """base_power_mode='interpolation' (the reference's default): a random grid written in the reference's own file formats is
    read by ITS PowerInterpolator; the oracle's restatement of interpolatePower / interpolateGridFast must give the same base
    power and signal at every update (N <= interp_nb_agents: above it the reference samples houses with `random`)."""
    import csv
    mg = _make_golden()
    rng = np.random.default_rng(47000 + idx)

    def axis(lo, hi, n, must=None):
        v = np.unique(np.round(np.sort(rng.uniform(lo, hi, n)), 3))
        if must is not None:
            v = np.unique(np.concatenate([v, np.asarray(must, dtype=np.float64)]))
        return [float(x) for x in v]
    axes = {"Ua_ratio": axis(0.8, 1.2, int(rng.integers(1, 4)), [1.0]), "Cm_ratio": axis(0.8, 1.2, int(rng.integers(1, 4)), [1.0]),
            "Ca_ratio": axis(0.8, 1.2, int(rng.integers(1, 3)), [1.0]), "Hm_ratio": axis(0.8, 1.2, int(rng.integers(1, 3)), [1.0]),
            "air_temp": axis(-5, 5, int(rng.integers(2, 5))), "mass_temp": axis(-5, 5, int(rng.integers(2, 4))),
            "OD_temp": axis(0, 20, int(rng.integers(2, 5))),
            "HVAC_power": [float(x) for x in sorted(set(rng.choice([10000, 12500, 15000, 17500, 20000], int(rng.integers(1, 4)))))],
            "hour": axis(0, 86399, int(rng.integers(2, 5))), "date": axis(0, 364, int(rng.integers(2, 4)))}
    dims = [len(v) for v in axes.values()]
    values = np.round(rng.uniform(0, 6000, int(np.prod(dims))))
    np.save(tmp_path / "grid.npy", values)
    (tmp_path / "params.json").write_text(json.dumps(axes))
    with open(tmp_path / "keys.csv", "w") as f:
        csv.writer(f).writerow(list(axes.keys()))
    ip = PG + "base_power_parameters.interpolation."
    N = int(rng.choice([1, 5, 12, 30]))
    patches = {
        PG + "base_power_mode": "interpolation", ip + "path_datafile": str(tmp_path / "grid.npy"),
        ip + "path_parameter_dict": str(tmp_path / "params.json"), ip + "path_dict_keys": str(tmp_path / "keys.csv"),
        ENV + "cluster_prop.nb_agents": N,
        ENV + "time_step": int(rng.choice([4, 7, 30, 60, 150, 300, 400])),
        ENV + "start_datetime_mode": str(rng.choice(["random", "fixed"])),
        ENV + "start_datetime": str(rng.choice(["2021-01-01 00:00:00", "2021-12-31 23:58:00", "2024-02-29 23:50:00", "2024-12-31 12:00:00",
                                                "2021-06-21 07:29:00"])),
        ENV + "cluster_prop.temp_mode": str(rng.choice(["noisy_sinusoidal", "noisy_sinusoidal_hot", "noisy_sinusoidal_heatwave", "constant"])),
        PG + "signal_mode": str(rng.choice(["flat", "sinusoidals", "regular_steps", "perlin"])),
        "noise_house_prop.noise_mode": str(rng.choice(["no_noise", "small_noise", "big_noise"])),
        "noise_hvac_prop.noise_mode": str(rng.choice(["no_noise", "small_noise", "big_noise"])),
    }
    seed = int(rng.integers(1, 10 ** 6))
    T = 90
    a = mg.run_scenario("live_interp_%d" % idx, patches, seed, T, "mixed", perlin="perlin" in patches[PG + "signal_mode"],
                        norm_steps=(0,), extra={"live": np.array(1)}, save=False)
    meta = json.loads(str(a["meta"]))
    cfg = gu._intkeys(meta["config"])
    env = mo.OracleEnv(cfg, nb_envs=1)
    env.seed, env.episode = seed, 0
    env.interp_grid = mo.InterpGrid(values, axes)
    params = {k: a["p_" + k][None, :] for k in ("Ta", "Tm", "target", "deadband", "Ua", "Cm", "Ca", "Hm", "capacity", "COP", "latent", "lockout")}
    params.update(t0=np.array([a["p_t0"]], dtype=np.int64), phase=np.array([a["p_phase"]]), ratio=np.array([a["p_ratio"]]))
    env.load_episode(params, od_table=a["od"][:, None])
    np.testing.assert_allclose(env.base_power[0], a["base_power"][0], rtol=1e-11)
    np.testing.assert_allclose(env.S[0], a["S"][0], rtol=1e-11, atol=1e-8)
    for t in range(T):
        r = env.step(a["actions"][t][None, :])
        np.testing.assert_allclose(env.base_power[0], a["base_power"][t + 1], rtol=1e-11, err_msg="base power @%d" % t)
        np.testing.assert_allclose(env.S[0], a["S"][t + 1], rtol=1e-11, atol=1e-8)
        np.testing.assert_allclose(env.Ta[0], a["Ta"][t], rtol=1e-10)
        np.testing.assert_allclose(r[0], a["reward"][t], rtol=1e-9, atol=1e-12)
