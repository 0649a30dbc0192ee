"""The single-env dict adapter (mdr_amd.MADemandResponseEnv) against the reference's golden vectors:
same dict surface (21 keys, order, types), values, normStateDict consumption, warnings, deepcopy."""
import copy
import random
import warnings

import numpy as np
import pytest

from tests import golden_util as gu

pytestmark = pytest.mark.gpu

OBS_KEYS = ["OD_temp", "datetime", "house_temp", "house_mass_temp", "hvac_turned_on", "hvac_seconds_since_off",
            "hvac_lockout", "house_target_temp", "house_deadband", "house_Ua", "house_Cm", "house_Ca", "house_Hm",
            "house_solar_gain", "hvac_COP", "hvac_cooling_capacity", "hvac_latent_cooling_fraction",
            "hvac_lockout_duration", "message", "reg_signal", "cluster_hvac_power"]
MSG_KEYS = ["current_temp_diff_to_target", "hvac_seconds_since_off", "hvac_curr_consumption", "hvac_max_consumption",
            "hvac_lockout_duration"]


def norm_state_vector(s, cfg):
    """Test-local restatement of the flat vector utils.normStateDict builds (utils.py:774-878) for the default
    state_properties / message_properties (all False) - the contract with Actor(num_state)."""
    env = cfg["default_env_prop"]
    norm = env["reward_prop"]["norm_reg_sig"]
    nb = env["cluster_prop"]["nb_agents"]
    L = s["hvac_lockout_duration"]
    v = [(s["house_temp"] - 20) / 5, (s["house_mass_temp"] - 20) / 5, (s["house_target_temp"] - 20) / 5,
         s["house_deadband"], s["hvac_cooling_capacity"] / cfg["default_hvac_prop"]["cooling_capacity"],
         1 if s["hvac_turned_on"] else 0, 1 if s["hvac_lockout"] else 0, s["hvac_seconds_since_off"] / L, L / L,
         s["reg_signal"] / (norm * nb), s["cluster_hvac_power"] / (norm * nb)]
    for m in s["message"]:
        v += [m["current_temp_diff_to_target"] / 5, m["hvac_seconds_since_off"] / L,
              m["hvac_curr_consumption"] / norm, m["hvac_max_consumption"] / norm]
    return np.array(v)


def make_env(g):
    import mdr_amd
    env = mdr_amd.MADemandResponseEnv(g.config, device="cuda:0", seed=g.seed, interp_grid=g.interp_grid())
    p = {k: v[0] if v.ndim > 1 else v for k, v in g.params().items()}
    obs = env.load_episode({k: np.asarray(v)[None] if np.asarray(v).ndim == 1 and k not in ("t0", "phase", "ratio") else v
                            for k, v in p.items()}, od_table=g.od_table(), seed=g.seed)
    return env, obs


@pytest.mark.parametrize("name", ["s1_c1_sinusoidals", "s3_c3_heterogeneous", "s9_single_house", "s12_interp_default_like"])
def test_dict_surface_matches_reference(name):
    g = gu.Golden(name)
    a = g.a
    env, obs = make_env(g)
    assert env.nb_agents == g.N and env.agent_ids == list(range(g.N))
    assert list(obs.keys()) == list(range(g.N))
    assert list(obs[0].keys()) == OBS_KEYS
    c = min(10, g.N - 1)
    assert len(obs[0]["message"]) == c and all(list(m.keys()) == MSG_KEYS for m in obs[0]["message"])
    assert obs[0]["cluster_hvac_power"] == 0 and obs[0]["house_solar_gain"] == 0
    steps = g.meta["norm_steps"]
    k = 0
    if steps[0] == 0:
        got = np.array([norm_state_vector(obs[i], g.config) for i in range(g.N)])
        np.testing.assert_allclose(got, a["norm_state"][0], rtol=2e-5, atol=2e-6)
        k = 1
    T = min(g.T, 300)
    for t in range(T):
        obs, rew, done, info = env.step({i: bool(a["actions"][t][i]) for i in range(g.N)})
        assert isinstance(rew[0], float) and done == {i: False for i in range(g.N)}
        assert info == {"cluster_hvac_power": a["P"][t]}
        assert [obs[i]["hvac_turned_on"] for i in range(g.N)] == [bool(x) for x in a["on"][t]]
        assert [obs[i]["hvac_lockout"] for i in range(g.N)] == [bool(x) for x in a["lock"][t]]
        assert [obs[i]["hvac_seconds_since_off"] for i in range(g.N)] == list(a["sso"][t])
        np.testing.assert_allclose([obs[i]["house_temp"] for i in range(g.N)], a["Ta"][t], rtol=1e-5)
        np.testing.assert_allclose([rew[i] for i in range(g.N)], a["reward"][t], rtol=1e-5, atol=1e-5)
        assert obs[0]["reg_signal"] == pytest.approx(a["S"][t + 1], rel=3e-6 if g.interp_grid() else 1e-9)
        assert obs[0]["OD_temp"] == pytest.approx(a["od"][t + 1], abs=5e-6)
        assert obs[0]["house_solar_gain"] == pytest.approx(a["solar"][t], rel=1e-6, abs=1e-4)
        assert obs[0]["datetime"] == env.datetime == env.start_datetime + (t + 1) * env.time_step
        if (t + 1) in steps:
            got = np.array([norm_state_vector(obs[i], g.config) for i in range(g.N)])
            np.testing.assert_allclose(got, a["norm_state"][k], rtol=2e-5, atol=2e-6)
            k += 1
    # neighbours are circular: floor(c/2) before, ceil(c/2) after (env 816-828)
    if g.N == 10:
        assert env.cluster.agent_communicators[1] == [7, 8, 9, 0, 2, 3, 4, 5, 6]
    assert env.power_grid.current_signal == pytest.approx(a["S"][T], rel=3e-6 if g.interp_grid() else 1e-9)
    if g.interp_grid():
        assert env.power_grid.base_power == pytest.approx(a["base_power"][T], rel=3e-6)
    assert env.cluster.houses[0].hvac.seconds_since_off == a["sso"][T - 1][0]


def test_missing_action_warns_and_means_off():
    g = gu.Golden("s1_c1_flat")
    env, _ = make_env(g)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        obs, *_ = env.step({i: True for i in range(1, g.N)})     # house 0 gets no command (env 1026-1032)
    assert any("did not receive any command" in str(x.message) for x in w)
    assert obs[0]["hvac_turned_on"] is False and obs[1]["hvac_turned_on"] is True


def test_reset_resamples_and_python_random_seeds_it():
    import mdr_amd
    cfg = gu.Golden("s1_c1_sinusoidals").config
    random.seed(5)
    e1 = mdr_amd.MADemandResponseEnv(cfg, device="cuda:0")
    o1 = e1.reset()
    random.seed(5)
    e2 = mdr_amd.MADemandResponseEnv(cfg, device="cuda:0")
    o2 = e2.reset()
    assert e1.start_datetime == e2.start_datetime
    assert [o1[i]["house_temp"] for i in o1] == [o2[i]["house_temp"] for i in o2]
    o3 = e1.reset()
    assert [o1[i]["house_temp"] for i in o1] != [o3[i]["house_temp"] for i in o3]
    assert all(o3[i]["house_temp"] >= 20 for i in o3)          # |gauss| start noise is one-sided (utils.py:628-636)


def test_deepcopy_snapshot_like_test_agents_use():
    """utils.test_ppo_agent & co. deep-copy the training env and roll the copy (utils.py:890-1008)."""
    g = gu.Golden("s1_c1_sinusoidals")
    env, obs = make_env(g)
    for t in range(20):
        obs, *_ = env.step({i: bool(g.a["actions"][t][i]) for i in range(g.N)})
    twin = copy.deepcopy(env)
    o_env, *_ = env.step({i: bool(g.a["actions"][20][i]) for i in range(g.N)})
    before = [twin.cluster.houses[i].current_temp for i in range(g.N)]
    o_twin, *_ = twin.step({i: bool(g.a["actions"][20][i]) for i in range(g.N)})
    assert before != [o_twin[i]["house_temp"] for i in range(g.N)]
    assert [o_env[i]["house_temp"] for i in range(g.N)] == [o_twin[i]["house_temp"] for i in range(g.N)]
    assert twin.datetime == env.datetime


@pytest.mark.gpu
def test_norm_states_equals_normStateDict_of_the_dicts():
    """env.norm_states() (device) == the reference's normStateDict applied to the adapter's own dicts, agent by agent."""
    import mdr_amd
    cfg = mdr_amd.default_config()
    cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = 14
    cfg["default_env_prop"]["power_grid_prop"]["base_power_mode"] = "constant"
    env = mdr_amd.MADemandResponseEnv(cfg, seed=11)
    obs = env.reset()
    for t in range(6):
        got = env.norm_states()
        want = np.array([norm_state_vector(obs[i], cfg) for i in range(14)])
        assert got.shape == want.shape == (14, 51)
        np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-6)
        obs, _, _, _ = env.step({i: (t + i) % 3 == 0 for i in obs})


@pytest.mark.parametrize("mode,defect", [("random_fixed", 0.0), ("random_sample", 0.0), ("neighbours", 0.3), ("random_sample", 0.4),
                                         ("closed_groups", 0.5)])
def test_norm_states_equals_normStateDict_under_random_gather(mode, defect):
    """ADVICE / VERDICT r1: the dict's `message` lists and norm_states() show the SAME link defects, `random_sample` senders and
    'random_fixed' table (device draws, one table per episode) - agent by agent, over steps and a reset."""
    import mdr_amd
    cfg = mdr_amd.default_config()
    cl = cfg["default_env_prop"]["cluster_prop"]
    cl.update(nb_agents=15, nb_agents_comm=4, agents_comm_mode=mode, comm_defect_prob=defect)
    cfg["default_env_prop"]["power_grid_prop"]["base_power_mode"] = "constant"
    cfg["noise_house_prop"]["noise_mode"] = "big_noise"
    env = mdr_amd.MADemandResponseEnv(cfg, seed=11)
    obs = env.reset()
    tables = []
    for episode in range(2):
        if mode == "random_fixed":
            tables.append(dict(env.cluster.agent_communicators))
            assert all(len(set(v)) == 4 and i not in v for i, v in tables[-1].items())
        dropped = 0
        for t in range(5):
            got = env.norm_states()
            want = np.array([norm_state_vector(obs[i], cfg) for i in range(15)])
            assert got.shape == want.shape == (15, 11 + 4 * 4)
            np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-6)
            live = np.array([[m["hvac_max_consumption"] != 0 for m in obs[i]["message"]] for i in range(15)])
            assert np.array_equal(live, got[:, 11:].reshape(15, 4, 4)[:, :, 3] != 0)
            dropped += int((~live).sum())
            obs, _, _, _ = env.step({i: (t + i) % 3 == 0 for i in obs})
        assert (dropped > 0) == (defect > 0)
        obs = env.reset()
    if mode == "random_fixed":
        assert tables[0] != tables[1]                       # re-drawn by build_environment (env 849-854)


def test_power_grid_noise_accumulators_match_reference():
    """PowerGrid.cumulated_abs_noise / nb_steps (env 1301-1302; SURVEY 8b attribute list) against the reference's own values."""
    g = gu.Golden("s7_perlin_wiring")
    env, obs = make_env(g)
    assert env.power_grid.nb_steps == 1
    assert env.power_grid.cumulated_abs_noise == pytest.approx(g.a["cumulated_abs_noise"][0], rel=1e-9)
    for t in range(60):
        env.step({i: bool(g.a["actions"][t][i]) for i in range(g.N)})
        assert env.power_grid.cumulated_abs_noise == pytest.approx(g.a["cumulated_abs_noise"][t + 1], rel=1e-9)
    assert env.power_grid.nb_steps == 61
    flat, _ = make_env(gu.Golden("s1_c1_flat"))
    flat.step({i: True for i in range(10)})
    assert flat.power_grid.nb_steps == 0 and flat.power_grid.cumulated_abs_noise == 0


def test_batched_metrics_equal_the_reference_formulas_on_the_dict_surface():
    """mdr_amd.metrics.BatchedMetrics (one update per step for all agents) against the per-agent running sums of the reference's
    metrics.Metrics.update / log (metrics.py:23-47), restated literally on the adapter's dicts."""
    import torch
    import mdr_amd
    from mdr_amd.metrics import BatchedMetrics
    cfg = mdr_amd.default_config()
    cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = 12
    cfg["default_env_prop"]["power_grid_prop"]["base_power_mode"] = "constant"
    cfg["noise_house_prop"]["noise_mode"] = "big_noise"
    env = mdr_amd.MADemandResponseEnv(cfg, seed=21)
    obs = env.reset()
    bm = BatchedMetrics(env._batched)
    lit = dict.fromkeys(["ret", "toff", "terr", "soff", "serr"], 0.0)
    n, T = env.nb_agents, 25
    for t in range(T):
        obs, rew, _, _ = env.step({i: obs[i]["house_temp"] > obs[i]["house_target_temp"] for i in obs})
        bm.update(env._batched, env._batched.t["reward"])
        for k in obs:                                                    # metrics.py:23-30, agent by agent
            lit["toff"] += (obs[k]["house_temp"] - obs[k]["house_target_temp"]) / n
            lit["terr"] += abs(obs[k]["house_temp"] - obs[k]["house_target_temp"]) / n
            lit["ret"] += rew[k] / n
            lit["soff"] += (obs[k]["reg_signal"] - obs[k]["cluster_hvac_power"]) / n ** 2
            lit["serr"] += abs(obs[k]["reg_signal"] - obs[k]["cluster_hvac_power"]) / n ** 2
    log = bm.log(T, T)
    assert list(log.keys()) == ["Mean train return", "Mean temperature offset", "Mean temperature error", "Mean next signal offset",
                                "Mean next signal error", "Mean signal error", "Mean signal offset", "Training steps"]
    assert log["Training steps"] == T
    for key, want in (("Mean train return", lit["ret"]), ("Mean temperature offset", lit["toff"]), ("Mean temperature error", lit["terr"]),
                      ("Mean signal offset", lit["soff"]), ("Mean next signal offset", lit["soff"]), ("Mean signal error", lit["serr"]),
                      ("Mean next signal error", lit["serr"])):
        assert float(log[key][0]) == pytest.approx(want / T, rel=1e-6, abs=1e-9), key
    bm.reset()
    assert float(bm.cumul_avg_reward.abs().sum()) == 0.0
