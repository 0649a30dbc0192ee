"""Pin the CPU oracle against the golden vectors captured from the reference.

fp64 against fp64: every quantity of every step must agree to ~1e-10; integer state exactly."""
import numpy as np
import pytest

from oracle import mdr_oracle as mo
from tests import golden_util as gu


def replay(g, use_perlin_seed=True):
    env = mo.OracleEnv(g.config, nb_envs=1)
    env.seed, env.episode = g.seed, 0        # selects the Perlin lattice the fixture was made with
    if g.interp_grid() is not None:
        env.interp_grid = mo.InterpGrid(*g.interp_grid())
    env.load_episode(g.params(), od_table=g.od_table())
    return env


@pytest.mark.parametrize("name", gu.names())
def test_oracle_reproduces_reference(name):
    g = gu.Golden(name)
    a = g.a
    env = replay(g)
    assert env.max_power[0] == pytest.approx(float(a["p_max_power"]), rel=1e-14)
    np.testing.assert_allclose(env.S[0], a["S"][0], rtol=1e-11, atol=1e-9)
    worst = dict(Ta=0.0, Tm=0.0, reward=0.0, S=0.0)
    for t in range(g.T):
        r = env.step(a["actions"][t][None, :])
        np.testing.assert_array_equal(env.on[0].astype(np.uint8), a["on"][t], err_msg="on @%d" % t)
        np.testing.assert_array_equal(env.lock[0].astype(np.uint8), a["lock"][t], err_msg="lock @%d" % t)
        np.testing.assert_array_equal(env.sso[0], a["sso"][t], err_msg="sso @%d" % t)
        assert env.P[0] == a["P"][t]
        np.testing.assert_allclose(env.solar[0], a["solar"][t], rtol=1e-12, atol=1e-10)
        np.testing.assert_allclose(env.Ta[0], a["Ta"][t], rtol=1e-10, atol=0)
        np.testing.assert_allclose(env.Tm[0], a["Tm"][t], rtol=1e-10, atol=0)
        np.testing.assert_allclose(r[0], a["reward"][t], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(env.S[0], a["S"][t + 1], rtol=1e-11, atol=1e-8)
        if "base_power" in a:
            np.testing.assert_allclose(env.base_power[0], a["base_power"][t + 1], rtol=1e-11)
        worst["Ta"] = max(worst["Ta"], float(np.max(np.abs(env.Ta[0] / a["Ta"][t] - 1))))
    assert worst["Ta"] < 1e-10


def test_anchor_digits_from_survey():
    """SURVEY.md 8c anchor: seed 1, N=10, sinusoidals, bang-bang, 1000 steps."""
    g = gu.Golden("s1_c1_sinusoidals")
    a = g.a
    assert int(a["p_t0"]) == mo.to_epoch_seconds(__import__("datetime").datetime(2021, 4, 22, 20, 45, 47))
    assert float(a["p_max_power"]) == 60000.0
    assert a["od"][0] == pytest.approx(28.714753130, abs=1e-9)
    assert a["S"][0] == pytest.approx(51113.936794, abs=1e-6)
    assert a["Ta"][0][0] == pytest.approx(20.748125187307, abs=1e-12)
    assert a["reward"][999][0] == pytest.approx(-0.936563432778, abs=1e-12)
    assert float(a["reward"].sum()) == pytest.approx(-22544.136763550, abs=1e-6)
    assert a["norm_state"].shape[-1] == 47


def test_bangbang_rule_matches_recorded_actions():
    """agents/bangbang_controllers.py:41-61 restated in OracleEnv.bangbang_actions."""
    g = gu.Golden("s1_c1_sinusoidals")
    env = replay(g)
    for t in range(200):
        np.testing.assert_array_equal(env.bangbang_actions()[0].astype(np.uint8), g.a["actions"][t])
        env.step(g.a["actions"][t][None, :])


@pytest.mark.parametrize("name,rule", [("s14_controller_deadband", "deadband_actions"), ("s14_controller_basic", "deadband_actions"),
                                       ("s14_controller_always_on", "always_on_actions"),
                                       ("s14_controller_greedy_myopic", "greedy_myopic_actions")])
def test_other_controllers_match_recorded_actions(name, rule):
    """agents/bangbang_controllers.py: DeadbandBangBangController 13-38, BasicController 64-88 (the same rule), AlwaysOnController
    1-10, and agents/greedy_myopic_controller.py (the centralised ranking + budget pass), restated in OracleEnv and held to the
    actions the reference's own controller objects took in the S14 fixtures."""
    g = gu.Golden(name)
    env = replay(g)
    held = 0
    for t in range(g.T):
        act = getattr(env, rule)()[0]
        np.testing.assert_array_equal(act.astype(np.uint8), g.a["actions"][t])
        if rule == "deadband_actions":
            half = env.deadband[0] / 2
            held += int(np.sum((env.Ta[0] >= env.target[0] - half) & (env.Ta[0] <= env.target[0] + half)))
        env.step(g.a["actions"][t][None, :])
    if rule == "deadband_actions":
        assert held > 50          # the hold band (keep what the HVAC is doing) is visited


def test_dynamic_obs_columns_match_normStateDict():
    """The 7 columns the kernels emit are entries 0,1,5,6,7,9,10 of the default normStateDict vector
    (utils.py:800-841; entry 2 = target, 3 = deadband, 4 = capacity ratio, 8 = constant 1)."""
    for name in ("s1_c1_sinusoidals", "s3_c3_heterogeneous"):
        g = gu.Golden(name)
        env = replay(g)
        steps = g.meta["norm_steps"]
        k = 0
        if steps[0] == 0:
            cols = env.dynamic_obs()[:, 0, :]
            ns = g.a["norm_state"][k]
            np.testing.assert_allclose(cols.T, ns[:, [0, 1, 5, 6, 7, 9, 10]], rtol=1e-11, atol=1e-12)
            k += 1
        for t in range(g.T):
            env.step(g.a["actions"][t][None, :])
            if (t + 1) in steps:
                cols = env.dynamic_obs()[:, 0, :]
                ns = g.a["norm_state"][k]
                np.testing.assert_allclose(cols.T, ns[:, [0, 1, 5, 6, 7, 9, 10]], rtol=1e-10, atol=1e-11)
                k += 1
        assert k == len(steps)
