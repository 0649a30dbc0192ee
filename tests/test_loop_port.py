"""The loop-form CPU port (bench.py's cpu_baseline) against the reference's golden vectors."""
import numpy as np
import pytest

from oracle import loop_port
from tests import golden_util as gu


@pytest.mark.parametrize("name", ["s1_c1_sinusoidals", "s1_c1_flat", "s1_c1_regular_steps", "s3_c3_heterogeneous", "s6_dt7_lockout45"])
def test_loop_port_reproduces_reference(name):
    g = gu.Golden(name)
    a = g.a
    p = {k[2:]: a[k] for k in a if k.startswith("p_")}
    env = loop_port.LoopPortEnv(g.config, params=p, od_table=a["od"])
    T = min(g.T, 200)
    for t in range(T):
        obs, rew, done, info = env.step({i: bool(a["actions"][t][i]) for i in range(g.N)})
        assert info["cluster_hvac_power"] == a["P"][t]
        np.testing.assert_allclose([obs[i]["house_temp"] for i in range(g.N)], a["Ta"][t], rtol=1e-11)
        np.testing.assert_allclose([rew[i] for i in range(g.N)], a["reward"][t], rtol=1e-9, atol=1e-12)
        np.testing.assert_array_equal([obs[i]["hvac_seconds_since_off"] for i in range(g.N)], a["sso"][t])
        assert obs[0]["reg_signal"] == pytest.approx(a["S"][t + 1], rel=1e-11)
    assert len(obs[0]) == 21 and len(obs[0]["message"]) == min(10, g.N - 1)


def test_loop_port_bangbang_closed_loop_matches_recorded_actions():
    g = gu.Golden("s1_c1_sinusoidals")
    p = {k[2:]: g.a[k] for k in g.a if k.startswith("p_")}
    env = loop_port.LoopPortEnv(g.config, params=p, od_table=g.a["od"])
    obs = env.observations()
    for t in range(100):
        act = loop_port.bangbang(obs)
        np.testing.assert_array_equal([act[i] for i in range(g.N)], g.a["actions"][t].astype(bool))
        obs, *_ = env.step(act)
