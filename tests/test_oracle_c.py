"""The plain-C restatement (oracle/mdr_oracle_c.c) against the reference's golden vectors."""
import numpy as np
import pytest

from oracle import c_port
from oracle import mdr_oracle as mo
from tests import golden_util as gu


@pytest.mark.parametrize("name", gu.names())
def test_c_port_reproduces_reference(name):
    g = gu.Golden(name)
    a = g.a
    ora = mo.OracleEnv(g.config, nb_envs=1)
    ora.seed, ora.episode = g.seed, 0
    if g.interp_grid() is not None:
        ora.interp_grid = mo.InterpGrid(*g.interp_grid())
    ora.load_episode(g.params(), od_table=g.od_table())
    port = c_port.CPort(ora)
    for t in range(g.T):
        od_old, sig_old = ora.OD.copy(), ora.S.copy()
        ora.step(a["actions"][t][None, :])                 # advances the per-env time functions
        port.step_arrays(a["actions"][t][None, :], od_old, ora.solar, sig_old)
        np.testing.assert_array_equal(port.a["on"][0], a["on"][t])
        np.testing.assert_array_equal(port.a["lock"][0], a["lock"][t])
        np.testing.assert_array_equal(port.a["sso"][0], a["sso"][t])
        assert port.a["P"][0] == a["P"][t]
        np.testing.assert_allclose(port.a["Ta"][0], a["Ta"][t], rtol=1e-11)
        np.testing.assert_allclose(port.a["Tm"][0], a["Tm"][t], rtol=1e-11)
        np.testing.assert_allclose(port.a["reward"][0], a["reward"][t], rtol=1e-9, atol=1e-12)


def test_c_port_baseline_runs():
    cfg = gu.reference_env_config()
    cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = 64
    cfg["default_env_prop"]["power_grid_prop"]["base_power_mode"] = "constant"
    cfg["default_env_prop"]["power_grid_prop"]["signal_mode"] = "flat"
    rate, steps, el = c_port.time_baseline(cfg, nb_envs=2, seconds=0.2)
    assert rate > 1e5 and steps > 0
