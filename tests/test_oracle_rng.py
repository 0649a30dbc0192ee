"""Philox4x32-10 known answers (Random123 kat vectors) and the sampling transforms of the reset path."""
import numpy as np
import pytest
from scipy import stats

from oracle import mdr_oracle as mo


def _px(c, k):
    return tuple(int(v) for v in mo.philox4x32_10(c[0], c[1], c[2], c[3], k[0], k[1]))


def test_philox_known_answers():
    assert _px((0, 0, 0, 0), (0, 0)) == (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)
    assert _px((0xFFFFFFFF,) * 4, (0xFFFFFFFF, 0xFFFFFFFF)) == (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)
    assert _px((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0)) == (
        0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)


def test_sampling_distributions():
    n = 200000
    x = mo.philox4x32_10(np.arange(n), 7, 0, 99, 123, 456)
    u = mo.u01(x[0])
    assert 0 < u.min() and u.max() < 1
    assert stats.kstest(u, "uniform").pvalue > 1e-3
    z = mo.gauss01(x[1], x[2])
    assert stats.kstest(z, "norm").pvalue > 1e-3
    tri = mo.triangular_mode1(u, 0.5, 1.5)
    assert stats.kstest(tri, stats.triang(c=0.5, loc=0.5, scale=1.0).cdf).pvalue > 1e-3
    assert np.all(mo.triangular_mode1(u, 1, 1) == 1.0)
    pick = mo.mulhi_pick(x[3], 5)
    assert pick.min() == 0 and pick.max() == 4
    assert stats.chisquare(np.bincount(pick, minlength=5)).pvalue > 1e-3


def test_oracle_reset_distributions_follow_the_reference_rules():
    """utils.apply_house_noise: |gauss| start/target offsets (one-sided), triangular factors; capacity list;
    randrange(364) days + randrange(86400) s after the configured start (utils.py:623-709)."""
    from tests.golden_util import reference_env_config
    cfg = reference_env_config()
    cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = 500
    cfg["default_env_prop"]["power_grid_prop"]["base_power_mode"] = "constant"
    cfg["default_env_prop"]["power_grid_prop"]["signal_mode"] = "flat"
    cfg["default_env_prop"]["power_grid_prop"]["artificial_signal_ratio_range"] = 3
    cfg["noise_house_prop"]["noise_mode"] = "big_noise"
    cfg["noise_hvac_prop"]["noise_mode"] = "big_noise"
    cfg["default_hvac_prop"]["lockout_noise"] = 10
    env = mo.OracleEnv(cfg, nb_envs=200).reset(seed=9, episode=2)
    assert env.Ta.min() >= 20 and env.target.min() >= 20
    assert stats.kstest((env.Ta - 20).ravel() / 5, stats.halfnorm.cdf).pvalue > 1e-3
    assert stats.kstest((env.target - 20).ravel() / 2, stats.halfnorm.cdf).pvalue > 1e-3
    for arr, d in ((env.Ua, 218.0), (env.Cm, 3.45e6), (env.Ca, 9.08e5), (env.Hm, 2.84e3)):
        f = arr.ravel() / d
        assert 0.8 <= f.min() and f.max() <= 1.2
        assert stats.kstest(f, stats.triang(c=0.5, loc=0.8, scale=0.4).cdf).pvalue > 1e-3
    assert set(np.unique(env.capacity)) == {10000.0, 12500.0, 15000.0, 17500.0, 20000.0}
    assert env.lockout.min() == 30 and env.lockout.max() == 50
    t0 = env.t0 - mo.to_epoch_seconds(__import__("datetime").datetime(2021, 1, 1))
    assert t0.min() >= 0 and t0.max() < 364 * 86400
    assert (1 / 3 <= env.ratio).all() and (env.ratio <= 3).all()
    other = mo.OracleEnv(cfg, nb_envs=200).reset(seed=9, episode=3)
    assert not np.array_equal(env.Ta, other.Ta)


def test_affine_form_equals_closed_form():
    """the 2x2 map the kernels apply IS update_temperature's closed form."""
    rng = np.random.default_rng(0)
    n = 4000
    f = lambda: rng.uniform(0.5, 1.5, n)
    Ua, Cm, Ca, Hm = 218 * f(), 3.45e6 * f(), 9.08e5 * f(), 2.84e3 * f()
    Ta, Tm, od = rng.uniform(15, 35, n), rng.uniform(15, 35, n), rng.uniform(20, 40, n)
    Qa = np.where(rng.random(n) < 0.5, -15000 / 1.35, 0.0) + rng.uniform(0, 1500, n)
    for dt in (4.0, 7.0, 60.0):
        a, b = mo.etp_closed_form(Ta, Tm, od, Qa, Ua, Cm, Ca, Hm, dt)
        m00, m01, m10, m11 = mo.etp_affine_coefficients(Ua, Cm, Ca, Hm, dt)
        tinf = od + Qa / Ua
        np.testing.assert_allclose(tinf + m00 * (Ta - tinf) + m01 * (Tm - tinf), a, rtol=0, atol=2e-11)
        np.testing.assert_allclose(tinf + m10 * (Ta - tinf) + m11 * (Tm - tinf), b, rtol=0, atol=2e-11)


def test_hvac_known_answers_from_reference_unit_tests():
    """env/unit_tests_MA_DemandResponse.py:36-77 (TestHVAC): Q, P and the 6-step lockout sequence."""
    assert -15000 / (1 + 0.35) == pytest.approx(-11111.111111111111)
    on, sso, L, dt = np.array([True]), np.array([12]), np.array([12]), 4
    seq = []
    for cmd in (True, False, True, True, True, True):
        on, lock, sso = mo.hvac_transition(on, sso, L, np.array([cmd]), dt)
        seq.append((bool(on[0]), bool(lock[0]), int(sso[0])))
    assert [s[:2] for s in seq] == [(True, False), (False, True), (False, True), (False, True), (True, False), (True, False)]
    assert [s[2] for s in seq][2:] == [4, 8, 0, 0]


def test_calendar_matches_python_datetime():
    import datetime as dt
    rng = np.random.default_rng(1)
    ts = rng.integers(0, 4_000_000_000, 5000)
    c = mo.civil_from_epoch(ts)
    for i in range(0, 5000, 7):
        d = dt.datetime(1970, 1, 1) + dt.timedelta(seconds=int(ts[i]))
        assert (c["year"][i], c["month"][i], c["day"][i], c["hour"][i], c["minute"][i], c["second"][i], c["yday"][i]) == (
            d.year, d.month, d.day, d.hour, d.minute, d.second, d.timetuple().tm_yday)
