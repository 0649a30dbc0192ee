"""GPU parity tests: the HIP path (through the C ABI) against the reference's golden vectors and the oracle.

Tolerances (BASELINE.json north_star: "within 1e-5 relative fp32 on temperatures/rewards"):
  * hvac_turned_on / hvac_lockout / seconds_since_off / cluster_hvac_power : bit-exact
  * house_temp, house_mass_temp (deg C)                                     : |d| <= 1e-5 * |T|
  * rewards : |d| <= 1e-5 * |r| + 1e-5.  The absolute floor is in normalised reward units (1 == a 1 degC
    temperature error, env 346-350); a pure relative bound is ill-posed where the penalty crosses zero.
  * regulation signal (fp64 tables) : 1e-9 relative ; solar gain / OD temperature (fp32 tables) : 1e-6
Actions are always the RECORDED sequence: a bang-bang threshold can flip on a 1e-7 difference and the
trajectories would then legitimately diverge (SURVEY.md section 7 "threshold-driven divergence").
"""
import numpy as np
import pytest
import torch

from tests import golden_util as gu

pytestmark = pytest.mark.gpu

T_RTOL = 1e-5
R_RTOL, R_ATOL = 1e-5, 1e-5


def _mdr():
    import mdr_amd
    return mdr_amd


def run_fixture(g, table_steps=64):
    mdr = _mdr()
    env = mdr.BatchedDemandResponseEnv(g.config, nb_envs=1, device="cuda:0", seed=g.seed, table_steps=table_steps,
                                       interp_grid=g.interp_grid())
    env.load_episode(g.params(), od_table=g.od_table(), seed=g.seed, episode=0)
    return env


@pytest.mark.parametrize("name", gu.names())
def test_hip_path_reproduces_reference_golden(name):
    g = gu.Golden(name)
    a = g.a
    env = run_fixture(g)
    # interpolated base power is built from fp32 house temperatures: the signal inherits ~1e-6 from them
    s_rtol = 3e-6 if g.interp_grid() is not None else 1e-9
    assert env.t["max_power"][0].item() == pytest.approx(float(a["p_max_power"]), rel=1e-12)
    assert env.reg_signal()[0].item() == pytest.approx(float(a["S"][0]), rel=s_rtol, abs=1e-6)
    acts = torch.from_numpy(a["actions"]).to("cuda:0")
    hist = {k: [] for k in ("Ta", "Tm", "sso", "flags", "reward", "P", "S", "solar")}
    for t in range(g.T):
        obs, reward, done, info = env.step(acts[t][None, :])
        hist["Ta"].append(env.house_temp()[0])
        hist["Tm"].append(env.house_mass_temp()[0])
        hist["sso"].append(env.t["sso"][0].clone())
        hist["flags"].append(env.t["flags"][0].clone())
        hist["reward"].append(reward[0].clone())
        hist["P"].append(info["cluster_hvac_power"][0].clone())
        hist["S"].append(env.reg_signal()[0].clone())
        hist["solar"].append(env.solar_gain()[0].clone())
        assert not bool(done.any())
    h = {k: torch.stack(v).cpu().numpy() for k, v in hist.items()}
    np.testing.assert_array_equal(h["flags"] & 1, a["on"])
    np.testing.assert_array_equal((h["flags"] >> 1) & 1, a["lock"])
    np.testing.assert_array_equal(h["sso"], a["sso"])
    np.testing.assert_array_equal(h["P"], a["P"])
    np.testing.assert_allclose(h["S"], a["S"][1:], rtol=s_rtol, atol=1e-6)
    np.testing.assert_allclose(h["solar"], a["solar"], rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(h["Ta"], a["Ta"], rtol=T_RTOL, atol=0)
    np.testing.assert_allclose(h["Tm"], a["Tm"], rtol=T_RTOL, atol=0)
    np.testing.assert_allclose(h["reward"], a["reward"], rtol=R_RTOL, atol=R_ATOL)


def test_rewards_hold_the_pure_relative_bound_away_from_the_zero_crossing():
    """north_star says "1e-5 relative on temperatures / rewards".  The absolute floor of R_ATOL exists for rewards whose temperature
    penalty is near zero - (T - deadband edge)^2 amplifies the temperature's relative error by 2 T / (T - edge) - where a relative bound is
    ill-posed.  Over every reward of every reference golden (r03, 131,580 rewards of 33 fixtures): |r| >= 1 - a temperature at least
    one degree past its band - holds the PURE relative 1e-5 (worst 6.3e-6 over 103,997 of them); below that the error is at most
    6.6e-6 ABSOLUTE, inside the floor the other parity tests allow (the worst relative error of a reward in 0.1 .. 1 is 1.14e-5: a
    house 0.55 degrees past its target, amplification 77)."""
    n_rel = n_abs = 0
    worst_rel = worst_abs = 0.0
    for name in gu.names():
        g = gu.Golden(name)
        if g.interp_grid() is not None:
            continue      # interpolated base power: the signal term inherits the fp32 state (tests/interp_util.py holds those)
        env = run_fixture(g)
        acts = torch.from_numpy(g.a["actions"]).to("cuda:0")
        got = []
        for t in range(g.T):
            _, reward, _, _ = env.step(acts[t][None, :])
            got.append(reward[0].clone())
        got = torch.stack(got).cpu().numpy().astype(np.float64)
        want = g.a["reward"].astype(np.float64)
        big = np.abs(want) >= 1.0
        if big.any():
            worst_rel = max(worst_rel, float((np.abs(got[big] - want[big]) / np.abs(want[big])).max()))
        if (~big).any():
            worst_abs = max(worst_abs, float(np.abs(got[~big] - want[~big]).max()))
        n_rel += int(big.sum())
        n_abs += int((~big).sum())
    assert n_rel > 100000 and worst_rel <= 1e-5, (n_rel, worst_rel)
    assert n_abs > 1000 and worst_abs <= 1e-5, (n_abs, worst_abs)


@pytest.mark.parametrize("table_steps", [1, 7, 64])
def test_time_table_chunking_is_invisible(table_steps):
    """Refilling the per-env time tables every K steps must not change a single bit."""
    g = gu.Golden("s3_c3_heterogeneous")
    ref = run_fixture(g, table_steps=1024)
    env = run_fixture(g, table_steps=table_steps)
    acts = torch.from_numpy(g.a["actions"]).to("cuda:0")
    for t in range(150):
        ref.step(acts[t][None, :])
        env.step(acts[t][None, :])
    for k in ("Ta", "Tm", "sso", "flags", "reward", "obs", "P"):
        assert torch.equal(ref.t[k], env.t[k]), k


def test_observation_columns_match_normStateDict():
    """The 7 planes the kernel writes == entries 0,1,5,6,7,9,10 of utils.normStateDict (utils.py:800-841)."""
    for name in ("s1_c1_sinusoidals", "s3_c3_heterogeneous"):
        g = gu.Golden(name)
        env = run_fixture(g)
        steps = g.meta["norm_steps"]
        cols = [0, 1, 5, 6, 7, 9, 10]
        k = 0
        acts = torch.from_numpy(g.a["actions"]).to("cuda:0")
        if steps[0] == 0:
            np.testing.assert_allclose(env.t["obs"][:, 0, :].cpu().numpy().T, g.a["norm_state"][0][:, cols], rtol=2e-5, atol=2e-6)
            k = 1
        for t in range(g.T):
            obs, *_ = env.step(acts[t][None, :])
            if (t + 1) in steps:
                np.testing.assert_allclose(obs[:, 0, :].cpu().numpy().T, g.a["norm_state"][k][:, cols], rtol=2e-5, atol=2e-6)
                k += 1
        assert k == len(steps)


def _oracle_pair(cfg, E, seed, **kw):
    from oracle import mdr_oracle as mo
    mdr = _mdr()
    env = mdr.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=seed, **kw)
    env.reset(episode=0)
    ora = mo.OracleEnv(cfg, nb_envs=E).reset(seed=seed, episode=0)
    return env, ora


def _cfg(n, **patches):
    mdr = _mdr()
    cfg = mdr.default_config()
    cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = n
    cfg["default_env_prop"]["power_grid_prop"]["base_power_mode"] = "constant"
    for dotted, v in patches.items():
        node = cfg
        parts = dotted.split(".")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = v
    return cfg


def _compare_state(env, ora, reward=None, r_ref=None):
    flags = env.t["flags"].cpu().numpy()
    np.testing.assert_array_equal((flags & 1).astype(bool), ora.on)
    np.testing.assert_array_equal((flags & 2).astype(bool), ora.lock)
    np.testing.assert_array_equal(env.t["sso"].cpu().numpy(), ora.sso)
    np.testing.assert_array_equal(env.t["P"].cpu().numpy(), ora.P)
    np.testing.assert_allclose(env.house_temp().cpu().numpy(), ora.Ta, rtol=T_RTOL, atol=0)
    np.testing.assert_allclose(env.house_mass_temp().cpu().numpy(), ora.Tm, rtol=T_RTOL, atol=0)
    np.testing.assert_allclose(env.reg_signal().cpu().numpy(), ora.S, rtol=1e-9, atol=1e-6)
    np.testing.assert_allclose(env.od_temp().cpu().numpy(), ora.OD, rtol=0, atol=5e-6)
    if reward is not None:
        np.testing.assert_allclose(reward.cpu().numpy(), r_ref, rtol=R_RTOL, atol=R_ATOL)


# every kernel family: sub-wave groups (N<=64), fused scalar, fused float4 with 1..4 tiles / 64..256 threads, split
SHAPES = [(37, 1), (33, 5), (16, 10), (9, 50), (5, 64), (6, 100), (4, 200), (5, 256), (3, 508), (3, 1021),
          (4, 1024), (2, 2048), (2, 3000), (2, 4096), (2, 4100), (1, 10001)]


@pytest.mark.parametrize("E,N", SHAPES)
def test_device_reset_and_steps_match_oracle(E, N):
    """Device-sampled episode (Philox streams) + 40 steps with random actions, every kernel family."""
    cfg = _cfg(N, **{"noise_house_prop.noise_mode": "big_noise", "noise_hvac_prop.noise_mode": "big_noise",
                     "default_hvac_prop.lockout_noise": 12, "default_house_prop.deadband": 0.5,
                     "default_env_prop.power_grid_prop.signal_mode": "perlin",
                     "default_env_prop.power_grid_prop.artificial_signal_ratio_range": 2,
                     "default_env_prop.cluster_prop.temp_mode": "noisy_sinusoidal_heatwave"})
    env, ora = _oracle_pair(cfg, E, seed=2024 + N)
    # sampled parameters, draw by draw
    for name, ref in (("Ua", ora.Ua), ("Cm", ora.Cm), ("Ca", ora.Ca), ("Hm", ora.Hm), ("capacity", ora.capacity)):
        np.testing.assert_allclose(env.t[name].cpu().numpy(), ref, rtol=2e-7)
    np.testing.assert_array_equal(env.t["lockout"].cpu().numpy(), ora.lockout)
    np.testing.assert_array_equal(env.t["t0"].cpu().numpy(), ora.t0)
    np.testing.assert_allclose(env.t["ratio"].cpu().numpy(), ora.ratio, rtol=1e-12)
    np.testing.assert_allclose(env.t["max_power"].cpu().numpy(), ora.max_power, rtol=1e-12)
    np.testing.assert_allclose(env.target_temp().cpu().numpy(), ora.target, rtol=2e-7)
    _compare_state(env, ora)
    rng = np.random.default_rng(N)
    for t in range(40):
        act = (rng.random((E, N)) < 0.55).astype(np.uint8)
        obs, reward, done, info = env.step(torch.from_numpy(act).to("cuda:0"))
        r_ref = ora.step(act)
        _compare_state(env, ora, reward, r_ref)
    np.testing.assert_allclose(obs.cpu().numpy(), ora.dynamic_obs(), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("mode", ["individual_L2", "common_L2", "common_max", "mixture"])
@pytest.mark.parametrize("N", [48, 1024, 5000])
def test_penalty_modes(mode, N):
    cfg = _cfg(N, **{"noise_house_prop.noise_mode": "big_noise", "default_house_prop.deadband": 1,
                     "default_env_prop.reward_prop.temp_penalty_mode": mode,
                     "default_env_prop.reward_prop.alpha_temp": 0.6, "default_env_prop.reward_prop.alpha_sig": 1.7,
                     "default_env_prop.power_grid_prop.signal_mode": "sinusoidals"})
    cfg["default_env_prop"]["reward_prop"]["temp_penalty_parameters"]["mixture"] = {
        "alpha_ind_L2": 1, "alpha_common_L2": 2, "alpha_common_max": 0.5}
    env, ora = _oracle_pair(cfg, 3, seed=99)
    rng = np.random.default_rng(1)
    for t in range(25):
        act = (rng.random((3, N)) < 0.5).astype(np.uint8)
        _, reward, _, _ = env.step(torch.from_numpy(act).to("cuda:0"))
        _compare_state(env, ora, reward, ora.step(act))


@pytest.mark.parametrize("signal", ["flat", "sinusoidals", "regular_steps", "perlin", "fast++_perlin"])
def test_signal_modes_long_horizon(signal):
    """Signals over 3 simulated hours crossing midnight (fixed start 22:30), fp64 tables vs oracle."""
    cfg = _cfg(64, **{"default_env_prop.power_grid_prop.signal_mode": signal,
                      "default_env_prop.start_datetime_mode": "fixed",
                      "default_env_prop.start_datetime": "2021-08-30 22:30:00",
                      "default_env_prop.time_step": 30})
    env, ora = _oracle_pair(cfg, 4, seed=5)
    act = np.ones((4, 64), dtype=np.uint8)
    dev_act = torch.from_numpy(act).to("cuda:0")
    for t in range(360):
        env.step(dev_act)
        ora.step(act)
        if t % 20 == 0:
            np.testing.assert_allclose(env.reg_signal().cpu().numpy(), ora.S, rtol=1e-9, atol=1e-6)
    _compare_state(env, ora)


def test_fused_bangbang_is_self_consistent_and_matches_oracle():
    """The in-kernel bang-bang rule: feed the actions the kernel reports to the oracle and compare; the rule
    itself (house_temp > target on the PRE-step observation) is checked against the kernel's own temperatures."""
    cfg = _cfg(1024, **{"noise_house_prop.noise_mode": "big_noise", "noise_hvac_prop.noise_mode": "big_noise"})
    env, ora = _oracle_pair(cfg, 4, seed=31)
    for t in range(60):
        pre = (env.t["Ta"] > env.t["target"]).to(torch.uint8)
        _, reward, _, _ = env.step_bangbang()
        taken = env.t["actions"]
        assert torch.equal(pre, taken)
        _compare_state(env, ora, reward, ora.step(taken.cpu().numpy()))
    # and the oracle's own bang-bang decision agrees wherever the temperature is not within fp32 noise of the target
    margin = np.abs(ora.Ta - ora.target) > 1e-4
    np.testing.assert_array_equal(ora.bangbang_actions()[margin], (env.t["Ta"] > env.t["target"]).cpu().numpy()[margin])


def test_rollout_equals_single_steps():
    cfg = _cfg(1024, **{"noise_house_prop.noise_mode": "big_noise"})
    mdr = _mdr()
    a = mdr.BatchedDemandResponseEnv(cfg, nb_envs=8, device="cuda:0", seed=7, table_steps=16)
    b = mdr.BatchedDemandResponseEnv(cfg, nb_envs=8, device="cuda:0", seed=7, table_steps=16)
    a.reset(episode=0)
    b.reset(episode=0)
    a.rollout(50)
    for _ in range(50):
        b.step_bangbang()
    for k in ("Ta", "Tm", "sso", "flags", "reward", "obs", "P", "actions"):
        assert torch.equal(a.t[k], b.t[k]), k
    assert a.steps_taken == b.steps_taken == 50


def test_env_partition_invariance():
    """C4-style sharding: envs [0,E) in one batch == the same envs in two batches with env_offset (no collective)."""
    cfg = _cfg(256, **{"noise_house_prop.noise_mode": "big_noise", "noise_hvac_prop.noise_mode": "small_noise",
                       "default_env_prop.power_grid_prop.artificial_signal_ratio_range": 3})
    mdr = _mdr()
    whole = mdr.BatchedDemandResponseEnv(cfg, nb_envs=12, device="cuda:0", seed=77)
    lo = mdr.BatchedDemandResponseEnv(cfg, nb_envs=5, device="cuda:0", seed=77, env_offset=0)
    hi = mdr.BatchedDemandResponseEnv(cfg, nb_envs=7, device="cuda:0", seed=77, env_offset=5)
    for e in (whole, lo, hi):
        e.reset(episode=3)
        e.rollout(80)
    for k in ("Ta", "Tm", "sso", "flags", "reward", "P", "t0"):
        assert torch.equal(whole.t[k][:5], lo.t[k]), k
        assert torch.equal(whole.t[k][5:], hi.t[k]), k
    assert torch.equal(whole.t["obs"][:, :5], lo.t["obs"]) and torch.equal(whole.t["obs"][:, 5:], hi.t["obs"])


def test_deepcopy_is_independent_snapshot():
    import copy
    cfg = _cfg(128)
    mdr = _mdr()
    env = mdr.BatchedDemandResponseEnv(cfg, nb_envs=3, device="cuda:0", seed=1)
    env.reset()
    env.rollout(70)       # past one table refill (K=64)
    twin = copy.deepcopy(env)
    env.rollout(30)
    before = twin.t["Ta"].clone()
    twin.rollout(30)
    assert not torch.equal(before, twin.t["Ta"])
    for k in ("Ta", "Tm", "sso", "flags", "reward", "P"):
        assert torch.equal(env.t[k], twin.t[k]), k


def test_errors_map_to_reference_exceptions():
    mdr = _mdr()
    with pytest.raises(ValueError):
        mdr.BatchedDemandResponseEnv(_cfg(8, **{"default_env_prop.power_grid_prop.signal_mode": "square"}), device="cuda:0")
    with pytest.raises(ValueError):
        mdr.BatchedDemandResponseEnv(_cfg(8, **{"default_env_prop.reward_prop.temp_penalty_mode": "L1"}), device="cuda:0")
    with pytest.raises(ValueError):   # HVAC.__init__: latent fraction outside [0, 1] (env 438-443)
        mdr.BatchedDemandResponseEnv(_cfg(8, **{"default_hvac_prop.latent_cooling_fraction": 1.5}), device="cuda:0")
    with pytest.raises(ValueError):   # lockout_duration + noise could go negative (env 444-449)
        mdr.BatchedDemandResponseEnv(_cfg(8, **{"default_hvac_prop.lockout_noise": 50}), device="cuda:0")
    env = mdr.BatchedDemandResponseEnv(_cfg(8), device="cuda:0")
    with pytest.raises(RuntimeError):  # step before reset
        env.step_bangbang()


def test_long_horizon_training_episode_length():
    """One full training episode (16,384 steps = 18.2 simulated hours, config.py:572-587) in fp32 against the fp64
    oracle with the oracle's own mostly-bang-bang action record: the error must stay bounded, not grow."""
    E, N, T = 4, 256, 16384
    cfg = _cfg(N, **{"noise_house_prop.noise_mode": "big_noise", "noise_hvac_prop.noise_mode": "big_noise",
                     "default_env_prop.power_grid_prop.signal_mode": "perlin"})
    env, ora = _oracle_pair(cfg, E, seed=404)
    rng = np.random.default_rng(4)
    acts = np.empty((T, E, N), dtype=np.uint8)
    worst_T, worst_r = 0.0, 0.0
    dev_acts = torch.empty((T, E, N), dtype=torch.uint8, device="cuda:0")
    chunk = 1024
    for c0 in range(0, T, chunk):
        ref_Ta, ref_r = [], []
        for t in range(c0, c0 + chunk):           # oracle first: its decisions become the recorded actions
            a = ora.bangbang_actions() ^ (rng.random((E, N)) < 0.1)
            acts[t] = a
            ref_r.append(ora.step(a).copy())
            ref_Ta.append(ora.Ta.copy())
        dev_acts[c0:c0 + chunk] = torch.from_numpy(acts[c0:c0 + chunk]).cuda()
        for t in range(c0, c0 + chunk):
            _, reward, _, _ = env.step(dev_acts[t])
            if (t + 1) % 256 == 0:
                i = t - c0
                worst_T = max(worst_T, float(np.max(np.abs(env.house_temp().cpu().numpy() / ref_Ta[i] - 1))))
                worst_r = max(worst_r, float(np.max(np.abs(reward.cpu().numpy() - ref_r[i]) / (1e-5 + 1e-5 * np.abs(ref_r[i])))))
        _compare_state(env, ora, reward, ref_r[-1])
    assert worst_T < 5e-6, worst_T           # observed ~5e-7
    assert worst_r < 1.0, worst_r


@pytest.mark.parametrize("E,N", [(5, 40), (3, 100), (4, 256), (2, 1024), (2, 5000)])
def test_interpolated_base_power_matches_oracle(E, N):
    """base_power_mode='interpolation' (the reference's default): device interpolatePower incl. the 100-house sampling
    for N > interp_nb_agents (Philox stream 8) against the oracle, across three update boundaries (every 75 steps)."""
    from oracle import mdr_oracle as mo
    values, axes = gu.Golden("s12_interp_default_like").interp_grid()
    cfg = _cfg(N, **{"noise_house_prop.noise_mode": "small_noise", "noise_hvac_prop.noise_mode": "big_noise",
                     "default_env_prop.power_grid_prop.base_power_mode": "interpolation",
                     "default_env_prop.power_grid_prop.signal_mode": "sinusoidals"})
    from tests.interp_util import DeviceFedOracle, base_power_bound
    mdr = _mdr()
    env = mdr.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=77, interp_grid=(values, axes))
    env.reset(episode=2)
    grid = mo.InterpGrid(values, axes)
    ora = mo.OracleEnv(cfg, nb_envs=E)
    fed = DeviceFedOracle(cfg, nb_envs=E)       # the oracle with the device's fp32 state as interpolatePower's query
    fed.device_env = env
    ora.interp_grid = fed.interp_grid = grid
    ora.reset(seed=77, episode=2)
    fed.reset(seed=77, episode=2)
    np.testing.assert_allclose(env.t["base_power"].cpu().numpy(), fed.base_power, rtol=1e-9)
    np.testing.assert_allclose(env.t["base_power"].cpu().numpy(), ora.base_power, rtol=3e-6)
    rng = np.random.default_rng(1)
    bound = 0.0
    for t in range(160):
        act = (rng.random((E, N)) < 0.5).astype(np.uint8)
        obs, reward, _, _ = env.step(torch.from_numpy(act).cuda())
        ora.step(act)
        r_fed = fed.step(act)
        if t % 75 in (73, 74, 0, 1):
            dev = env.t["base_power"].cpu().numpy()
            # same query, both lookups in fp64: rounding only; rewards then hold north_star's 1e-5
            np.testing.assert_allclose(dev, fed.base_power, rtol=1e-9)
            np.testing.assert_allclose(env.reg_signal().cpu().numpy(), fed.S, rtol=1e-9)
            np.testing.assert_allclose(obs[5].cpu().numpy(), np.broadcast_to((fed.S / (7500.0 * N))[:, None], (E, N)), rtol=1e-6)
            np.testing.assert_allclose(reward.cpu().numpy(), r_fed, rtol=R_RTOL, atol=R_ATOL)
            # pure fp64 oracle: inside what the fp32 temperatures can move the lookup (steepest grid edge x state difference)
            bound = max(bound, base_power_bound(grid, env, ora, N))
            assert np.all(np.abs(dev - ora.base_power) <= bound * (1 + 1e-9) + 1e-6)
            np.testing.assert_allclose(dev, ora.base_power, rtol=3e-6)
            np.testing.assert_array_equal(env.t["P"].cpu().numpy(), ora.P)
    assert len(np.unique(env.t["base_power"].cpu().numpy())) == E


def test_interpolation_mode_needs_a_grid():
    mdr = _mdr()
    from mdr_amd.config import InterpolationGridMissing
    cfg = _cfg(8, **{"default_env_prop.power_grid_prop.base_power_mode": "interpolation"})
    with pytest.raises(InterpolationGridMissing):      # the reference does not ship mergedGridSearchResultFinal.npy
        mdr.BatchedDemandResponseEnv(cfg, device="cuda:0", regenerate_missing_grid=False)


def test_single_house_envs_in_bulk_match_the_group_kernel_and_the_oracle():
    """N = 1 (config.py's literal default) at large E takes the env-vectorised kernel; it must equal the generic
    one-lane-per-env kernel bit for bit, and the oracle on a sample."""
    from oracle import mdr_oracle as mo
    mdr = _mdr()
    cfg = _cfg(1, **{"noise_house_prop.noise_mode": "big_noise", "noise_hvac_prop.noise_mode": "big_noise",
                     "default_env_prop.reward_prop.temp_penalty_mode": "mixture"})
    E = 262144
    big = mdr.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=9)
    small = mdr.BatchedDemandResponseEnv(cfg, nb_envs=4096, device="cuda:0", seed=9, env_offset=E - 4096)   # generic kernel
    ora = mo.OracleEnv(cfg, nb_envs=64, env_offset=E - 64).reset(seed=9, episode=0)
    big.reset(episode=0)
    small.reset(episode=0)
    gen = torch.Generator(device="cuda").manual_seed(2)
    for t in range(30):
        act = (torch.rand((E, 1), device="cuda", generator=gen) < 0.5).to(torch.uint8)
        _, r_big, _, _ = big.step(act)
        _, r_small, _, _ = small.step(act[E - 4096:].contiguous())
        r_ref = ora.step(act[E - 64:].cpu().numpy())
    for k in ("Ta", "Tm", "sso", "flags", "reward", "P"):
        assert torch.equal(big.t[k][E - 4096:], small.t[k]), k
    assert torch.equal(big.t["obs"][:, E - 4096:], small.t["obs"])
    np.testing.assert_allclose(big.house_temp()[E - 64:].cpu().numpy(), ora.Ta, rtol=T_RTOL)
    np.testing.assert_allclose(r_big[E - 64:].cpu().numpy(), r_ref, rtol=R_RTOL, atol=R_ATOL)
    np.testing.assert_array_equal(big.t["P"][E - 64:].cpu().numpy(), ora.P)
    big.step_bangbang()
    small.step_bangbang()
    assert torch.equal(big.t["Ta"][E - 4096:], small.t["Ta"]) and torch.equal(big.t["actions"][E - 4096:], small.t["actions"])


def test_c2_literal_1024_envs_x_50_houses_1000_steps():
    """BASELINE.json configs[1] exactly as SURVEY.md 8(d) spells it out: E = 1024, N = 50, no house / HVAC noise (uniform
    parameters), constant outdoor temperature (no weather noise), no solar gain, flat signal on constant base power, fixed start
    date, initial Ta = Tm = 20 + |N(0, 5)| from seed 1234, recorded Bernoulli(0.5) actions from seed 1234, T = 1000 - the HIP
    path against the oracle at EVERY step: integer state and cluster power exactly, temperatures and rewards to 1e-5."""
    from oracle import mdr_oracle as mo
    E, N, T = 1024, 50, 1000
    cfg = _cfg(N, **{"noise_house_prop.noise_mode": "no_noise", "noise_hvac_prop.noise_mode": "no_noise",
                     "default_env_prop.cluster_prop.temp_mode": "constant", "default_house_prop.solar_gain_bool": False,
                     "default_env_prop.power_grid_prop.signal_mode": "flat",
                     "default_env_prop.start_datetime_mode": "fixed", "default_env_prop.start_datetime": "2021-06-15 12:00:00"})
    rng = np.random.default_rng(1234)
    house, hvac = cfg["default_house_prop"], cfg["default_hvac_prop"]
    start = 20.0 + np.abs(rng.normal(0.0, 5.0, size=(E, N)))
    full = lambda v, dt=np.float64: np.full((E, N), v, dtype=dt)
    params = dict(Ta=start, Tm=start.copy(), target=full(house["target_temp"]), deadband=full(house["deadband"]),
                  Ua=full(house["Ua"]), Cm=full(house["Cm"]), Ca=full(house["Ca"]), Hm=full(house["Hm"]),
                  capacity=full(hvac["cooling_capacity"]), COP=full(hvac["COP"]), latent=full(hvac["latent_cooling_fraction"]),
                  lockout=full(hvac["lockout_duration"], np.int64),
                  t0=np.full(E, mo.to_epoch_seconds(__import__("datetime").datetime(2021, 6, 15, 12, 0, 0)), dtype=np.int64),
                  phase=np.zeros(E), ratio=np.ones(E))
    actions = (rng.random((T, E, N)) < 0.5).astype(np.uint8)
    env = _mdr().BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=1234)
    env.load_episode(params, seed=1234, episode=0)
    ora = mo.OracleEnv(cfg, nb_envs=E)
    ora.seed, ora.episode = 1234, 0
    ora.load_episode(params)
    assert float(ora.OD.std()) == 0.0 and float(ora.S.std()) == 0.0          # fixed outdoor temperature, flat signal
    acts = torch.from_numpy(actions).to("cuda:0")
    worst_T = worst_r = 0.0
    for t in range(T):
        obs, reward, done, info = env.step(acts[t])
        r_ref = ora.step(actions[t])
        flags = env.t["flags"].cpu().numpy()
        assert np.array_equal((flags & 1).astype(bool), ora.on) and np.array_equal((flags & 2).astype(bool), ora.lock), t
        assert np.array_equal(env.t["sso"].cpu().numpy(), ora.sso), t
        assert np.array_equal(info["cluster_hvac_power"].cpu().numpy(), ora.P), t
        Ta = env.house_temp().cpu().numpy()
        worst_T = max(worst_T, float(np.max(np.abs(Ta - ora.Ta) / np.abs(ora.Ta))),
                      float(np.max(np.abs(env.house_mass_temp().cpu().numpy() - ora.Tm) / np.abs(ora.Tm))))
        worst_r = max(worst_r, float(np.max(np.abs(reward.cpu().numpy() - r_ref) - R_RTOL * np.abs(r_ref))))
        assert worst_T <= T_RTOL and worst_r <= R_ATOL, (t, worst_T, worst_r)
    np.testing.assert_allclose(obs.cpu().numpy(), ora.dynamic_obs(), rtol=2e-5, atol=2e-6)
    assert worst_T < 2e-6            # observed ~5e-7: the difference form holds fp32 an order below the bar over 1000 steps


def test_reset_drops_a_loaded_outdoor_temperature_table():
    """load_episode(od_table=...) followed by reset() must not keep replaying the recorded outdoor temperatures."""
    g = gu.Golden("s1_c1_flat")
    env = run_fixture(g)
    env.step_bangbang()
    env.reset(seed=77, episode=3)
    fresh = _mdr().BatchedDemandResponseEnv(g.config, nb_envs=1, device="cuda:0", seed=77)
    fresh.reset(episode=3)
    assert torch.equal(env.t["tab_od"], fresh.t["tab_od"])
    for _ in range(5):
        env.step_bangbang()
        fresh.step_bangbang()
    assert torch.equal(env.t["Ta"], fresh.t["Ta"]) and torch.equal(env.od_temp(), fresh.od_temp())
