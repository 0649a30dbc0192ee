"""BASELINE.json's full size (4096 envs x 1024 houses = 4,194,304 houses) through size-independent properties -
the oracle cannot run this size in test time, so each check recomputes an invariant of the reference's model from
the device state itself (torch fp64) or compares two device runs bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu

E, N = 4096, 1024


def _cfg(**patches):
    import mdr_amd
    cfg = mdr_amd.default_config()
    env = cfg["default_env_prop"]
    env["cluster_prop"]["nb_agents"] = N
    env["power_grid_prop"]["base_power_mode"] = "constant"
    cfg["noise_house_prop"]["noise_mode"] = "house_big_noise"
    cfg["noise_hvac_prop"]["noise_mode"] = "big_noise"
    for dotted, v in patches.items():
        node = cfg
        parts = dotted.split(".")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = v
    return cfg


def _env(cfg, **kw):
    import mdr_amd
    return mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=2024, **kw)


def test_aggregate_reward_and_observation_identities():
    """cluster_hvac_power == sum(on * max_consumption) exactly (env 1042-1050); reward == -(penalty + signal term)
    (env 364-372) and the observation planes == their normStateDict formulas (utils.py:800-841), from the device state."""
    env = _env(_cfg(**{"default_house_prop.deadband": 0.4}))
    env.reset(episode=0)
    gen = torch.Generator(device="cuda").manual_seed(0)
    for step in range(3):
        act = (torch.rand((E, N), device="cuda", generator=gen) < 0.6).to(torch.uint8)
        s_old = env.reg_signal().clone()
        obs, reward, done, info = env.step(act)
    t = env.t
    on = (t["flags"] & 1).bool()
    P = torch.where(on, t["P_max"].double(), torch.zeros((), dtype=torch.float64, device="cuda")).sum(dim=1)
    assert torch.equal(P, info["cluster_hvac_power"])
    hi, lo = t["target"] + 0.5 * t["deadband"], t["target"] - 0.5 * t["deadband"]
    pen = torch.where(t["Ta"] > hi, (t["Ta"] - hi) ** 2, torch.where(t["Ta"] < lo, (lo - t["Ta"]) ** 2, torch.zeros_like(hi)))
    sig = (((P - s_old) / N) ** 2 / 3515625.0).float()
    torch.testing.assert_close(reward, -(pen + sig[:, None]), rtol=2e-6, atol=1e-6)
    assert not bool(done.any())
    shift = env.spec.temp_ref - 20.0
    torch.testing.assert_close(obs[0], (t["Ta"] + shift) * 0.2, rtol=0, atol=0)
    assert torch.equal(obs[2], on.float()) and torch.equal(obs[3], ((t["flags"] & 2) != 0).float())
    torch.testing.assert_close(obs[4], t["sso"].float() / t["lockout"].float(), rtol=0, atol=0)
    torch.testing.assert_close(obs[6], (P / (7500.0 * N)).float()[:, None].expand(E, N), rtol=1e-7, atol=0)
    torch.testing.assert_close(obs[5], (env.reg_signal() / (7500.0 * N)).float()[:, None].expand(E, N), rtol=1e-7, atol=0)
    # HVAC lockout invariants (env 463-492): on => sso == 0 and not locked; locked => off
    assert bool((t["sso"][on] == 0).all()) and not bool(((t["flags"] & 3) == 3).any())


def test_thermal_fixed_point_and_lockout_clock():
    """With the HVAC held off, no solar gain and a constant outdoor temperature, T_air = T_mass = T_od is a fixed point
    of update_temperature (env 664-738) - exactly, in fp32, for every heterogeneous house - and seconds_since_off
    advances by time_step per step (env 475-476)."""
    cfg = _cfg(**{"default_house_prop.solar_gain_bool": False, "default_env_prop.cluster_prop.temp_mode": "constant",
                  "default_house_prop.init_air_temp": 26.5, "default_house_prop.init_mass_temp": 26.5,
                  "default_env_prop.power_grid_prop.signal_mode": "flat"})
    env = _env(cfg)
    env.reset(episode=0)
    ta0, sso0 = env.t["Ta"].clone(), env.t["sso"].clone()
    assert bool((ta0 == 6.5).all())
    off = torch.zeros((E, N), dtype=torch.uint8, device="cuda")
    env.rollout(40, off)
    assert torch.equal(env.t["Ta"], ta0) and torch.equal(env.t["Tm"], ta0)
    assert torch.equal(env.t["sso"], sso0 + 40 * 4)
    assert not bool(env.t["flags"].any()) and float(env.t["P"].abs().max()) == 0.0


def test_determinism_and_table_chunking_at_full_size():
    cfg = _cfg()
    a, b = _env(cfg, table_steps=64), _env(cfg, table_steps=7)
    a.reset(episode=1)
    b.reset(episode=1)
    a.rollout(30)
    b.rollout(30)
    for k in ("Ta", "Tm", "sso", "flags", "reward", "obs", "P"):
        assert torch.equal(a.t[k], b.t[k]), k


def test_fused_rollout_and_partition_invariance_at_full_size():
    """Fused multi-step rollout == single steps, and the second half of the envs run as their own batch
    (env_offset) reproduces rows [E/2, E) - independent replicas, no collective."""
    import mdr_amd
    cfg = _cfg()
    whole = _env(cfg)
    whole.reset(episode=0)
    whole.rollout(20)
    fused = _env(cfg)
    fused.reset(episode=0)
    fused.rollout_fused(20, accumulate=False)
    for k in ("Ta", "Tm", "sso", "flags", "reward", "obs", "P"):
        assert torch.equal(whole.t[k], fused.t[k]), k
    del fused
    half = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E // 2, device="cuda:0", seed=2024, env_offset=E // 2)
    half.reset(episode=0)
    half.rollout(20)
    for k in ("Ta", "Tm", "sso", "flags", "reward", "P"):
        assert torch.equal(whole.t[k][E // 2:], half.t[k]), k


def test_obs_vector_layouts_agree_at_full_size():
    env = _env(_cfg())
    env.reset(episode=0)
    env.rollout(5)
    planes = env.obs_vector("planes")
    rows = env.obs_vector("rows")
    assert torch.equal(planes.permute(1, 2, 0), rows)      # every obs kernel rounds every column the same way
    # own columns against the seven planes the step kernel wrote
    step_planes = env.t["obs"]
    for col, plane in ((0, 0), (1, 1), (5, 2), (6, 3), (7, 4), (9, 5), (10, 6)):
        assert torch.equal(rows[..., col], step_planes[plane]), (col, plane)
    # message m of house h comes from its circular neighbour (env 816-828): column 11 + 4 m + 3 is the sender's max consumption
    pmax = env.t["P_max"] / 7500.0
    for m, shift in ((0, -5), (4, -1), (5, 1), (9, 5)):
        torch.testing.assert_close(rows[..., 11 + 4 * m + 3], torch.roll(pmax, -shift, dims=1), rtol=1e-6, atol=0)
