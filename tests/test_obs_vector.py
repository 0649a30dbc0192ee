"""utils.normStateDict restated: the oracle's vectorised form against the reference's own vectors (CPU), and the
HIP kernel k_obs_vector against both (GPU), for every golden scenario - default columns, every optional
state/message column, every static communication topology."""
import numpy as np
import pytest

from oracle import mdr_oracle as mo
from tests import golden_util as gu


def _replay(g):
    env = mo.OracleEnv(g.config, nb_envs=1)
    env.seed, env.episode = g.seed, 0
    if g.interp_grid() is not None:
        env.interp_grid = mo.InterpGrid(*g.interp_grid())
    env.load_episode(g.params(), od_table=g.od_table())
    return env


@pytest.mark.parametrize("name", gu.names())
def test_oracle_norm_state_matches_reference(name):
    g = gu.Golden(name)
    env = _replay(g)
    links = g.a["links"].astype(np.int64) if "links" in g.a else None
    steps = g.meta["norm_steps"]
    k = 0
    for t in range(g.T + 1):
        if t in steps:
            got = env.norm_state(g.config, links)[0]
            assert got.shape == g.a["norm_state"][k].shape, (got.shape, g.a["norm_state"][k].shape)
            np.testing.assert_allclose(got, g.a["norm_state"][k], rtol=1e-10, atol=1e-12)
            k += 1
        if t < g.T:
            env.step(g.a["actions"][t][None, :])
    assert k == len(steps)


@pytest.mark.gpu
@pytest.mark.parametrize("name", gu.names())
def test_hip_obs_vector_matches_reference(name):
    import torch
    import mdr_amd
    g = gu.Golden(name)
    env = mdr_amd.BatchedDemandResponseEnv(g.config, nb_envs=1, device="cuda:0", seed=g.seed, interp_grid=g.interp_grid())
    env.load_episode(g.params(), od_table=g.od_table(), seed=g.seed, episode=0)
    if "links" in g.a and g.config["default_env_prop"]["cluster_prop"]["agents_comm_mode"] == "random_fixed":
        env.set_comm_links(g.a["links"])
    steps = g.meta["norm_steps"]
    acts = torch.from_numpy(g.a["actions"]).to("cuda:0")
    F = g.a["norm_state"].shape[-1]
    assert env.obs_vector_length() == F
    k = 0
    for t in range(g.T + 1):
        if t in steps:
            planes = env.obs_vector("planes")
            rows = env.obs_vector("rows")
            assert planes.shape == (F, 1, g.N) and rows.shape == (1, g.N, F)
            torch.testing.assert_close(planes[:, 0, :].t().contiguous(), rows[0], rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(rows[0].cpu().numpy(), g.a["norm_state"][k], rtol=2e-5, atol=2e-6)
            k += 1
        if t < g.T:
            env.step(acts[t][None, :])
    assert k == len(steps)


@pytest.mark.gpu
@pytest.mark.parametrize("E,N,comm", [(3, 1024, 10), (2, 1000, 7), (5, 64, 10), (7, 11, 10), (2, 4100, 4), (70000, 20, 10), (300, 50, 10),
                                      (33, 12, 10), (9, 300, 10), (5, 2048, 10)])
def test_hip_obs_vector_matches_oracle_batched(E, N, comm):
    import torch
    import mdr_amd
    cfg = mdr_amd.default_config()
    env_p = cfg["default_env_prop"]
    env_p["cluster_prop"]["nb_agents"] = N
    env_p["cluster_prop"]["nb_agents_comm"] = comm
    env_p["power_grid_prop"]["base_power_mode"] = "constant"
    if N not in (20, 50, 12, 300, 2048):    # those shapes keep the DEFAULT observation (F = 51): the specialised kernels
        env_p["state_properties"].update(hour=True, day=True, solar_gain=True, thermal=True, hvac=True)
        env_p["message_properties"].update(thermal=True, hvac=True)
    cfg["noise_house_prop"]["noise_mode"] = "big_noise"
    cfg["noise_hvac_prop"]["noise_mode"] = "big_noise"
    env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=17)
    env.reset(episode=0)
    ora = mo.OracleEnv(cfg, nb_envs=E).reset(seed=17, episode=0)
    rng = np.random.default_rng(3)
    for t in range(12):
        act = (rng.random((E, N)) < 0.5).astype(np.uint8)
        env.step(torch.from_numpy(act).cuda())
        ora.step(act)
    ref = ora.norm_state(cfg)
    np.testing.assert_allclose(env.obs_vector("rows").cpu().numpy(), ref, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(env.obs_vector("planes").cpu().numpy(), np.moveaxis(ref, -1, 0), rtol=2e-5, atol=2e-6)


@pytest.mark.gpu
def test_comm_defects_zero_whole_messages_at_the_configured_rate():
    import torch
    import mdr_amd
    cfg = mdr_amd.default_config()
    cfg["default_env_prop"]["cluster_prop"].update(nb_agents=256, comm_defect_prob=0.3)
    cfg["default_env_prop"]["power_grid_prop"]["base_power_mode"] = "constant"
    env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=64, device="cuda:0", seed=5)
    env.reset()
    env.rollout(3)
    rows = env.obs_vector("rows")
    msgs = rows[..., 11:].reshape(64, 256, 10, 4)
    dead = (msgs[..., 3] == 0)                    # hvac_max_consumption is never 0 in a live message
    assert bool(((msgs == 0).all(-1) == dead).all())
    rate = dead.float().mean().item()
    assert abs(rate - 0.3) < 0.01, rate
    again = env.obs_vector("rows").clone()        # same step -> same draws
    assert torch.equal(again, rows)
    env.rollout(1)
    dead2 = env.obs_vector("rows")[..., 11:].reshape(64, 256, 10, 4)[..., 3] == 0
    assert not torch.equal(dead, dead2)


@pytest.mark.gpu
def test_random_sample_links_are_distinct_uniform_and_redrawn():
    """agents_comm_mode='random_sample' (env 976-983): nb_comm distinct senders among the other houses, uniformly,
    anew at every step - checked through the sender's hvac_max_consumption column, made unique per house."""
    import torch
    import mdr_amd
    from scipy import stats
    N, E, c = 40, 512, 10
    cfg = mdr_amd.default_config()
    cfg["default_env_prop"]["cluster_prop"].update(nb_agents=N, agents_comm_mode="random_sample")
    cfg["default_env_prop"]["power_grid_prop"]["base_power_mode"] = "constant"
    env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=11)
    env.reset()
    env.t["P_max"].copy_((torch.arange(N, device="cuda", dtype=torch.float32) + 1.0)[None, :].expand(E, N) * 7500.0)   # id + 1 after /7500
    def senders():
        for layout in ("rows", "planes"):
            v = env.obs_vector(layout)
            rows = v if layout == "rows" else v.permute(1, 2, 0)
            ids = (rows[..., 11:].reshape(E, N, c, 4)[..., 3]).round().long() - 1
            yield ids
    a, b = list(senders())
    assert torch.equal(a, b)                                         # both layouts draw the same links
    own = torch.arange(N, device="cuda")[None, :, None]
    assert bool(((a >= 0) & (a < N) & (a != own)).all())
    srt = a.sort(dim=-1).values
    assert bool((srt[..., 1:] != srt[..., :-1]).all())               # distinct within a house
    counts = torch.bincount(((a - (a > own).long())).reshape(-1), minlength=N - 1).cpu().numpy()   # index among the others
    assert stats.chisquare(counts).pvalue > 1e-4
    first = torch.bincount((a[..., 0] - (a[..., 0] > own[..., 0]).long()).reshape(-1), minlength=N - 1).cpu().numpy()
    assert stats.chisquare(first).pvalue > 1e-4                      # every slot is uniform, not only the set
    env.rollout(1)
    env.t["P_max"].copy_((torch.arange(N, device="cuda", dtype=torch.float32) + 1.0)[None, :].expand(E, N) * 7500.0)
    c2 = next(iter(senders()))
    assert not torch.equal(a, c2)                                    # re-drawn at the next step
