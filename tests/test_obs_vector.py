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
            keep = None
            if "msg_keep" in g.a:       # the reference's own `random.sample` senders and `np.random.rand()` outcomes of this step
                links, keep = g.a["msg_senders"][k][None].astype(np.int64), g.a["msg_keep"][k][None].astype(bool)
            got = env.norm_state(g.config, links, keep)[0]
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
    # link defects / random_sample senders are drawn from MT19937 in the reference and from Philox here: those fixtures pin the
    # ORACLE with the reference's recorded draws (test above); the device is then held to the oracle draw by draw
    # (test_comm_draws_match_the_oracle_streams) and element by element (test_hip_obs_vector_random_gather_matches_oracle)
    ora = _replay(g) if "msg_keep" in g.a else None
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
            static = g.a["links"].astype(np.int64) if "links" in g.a else None      # the episode's table (None: circular / random_sample)
            want = g.a["norm_state"][k] if ora is None else ora.norm_state(g.config, static)[0]
            np.testing.assert_allclose(rows[0].cpu().numpy(), want, rtol=2e-5, atol=2e-6)
            if ora is not None:       # and the columns that do not depend on the draws still match the reference itself
                np.testing.assert_allclose(rows[0, :, :11].cpu().numpy(), g.a["norm_state"][k][:, :11], rtol=2e-5, atol=2e-6)
            k += 1
        if t < g.T:
            env.step(acts[t][None, :])
            if ora is not None:
                ora.step(g.a["actions"][t][None, :])
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


# ------------------------------------------------------------------------------------------------------------------
# The random part of the message gather (env 976-1002), pinned draw by draw: Philox stream 7 (link defects) and
# stream 9 (`random_sample` senders) restated in oracle/mdr_oracle.py (link_keep / sampled_senders)
# ------------------------------------------------------------------------------------------------------------------
def test_oracle_cumulated_abs_noise_matches_reference():
    """PowerGrid.cumulated_abs_noise / nb_steps (env 1301-1302) in the perlin families, incl. the call inside build_environment."""
    for name in ("s7_perlin_wiring", "s7_fastpp_perlin_wiring", "s12_interp_perlin_noon"):
        g = gu.Golden(name)
        env = _replay(g)
        np.testing.assert_allclose(env.cumulated_abs_noise[0], g.a["cumulated_abs_noise"][0], rtol=1e-10)
        for t in range(g.T):
            env.step(g.a["actions"][t][None, :])
            np.testing.assert_allclose(env.cumulated_abs_noise[0], g.a["cumulated_abs_noise"][t + 1], rtol=1e-10)
        assert env.grid_steps == int(g.a["grid_nb_steps"]) == g.T + 1


def test_oracle_sampled_senders_are_an_ordered_sample_without_replacement():
    cfg = gu.reference_env_config()
    cfg["default_env_prop"]["cluster_prop"].update(nb_agents=9, nb_agents_comm=8, agents_comm_mode="random_sample")
    cfg["default_env_prop"]["power_grid_prop"]["base_power_mode"] = "constant"
    env = mo.OracleEnv(cfg, nb_envs=50).reset(seed=3, episode=2)
    s0 = env.sampled_senders(8)
    own = np.arange(9)[None, :, None]
    assert s0.shape == (50, 9, 8) and ((s0 >= 0) & (s0 < 9) & (s0 != own)).all()
    assert (np.sort(s0, axis=-1) == np.sort(np.where(np.arange(9)[None, :] == np.arange(9)[:, None], -1, np.arange(9)[None, :]), axis=-1)[:, 1:][None]).all()
    env.step(np.zeros((50, 9), dtype=bool))
    assert not np.array_equal(env.sampled_senders(8), s0)          # re-drawn every step
    again = mo.OracleEnv(cfg, nb_envs=50).reset(seed=3, episode=2)
    assert np.array_equal(again.sampled_senders(8), s0)            # a pure function of (seed, episode, env, house, step)
    assert not np.array_equal(mo.OracleEnv(cfg, nb_envs=50).reset(seed=3, episode=3).sampled_senders(8), s0)


def _gather_cfg(N, mode, nb_comm, defect, flags=False):
    import mdr_amd
    cfg = mdr_amd.default_config()
    env_p = cfg["default_env_prop"]
    cl = env_p["cluster_prop"]
    cl.update(nb_agents=N, agents_comm_mode=mode, nb_agents_comm=nb_comm, comm_defect_prob=defect)
    if mode == "neighbours_2D":
        cl["agents_comm_parameters"]["neighbours_2D"] = {"row_size": 10, "distance_comm": 2}
    env_p["power_grid_prop"]["base_power_mode"] = "constant"
    if flags:
        env_p["state_properties"].update(hour=True, day=True, solar_gain=True, thermal=True, hvac=True)
        env_p["message_properties"].update(thermal=True, hvac=True)
    cfg["noise_house_prop"]["noise_mode"] = "big_noise"
    cfg["noise_hvac_prop"]["noise_mode"] = "big_noise"
    return cfg


def _oracle_links(cfg, seed, episode):
    """The static link table the product derives for this episode ('random_fixed': from (seed, episode)); None = the mode's own."""
    from mdr_amd.comm import links_array
    cl = cfg["default_env_prop"]["cluster_prop"]
    if cl["agents_comm_mode"] in ("neighbours", "random_sample", "no_message"):
        return None
    return links_array(cl, seed_episode=(seed, episode)).astype(np.int64)


@pytest.mark.gpu
@pytest.mark.parametrize("E,N,c,mode,defect", [
    (3, 64, 10, "random_sample", 0.1), (2, 17, 16, "random_sample", 0.5), (5, 1000, 10, "random_sample", 0.0), (1, 4100, 3, "random_sample", 0.3),
    (4, 50, 10, "neighbours", 0.1), (2, 1024, 10, "neighbours", 0.5), (3, 60, 5, "closed_groups", 0.5), (2, 40, 7, "random_fixed", 0.1),
    (2, 100, 12, "neighbours_2D", 0.5), (2, 300, 13, "neighbours", 0.9)])
def test_comm_draws_match_the_oracle_streams(E, N, c, mode, defect):
    """mdr_env_comm_draws (the very draws the observation kernels use) == the oracle's restatement of Philox streams 7 and 9:
    which sender every slot listens to and which links drop their message, exactly, over steps, episodes and env offsets."""
    import torch
    import mdr_amd
    cfg = _gather_cfg(N, mode, c, defect)
    for episode, env_offset in ((0, 0), (3, 11)):
        env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=(1 << 40) + 77, env_offset=env_offset)
        env.reset(episode=episode)
        ora = mo.OracleEnv(cfg, nb_envs=E, env_offset=env_offset).reset(seed=(1 << 40) + 77, episode=episode)
        for t in range(4):
            senders, keep = env.comm_draws()
            cc = senders.shape[2]
            np.testing.assert_array_equal(keep.cpu().numpy(), ora.link_keep(cc, defect))
            if mode == "random_sample":
                np.testing.assert_array_equal(senders.cpu().numpy(), ora.sampled_senders(cc))
            else:
                table = _oracle_links(cfg, (1 << 40) + 77, episode)
                table = ora.circular_links(cc) if table is None else table
                np.testing.assert_array_equal(senders.cpu().numpy(), np.broadcast_to(table[None], (E, N, cc)))
            act = torch.zeros((E, N), dtype=torch.uint8, device="cuda:0")
            env.step(act)
            ora.step(np.zeros((E, N), dtype=bool))


@pytest.mark.gpu
@pytest.mark.parametrize("defect", [0.1, 0.5])
@pytest.mark.parametrize("mode,N,c,flags", [("neighbours", 1024, 10, False), ("neighbours", 52, 10, False), ("neighbours", 300, 7, True),
                                            ("closed_groups", 66, 5, False), ("random_fixed", 48, 6, True), ("neighbours_2D", 100, 12, False),
                                            ("random_sample", 64, 10, False), ("random_sample", 1001, 16, True)])
def test_hip_obs_vector_random_gather_matches_oracle(mode, N, c, flags, defect):
    """The flat normStateDict vector with link defects and `random_sample` senders, element for element against the oracle:
    every zeroed message and every sender is the one the restated Philox streams give - unsharded, rows and planes, and as
    three house shards (the sharded gather draws per GLOBAL house index)."""
    import torch
    import mdr_amd
    from mdr_amd.sharding import LocalShardGroup
    cfg = _gather_cfg(N, mode, c, defect, flags)
    E, seed = 3, 909
    env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=seed)
    group = LocalShardGroup(cfg, nb_envs=E, nb_shards=3, devices=("cuda:0",), seed=seed)
    env.reset(episode=1)
    group.reset(episode=1)
    ora = mo.OracleEnv(cfg, nb_envs=E).reset(seed=seed, episode=1)
    links = _oracle_links(cfg, seed, 1)
    rng = np.random.default_rng(5)
    for t in range(4):
        want = ora.norm_state(cfg, links)
        rows = env.obs_vector("rows").cpu().numpy()
        planes = env.obs_vector("planes").cpu().numpy()
        np.testing.assert_allclose(rows, want, rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(np.moveaxis(planes, 0, -1), want, rtol=2e-5, atol=2e-6)
        cc = ora.link_keep(c, defect).shape[2] if mode != "neighbours_2D" else links.shape[1]
        mf = 4 + (7 if flags else 0)
        own = want.shape[-1] - cc * mf
        dev_msgs, ora_msgs = rows[..., own:].reshape(E, N, cc, mf), want[..., own:].reshape(E, N, cc, mf)
        dead = dev_msgs[..., 3] == 0                    # hvac_max_consumption is never 0 in a delivered message
        assert np.array_equal(dead, ora_msgs[..., 3] == 0), "dropped messages differ at step %d" % t
        assert np.array_equal(dead, ~ora.link_keep(cc, defect)) and (dev_msgs[dead] == 0).all()
        sharded = torch.cat(group.obs_vector("rows"), dim=1).cpu().numpy()
        assert np.array_equal(sharded, rows), "3 shards differ from the unsharded vector at step %d" % t
        act = (rng.random((E, N)) < 0.5).astype(np.uint8)
        env.step(torch.from_numpy(act).cuda())
        parts = [torch.from_numpy(np.ascontiguousarray(act[:, s.house_offset:s.house_offset + s.nb_houses])).cuda() for s in group.shards]
        group.step(parts)
        ora.step(act)


@pytest.mark.gpu
def test_random_fixed_links_are_redrawn_per_episode_and_shared_by_every_view():
    """one 'random_fixed' table per episode, derived from (seed, episode): re-drawn on reset (the reference re-draws it
    in build_environment), identical for the flat vector, the shard plan and a deep copy."""
    import copy
    import mdr_amd
    cfg = _gather_cfg(40, "random_fixed", 6, 0.0)
    env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=2, device="cuda:0", seed=5)
    env.reset(episode=0)
    t0 = env.comm_links_array().copy()
    own = np.arange(40)[:, None]
    assert t0.shape == (40, 6) and (t0 != own).all() and all(len(set(r)) == 6 for r in t0.tolist())
    np.testing.assert_array_equal(env.comm_draws()[0].cpu().numpy(), np.broadcast_to(t0[None], (2, 40, 6)))
    twin = copy.deepcopy(env)
    np.testing.assert_array_equal(twin.comm_links_array(), t0)
    assert np.array_equal(twin.obs_vector("rows").cpu().numpy(), env.obs_vector("rows").cpu().numpy())
    env.reset()                                    # episode 1
    t1 = env.comm_links_array()
    assert not np.array_equal(t0, t1)
    other = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=2, device="cuda:0", seed=5)
    other.reset(episode=1)
    np.testing.assert_array_equal(other.comm_links_array(), t1)       # what another rank with the same seed derives
    ora = mo.OracleEnv(cfg, nb_envs=2).reset(seed=5, episode=1)
    np.testing.assert_allclose(env.obs_vector("rows").cpu().numpy(), ora.norm_state(cfg, t1.astype(np.int64)), rtol=2e-5, atol=2e-6)
