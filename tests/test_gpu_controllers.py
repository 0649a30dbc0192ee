"""The reference's rule-based controllers evaluated in-kernel (agents/bangbang_controllers.py: BangBangController 41-61,
DeadbandBangBangController 13-38 == BasicController 64-88, AlwaysOnController 1-10; main-deploy.py:57-104 drives them): the
closed loop on the device against the S14 fixtures - the reference env under the reference's own controller objects - and the
rollout forms (single launches, fused, split, persistent) against single steps."""
import numpy as np
import pytest
import torch

from tests import golden_util as gu

pytestmark = pytest.mark.gpu

KIND = {"s14_controller_deadband": "deadband", "s14_controller_basic": "basic", "s14_controller_always_on": "always_on"}


def _env_for(g):
    import mdr_amd
    env = mdr_amd.BatchedDemandResponseEnv(g.config, nb_envs=1, device="cuda:0", seed=g.seed, table_steps=64)
    env.load_episode(g.params(), od_table=g.od_table(), seed=g.seed, episode=0)
    return env


@pytest.mark.parametrize("name", sorted(KIND))
def test_closed_loop_reproduces_the_reference_under_its_own_controller(name):
    """No actions are fed: the kernel decides from its own fp32 state.  Actions, HVAC state and power bit for bit, temperatures and
    rewards to the tolerances of the recorded-action fixtures (a decision within fp32 noise of a band edge would fork the
    trajectory: these fixtures hold none - the assertion on the actions would say so)."""
    g = gu.Golden(name)
    a = g.a
    env = _env_for(g)
    env.set_controller(KIND[name])
    for t in range(g.T):
        _, reward, _, info = env.step_controller()
        assert np.array_equal(env.t["actions"][0].cpu().numpy(), a["actions"][t]), t
        fl = env.t["flags"][0].cpu().numpy()
        assert np.array_equal(fl & 1, a["on"][t]) and np.array_equal((fl >> 1) & 1, a["lock"][t]), t
        assert info["cluster_hvac_power"][0].item() == a["P"][t]
        np.testing.assert_allclose(env.house_temp()[0].cpu().numpy(), a["Ta"][t], rtol=1e-5, atol=0)
        np.testing.assert_allclose(reward[0].cpu().numpy(), a["reward"][t], rtol=1e-5, atol=1e-5)


def _cfg(n, deadband=1.5, mode="individual_L2"):
    import mdr_amd
    cfg = mdr_amd.default_config()
    env = cfg["default_env_prop"]
    env["cluster_prop"]["nb_agents"] = n
    env["power_grid_prop"]["base_power_mode"] = "constant"
    env["power_grid_prop"]["signal_mode"] = "sinusoidals"
    env["reward_prop"]["temp_penalty_mode"] = mode
    cfg["default_house_prop"]["deadband"] = deadband
    cfg["noise_house_prop"]["noise_mode"] = "big_noise"
    cfg["noise_hvac_prop"]["noise_mode"] = "big_noise"
    return cfg


STATE = ("Ta", "Tm", "sso", "flags", "actions", "reward", "obs", "P")


@pytest.mark.parametrize("kind", ["deadband", "always_on"])
@pytest.mark.parametrize("E,N", [(64, 1024), (3, 20000), (4096, 50), (2000, 20), (512, 36)])
def test_rollouts_apply_the_controller_as_single_steps_do(kind, E, N):
    """rollout (one launch per step / split / multi / packed forms by shape) and rollout_fused under a controller end bit for bit
    where T step_controller() calls end; and the rule itself, on the kernel's own pre-step state."""
    import mdr_amd
    cfg = _cfg(N)
    envs = [mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=13, table_steps=16) for _ in range(3)]
    for e in envs:
        e.reset(episode=1)
        e.set_controller(kind)
    one, roll, fused = envs
    T = 37
    for t in range(T):
        Ta, tg, db, on = one.t["Ta"].clone(), one.t["target"], one.t["deadband"], (one.t["flags"] & 1).bool()
        one.step_controller()
        want = torch.ones_like(on) if kind == "always_on" else torch.where(Ta < tg - 0.5 * db, torch.zeros_like(on),
                                                                           torch.where(Ta > tg + 0.5 * db, torch.ones_like(on), on))
        assert torch.equal(one.t["actions"].bool(), want), t
    roll.rollout(T)
    got = fused.rollout_fused(T)
    for name in STATE:
        assert torch.equal(roll.t[name], one.t[name]), name
        if got is not None:
            assert torch.equal(fused.t[name], one.t[name]), name
    if kind == "deadband":      # the hold band is in play: neither everything on nor the plain bang-bang rule
        bang = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=13, table_steps=16)
        bang.reset(episode=1)
        bang.rollout(T)
        assert not torch.equal(bang.t["flags"], one.t["flags"])


def test_persistent_rollout_applies_the_controller():
    import mdr_amd
    cfg = _cfg(20000, mode="mixture")
    a = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=2, device="cuda:0", seed=5, table_steps=16)
    b = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=2, device="cuda:0", seed=5, table_steps=16)
    for e in (a, b):
        e.reset(episode=0)
        e.set_controller("deadband")
    a.rollout(40)
    b.rollout_persistent(40)
    for name in STATE:
        assert torch.equal(a.t[name], b.t[name]), name


def test_deepcopy_keeps_the_controller_and_unknown_names_are_refused():
    import copy
    import mdr_amd
    env = mdr_amd.BatchedDemandResponseEnv(_cfg(64), nb_envs=4, device="cuda:0", seed=1)
    env.reset(episode=0)
    with pytest.raises(ValueError):
        env.set_controller("mpc")
    env.set_controller("deadband")
    env.rollout(5)
    twin = copy.deepcopy(env)
    env.rollout(9)
    twin.rollout(9)
    assert torch.equal(env.t["Ta"], twin.t["Ta"]) and torch.equal(env.t["flags"], twin.t["flags"])


def test_sharded_steps_and_odd_house_counts_under_the_deadband_controller():
    """The records path of sharded houses (three in-process shards) and the persistent rollout with one house per lane
    (nb_houses % 4 != 0) apply the controller as the unsharded single steps do."""
    import mdr_amd
    from mdr_amd.sharding import LocalShardGroup
    N = 14001
    cfg = _cfg(N, mode="common_L2")
    whole = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=2, device="cuda:0", seed=21, table_steps=16)
    whole.reset(episode=0)
    whole.set_controller("deadband")
    group = LocalShardGroup(cfg, nb_envs=2, nb_shards=3, seed=21, table_steps=16)
    group.reset(episode=0)
    per = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=2, device="cuda:0", seed=21, table_steps=16)
    per.reset(episode=0)
    per.set_controller("deadband")
    for _ in range(20):
        whole.step_controller()
        group.step_controller("deadband")
    per.rollout_persistent(20)
    lo = 0
    for shard in group.shards:
        hi = lo + shard.nb_houses
        for name in ("Ta", "Tm", "sso", "flags", "actions"):
            assert torch.equal(shard.t[name], whole.t[name][:, lo:hi]), name
        assert torch.equal(shard.t["P"], whole.t["P"])
        lo = hi
    for name in STATE:
        assert torch.equal(per.t[name], whole.t[name]), name


def test_greedy_myopic_closed_loop_reproduces_the_reference():
    """agents/greedy_myopic_controller.py on the device (mdr_env_greedy_myopic_actions: per-env ranking in LDS + the budget pass)
    against the S14 fixture - the reference env under the reference's own GreedyMyopic objects (pandas sort_values + iterrows)."""
    g = gu.Golden("s14_controller_greedy_myopic")
    a = g.a
    env = _env_for(g)
    for t in range(g.T):
        acts = env.greedy_myopic_actions()
        assert np.array_equal(acts[0].cpu().numpy(), a["actions"][t]), t
        _, reward, _, info = env.step(acts)
        fl = env.t["flags"][0].cpu().numpy()
        assert np.array_equal(fl & 1, a["on"][t]) and np.array_equal((fl >> 1) & 1, a["lock"][t]), t
        assert info["cluster_hvac_power"][0].item() == a["P"][t]
        np.testing.assert_allclose(env.house_temp()[0].cpu().numpy(), a["Ta"][t], rtol=1e-5, atol=0)


@pytest.mark.parametrize("E,N", [(1001, 10), (501, 7), (333, 13), (300, 20), (77, 32), (3, 1), (64, 50), (41, 100), (23, 200), (9, 400), (16, 1024), (3, 2048),
                                  (5, 777), (7, 1500)])
def test_greedy_myopic_matches_the_oracle_on_batches(E, N):
    """Every kernel form (8 / 4 / 2 small envs or one env per wavefront with 1 / 2 / 4 / 8 / 16 sorted positions per lane, the LDS workgroup form
    above 1024 houses, padded sorts, a last workgroup with idle waves) against the rule restated on the device's own state."""
    import mdr_amd
    from oracle import mdr_oracle as mo
    cfg = _cfg(N)
    env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=11, table_steps=16)
    env.reset(episode=0)
    ora = mo.OracleEnv.mirror(env) if hasattr(mo.OracleEnv, "mirror") else None
    for t in range(12):
        acts = env.greedy_myopic_actions().clone()
        # the rule itself, restated on the device's own state (fp64 pass over the fp32 powers, houses ranked by the fp32 difference)
        Ta, tg = env.t["Ta"].cpu().numpy(), env.t["target"].cpu().numpy()
        pw = env.t["P_max"].cpu().numpy().astype(np.float64)
        lock = ((env.t["flags"].cpu().numpy() >> 1) & 1).astype(bool)
        sig = env.reg_signal().cpu().numpy()
        want = np.zeros((E, N), dtype=np.uint8)
        for e in range(E):
            order = np.argsort(-(Ta[e] - tg[e]), kind="stable")
            total = 0.0
            for h in order:
                p = pw[e, h]
                if p + total < sig[e] or (abs(p + total - sig[e]) < abs(total - sig[e]) and not lock[e, h]):
                    total += p
                    want[e, h] = 1
        assert np.array_equal(acts.cpu().numpy(), want), t
        env.step(acts)
    big = mdr_amd.BatchedDemandResponseEnv(_cfg(4096), nb_envs=1, device="cuda:0", seed=1)
    big.reset(episode=0)
    with pytest.raises(Exception):
        big.greedy_myopic_actions()


@pytest.mark.parametrize("E,N", [(37, 10), (21, 50), (6, 300), (5, 1024)])
def test_greedy_myopic_keeps_the_sequential_totals_when_prefix_sums_would_round(E, N):
    """Powers spread over sixteen decades (and zeros) in every other env: fp64 prefix sums of those are not exact, so the kernel must
    walk such an env house by house, as the reference's loop does; the envs in between keep the run-wise pass."""
    import mdr_amd
    env = mdr_amd.BatchedDemandResponseEnv(_cfg(N), nb_envs=E, device="cuda:0", seed=5, table_steps=16)
    env.reset(episode=0)
    rng = np.random.default_rng(N)
    pw32 = env.t["P_max"].cpu().numpy().copy()
    wild = (10.0 ** rng.uniform(-8, 8, size=(E, N))).astype(np.float32)
    wild[rng.random((E, N)) < 0.1] = 0.0
    pw32[::2] = wild[::2]
    env.t["P_max"].copy_(torch.from_numpy(pw32).to(env.t["P_max"].device))
    for t in range(4):
        acts = env.greedy_myopic_actions().clone().cpu().numpy()
        Ta, tg = env.t["Ta"].cpu().numpy(), env.t["target"].cpu().numpy()
        lock = ((env.t["flags"].cpu().numpy() >> 1) & 1).astype(bool)
        sig = env.reg_signal().cpu().numpy()
        if t % 2:                          # budgets inside the wild sums as well
            sig = sig * 0 + np.float64(pw32.astype(np.float64).sum(axis=1) * rng.uniform(0.0, 1.0, size=E))
            tab = env.table("tab_signal")
            tab[:] = torch.from_numpy(sig).to(tab.device)[None, :]
            acts = env.greedy_myopic_actions().clone().cpu().numpy()
        want = np.zeros((E, N), dtype=np.uint8)
        for e in range(E):
            total = 0.0
            for h in np.argsort(-(Ta[e] - tg[e]), kind="stable"):
                p = float(pw32[e, h])
                if p + total < sig[e] or (abs(p + total - sig[e]) < abs(total - sig[e]) and not lock[e, h]):
                    total += p
                    want[e, h] = 1
        assert np.array_equal(acts, want), t
        env.step(torch.from_numpy((rng.random((E, N)) < 0.5).astype(np.uint8)).to(env.t["actions"].device))


def test_deploy_controller_accumulates_what_the_stepwise_loop_does():
    """rollout.deploy_controller (main-deploy.py's loop under a rule-based agent): the fused rollout's accumulators for the in-kernel
    rules, a step loop for GreedyMyopic - against plain stepping with the sums kept on the side."""
    import mdr_amd
    from mdr_amd.rollout import deploy_controller
    for kind in ("deadband", "greedy_myopic"):
        cfg = _cfg(100)
        a = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=40, device="cuda:0", seed=8, table_steps=16)
        b = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=40, device="cuda:0", seed=8, table_steps=16)
        a.reset(episode=0)
        b.reset(episode=0)
        got = deploy_controller(a, kind, 30)
        if kind != "greedy_myopic":
            b.set_controller(kind)
        rsum = torch.zeros_like(got["reward_sum"])
        serr = torch.zeros_like(got["sq_signal_error_sum"])
        for _ in range(30):
            _, r, _, info = b.step_greedy_myopic() if kind == "greedy_myopic" else b.step_controller()
            rsum += r
            serr += (b.reg_signal() - info["cluster_hvac_power"]) ** 2
        assert torch.equal(a.t["Ta"], b.t["Ta"]) and torch.equal(a.t["flags"], b.t["flags"]), kind
        assert torch.equal(got["reward_sum"], rsum), kind
        torch.testing.assert_close(got["sq_signal_error_sum"], serr, rtol=1e-12, atol=0)
