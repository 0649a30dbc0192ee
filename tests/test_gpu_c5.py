"""BASELINE config 5 at its full size: ONE env x 1,000,000 houses.

On one GPU: (a) unsharded through the split path against the fp64 oracle; (b) the eight 125,000-house shards of the
8-GPU layout rehearsed in one process (sharding.LocalShardGroup: same step_begin / [world][3][E] block /
step_end_gathered sequence the RCCL path runs, the all-gather replaced by device copies) against (a)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N_C5 = 1_000_000


def _cfg(mode="individual_L2"):
    import mdr_amd
    cfg = mdr_amd.default_config()
    env = cfg["default_env_prop"]
    env["cluster_prop"]["nb_agents"] = N_C5
    env["power_grid_prop"]["base_power_mode"] = "constant"
    env["power_grid_prop"]["signal_mode"] = "sinusoidals"
    env["reward_prop"]["temp_penalty_mode"] = mode
    cfg["noise_house_prop"]["noise_mode"] = "big_noise"
    cfg["noise_hvac_prop"]["noise_mode"] = "big_noise"
    return cfg


def test_c5_unsharded_matches_oracle():
    import mdr_amd
    from oracle import mdr_oracle as mo
    cfg = _cfg("mixture")
    env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=1, device="cuda:0", seed=5)
    env.reset(episode=0)
    ora = mo.OracleEnv(cfg, nb_envs=1).reset(seed=5, episode=0)
    assert float(env.t["max_power"][0]) == float(ora.max_power[0])          # 1e6 whole-watt terms: fp64 sum is exact
    rng = np.random.default_rng(0)
    for t in range(6):
        act = (rng.random((1, N_C5)) < 0.5).astype(np.uint8)
        _, reward, _, _ = env.step(torch.from_numpy(act).cuda())
        r_ref = ora.step(act)
        assert float(env.t["P"][0]) == float(ora.P[0])                      # 6e9 W: beyond fp32's exact integers, exact in fp64
        np.testing.assert_array_equal(env.t["sso"].cpu().numpy(), ora.sso)
        np.testing.assert_allclose(env.house_temp().cpu().numpy(), ora.Ta, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(env.reg_signal().cpu().numpy(), ora.S, rtol=1e-9)
        np.testing.assert_allclose(reward.cpu().numpy(), r_ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("mode,shards", [("individual_L2", 8), ("mixture", 8), ("common_max", 3)])
def test_c5_eight_shards_in_one_process_match_unsharded(mode, shards):
    import mdr_amd
    from mdr_amd.sharding import LocalShardGroup, house_shard
    cfg = _cfg(mode)
    whole = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=1, device="cuda:0", seed=9)
    whole.reset(episode=1)
    group = LocalShardGroup(cfg, nb_envs=1, nb_shards=shards, devices=("cuda:0",), seed=9)
    group.reset(episode=1)
    assert [e.nb_houses for e in group.shards] == [house_shard(N_C5, shards, r)[1] for r in range(shards)]
    assert sum(e.nb_houses for e in group.shards) == N_C5
    for e in group.shards:
        assert torch.equal(e.t["max_power"], whole.t["max_power"])
    assert torch.equal(group.gather("Ta"), whole.t["Ta"])                   # Philox counters are global house indices
    for t in range(8):
        if t % 2:
            whole.step_bangbang()
            group.step_bangbang()
        else:
            act = (torch.rand((1, N_C5), device="cuda:0") < 0.5).to(torch.uint8)
            whole.step(act)
            bounds = np.cumsum([0] + [e.nb_houses for e in group.shards])
            group.step([act[:, bounds[r]:bounds[r + 1]].contiguous() for r in range(shards)])
        assert torch.equal(group.cluster_hvac_power(), whole.t["P"])
        for e in group.shards:
            assert torch.equal(e.t["P"], whole.t["P"])
            assert torch.equal(e.reg_signal(), whole.reg_signal())
        for name in ("Ta", "Tm", "sso", "flags"):
            assert torch.equal(group.gather(name), whole.t[name]), name
        # the penalty sum is reduced in a different order (per shard, then over shards): rewards agree to fp32 rounding
        torch.testing.assert_close(group.gather("reward"), whole.t["reward"], rtol=1e-6, atol=1e-6)
    with pytest.raises(RuntimeError):
        group.shards[0].step_bangbang()                                     # a lone shard cannot step
