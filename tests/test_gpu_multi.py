"""Device-filling batches of small envs with N % 4 == 2 - the reference deploys with 50 houses (cli.py:629) - take the
step kernel that puts TWO envs into one lane group, four flat houses per lane on 16-byte accesses (csrc/mdr_multi.hip), with
the totals of the one-env-per-group mapping whose multi-step kernel is its closed loop.  Against the oracle, against the
one-env-per-group kernels a small batch of the same global envs takes, and fused rollout == single steps bit for bit
(env/MA_DemandResponse.py:1005-1055, 234-373)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfg(n, mode):
    import mdr_amd
    cfg = mdr_amd.default_config()
    env = cfg["default_env_prop"]
    env["cluster_prop"]["nb_agents"] = n
    env["power_grid_prop"]["base_power_mode"] = "constant"
    env["power_grid_prop"]["signal_mode"] = "sinusoidals"
    env["reward_prop"]["temp_penalty_mode"] = mode
    cfg["noise_house_prop"]["noise_mode"] = "big_noise"
    cfg["noise_hvac_prop"]["noise_mode"] = "big_noise"
    cfg["default_hvac_prop"]["lockout_noise"] = 20
    return cfg


def _envs(N, odd=False):
    E = -(-270000 // N)
    if odd:
        E += 1 if (E * N) % 4 == 0 else 0
        E += 1 if N % 4 == 0 and E % 2 == 0 else 0      # packed envs: a last wavefront that is not full
    return E


@pytest.mark.parametrize("N,mode,odd", [(50, "mixture", False), (50, "individual_L2", True), (10, "common_L2", False), (6, "mixture", True),
                                        (30, "common_max", False), (126, "mixture", True), (14, "individual_L2", False), (66, "common_L2", True),
                                        # N % 4 == 0 with N / 4 lanes not a power of two: whole envs packed into the wavefront (k_step_packed)
                                        (36, "individual_L2", False), (36, "mixture", True), (68, "common_L2", False), (72, "common_max", True),
                                        (76, "mixture", False), (20, "mixture", False), (20, "individual_L2", True), (40, "common_L2", True)])
def test_multi_env_groups_match_oracle_and_single_env_groups(N, mode, odd):
    import mdr_amd
    from oracle import mdr_oracle as mo
    E = _envs(N, odd)
    cfg = _cfg(N, mode)
    env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=8, table_steps=8)
    env.reset(episode=1)
    k = 37                                    # the LAST envs of the batch (incl. the lane that holds fewer than four houses)
    off = E - k
    small = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=k, device="cuda:0", seed=8, table_steps=8, env_offset=off)
    small.reset(episode=1)
    ora = mo.OracleEnv(cfg, nb_envs=k, env_offset=off).reset(seed=8, episode=1)
    rng = np.random.default_rng(5)
    sl = slice(off, off + k)
    for t in range(12):
        if t % 3 == 2:
            env.step_bangbang()
            small.step_bangbang()
            act = small.t["actions"].cpu().numpy()
        else:
            act = (rng.random((k, N)) < 0.5).astype(np.uint8)
            full = (torch.rand((E, N), device="cuda") < 0.5).to(torch.uint8)
            full[sl] = torch.from_numpy(act).cuda()
            env.step(full)
            small.step(torch.from_numpy(act).cuda())
        r_ref = ora.step(act)
        for name in ("Ta", "Tm", "sso", "flags", "actions", "P"):
            assert torch.equal(env.t[name][sl], small.t[name]), (t, name)
        assert torch.equal(env.t["obs"][:5, sl], small.t["obs"][:5]) and torch.equal(env.t["obs"][5:, sl], small.t["obs"][5:])
        if mode == "individual_L2":
            assert torch.equal(env.t["reward"][sl], small.t["reward"])
        else:                                 # the penalty sum meets the houses in pairs / fours here, one by one there
            torch.testing.assert_close(env.t["reward"][sl], small.t["reward"], rtol=1e-6, atol=1e-6)
        np.testing.assert_array_equal(env.t["sso"][sl].cpu().numpy(), ora.sso)
        assert np.array_equal(env.t["P"][sl].cpu().numpy(), ora.P)
        np.testing.assert_allclose(env.house_temp()[sl].cpu().numpy(), ora.Ta, rtol=1e-5, atol=0)
        np.testing.assert_allclose(env.t["reward"][sl].cpu().numpy(), r_ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("N,mode,odd", [(50, "mixture", False), (50, "individual_L2", True), (10, "common_max", False), (126, "common_L2", True),
                                        (36, "mixture", False), (36, "individual_L2", True), (72, "common_L2", True), (76, "common_max", False),
                                        (20, "common_max", True), (40, "mixture", False)])
def test_multi_env_fused_rollout_equals_single_steps(N, mode, odd):
    import mdr_amd
    E = _envs(N, odd)
    cfg = _cfg(N, mode)
    a = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=2, table_steps=8)
    b = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=2, table_steps=8, obs_planes=False)
    a.reset(episode=0)
    b.reset(episode=0)
    rsum = torch.zeros_like(a.t["reward"])
    terr = torch.zeros(E, dtype=torch.float64, device="cuda")
    serr = torch.zeros(E, dtype=torch.float64, device="cuda")
    trace = []
    for _ in range(21):
        a.step_bangbang()
        rsum = rsum + a.t["reward"]
        d = a.t["Ta"] - a.t["target"]
        terr += (d * d).double().sum(dim=1)
        serr += (a.reg_signal() - a.t["P"]) ** 2
        trace.append(a.t["P"].clone())
    out = b.rollout_fused(21, power_trace=True)
    for name in ("Ta", "Tm", "sso", "flags", "actions", "reward", "P"):
        assert torch.equal(a.t[name], b.t[name]), name
    assert torch.equal(out["reward_sum"], rsum)
    assert torch.equal(out["power_trace"], torch.stack(trace))
    torch.testing.assert_close(out["sq_signal_error_sum"], serr, rtol=1e-12, atol=0)
    torch.testing.assert_close(out["sq_temp_error_sum"], terr, rtol=1e-6, atol=0)
    b.set_obs_planes(True)
    assert torch.equal(a.t["obs"], b.t["obs"])
