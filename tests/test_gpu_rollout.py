"""The fused multi-step rollout (houses resident in registers across steps) must end bit-for-bit where the same
number of single bang-bang steps ends, and its accumulators must equal the stepwise sums (main-deploy.py:124-152)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfg(n, **patches):
    import mdr_amd
    cfg = mdr_amd.default_config()
    cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = n
    cfg["default_env_prop"]["power_grid_prop"]["base_power_mode"] = "constant"
    cfg["noise_house_prop"]["noise_mode"] = "big_noise"
    cfg["noise_hvac_prop"]["noise_mode"] = "big_noise"
    for dotted, v in patches.items():
        node = cfg
        parts = dotted.split(".")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = v
    return cfg


@pytest.mark.parametrize("E,N,mode", [(6, 1024, "individual_L2"), (3, 2048, "mixture"), (5, 256, "common_L2"), (4, 300, "individual_L2"),
                                      (40, 50, "individual_L2"), (70, 10, "common_max"), (9, 1, "individual_L2"), (3, 64, "mixture")])
def test_fused_rollout_equals_single_steps(E, N, mode):
    import mdr_amd
    cfg = _cfg(N, **{"default_env_prop.reward_prop.temp_penalty_mode": mode, "default_house_prop.deadband": 0.5})
    T = 150                                   # crosses two table refills at K = 64
    a = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=8)
    b = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=8)
    a.reset(episode=0)
    b.reset(episode=0)
    res = a.rollout_fused(T, power_trace=True)
    assert res is not None
    rsum = torch.zeros((E, N), dtype=torch.float32, device="cuda:0")
    terr = torch.zeros(E, dtype=torch.float64, device="cuda:0")
    serr = torch.zeros(E, dtype=torch.float64, device="cuda:0")
    trace = []
    for t in range(T):
        _, reward, _, info = b.step_bangbang()
        rsum += reward
        d = (b.t["Ta"] - b.t["target"]).double()
        terr += (d * d).sum(dim=1)
        serr += (b.reg_signal() - info["cluster_hvac_power"]) ** 2
        trace.append(info["cluster_hvac_power"].clone())
    for k in ("Ta", "Tm", "sso", "flags", "reward", "obs", "P", "actions"):
        assert torch.equal(a.t[k], b.t[k]), k
    assert a.steps_taken == b.steps_taken == T
    assert torch.equal(res["power_trace"], torch.stack(trace))
    assert torch.equal(res["reward_sum"], rsum)
    torch.testing.assert_close(res["sq_signal_error_sum"], serr, rtol=1e-12, atol=0)
    torch.testing.assert_close(res["sq_temp_error_sum"], terr, rtol=1e-6, atol=0)   # fp32 squares, different summation tree
    a.rollout_fused(10)                        # keeps going from where it stopped
    b.rollout(10)
    assert torch.equal(a.t["Ta"], b.t["Ta"]) and torch.equal(a.t["reward"], b.t["reward"])


@pytest.mark.parametrize("N", [5000, 2052, 513, 1023])
def test_fused_rollout_shapes_without_a_fused_kernel_step_inside_the_library(N):
    """N > 2048, or N > 512 with N % 4 != 0: single steps + k_rollout_accumulate, same accumulators as the fused kernels."""
    import mdr_amd
    env = mdr_amd.BatchedDemandResponseEnv(_cfg(N), nb_envs=3, device="cuda:0", seed=1, table_steps=8)
    twin = mdr_amd.BatchedDemandResponseEnv(_cfg(N), nb_envs=3, device="cuda:0", seed=1, table_steps=8)
    env.reset(episode=0)
    twin.reset(episode=0)
    res = env.rollout_fused(20, power_trace=True)
    rsum = torch.zeros_like(res["reward_sum"])
    terr = torch.zeros(3, dtype=torch.float64, device="cuda:0")
    serr = torch.zeros(3, dtype=torch.float64, device="cuda:0")
    for t in range(20):
        _, r, _, _ = twin.step_bangbang()
        rsum += r
        d = (twin.t["Ta"] - twin.t["target"]).double()
        terr += (d * d).sum(dim=1)
        serr += (twin.reg_signal() - twin.t["P"]) ** 2
        assert torch.equal(res["power_trace"][t], twin.t["P"])
    assert torch.equal(env.t["Ta"], twin.t["Ta"]) and env.steps_taken == 20
    assert torch.equal(res["reward_sum"], rsum)
    torch.testing.assert_close(res["sq_signal_error_sum"], serr, rtol=1e-12, atol=0)
    torch.testing.assert_close(res["sq_temp_error_sum"], terr, rtol=1e-6, atol=0)


def test_fused_rollout_in_interpolation_mode():
    """Chunks stop at every interpolatePower update (75 steps); the deferred signal-error term uses the final signal."""
    import mdr_amd
    from tests import golden_util as gu
    grid = gu.Golden("s12_interp_default_like").interp_grid()
    cfg = _cfg(256, **{"default_env_prop.power_grid_prop.base_power_mode": "interpolation",
                       "default_env_prop.power_grid_prop.signal_mode": "perlin"})
    a = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=6, device="cuda:0", seed=4, interp_grid=grid)
    b = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=6, device="cuda:0", seed=4, interp_grid=grid)
    a.reset(episode=0)
    b.reset(episode=0)
    res = a.rollout_fused(200)
    serr = torch.zeros(6, dtype=torch.float64, device="cuda:0")
    for t in range(200):
        _, _, _, info = b.step_bangbang()
        serr += (b.reg_signal() - info["cluster_hvac_power"]) ** 2
    for k in ("Ta", "Tm", "sso", "flags", "reward", "obs", "P", "base_power", "tab_signal"):
        assert torch.equal(a.t[k], b.t[k]), k
    torch.testing.assert_close(res["sq_signal_error_sum"], serr, rtol=1e-12, atol=0)
