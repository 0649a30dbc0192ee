"""mdr_buffers_t.obs == NULL: the step kernels skip the seven per-step observation planes (71 instead of 99 bytes per house-step)
for loops that observe through normStateDict / observe -> act (train_ppo.py:69-72); everything else stays bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfg(n, mode="individual_L2"):
    import mdr_amd
    cfg = mdr_amd.default_config()
    env = cfg["default_env_prop"]
    env["cluster_prop"]["nb_agents"] = n
    env["power_grid_prop"]["base_power_mode"] = "constant"
    env["power_grid_prop"]["signal_mode"] = "perlin"
    env["reward_prop"]["temp_penalty_mode"] = mode
    cfg["noise_house_prop"]["noise_mode"] = "big_noise"
    cfg["noise_hvac_prop"]["noise_mode"] = "big_noise"
    return cfg


@pytest.mark.parametrize("E,N,mode", [(64, 1024, "individual_L2"), (300, 20, "mixture"), (200, 50, "common_L2"), (3, 5000, "common_max"),
                                      (5, 1001, "individual_L2"), (262144, 1, "individual_L2"), (2, 3000, "mixture")])
def test_steps_without_planes_leave_the_same_state_and_rewards(E, N, mode):
    import mdr_amd
    cfg = _cfg(N, mode)
    ref = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=4, table_steps=8)
    bare = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=4, table_steps=8, obs_planes=False)
    assert bare.t["obs"].numel() == 0
    ref.reset(episode=0)
    bare.reset(episode=0)
    gen = torch.Generator(device="cuda").manual_seed(1)
    for t in range(20):
        if t % 3 == 2:
            ref.step_bangbang()
            bare.step_bangbang()
        else:
            act = (torch.rand((E, N), device="cuda", generator=gen) < 0.5).to(torch.uint8)
            ref.step(act)
            bare.step(act)
        for k in ("Ta", "Tm", "sso", "flags", "reward", "P", "actions"):
            assert torch.equal(ref.t[k], bare.t[k]), (t, k)
    ref.rollout(9)
    bare.rollout(9)
    if N <= 2048 or N > 4096:
        a = ref.rollout_fused(11) if N <= 2048 else ref.rollout_persistent(11)
        b = bare.rollout_fused(11) if N <= 2048 else bare.rollout_persistent(11)
        for k in a:
            assert torch.equal(a[k], b[k]), k
    for k in ("Ta", "Tm", "sso", "flags", "reward", "P"):
        assert torch.equal(ref.t[k], bare.t[k]), k
    assert torch.equal(ref.obs_vector("rows"), bare.obs_vector("rows"))      # the flat observation reads the state, not the planes
    bare.set_obs_planes(True)                                               # planes bound later: brought up to date from the state
    assert torch.equal(bare.t["obs"], ref.t["obs"])
    ref.step_bangbang()
    bare.step_bangbang()
    assert torch.equal(bare.t["obs"], ref.t["obs"])
    ref.set_obs_planes(False)                                               # and off again: the buffer keeps its last contents, state moves on
    before = ref.t["obs"].clone()
    ref.step_bangbang()
    bare.step_bangbang()
    assert torch.equal(ref.t["obs"], before) and torch.equal(ref.t["Ta"], bare.t["Ta"])
    ref.set_obs_planes(True)
    assert torch.equal(ref.t["obs"], bare.t["obs"])


def test_rollout_collection_skips_the_planes_and_returns_the_same_transitions():
    import mdr_amd
    from mdr_amd.rollout import ActorMLP, collect_ppo_rollout
    cfg = _cfg(64)
    torch.manual_seed(0)
    outs = []
    for planes in (True, False):
        env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=96, device="cuda:0", seed=9)
        env.reset(episode=0)
        torch.manual_seed(0)
        actor = ActorMLP(env.obs_vector_length()).cuda()
        outs.append((collect_ppo_rollout(env, actor, 6, seed=3, obs_planes=planes), env))
    (a, ea), (b, eb) = outs
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert eb._obs_planes_on and torch.equal(ea.t["obs"], eb.t["obs"])      # switched back on and refreshed at the end
