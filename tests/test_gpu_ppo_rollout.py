"""Row f-2: device-resident PPO rollout collection (batched actor forward, sampling, transition tensors, returns)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _env(E, N):
    import mdr_amd
    cfg = mdr_amd.default_config()
    cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = N
    cfg["default_env_prop"]["power_grid_prop"]["base_power_mode"] = "constant"
    env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=3)
    env.reset(episode=0)
    return env


def test_actor_layout_matches_reference_state_dict_keys():
    from mdr_amd.rollout import ActorMLP, CriticMLP
    a = ActorMLP(51, 2, [100, 100])
    assert list(a.state_dict().keys()) == ["fc.0.weight", "fc.0.bias", "fc.1.weight", "fc.1.bias", "fc.2.weight", "fc.2.bias"]
    assert a.state_dict()["fc.0.weight"].shape == (100, 51) and a.state_dict()["fc.2.weight"].shape == (2, 100)
    assert CriticMLP(51, [100, 100]).state_dict()["fc.2.weight"].shape == (1, 100)
    p = a(torch.randn(7, 51))
    torch.testing.assert_close(p.sum(1), torch.ones(7))


def test_collect_rollout_shapes_sampling_and_returns():
    from mdr_amd.rollout import ActorMLP, CriticMLP, collect_ppo_rollout, discounted_returns
    E, N, T = 64, 50, 12
    env = _env(E, N)
    F = env.obs_vector_length()
    assert F == 51
    torch.manual_seed(0)
    actor = ActorMLP(F).cuda()
    critic = CriticMLP(F).cuda()
    gen = torch.Generator(device="cuda").manual_seed(1)
    ro = collect_ppo_rollout(env, actor, T, gamma=0.9, critic=critic, generator=gen)
    assert ro["state"].shape == (T + 1, E * N, F) and ro["action"].shape == (T, E * N)
    assert env.steps_taken == T
    # the stored probability is the actor's probability of the action that was taken, on the stored state
    p = actor(ro["state"][3])
    torch.testing.assert_close(p.gather(1, ro["action"][3][:, None]).squeeze(1), ro["a_prob"][3])
    # actions are drawn from those probabilities
    p1 = torch.stack([actor(ro["state"][t])[:, 1] for t in range(T)])
    assert abs(ro["action"].float().mean().item() - p1.mean().item()) < 0.01
    # state[t+1] is what the env shows after step t: its reward column dependencies are consistent with the env
    torch.testing.assert_close(ro["state"][T], env.obs_vector("rows").view(E * N, F))
    # returns: the reference's backward loop (agents/ppo.py:123-134) on one agent
    r = ro["reward"][:, 5].cpu().numpy()
    v = critic(ro["state"][T][5:6]).item()
    R, ref = v, []
    for t in reversed(range(T)):
        if t != T - 1:
            pass
        R = r[t] + 0.9 * R
        ref.insert(0, R)
    np.testing.assert_allclose(ro["return"][:, 5].cpu().numpy(), ref, rtol=1e-5)
    # done restarts the running return
    rew = torch.tensor([[1.0], [2.0], [3.0], [4.0]])
    done = torch.tensor([[False], [True], [False], [True]])
    got = discounted_returns(rew, done, 0.5)
    assert got.squeeze(1).tolist() == [1 + 0.5 * 2, 2.0, 3 + 0.5 * 4, 4.0]
    # same generator seed -> same rollout
    env2 = _env(E, N)
    ro2 = collect_ppo_rollout(env2, actor, T, gamma=0.9, critic=critic, generator=torch.Generator(device="cuda").manual_seed(1))
    assert torch.equal(ro["action"], ro2["action"]) and torch.equal(ro["reward"], ro2["reward"])


def test_collect_rollout_with_the_fused_policy_kernel():
    """fused=True (the default for the reference's actor shape): actor forward + Categorical.sample in one MFMA kernel."""
    from mdr_amd.rollout import ActorMLP, CriticMLP, collect_ppo_rollout
    E, N, T = 96, 50, 10
    env = _env(E, N)
    F = env.obs_vector_length()
    torch.manual_seed(3)
    actor = ActorMLP(F).cuda()
    critic = CriticMLP(F).cuda()
    ro = collect_ppo_rollout(env, actor, T, gamma=0.9, critic=critic, seed=5)
    assert hasattr(actor, "_mdr_fused_observe")                           # the fused path ran (observe -> act: default observation, N = 50)
    assert ro["action"].dtype == torch.int64 and set(ro["action"].unique().tolist()) <= {0, 1}
    # the stored probability is the actor's (torch fp32) probability of the stored action on the stored state
    for t in (0, 4, T - 1):
        p = actor(ro["state"][t])
        torch.testing.assert_close(p.gather(1, ro["action"][t][:, None]).squeeze(1), ro["a_prob"][t], rtol=1e-5, atol=2e-6)
    p1 = torch.stack([actor(ro["state"][t])[:, 1] for t in range(T)])
    assert abs(ro["action"].float().mean().item() - p1.mean().item()) < 0.01
    # state[t + 1] is the env's observation after step t, written straight into the buffer
    torch.testing.assert_close(ro["state"][T], env.obs_vector("rows").view(E * N, F))
    twin = _env(E, N)
    twin_obs = twin.obs_vector("rows").view(E * N, F)
    assert torch.equal(ro["state"][0], twin_obs)
    twin.step(ro["action"][0].to(torch.uint8).view(E, N))
    assert torch.equal(ro["state"][1], twin.obs_vector("rows").view(E * N, F))
    assert torch.equal(ro["reward"][0], twin.t["reward"].reshape(-1))
    # counter-based draws: same seed and step counters -> same rollout; another seed -> another one
    ro2 = collect_ppo_rollout(_env(E, N), actor, T, gamma=0.9, critic=critic, seed=5)
    ro3 = collect_ppo_rollout(_env(E, N), actor, T, gamma=0.9, critic=critic, seed=6)
    assert torch.equal(ro["action"], ro2["action"]) and torch.equal(ro["reward"], ro2["reward"])
    assert not torch.equal(ro["action"], ro3["action"])
    # a weight update invalidates the packed copy
    packed = actor._mdr_fused_observe[1]
    with torch.no_grad():
        actor.fc[2].bias.add_(1.0)
    collect_ppo_rollout(_env(E, N), actor, 2, seed=5)
    assert actor._mdr_fused_observe[1] is not packed


def test_collect_rollout_bf16x3_policy_stays_close_to_the_fp32_actor():
    from mdr_amd.rollout import ActorMLP, collect_ppo_rollout
    E, N, T = 64, 50, 6
    env = _env(E, N)
    F = env.obs_vector_length()
    torch.manual_seed(4)
    actor = ActorMLP(F).cuda()
    ro = collect_ppo_rollout(env, actor, T, seed=9, policy_precision="bf16x3")
    for t in range(T):
        p = actor(ro["state"][t]).gather(1, ro["action"][t][:, None]).squeeze(1)
        torch.testing.assert_close(ro["a_prob"][t], p, rtol=1e-3, atol=2e-5)
        assert float((ro["a_prob"][t] - p).abs().mean()) < 5e-6
    with pytest.raises(ValueError):
        collect_ppo_rollout(_env(E, N), actor, 2, policy_precision="fp8")


@pytest.mark.parametrize("with_bootstrap", [False, True])
def test_discounted_returns_kernel_equals_the_torch_scan(with_bootstrap):
    from mdr_amd import rollout
    T, A = 37, 10007
    g = torch.Generator(device="cuda").manual_seed(0)
    reward = torch.randn((T, A), device="cuda", generator=g)
    done = torch.rand((T, A), device="cuda", generator=g) < 0.1
    done[T - 1] = True
    boot = torch.randn((T, A), device="cuda", generator=g) if with_bootstrap else None
    got = rollout.discounted_returns(reward, done, 0.97, boot)
    ref = rollout.discounted_returns(reward.cpu(), done.cpu(), 0.97, boot.cpu() if boot is not None else None)     # the torch loop
    assert torch.equal(got.cpu(), ref)
    # 3-D shapes ([T, E, N]) and the reference's backward loop in double precision
    got3 = rollout.discounted_returns(reward.view(T, 1, A), done.view(T, 1, A), 0.97, boot.view(T, 1, A) if boot is not None else None)
    assert torch.equal(got3.view(T, A), got)
    r64, R, exp = reward[:, 5].double().cpu().numpy(), 0.0, []
    for t in reversed(range(T)):
        if bool(done[t, 5]):
            R = float(boot[t, 5]) if with_bootstrap else 0.0
        R = r64[t] + 0.97 * R
        exp.insert(0, R)
    np.testing.assert_allclose(got[:, 5].cpu().numpy(), exp, rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_dqn_transitions_follow_the_greedy_network_and_the_global_coin():
    """rollout.collect_dqn_transitions (train_dqn.py:55-91): with epsilon 0 every action is the argmax of the Q-network on the stored
    state (the plain fp32 torch forward; agents whose two Q-values lie within 1e-5 of each other aside), with epsilon 1 every env explores,
    in between whole envs explore or exploit; next_state[t] is state[t + 1] and what the env observes after the step."""
    import mdr_amd
    from mdr_amd.rollout import ActorMLP, collect_dqn_transitions
    cfg = mdr_amd.default_config()
    cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = 50
    cfg["default_env_prop"]["power_grid_prop"]["base_power_mode"] = "constant"
    torch.manual_seed(3)
    env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=64, device="cuda:0", seed=4)
    env.reset(episode=0)
    q = ActorMLP(env.obs_vector_length()).to("cuda:0")      # DQN_network has the Actor's stack of Linear layers (agents/network.py:58-77)
    out = collect_dqn_transitions(env, q, 6, epsilon=0.0, seed=1)
    assert not bool(out["explored"].any())
    with torch.no_grad():
        x = out["state"][:-1]
        for i, layer in enumerate(q.fc):
            x = layer(x)
            if i < 2:
                x = torch.relu(x)
    clear = (x[..., 0] - x[..., 1]).abs() > 1e-5
    assert torch.equal(out["action"][clear], x.argmax(dim=-1)[clear]) and float(clear.float().mean()) > 0.99
    assert torch.equal(out["state"][-1].view(64, 50, -1), env.obs_vector("rows"))
    assert torch.equal(out["reward"][-1].view(64, 50), env.t["reward"])
    every = collect_dqn_transitions(env, q, 4, epsilon=1.0, seed=2)
    assert bool(every["explored"].all()) and 0.4 < float(every["action"].float().mean()) < 0.6
    mixed = collect_dqn_transitions(env, q, 8, epsilon=0.5, epsilon_decay=0.9, min_epsilon=0.1, seed=3)
    frac = float(mixed["explored"].float().mean())
    assert 0.2 < frac < 0.6 and abs(mixed["epsilon"] - 0.5 * 0.9 ** 8) < 1e-12


def test_mappo_transitions_carry_the_other_agents_actions():
    """train_mappo.py:79-86 for all envs at once: `others_actions` of agent k is the step's action dict without k, in agent order
    (deepcopy(action); pop(k); list(values())) - the centralised critic's extra input (agents/mappo.py:64, 87)."""
    from mdr_amd.rollout import ActorMLP, CriticMLP, collect_ppo_rollout
    E, N, T = 5, 20, 4
    env = _env(E, N)
    torch.manual_seed(2)
    actor = ActorMLP(env.obs_vector_length()).cuda()
    ro = collect_ppo_rollout(env, actor, T, with_others_actions=True, seed=9)
    oa = ro["others_actions"]
    assert oa.shape == (T, E * N, N - 1) and oa.dtype == ro["action"].dtype
    act = ro["action"].cpu().numpy().reshape(T, E, N)
    got = oa.cpu().numpy().reshape(T, E, N, N - 1)
    for t in range(T):
        for e in range(E):
            action = {k: int(act[t, e, k]) for k in range(N)}
            for k in range(N):
                action_k = dict(action)
                action_k.pop(k)
                assert got[t, e, k].tolist() == list(action_k.values())
    # the critic of agents/mappo.py:87 takes state and others_actions side by side
    critic = CriticMLP(env.obs_vector_length() + N - 1).cuda()
    v = critic(torch.cat((ro["state"][0], oa[0].float()), dim=1))
    assert v.shape == (E * N, 1)
    with pytest.raises(ValueError):
        collect_ppo_rollout(env, actor, 2, critic=critic, with_others_actions=True)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_wide_observations_go_to_the_policy_as_planes_with_the_same_result(precision):
    """Message columns on (config.py message_properties: F = 121): without kept states the rollout hands the policy kernel feature
    planes instead of rows - same actions, probabilities and rewards as the rows path (store_states=True takes rows)."""
    import mdr_amd
    from mdr_amd.rollout import ActorMLP, CriticMLP, collect_ppo_rollout
    cfg = mdr_amd.default_config()
    cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = 20
    cfg["default_env_prop"]["power_grid_prop"]["base_power_mode"] = "constant"
    cfg["default_env_prop"]["message_properties"]["thermal"] = True
    cfg["default_env_prop"]["message_properties"]["hvac"] = True
    out = []
    torch.manual_seed(4)
    actor, critic = None, None
    for keep in (True, False):
        env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=37, device="cuda:0", seed=8)
        env.reset(episode=0)
        F = env.obs_vector_length()
        assert F == 11 + 10 * 11
        if actor is None:
            actor, critic = ActorMLP(F).cuda(), CriticMLP(F).cuda()
        out.append(collect_ppo_rollout(env, actor, 6, critic=critic, store_states=keep, seed=5, policy_precision=precision))
    kept, lean = out
    assert torch.equal(kept["action"], lean["action"]) and torch.equal(kept["a_prob"], lean["a_prob"])
    assert torch.equal(kept["reward"], lean["reward"]) and torch.equal(kept["return"], lean["return"])
    with torch.no_grad():
        p = actor(kept["state"][2]).gather(1, kept["action"][2][:, None]).squeeze(1)
    torch.testing.assert_close(kept["a_prob"][2], p, rtol=2e-3 if precision == "bf16x3" else 1e-5, atol=2e-5)


def test_dqn_transitions_one_kernel_form_stores_the_rows_of_the_rows_form():
    """With the default observation the DQN loop observes, evaluates and takes the argmax in one kernel and the rows reach `state`
    on the side: with epsilon 0 and the same seeds, states / actions / rewards equal those of an env with message columns OFF forced
    through observation rows (sharded envs and wide observations take that form) wherever no Q-value pair is a near tie."""
    import mdr_amd
    from mdr_amd import rollout as ro
    cfg = mdr_amd.default_config()
    cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = 20
    cfg["default_env_prop"]["power_grid_prop"]["base_power_mode"] = "constant"
    torch.manual_seed(3)
    q = None
    outs = []
    for force_rows in (False, True):
        env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=33, device="cuda:0", seed=4)
        env.reset(episode=0)
        if q is None:
            q = ro.ActorMLP(env.obs_vector_length()).to("cuda:0")
        keep = ro._observe_act_supported
        if force_rows:
            ro._observe_act_supported = lambda e, a: False
        try:
            outs.append(ro.collect_dqn_transitions(env, q, 1, epsilon=0.0, seed=1))
        finally:
            ro._observe_act_supported = keep
    one, rows = outs
    assert getattr(q, "_mdr_fused_observe", None) is not None
    assert torch.equal(one["state"][0], rows["state"][0])                  # the rows written on the side ARE obs_vector's rows
    with torch.no_grad():
        x = q.fc[2](torch.relu(q.fc[1](torch.relu(q.fc[0](one["state"][0])))))
    clear = (x[:, 0] - x[:, 1]).abs() > 1e-5
    assert torch.equal(one["action"][0][clear], rows["action"][0][clear]) and float(clear.float().mean()) > 0.99
    if bool((one["action"][0] == rows["action"][0]).all()):
        assert torch.equal(one["state"][1], rows["state"][1]) and torch.equal(one["reward"], rows["reward"])
