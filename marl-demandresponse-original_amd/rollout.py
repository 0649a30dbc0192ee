"""Device-resident PPO rollout collection on top of the batched env.

What train_ppo.py:60-108 does per step for ONE env - ``normStateDict`` -> ``PPO.select_action`` (actor forward on
a batch of 1, ``Categorical.sample``) -> ``env.step`` -> ``store_transition`` - is done here for all E x N agents
at once with tensors that never leave the GPU:

    obs rows [E*N, F]  (HIP: mdr_env_obs_vector)  ->  actor MLP (rocBLAS GEMMs via torch)  ->  multinomial
    ->  actions uint8 [E, N]  ->  mdr_env_step  ->  reward [E, N]

PyTorch is used for the policy network only (tiny dense layers: plain library GEMMs); the env arithmetic stays in
libmdr_hip.so.  ``ActorMLP`` / ``CriticMLP`` keep the reference's module layout (agents/network.py:14-57:
``fc`` ModuleList, ReLU, softmax head) so a reference ``actor.pth`` state_dict loads unchanged.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F


class ActorMLP(nn.Module):
    """agents/network.py:14-33 - Linear/ReLU stack with a softmax head; state_dict keys ``fc.<i>.weight|bias``."""

    def __init__(self, num_state: int, num_action: int = 2, layers: Sequence[int] = (100, 100)):
        super().__init__()
        dims = [num_state] + [int(x) for x in layers]
        self.layers = list(layers)
        self.fc = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)])
        self.fc.append(nn.Linear(dims[-1], num_action))

    def forward(self, x):
        for lin in self.fc[:-1]:
            x = F.relu(lin(x))
        return F.softmax(self.fc[-1](x), dim=1)


class CriticMLP(nn.Module):
    """agents/network.py:36-57."""

    def __init__(self, num_state: int, layers: Sequence[int] = (100, 100)):
        super().__init__()
        dims = [num_state] + [int(x) for x in layers]
        self.layers = list(layers)
        self.fc = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)])
        self.fc.append(nn.Linear(dims[-1], 1))

    def forward(self, x):
        for lin in self.fc[:-1]:
            x = F.relu(lin(x))
        return self.fc[-1](x)


def _discounted_returns_hip(reward, done, gamma, bootstrap):
    import ctypes as C
    from . import _native as nat
    lib = nat.load()
    out = torch.empty_like(reward)
    d8 = done.contiguous().view(torch.uint8) if done.dtype == torch.bool else done.to(torch.uint8).contiguous()
    boot = bootstrap.to(torch.float32).contiguous() if bootstrap is not None else None
    T = reward.shape[0]
    A = reward[0].numel()
    with torch.cuda.device(reward.device):
        rc = lib.mdr_discounted_returns(C.c_void_p(reward.data_ptr()), C.c_void_p(d8.data_ptr()),
                                        C.c_void_p(boot.data_ptr()) if boot is not None else None, C.c_float(gamma), T, A,
                                        C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream(reward.device).cuda_stream))
    if rc != 0:
        raise RuntimeError("mdr_discounted_returns failed: %s" % lib.mdr_status_string(rc).decode())
    return out


def _fused_policy(actor, dev, precision: str = "fp32", observe: bool = False, msg_floats: int = 40, greedy: bool = False):
    """FusedActor of `actor`, re-packed only when a parameter changed (torch bumps `_version` on in-place updates).
    ``observe``: packed for ``FusedActor.sample_env`` (W1's columns in the order the observe -> act kernels stage the features)."""
    from .policy import BF16X3, FEATURES_NORMSTATE, FEATURES_OBSERVE, FusedActor
    if precision not in ("fp32", "bf16x3"):
        raise ValueError("policy_precision must be 'fp32' or 'bf16x3'")
    key = tuple((p.data_ptr(), p._version) for p in actor.parameters()) + (str(dev), precision, msg_floats if observe else 0, bool(greedy))
    slot = "_mdr_fused_observe" if observe else "_mdr_fused"
    cached = getattr(actor, slot, None)
    if cached is None or cached[0] != key:
        layout = BF16X3 if precision == "bf16x3" and actor.fc[0].in_features <= 128 else None      # None: the exact-fp32 form that fits
        cached = (key, FusedActor.from_module(actor, device=dev, layout=layout, greedy=greedy,
                                              feature_order=FEATURES_OBSERVE if observe else FEATURES_NORMSTATE, observe_msg_floats=msg_floats))
        setattr(actor, slot, cached)
    return cached[1]


def _observe_act_supported(env, actor) -> bool:
    """Can ``FusedActor.sample_env`` serve this env / actor?  At most 13 senders (fewer than the houses) - the circular neighbours,
    a link table or random_sample - with 4-field messages, any of the optional STATE columns, link defects, at most 64 features,
    unsharded houses."""
    mp = env.config["default_env_prop"]["message_properties"]
    c = _observe_senders(env)
    F = env.obs_vector_length()
    return (not env.sharded and not mp["thermal"] and not mp["hvac"] and c <= 13 and env.nb_houses >= c + 1 and F <= 64
            and actor.fc[0].in_features == F)


def _observe_senders(env) -> int:
    """Message slots per house of this env's observation (the width of its link table where it has one)."""
    return int(env._obs_spec("rows").nb_comm)


def _fusable(actor) -> bool:
    fc = getattr(actor, "fc", None)
    if fc is None or len(fc) != 3 or not all(isinstance(m, nn.Linear) for m in fc):
        return False
    return fc[0].out_features <= 127 and fc[1].out_features <= 127 and fc[2].out_features == 2 and fc[0].weight.is_cuda


def discounted_returns(reward: torch.Tensor, done: torch.Tensor, gamma: float,
                       bootstrap: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The Monte-Carlo return scan of PPO.update (agents/ppo.py:123-134) along dim 0 of ``reward`` [T, ...]:
    R_t = r_t + gamma * R_{t+1}; where ``done[t]`` the running return restarts from ``bootstrap[t]`` (the critic's value
    of the next state, or 0 with zero_eoepisode_return) before adding r_t."""
    T = reward.shape[0]
    if reward.is_cuda and reward.dtype == torch.float32 and reward.is_contiguous() and done.shape == reward.shape and T > 0:
        return _discounted_returns_hip(reward, done, float(gamma), bootstrap)     # one launch instead of 4 T
    out = torch.empty_like(reward)
    running = torch.zeros_like(reward[0])
    for t in range(T - 1, -1, -1):
        if bootstrap is not None:
            running = torch.where(done[t], bootstrap[t], running)
        else:
            running = torch.where(done[t], torch.zeros_like(running), running)
        running = reward[t] + gamma * running
        out[t] = running
    return out


def others_actions(action: torch.Tensor, nb_envs: int, nb_houses: int) -> torch.Tensor:
    """MAPPO's ``Transition.others_actions`` (train_mappo.py:79-84: the step's action dict without agent k, in agent order) for the
    flat per-agent layout of ``collect_ppo_rollout``: ``action`` [T, E*N] -> [T, E*N, N-1] of the same dtype, entry j of agent i
    being the action of agent j (j < i) or j + 1 (j >= i) of the same env - what agents/mappo.py:64,87 concatenates to the state as
    the centralised critic's input.  One gather on the device; meant for the reference's cluster sizes (N - 1 values per agent-step)."""
    T = action.shape[0]
    E, N = int(nb_envs), int(nb_houses)
    if action.shape[1] != E * N:
        raise ValueError("action must be [T, nb_envs * nb_houses]")
    i = torch.arange(N, device=action.device)[:, None]
    j = torch.arange(N - 1, device=action.device)[None, :]
    idx = j + (j >= i).to(j.dtype)                                   # [N, N-1]: the other agents of agent i, in agent order
    return action.view(T, E, N)[:, :, idx].reshape(T, E * N, N - 1)


@torch.no_grad()
def collect_ppo_rollout(env, actor: nn.Module, nb_steps: int, gamma: float = 0.99, critic: Optional[nn.Module] = None,
                        generator: Optional[torch.Generator] = None, store_states: bool = True,
                        fused: Optional[bool] = None, seed: int = 0, policy_precision: str = "fp32",
                        observe_act: Optional[bool] = None, obs_planes: Optional[bool] = None,
                        with_others_actions: bool = False) -> Dict[str, torch.Tensor]:
    """Roll ``nb_steps`` with actions sampled from ``actor`` for every agent of every env.

    ``fused`` (default: whenever the actor has the reference's shape - two hidden layers of <= 127 units, two actions -
    and no torch ``generator`` is given): the actor forward, softmax and ``Categorical.sample`` run as ONE HIP kernel on
    the matrix cores in exact fp32 (``mdr_amd.policy.FusedActor``), the draws coming from Philox4x32-10 keyed by
    ``seed`` with the env's step counter in the counter; otherwise torch GEMMs + ``torch.multinomial``.
    ``policy_precision="bf16x3"`` runs the fused kernel on bf16 matrix instructions with every operand split into a
    bf16 head and tail (16 significand bits; probabilities within ~1e-5 of the fp32 forward) - about 2.7x faster.
    ``observe_act`` (default: for every shape the kernels cover - the reference's default observation of 51 features and 10
    circular neighbours, optional state columns, up to 13 senders - circular neighbours, a link table or random_sample -, link defects):
    observation and policy are ONE kernel (``FusedActor.sample_env``): the 51 features of every agent are built in LDS from the
    compact state and fed to the matrix cores from there; with ``store_states`` the same kernel copies the rows into the
    transition buffer on the side (written once, never read back by the policy), without it they are not materialised at all.

    ``obs_planes`` (default: False whenever observation and policy are one kernel): with False the env steps WITHOUT writing its
    seven per-step observation planes during the collection - nothing here reads them (71 instead of 99 bytes per house-step,
    train_ppo.py:69-72 observes through normStateDict) - and brings them up to date once at the end.

    Returns tensors with the agents flattened as [T, E*N, ...] in the reference's per-agent order:
    ``state`` [T+1, E*N, F] (``state[t+1]`` is ``next_state[t]``; omitted if ``store_states`` is False), ``action`` int64,
    ``a_prob`` (probability of the taken action - the reference stores the probability, not its log: agents/ppo.py:75),
    ``reward``, ``done`` (True on the last step: train_ppo.py:84) and ``return`` (discounted_returns).
    ``with_others_actions``: MAPPO's transitions (train_mappo.py:46, 79-86) - adds ``others_actions`` [T, E*N, N-1] (``others_actions()``);
    the bootstrap through ``critic`` is PPO's (a critic over the state alone) and is refused together with it."""
    E, N = env.nb_envs, env.nb_houses
    if with_others_actions and critic is not None:
        raise ValueError("MAPPO's critic takes state + others_actions (agents/mappo.py:87); the reference does not bootstrap it either")
    F_len = env.obs_vector_length()
    dev = env.device
    T = int(nb_steps)
    states = torch.empty((T + 1, E * N, F_len), dtype=torch.float32, device=dev) if store_states else None
    action = torch.empty((T, E * N), dtype=torch.int64, device=dev)
    a_prob = torch.empty((T, E * N), dtype=torch.float32, device=dev)
    reward = torch.empty((T, E * N), dtype=torch.float32, device=dev)
    policy = None
    if fused is None:
        fused = generator is None and _fusable(actor)
    if observe_act is None:
        # every shape the one-kernel form covers (the default observation; optional state columns, another neighbour count, link
        # defects): it is the faster one in both precisions (r03, 4096 x 1024, F = 58: 1338 against 1507 us per step in exact fp32,
        # 634 against 860 us with the split-bf16 policy)
        observe_act = bool(fused) and _observe_act_supported(env, actor)
    if observe_act and not fused:
        raise ValueError("observe_act needs the fused policy")
    if fused:
        policy = _fused_policy(actor, dev, policy_precision, observe=observe_act,
                               msg_floats=4 * _observe_senders(env))
        act_u8 = torch.empty((T, E * N), dtype=torch.uint8, device=dev)

    # wide observations (the optional message columns: 81 / 91 / 121 features) that nobody keeps go to the policy kernel as feature
    # PLANES [F][E][N]: a feature load then touches 4 cache lines instead of 64 (F = 121 at 4096 x 1024: actor 2035 -> 1634 us in
    # exact fp32, 1262 -> 1083 us bf16x3, and the planes kernel writes 383 us against 446 for rows)
    use_planes = policy is not None and not observe_act and not store_states and F_len > 64

    def observe(t):      # straight into the transition buffer when states are kept: no 4 F bytes/agent copy per step
        if store_states:
            return env.obs_vector("rows", out=states[t].view(E, N, F_len)).view(E * N, F_len)
        if use_planes:
            return env.obs_vector("planes")
        return env.obs_vector("rows").view(E * N, F_len)

    if obs_planes is None:
        obs_planes = not observe_act
    planes_were_on = bool(getattr(env, "_obs_planes_on", True))
    if not obs_planes and planes_were_on and not env.sharded:
        env.set_obs_planes(False)
    obs = None if observe_act else observe(0)
    step0 = env.steps_taken
    for t in range(T):
        if observe_act:             # normStateDict + select_action for all agents in ONE kernel; `state` stored on the side if wanted
            policy.sample_env(env, seed, step0 + t, action=act_u8[t], a_prob=a_prob[t], rows_out=states[t] if store_states else None)
            _, r, _, _ = env.step(act_u8[t].view(E, N))
        elif policy is not None:    # agents/ppo.py:68-75 for all agents: one kernel, action and a_prob written in place
            policy.sample(obs, seed, step0 + t, action=act_u8[t], a_prob=a_prob[t])
            _, r, _, _ = env.step(act_u8[t].view(E, N))
        else:
            probs = actor(obs)
            a = torch.multinomial(probs, 1, generator=generator).squeeze(1)        # Categorical(action_prob).sample()
            action[t] = a
            a_prob[t] = probs.gather(1, a[:, None]).squeeze(1)
            _, r, _, _ = env.step(a.to(torch.uint8).view(E, N))
        reward[t] = r.reshape(-1)
        if not observe_act:
            obs = observe(t + 1)
    if (observe_act and (store_states or critic is not None)) or (use_planes and critic is not None):
        use_planes = False
        obs = observe(T)            # next_state of the last transition / the critic's bootstrap input (rows)
    if planes_were_on and not env._obs_planes_on:
        env.set_obs_planes(True)    # one pass over the state: the planes of the last step
    if policy is not None:
        action.copy_(act_u8)        # one widening pass at the end (the reference stores Categorical's int64)
    done = torch.zeros((T, E * N), dtype=torch.bool, device=dev)
    done[T - 1] = True
    bootstrap = None
    if critic is not None:
        bootstrap = torch.zeros((T, E * N), dtype=torch.float32, device=dev)
        bootstrap[T - 1] = critic(obs).squeeze(1)
    out = {"action": action, "a_prob": a_prob, "reward": reward, "done": done,
           "return": discounted_returns(reward, done, gamma, bootstrap)}
    if store_states:
        out["state"] = states
    if with_others_actions:
        out["others_actions"] = others_actions(action, E, N)
    return out


@torch.no_grad()
def collect_dqn_transitions(env, q_net: nn.Module, nb_steps: int, epsilon: float = 1.0, epsilon_decay: float = 1.0,
                            min_epsilon: float = 0.0, seed: int = 0, policy_precision: str = "fp32") -> Dict[str, torch.Tensor]:
    """The interaction loop of train_dqn.py:55-91 for all envs at once, transitions kept on the GPU: every step the epsilon-greedy
    action of ``q_net`` (agents/network.py DQN_network: the same MLP stack as the PPO actor; ``DQNAgent.select_action`` is an argmax -
    agents/dqn.py:52-56 - evaluated by the fused policy kernel in its greedy form), ``env.step``, and the transition
    (state, action, reward, next_state) of ``store_transition`` (agents/dqn.py:58-63).  Exploration as the reference's loop ends up doing it:
    its per-agent draw (train_dqn.py:61-65) is overwritten by ONE draw per step (67-70) - with probability epsilon every agent of the env
    acts at random, otherwise every agent acts greedily; here one such draw per env and step.  epsilon <- max(epsilon * decay, min)
    after every step (19-22, 91).  Returns ``state`` [T + 1, E*N, F] (``state[t + 1]`` is ``next_state[t]``), ``action`` int64,
    ``reward`` [T, E*N], ``explored`` bool [T, E] and the final ``epsilon``."""
    from .policy import FusedActor
    E, N = env.nb_envs, env.nb_houses
    dev = env.device
    F_len = env.obs_vector_length()
    T = int(nb_steps)
    layout = None
    if policy_precision == "bf16x3" and q_net.fc[0].in_features <= 128:
        from .policy import BF16X3
        layout = BF16X3
    # observation, Q-network and argmax in ONE kernel wherever the rollout collection has that form; the rows land in `state` on the side
    one_kernel = _fusable(q_net) and _observe_act_supported(env, q_net)
    if one_kernel:
        greedy = _fused_policy(q_net, dev, policy_precision, observe=True, msg_floats=4 * _observe_senders(env), greedy=True)
    else:
        greedy = FusedActor.from_module(q_net, device=dev, layout=layout, greedy=True)
    state = torch.empty((T + 1, E * N, F_len), dtype=torch.float32, device=dev)
    action = torch.empty((T, E * N), dtype=torch.int64, device=dev)
    reward = torch.empty((T, E * N), dtype=torch.float32, device=dev)
    explored = torch.empty((T, E), dtype=torch.bool, device=dev)
    act = torch.empty(E * N, dtype=torch.uint8, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed))
    eps = float(epsilon)
    if not one_kernel:
        env.obs_vector("rows", out=state[0].view(E, N, F_len))
    for t in range(T):
        if one_kernel:
            greedy.sample_env(env, seed, env.steps_taken, action=act, rows_out=state[t])
        else:
            greedy.sample(state[t], seed, env.steps_taken, action=act)
        coin = torch.rand(E, device=dev, generator=gen) < eps
        random_act = torch.randint(0, 2, (E, N), device=dev, generator=gen, dtype=torch.uint8)
        chosen = torch.where(coin[:, None], random_act, act.view(E, N))
        _, r, _, _ = env.step(chosen)
        action[t] = chosen.view(-1).to(torch.int64)
        reward[t] = r.view(-1)
        explored[t] = coin
        if not one_kernel or t == T - 1:
            env.obs_vector("rows", out=state[t + 1].view(E, N, F_len))
        eps = max(eps * float(epsilon_decay), float(min_epsilon))
    return {"state": state, "action": action, "reward": reward, "explored": explored, "epsilon": eps}


def deploy_controller(env, kind: str, nb_steps: int) -> Dict[str, torch.Tensor]:
    """The evaluation loop of main-deploy.py:99-152 under one of the reference's rule-based agents (``agents_dict`` of
    main-deploy.py:22-34) for all envs at once: "bangbang", "deadband", "basic", "always_on" run inside the step kernels
    (``env.set_controller``; the houses stay in registers across steps where a fused rollout kernel exists), "greedy_myopic" ranks
    the houses of every env on the device before each step.  Returns the metrics the script accumulates: ``reward_sum`` [E, N],
    ``sq_temp_error_sum`` [E], ``sq_signal_error_sum`` [E]."""
    if kind != "greedy_myopic":
        env.set_controller(kind)
        out = None if env.sharded else env.rollout_fused(int(nb_steps))
        if out is not None:
            return out
    E, N = env.nb_envs, env.nb_houses
    out = {"reward_sum": torch.zeros((E, N), dtype=torch.float32, device=env.device),
           "sq_temp_error_sum": torch.zeros(E, dtype=torch.float64, device=env.device),
           "sq_signal_error_sum": torch.zeros(E, dtype=torch.float64, device=env.device)}
    for _ in range(int(nb_steps)):
        if kind == "greedy_myopic":
            _, r, _, info = env.step_greedy_myopic()
        else:
            _, r, _, info = env.step_controller()
        out["reward_sum"] += r
        d = (env.t["Ta"] - env.t["target"]).double()
        out["sq_temp_error_sum"] += (d * d).sum(dim=1)
        out["sq_signal_error_sum"] += (env.reg_signal() - info["cluster_hvac_power"]) ** 2
    return out


def deploy_policy(env, policy, nb_steps: int, seed: int = 0, use_graph: Optional[bool] = None, policy_precision: str = "fp32",
                  greedy: bool = False) -> Dict[str, torch.Tensor]:
    """The evaluation loop of main-deploy.py:99-152 with a learned agent (PPOAgent / DQNAgent, agents/rl_controllers.py) for all
    envs at once: every step observation -> ``policy`` -> ``env.step``, with the metrics the script accumulates: ``reward_sum``
    [E, N], ``sq_temp_error_sum`` [E] (sum over steps and houses of (house_temp - target)^2), ``sq_signal_error_sum`` [E] (sum over
    steps of (reg_signal - cluster_hvac_power)^2).

    ``policy``: the network itself (``ActorMLP`` / the reference's ``Actor``; ``greedy=True`` for a ``DQN_network``: argmax) - packed
    here, and observation and policy are then ONE kernel wherever ``collect_ppo_rollout`` would make them one (no observation rows at
    all) - or a ready ``FusedActor``: one packed with ``feature_order=FEATURES_OBSERVE`` takes the same one-kernel path, any other
    gets observation rows.

    ``use_graph`` (default: when the env was built with ``graph_mode=True``): the step is captured once in a
    ``torch.cuda.CUDAGraph`` and replayed - the launch-bound regime of small batches."""
    from .policy import FEATURES_OBSERVE
    E, N = env.nb_envs, env.nb_houses
    dev = env.device
    F_len = env.obs_vector_length()
    if isinstance(policy, nn.Module):
        if not _fusable(policy):
            raise ValueError("deploy_policy takes Linear(F,H1) - Linear(H1,H2) - Linear(H2,2) networks on the device (or a FusedActor)")
        policy = _fused_policy(policy, dev, policy_precision, observe=_observe_act_supported(env, policy),
                               msg_floats=4 * _observe_senders(env), greedy=greedy)
    observe_act = getattr(policy, "feature_order", 0) == FEATURES_OBSERVE
    obs = None if observe_act else torch.empty((E, N, F_len), dtype=torch.float32, device=dev)
    act = torch.empty(E * N, dtype=torch.uint8, device=dev)
    out = {"reward_sum": torch.zeros((E, N), dtype=torch.float32, device=dev),
           "sq_temp_error_sum": torch.zeros(E, dtype=torch.float64, device=dev),
           "sq_signal_error_sum": torch.zeros(E, dtype=torch.float64, device=dev)}
    graph_mode = bool(getattr(env, "graph_mode", False))
    if use_graph is None:
        use_graph = graph_mode
    if use_graph and not graph_mode:
        raise ValueError("use_graph needs an env built with graph_mode=True")
    step0 = 0 if graph_mode else env.steps_taken
    step_dev = env.device_time_index if graph_mode else None

    def one_step(t):
        if observe_act:      # normStateDict + act for all agents in one kernel: no observation rows
            policy.sample_env(env, seed, step0 + t, action=act, step_dev=step_dev)
        else:
            env.obs_vector("rows", out=obs)
            policy.sample(obs.view(E * N, F_len), seed, step0 + t, action=act, step_dev=step_dev)
        _, r, _, info = env.step(act.view(E, N))
        out["reward_sum"] += r
        d = (env.t["Ta"] - env.t["target"]).double()
        out["sq_temp_error_sum"] += (d * d).sum(dim=1)
        out["sq_signal_error_sum"] += (env.reg_signal() - info["cluster_hvac_power"]) ** 2

    if not use_graph or nb_steps < 3:
        for t in range(nb_steps):
            one_step(0 if graph_mode else t)
        return out
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):          # one eager step on the capture stream: warms up and syncs the device cursor
        one_step(0)
    torch.cuda.current_stream(dev).wait_stream(side)
    done = 1
    g = None
    while done < nb_steps:
        env.graph_replayed(0)                  # tables refilled / device cursor current before the room is read
        n = min(env.graph_room(), nb_steps - done)
        if n < 1:                              # the next step lands on an interpolatePower update: host work, never replayed
            one_step(0)
            done += 1
            continue
        if g is None:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                one_step(0)
        for _ in range(n):
            g.replay()
        env.graph_replayed(n)
        done += n
    return out
