"""Observe -> act in one kernel (mdr_env_actor_sample): utils.normStateDict of every agent built in LDS from the
compact state and fed to the matrix-core Actor forward, against (a) the two-kernel path - mdr_env_obs_vector rows + mdr_actor_sample -
whose observation is pinned on the reference's normStateDict vectors (tests/test_obs_vector.py) and (b) a plain PyTorch fp32 forward
of the same Actor on those rows.  The staged features are the rows' bits; only the k order of layer 1 differs (messages first), so
the probabilities agree to fp32 summation-order rounding (2e-6) - bf16x3: to its own 2e-5."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfg(N, **patches):
    import mdr_amd
    cfg = mdr_amd.default_config()
    cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = N
    cfg["default_env_prop"]["power_grid_prop"]["base_power_mode"] = "constant"
    cfg["default_env_prop"]["power_grid_prop"]["signal_mode"] = "perlin"
    cfg["noise_house_prop"]["noise_mode"] = "big_noise"
    cfg["noise_hvac_prop"]["noise_mode"] = "big_noise"
    cfg["default_hvac_prop"]["lockout_noise"] = 15
    for dotted, v in patches.items():
        node = cfg
        parts = dotted.split(".")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = v
    return cfg


def _actor(seed=0, scale=2.0, layers=(100, 100)):
    from mdr_amd.rollout import ActorMLP
    torch.manual_seed(seed)
    actor = ActorMLP(51, 2, layers).to("cuda:0")
    with torch.no_grad():
        for lin in actor.fc:
            lin.weight.mul_(scale)
            lin.bias.uniform_(-0.5, 0.5)
    return actor


def _walk(env, steps, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    for _ in range(steps):
        env.step((torch.rand((env.nb_envs, env.nb_houses), generator=g) < 0.5).to(torch.uint8).cuda())


@pytest.mark.parametrize("E,N", [(3, 32), (5, 64), (2, 1024), (33, 96), (1, 4096), (700, 160),
                                 # any cluster size: tiles that start inside an env, span up to four envs, end in a partial tile
                                 (100, 20), (37, 50), (5, 11), (3, 33), (9, 100), (1, 1000), (7, 31), (64, 12), (2, 4100), (1, 13)])
@pytest.mark.parametrize("layout,layers", [(1, (100, 100)), (2, (100, 100)), (1, (127, 120)), (2, (64, 32)), (3, (100, 100)), (3, (97, 99))])
def test_observe_act_equals_rows_then_actor(E, N, layout, layers):
    import mdr_amd
    from mdr_amd.policy import FEATURES_OBSERVE, FusedActor
    env = mdr_amd.BatchedDemandResponseEnv(_cfg(N), nb_envs=E, device="cuda:0", seed=5 + N)
    env.reset(episode=0)
    actor = _actor(seed=E, layers=layers)
    by_rows = FusedActor.from_module(actor, layout=layout)
    by_state = FusedActor.from_module(actor, layout=layout, feature_order=FEATURES_OBSERVE)
    for k in range(3):                                   # reset state (all off, sso = lockout), then two walked states
        rows = env.obs_vector("rows").view(E * N, 51)
        a0, p0, probs0 = by_rows.sample(rows, seed=9, step=k, want_probs=True)
        a1, p1, probs1 = by_state.sample_env(env, seed=9, step=k, want_probs=True)
        with torch.no_grad():
            ref = actor(rows)
        if layout == 2:
            torch.testing.assert_close(probs1, ref, rtol=2e-3, atol=2e-5)
            torch.testing.assert_close(probs1, probs0, rtol=2e-3, atol=2e-5)
        else:
            torch.testing.assert_close(probs1, ref, rtol=1e-5, atol=2e-6)
            torch.testing.assert_close(probs1, probs0, rtol=1e-5, atol=2e-6)
        assert torch.equal(p1, probs1.gather(1, a1.long()[:, None]).squeeze(1))
        # the rows written on the side are the rows of mdr_env_obs_vector, bit for bit, and storing them changes nothing else
        kept = torch.full((E * N, 51), float("nan"), device="cuda:0")
        a2, p2, probs2 = by_state.sample_env(env, seed=9, step=k, want_probs=True, rows_out=kept)
        assert torch.equal(kept, rows), "rows_out differs from obs_vector('rows')"
        assert torch.equal(a2, a1) and torch.equal(probs2, probs1)
        # same Philox draw per agent: the actions differ only where u falls between the two (nearly equal) probabilities
        differ = a0 != a1
        assert int(differ.sum()) <= max(2, E * N // 20000)
        if bool(differ.any()):
            assert float((probs0[differ, 0] - probs1[differ, 0]).abs().max()) < (1e-4 if layout == 2 else 1e-5)
        _walk(env, 7, seed=k)


from tests.conftest import poison_lds as _poison_lds  # noqa: E402

_STATE_FLAGS = ("hour", "day", "solar_gain", "thermal", "hvac")


def _shape_cfg(N, flags=(), nb_comm=10, defects=0.0):
    patches = {"default_env_prop.cluster_prop.nb_agents_comm": nb_comm, "default_env_prop.cluster_prop.comm_defect_prob": defects,
               "default_house_prop.solar_gain_bool": True}
    for f in flags:
        patches["default_env_prop.state_properties." + f] = True
    return _cfg(N, **patches)


@pytest.mark.parametrize("E,N,flags,nb_comm,defects", [
    (6, 64, ("thermal", "hvac"), 10, 0.0),                      # F = 58
    (3, 1024, _STATE_FLAGS, 10, 0.0),                           # every optional state column: F = 63
    (40, 50, ("hour", "day"), 10, 0.1),                         # the reference's deployment size, 10 % link defects
    (9, 96, (), 10, 0.5),                                       # default shape, every second link dead
    (100, 20, ("solar_gain",), 6, 0.0),                         # fewer neighbours
    (17, 33, ("solar_gain",), 13, 0.2),                         # 13 neighbours: 6 before, 7 after (F = 64)
    (5, 12, (), 11, 0.0),                                       # an odd count in the smallest env that has it
    (64, 14, _STATE_FLAGS, 4, 0.3),
    (2, 4100, ("thermal",), 12, 0.1),
    (8, 40, (), 0, 0.0),                                        # no neighbours at all: the 11 own features
    (1, 1000, ("day",), 1, 0.0),
])
@pytest.mark.parametrize("layout", [3, 2, 1])
def test_observe_act_extended_shapes(E, N, flags, nb_comm, defects, layout):
    """Optional state columns (utils.py:774-830), any number of circular neighbours (env 816-828), link defects (env 988-1002):
    rows written on the side == mdr_env_obs_vector bit for bit, probabilities == the actor on those rows."""
    import mdr_amd
    from mdr_amd.policy import FEATURES_OBSERVE, FusedActor
    from mdr_amd.rollout import ActorMLP
    env = mdr_amd.BatchedDemandResponseEnv(_shape_cfg(N, flags, nb_comm, defects), nb_envs=E, device="cuda:0", seed=5 + N)
    env.reset(episode=1)
    F = env.obs_vector_length()
    assert F == 11 + 4 * nb_comm + 2 * ("hour" in flags) + 2 * ("day" in flags) + ("solar_gain" in flags) + 5 * ("thermal" in flags) + 2 * ("hvac" in flags)
    torch.manual_seed(E)
    actor = ActorMLP(F, 2, (100, 100)).to("cuda:0")
    with torch.no_grad():
        for lin in actor.fc:
            lin.weight.mul_(2.0)
            lin.bias.uniform_(-0.5, 0.5)
    by_rows = FusedActor.from_module(actor, layout=layout)
    by_state = FusedActor.from_module(actor, layout=layout, feature_order=FEATURES_OBSERVE, observe_msg_floats=4 * nb_comm)
    for k in range(3):
        rows = env.obs_vector("rows").view(E * N, F)
        if defects > 0:
            msgs = rows[:, F - 4 * nb_comm:].view(E * N, nb_comm, 4)
            dead = (msgs == 0).all(dim=2).float().mean()
            assert abs(float(dead) - defects) < 0.05 + 2.0 / (E * N * nb_comm) ** 0.5      # the defects really are in the rows
        a0, p0, probs0 = by_rows.sample(rows, seed=9, step=k, want_probs=True)
        kept = torch.full((E * N, F), float("nan"), device="cuda:0")
        _poison_lds()
        a1, p1, probs1 = by_state.sample_env(env, seed=9, step=k, want_probs=True, rows_out=kept)
        assert torch.equal(kept, rows), "rows_out differs from obs_vector('rows')"
        _poison_lds()
        a2, p2, probs2 = by_state.sample_env(env, seed=9, step=k, want_probs=True)
        assert torch.equal(a2, a1) and torch.equal(probs2, probs1)
        with torch.no_grad():
            ref = actor(rows)
        tol = dict(rtol=2e-3, atol=2e-5) if layout == 2 else dict(rtol=1e-5, atol=2e-6)
        torch.testing.assert_close(probs1, ref, **tol)
        torch.testing.assert_close(probs1, probs0, **tol)
        differ = a0 != a1
        assert int(differ.sum()) <= max(2, E * N // 20000)
        _walk(env, 7, seed=k)


@pytest.mark.parametrize("E,N,mode,nb_comm,flags,defects", [
    (6, 64, "closed_groups", 10, (), 0.0),                      # groups of 11 that all hear each other (env 830-857)
    (40, 50, "closed_groups", 4, ("hour", "day"), 0.1),         # the deployment size, tiles across envs, link defects on top
    (3, 1024, "random_fixed", 10, ("thermal", "hvac"), 0.0),    # one table drawn per episode, senders anywhere in the env (env 866-878)
    (9, 96, "random_fixed", 13, (), 0.3),
    (25, 36, "neighbours_2D", 8, ("solar_gain",), 0.0),         # the 2-D neighbourhood table (env 859-897)
    (7, 33, "random_sample", 10, (), 0.0),                      # senders re-drawn per house and step (env 976-983)
    (2, 1000, "random_sample", 5, ("day",), 0.2),
    (100, 20, "random_sample", 3, (), 0.0),
])
@pytest.mark.parametrize("layout", [3, 2, 1])
def test_observe_act_link_tables(E, N, mode, nb_comm, flags, defects, layout):
    """Senders that are not the circular neighbours (ClusterHouses.build_agent_comm_links, env 806-902; random_sample 976-983):
    the actor kernel gathers their message records through the table (mdr_env_actor_sample_links).  Rows written on the side ==
    mdr_env_obs_vector bit for bit, probabilities == the actor on those rows."""
    import mdr_amd
    from mdr_amd.policy import FEATURES_OBSERVE, FusedActor
    from mdr_amd.rollout import ActorMLP
    cfg = _shape_cfg(N, flags, nb_comm, defects)
    cluster = cfg["default_env_prop"]["cluster_prop"]
    cluster["agents_comm_mode"] = mode
    if mode == "neighbours_2D":
        cluster["agents_comm_parameters"]["neighbours_2D"].update({"row_size": 6, "distance_comm": 2})
    env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=5 + N)
    env.reset(episode=1)
    F = env.obs_vector_length()
    c = (F - 11 - 2 * ("hour" in flags) - 2 * ("day" in flags) - ("solar_gain" in flags) - 5 * ("thermal" in flags) - 2 * ("hvac" in flags)) // 4
    torch.manual_seed(E)
    actor = ActorMLP(F, 2, (100, 100)).to("cuda:0")
    with torch.no_grad():
        for lin in actor.fc:
            lin.weight.mul_(2.0)
            lin.bias.uniform_(-0.5, 0.5)
    by_rows = FusedActor.from_module(actor, layout=layout)
    by_state = FusedActor.from_module(actor, layout=layout, feature_order=FEATURES_OBSERVE, observe_msg_floats=4 * c)
    for k in range(3):
        rows = env.obs_vector("rows").view(E * N, F)
        a0, p0, probs0 = by_rows.sample(rows, seed=9, step=k, want_probs=True)
        kept = torch.full((E * N, F), float("nan"), device="cuda:0")
        _poison_lds()
        a1, p1, probs1 = by_state.sample_env(env, seed=9, step=k, want_probs=True, rows_out=kept)
        assert torch.equal(kept, rows), "rows_out differs from obs_vector('rows')"
        _poison_lds()
        a2, p2, probs2 = by_state.sample_env(env, seed=9, step=k, want_probs=True)
        assert torch.equal(a2, a1) and torch.equal(probs2, probs1)
        with torch.no_grad():
            ref = actor(rows)
        tol = dict(rtol=2e-3, atol=2e-5) if layout == 2 else dict(rtol=1e-5, atol=2e-6)
        torch.testing.assert_close(probs1, ref, **tol)
        torch.testing.assert_close(probs1, probs0, **tol)
        assert int((a0 != a1).sum()) <= max(2, E * N // 20000)
        _walk(env, 7, seed=k)


def test_observe_act_refuses_what_it_does_not_cover():
    import mdr_amd
    from mdr_amd.policy import FEATURES_OBSERVE, FusedActor
    from mdr_amd.rollout import ActorMLP
    actor = _actor()
    by_state = FusedActor.from_module(actor, layout=1, feature_order=FEATURES_OBSERVE)
    # what stays with rows + actor: the optional MESSAGE columns (with 10 senders beyond the 64 features of a staged row), more than 64 features
    for patches in ({"default_env_prop.message_properties.thermal": True},
                    {"default_env_prop.message_properties.hvac": True}):
        env = mdr_amd.BatchedDemandResponseEnv(_cfg(64, **patches), nb_envs=4, device="cuda:0", seed=1)
        env.reset(episode=0)
        with pytest.raises(NotImplementedError):
            by_state.sample_env(env, 0, 0)
    env = mdr_amd.BatchedDemandResponseEnv(_cfg(10), nb_envs=4, device="cuda:0", seed=1)          # 9 neighbours: an actor packed for 10 does not fit
    env.reset(episode=0)
    with pytest.raises(NotImplementedError):
        by_state.sample_env(env, 0, 0)
    env = mdr_amd.BatchedDemandResponseEnv(_cfg(64), nb_envs=4, device="cuda:0", seed=1)
    env.reset(episode=0)
    with pytest.raises(NotImplementedError):          # an actor packed for observation rows
        FusedActor.from_module(actor, layout=1).sample_env(env, 0, 0)
    with pytest.raises(RuntimeError):                 # and the other way round
        by_state.sample(env.obs_vector("rows").view(-1, 51), 0, 0)
    with pytest.raises(ValueError):                   # 14 neighbours + 11 own features: beyond the 64 features of the staged row
        FusedActor.from_module(ActorMLP(67).cuda(), feature_order=FEATURES_OBSERVE, observe_msg_floats=56)
    with pytest.raises(ValueError):                   # 47 features cannot start with 40 message floats
        FusedActor.from_module(ActorMLP(47).cuda(), feature_order=FEATURES_OBSERVE)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_collect_rollout_observe_act_vs_rows(precision):
    """collect_ppo_rollout(store_states=False) takes the observe -> act path by itself; against the rows path the first step agrees
    to rounding, the trajectories then differ only through the handful of agents whose draw sat between the two probabilities."""
    import mdr_amd
    from mdr_amd.rollout import collect_ppo_rollout
    E, N, T = (16, 256, 12) if precision == "fp32" else (81, 50, 12)       # 50 houses: the general staging path
    cfg = _cfg(N)
    actor = _actor(seed=3)
    outs = []
    for observe in (True, False, None):
        env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=4)
        env.reset(episode=0)
        outs.append(collect_ppo_rollout(env, actor, T, store_states=False, seed=11, policy_precision=precision, observe_act=observe))
        assert env.steps_taken == T
    a, b, c = outs
    for k in ("action", "a_prob", "reward", "return"):
        assert torch.equal(a[k], c[k]), k                      # None == True here: the default picks the fused path
    tol = 3e-5 if precision == "bf16x3" else 3e-6
    assert float((a["a_prob"][0] - b["a_prob"][0]).abs().max()) < tol or int((a["action"][0] != b["action"][0]).sum()) <= 2
    assert float((a["action"] != b["action"]).float().mean()) < 2e-3
    torch.testing.assert_close(a["reward"].mean(), b["reward"].mean(), rtol=1e-3, atol=1e-4)
    # with the states kept: the transition buffer's `state` is what the rows kernel would have written at every step of ITS trajectory
    env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=4)
    env.reset(episode=0)
    kept = collect_ppo_rollout(env, actor, T, store_states=True, seed=11, policy_precision=precision)      # observe -> act + rows on the side
    for k in ("action", "a_prob", "reward", "return"):
        assert torch.equal(kept[k], a[k]), k
    replay = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=4)
    replay.reset(episode=0)
    for s in range(T):
        assert torch.equal(kept["state"][s], replay.obs_vector("rows").view(E * N, 51)), s
        replay.step(kept["action"][s].to(torch.uint8).view(E, N))
    assert torch.equal(kept["state"][T], replay.obs_vector("rows").view(E * N, 51))


def test_observe_act_in_a_replayed_graph():
    """Graph mode: the fused kernel takes the table row of the regulation signal and its Philox step from the device cursor."""
    import mdr_amd
    from mdr_amd.policy import FEATURES_OBSERVE, FusedActor
    E, N, T = 8, 64, 40
    cfg = _cfg(N)
    policy = FusedActor.from_module(_actor(seed=1), layout=2, feature_order=FEATURES_OBSERVE)
    envs = [mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=2, table_steps=16, graph_mode=True) for _ in range(2)]
    plain = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=2, table_steps=16)
    acts = [torch.empty(E * N, dtype=torch.uint8, device="cuda:0") for _ in range(3)]
    for e in envs + [plain]:
        e.reset(episode=0)

    def one(env, act):
        policy.sample_env(env, 77, 0, action=act, step_dev=env.device_time_index)
        env.step(act.view(E, N))

    for t in range(T):
        one(envs[0], acts[0])
        policy.sample_env(plain, 77, t, action=acts[2])
        plain.step(acts[2].view(E, N))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        one(envs[1], acts[1])
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        one(envs[1], acts[1])
    done = 1
    while done < T:
        n = min(envs[1].graph_room(), T - done)
        for _ in range(n):
            g.replay()
        envs[1].graph_replayed(n)
        done += n
    for k in ("Ta", "sso", "flags", "reward", "P"):
        assert torch.equal(envs[0].t[k], plain.t[k]) and torch.equal(envs[1].t[k], plain.t[k]), k
    assert torch.equal(acts[0], acts[2]) and torch.equal(acts[1], acts[2])


def test_general_staging_equals_the_lean_one_where_both_apply():
    """MDR_OBSERVE_GEN=1 sends nb_houses % 32 == 0 shapes through the any-cluster-size staging too: same rows, same probabilities
    (a child process: the library reads the knob once)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = r"""
import torch, mdr_amd
from mdr_amd.policy import FEATURES_OBSERVE, FusedActor
from mdr_amd.rollout import ActorMLP
from tests.test_gpu_observe_act import _cfg, _actor, _walk
for E, N in ((40, 64), (3, 1024), (9, 96)):
    env = mdr_amd.BatchedDemandResponseEnv(_cfg(N), nb_envs=E, device="cuda:0", seed=3)
    env.reset(episode=0)
    _walk(env, 5, seed=1)
    rows = env.obs_vector("rows").view(E * N, 51)
    actor = _actor(seed=N)
    for layout in (1, 2, 3):
        a0, p0, probs0 = FusedActor.from_module(actor, layout=layout).sample(rows, 4, 1, want_probs=True)
        kept = torch.empty_like(rows)
        a1, p1, probs1 = FusedActor.from_module(actor, layout=layout, feature_order=FEATURES_OBSERVE).sample_env(env, 4, 1, want_probs=True, rows_out=kept)
        assert torch.equal(kept, rows)
        torch.testing.assert_close(probs1, probs0, rtol=2e-3 if layout == 2 else 1e-5, atol=2e-5 if layout == 2 else 2e-6)
print("gen ok")
"""
    env = dict(os.environ, MDR_OBSERVE_GEN="1", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    res = subprocess.run([sys.executable, "-c", child], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "gen ok" in res.stdout, res.stdout[-1500:] + res.stderr[-3000:]


@pytest.mark.parametrize("idx", range(40))
def test_fuzz_observe_act_vs_rows(idx):
    """Random cluster sizes (11 ... 300, every tile / env alignment), batch sizes, lockout noise and walked states: the rows the fused
    kernel copies out are bit for bit those of mdr_env_obs_vector, its probabilities those of the rows path (both layouts)."""
    import mdr_amd
    from mdr_amd.policy import FEATURES_OBSERVE, FusedActor
    rng = np.random.default_rng(5100 + idx)
    N = int(rng.choice([11, 12, 13, 15, 16, 17, 20, 21, 22, 31, 32, 33, 37, 42, 47, 48, 50, 63, 64, 65, 100, 127, 200, 300]))
    E = int(rng.integers(1, max(2, 6000 // N)))
    cfg = _cfg(N, **{"default_hvac_prop.lockout_noise": int(rng.choice([0, 15, 39])),
                     "default_env_prop.time_step": int(rng.choice([4, 7, 60]))})
    env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=int(rng.integers(0, 2 ** 31)))
    env.reset(episode=int(rng.integers(0, 4)))
    _walk(env, int(rng.integers(0, 9)), seed=idx)
    rows = env.obs_vector("rows").view(E * N, 51)
    actor = _actor(seed=idx)
    for layout in (1, 2, 3):
        kept = torch.full((E * N, 51), float("nan"), device="cuda:0")
        a0, _, probs0 = FusedActor.from_module(actor, layout=layout).sample(rows, 3, idx, want_probs=True)
        a1, _, probs1 = FusedActor.from_module(actor, layout=layout, feature_order=FEATURES_OBSERVE).sample_env(env, 3, idx, want_probs=True, rows_out=kept)
        assert torch.equal(kept, rows), "case %d (E=%d N=%d layout %d): rows differ" % (idx, E, N, layout)
        torch.testing.assert_close(probs1, probs0, rtol=2e-3 if layout == 2 else 1e-5, atol=2e-5 if layout == 2 else 2e-6)
        assert int((a0 != a1).sum()) <= max(2, E * N // 20000)


@pytest.mark.parametrize("N,nb_comm", [(20, 6), (64, 6), (20, 2), (50, 0), (96, 4)])
def test_bf16_extended_rows_never_read_past_their_window(N, nb_comm):
    """The bf16x3 forward reads 64 floats per row whatever the row stride; with strides below 48 floats (F <= 42) the last row of
    the last wave's window used to reach past the workgroup's LDS, where a NaN left by an earlier kernel times a zero weight made a
    NaN logit.  Poison: an actor with NaN weights whose fragments fill 150 KB of every CU's LDS; then every probability of the
    small-row shapes must still be finite and equal with and without rows_out."""
    import mdr_amd
    from mdr_amd.policy import FEATURES_OBSERVE, FusedActor
    from mdr_amd.rollout import ActorMLP
    env = mdr_amd.BatchedDemandResponseEnv(_shape_cfg(N, ("solar_gain",), nb_comm, 0.0), nb_envs=100, device="cuda:0", seed=N)
    env.reset(episode=0)
    F = env.obs_vector_length()
    torch.manual_seed(N)
    actor = ActorMLP(F, 2, (100, 100)).to("cuda:0")
    by_state = FusedActor.from_module(actor, layout=2, feature_order=FEATURES_OBSERVE, observe_msg_floats=4 * nb_comm)
    for k in range(3):
        _poison_lds()                                                # every CU's LDS now holds NaN up to ~150 KB
        a1, _, probs1 = by_state.sample_env(env, seed=3, step=k, want_probs=True)
        _poison_lds()
        kept = torch.empty((100 * N, F), device="cuda:0")
        a2, _, probs2 = by_state.sample_env(env, seed=3, step=k, want_probs=True, rows_out=kept)
        assert bool(torch.isfinite(probs1).all()) and bool(torch.isfinite(probs2).all())
        assert torch.equal(a1, a2) and torch.equal(probs1, probs2)
        _walk(env, 5, seed=k)
