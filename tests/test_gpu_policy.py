"""Fused policy forward + sampling (include/mdr_policy.h) against a plain PyTorch fp32 forward of the same Actor.

The reference's PPO.select_action (agents/ppo.py:68-75): probs = actor_net(state); action ~ Categorical(probs);
returns (action, probs[action]).  Tolerance: fp32 MFMA is a k-ordered fp32 fma chain, torch's GEMM sums in another
order -> probabilities to 2e-6 absolute / 1e-5 relative."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _actor(F, layers, seed=0, scale=1.0):
    from mdr_amd.rollout import ActorMLP
    torch.manual_seed(seed)
    actor = ActorMLP(F, 2, layers).to("cuda:0")
    with torch.no_grad():
        for lin in actor.fc:
            lin.weight.mul_(scale)
            lin.bias.uniform_(-0.5, 0.5)
    return actor


@pytest.mark.parametrize("A,F,layers", [(1, 51, (100, 100)), (31, 51, (100, 100)), (33, 51, (100, 100)), (1000, 51, (100, 100)),
                                        (4097, 47, (100, 100)), (777, 133, (100, 100)), (500, 11, (64, 32)), (300, 51, (127, 127)),
                                        (300, 50, (1, 1)), (300000, 51, (100, 100)), (1000, 51, (97, 100)), (999, 33, (100, 98)),
                                        (50, 64, (99, 97)), (1000, 65, (100, 100)), (777, 81, (100, 100)), (4097, 91, (100, 100)),
                                        (3000, 121, (100, 100)), (513, 128, (127, 113)), (200, 97, (64, 100))])
@pytest.mark.parametrize("layout", [0, 1, 2, 3])
def test_fused_actor_matches_torch_forward(A, F, layers, layout):
    from mdr_amd.policy import FusedActor
    if layout == 3 and not all(97 <= h <= 100 for h in layers):      # the 4x4-tail form: hidden layers of 6 x 16 + (1..4) units
        with pytest.raises(ValueError):
            FusedActor.from_module(_actor(min(F, 64), layers), layout=3)
        return
    if layout >= 1 and F > 128:      # the 16-agent forms hold at most 32 feature registers per lane
        with pytest.raises(RuntimeError):
            FusedActor.from_module(_actor(F, layers), layout=1).sample(torch.zeros((4, F), device="cuda:0"), 0, 0)
        return
    actor = _actor(F, layers, seed=A, scale=3.0)
    fused = FusedActor.from_module(actor, layout=layout)
    g = torch.Generator(device="cuda:0").manual_seed(1)
    obs = torch.randn((A, F), device="cuda:0", generator=g) * 2.0
    with torch.no_grad():
        ref = actor(obs)
    action, a_prob, probs = fused.sample(obs, seed=7, step=3, want_probs=True)
    if layout == 2:      # bf16 head + tail operands: 16 significand bits, logits of magnitude ~10 -> probabilities to ~1e-4
        torch.testing.assert_close(probs, ref, rtol=2e-3, atol=2e-5)
        assert float((probs - ref).abs().mean()) < 5e-6
    else:
        torch.testing.assert_close(probs, ref, rtol=1e-5, atol=2e-6)
    assert torch.equal(a_prob, probs.gather(1, action.long()[:, None]).squeeze(1))       # the probability of the action taken
    assert set(action.unique().tolist()) <= {0, 1}
    assert float((probs.sum(1) - 1).abs().max()) < 1e-6


@pytest.mark.parametrize("layout,H1,H2", [(0, 100, 100), (1, 100, 100), (2, 100, 100), (3, 100, 100), (3, 98, 99), (3, 97, 100), (1, 98, 99),
                                          (1, 101, 37)])
def test_weight_layout_is_checked_with_asymmetric_integer_data(layout, H1, H2):
    """Exact small-integer weights and inputs (every product and sum exact in fp32): the logits' difference must be exact,
    which a swapped A/B operand, a wrong k order between the layers or a transposed block would not survive."""
    from mdr_amd.policy import FusedActor
    F, A = 51, 257
    rng = np.random.default_rng(0)
    w1 = rng.integers(-2, 3, (H1, F)).astype(np.float32)
    b1 = rng.integers(-3, 4, H1).astype(np.float32)
    w2 = rng.integers(-1, 2, (H2, H1)).astype(np.float32) * (rng.random((H2, H1)) < 0.2)
    b2 = rng.integers(-3, 4, H2).astype(np.float32)
    w3 = rng.integers(-1, 2, (2, H2)).astype(np.float32) * (rng.random((2, H2)) < 0.3)
    b3 = np.array([1.0, -2.0], dtype=np.float32)
    if H1 > 96 and H2 > 96:      # the units of the partial last block count, every one differently
        w2[:, 96:H1] = rng.integers(1, 3, (H2, H1 - 96)) * np.array([1, -1, 1, -1])[:H1 - 96]
        w3[0, 96:H2] = np.arange(1, H2 - 95)
        w3[1, 96:H2] = -1
    x = rng.integers(-2, 3, (A, F)).astype(np.float32)
    h1 = np.maximum(x.astype(np.float64) @ w1.T + b1, 0)
    h2 = np.maximum(h1 @ w2.T + b2, 0)
    logits = h2 @ w3.T + b3
    assert np.abs(h2).max() < 2 ** 22                    # everything stays in fp32's exact-integer range
    d = logits[:, 0] - logits[:, 1]
    p0 = 1.0 / (1.0 + np.exp(-d))
    fused = FusedActor(w1, b1, w2.astype(np.float32), b2, w3.astype(np.float32), b3, layout=layout)
    _, _, probs = fused.sample(torch.from_numpy(x).cuda(), seed=1, step=0, want_probs=True)
    np.testing.assert_allclose(probs[:, 0].cpu().numpy(), p0, rtol=2e-6, atol=1e-30)
    np.testing.assert_allclose(probs[:, 1].cpu().numpy(), 1.0 / (1.0 + np.exp(d)), rtol=2e-6, atol=1e-30)


def test_sampling_follows_the_probabilities_and_the_step_counter():
    from mdr_amd.policy import FusedActor
    A, F = 1 << 20, 51
    actor = _actor(F, (100, 100), seed=5, scale=0.0)          # zero weights: the same probabilities for every agent
    with torch.no_grad():
        actor.fc[2].bias.copy_(torch.tensor([0.3, -0.55]))
        for lin in actor.fc[:2]:
            lin.bias.zero_()
    fused = FusedActor.from_module(actor)
    obs = torch.zeros((A, F), device="cuda:0")
    a0, p0, probs = fused.sample(obs, seed=11, step=0, want_probs=True)
    p = float(probs[0, 0])
    assert abs(p - 1 / (1 + np.exp(-0.85))) < 1e-6
    frac = float((a0 == 0).float().mean())
    assert abs(frac - p) < 5 * np.sqrt(p * (1 - p) / A)         # binomial 5 sigma
    a1, _ = fused.sample(obs, seed=11, step=1)
    a0b, _ = fused.sample(obs, seed=11, step=0)
    a2, _ = fused.sample(obs, seed=12, step=0)
    assert torch.equal(a0, a0b)                                  # counter-based: same (seed, step) -> same draws
    assert 0.3 < float((a0 != a1).float().mean()) < 0.6         # independent draws differ with prob 2 p (1 - p) = 0.42
    assert 0.3 < float((a0 != a2).float().mean()) < 0.6
    # neighbouring agents are independent
    x = (a0[:-1] == 0).float() - p
    y = (a0[1:] == 0).float() - p
    assert abs(float((x * y).mean())) < 5 * p * (1 - p) / np.sqrt(A)


def test_fused_actor_argument_checks():
    from mdr_amd.policy import FusedActor
    from mdr_amd.rollout import ActorMLP
    with pytest.raises(ValueError):
        FusedActor.from_module(ActorMLP(51, 2, (100,)).to("cuda:0"))
    with pytest.raises(ValueError):
        FusedActor.from_module(ActorMLP(51, 2, (128, 100)).to("cuda:0"))
    fused = FusedActor.from_module(ActorMLP(51, 2, (100, 100)).to("cuda:0"))
    with pytest.raises(ValueError):
        fused.sample(torch.zeros((4, 50), device="cuda:0"), 0, 0)
    with pytest.raises(ValueError):
        fused.sample(torch.zeros((4, 51), device="cuda:0", dtype=torch.float64), 0, 0)


@pytest.mark.parametrize("layout", [0, 1, 2, 3])
@pytest.mark.parametrize("A,F", [(1000, 51), (4097, 47), (33, 11), (2050, 91)])
def test_feature_plane_input_gives_the_same_bits_as_rows(A, F, layout):
    """obs as feature planes [F][stride] (what mdr_env_obs_vector MDR_OBS_PLANES writes) instead of rows [A][F]."""
    from mdr_amd.policy import FusedActor
    actor = _actor(F, (100, 100), seed=2, scale=3.0)
    fused = FusedActor.from_module(actor, layout=layout)
    rows = torch.randn((A, F), device="cuda:0")
    a0, p0, pr0 = fused.sample(rows, 3, 4, want_probs=True)
    planes = rows.t().contiguous()                                   # [F, A]
    a1, p1, pr1 = fused.sample(planes, 3, 4, want_probs=True)
    assert torch.equal(a0, a1) and torch.equal(pr0, pr1) and torch.equal(p0, p1)
    padded = torch.zeros((F, A + 576), device="cuda:0")
    padded[:, :A] = planes
    a2, _, pr2 = fused.sample(padded[:, :A].view(F, A) if False else padded.as_strided((F, A), (A + 576, 1)), 3, 4, want_probs=True)
    assert torch.equal(a0, a2) and torch.equal(pr0, pr2)
    a3, _, pr3 = fused.sample(planes.t(), 3, 4, want_probs=True)     # the transposed [A, F] view of the planes
    assert torch.equal(a0, a3) and torch.equal(pr0, pr3)
    env_like = planes.view(F, 1, A)                                  # [F, E, N] as obs_vector("planes") returns it
    a4, _ = fused.sample(env_like, 3, 4)
    assert torch.equal(a0, a4)


@pytest.mark.parametrize("layout", [0, 1, 2, 3])
def test_greedy_mode_is_the_argmax_of_a_dqn_network(layout):
    """DQNAgent.act (agents/rl_controllers.py:53-60): the same Linear/ReLU stack, action = argmax of its two outputs."""
    from mdr_amd.policy import FusedActor
    net = _actor(51, (100, 100), seed=8, scale=2.0)              # ActorMLP's fc stack == DQN_network's
    fused = FusedActor.from_module(net, layout=layout, greedy=True)
    obs = torch.randn((20000, 51), device="cuda:0")
    with torch.no_grad():
        x = obs
        for lin in net.fc[:-1]:
            x = torch.relu(lin(x))
        q = net.fc[-1](x)
    a, _ = fused.sample(obs, 1, 1)
    a2, _ = fused.sample(obs, 99, 7)
    assert torch.equal(a, a2)                                     # no randomness involved
    clear = (q[:, 0] - q[:, 1]).abs() > (1e-3 if layout == 2 else 1e-5)
    assert clear.float().mean() > 0.99
    assert torch.equal(a[clear].long(), q.argmax(1)[clear])


@pytest.mark.parametrize("idx", range(64))
def test_fuzz_random_network_shapes(idx):
    """Random (F, H1, H2, A) at every padding edge of the four layouts against the torch forward."""
    from mdr_amd.policy import FusedActor
    rng = np.random.default_rng(100 + idx)
    layout = idx % 4
    F = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 47, 50, 51, 52, 62, 63, 64] + ([65, 100, 133] if layout == 0 else [])
                       + ([65, 66, 79, 80, 81, 91, 95, 96, 97, 100, 121, 127, 128] if layout >= 1 else [])))
    H1 = int(rng.choice([1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 95, 96, 97, 100, 111, 112, 113, 126, 127]))
    H2 = int(rng.choice([1, 3, 16, 17, 32, 48, 64, 99, 100, 111, 112, 113, 127]))
    if layout == 3:
        H1, H2 = int(rng.integers(97, 101)), int(rng.integers(97, 101))
    A = int(rng.choice([1, 15, 16, 17, 31, 32, 33, 63, 64, 65, 1000, 4099]))
    actor = _actor(F, (H1, H2), seed=idx, scale=2.0)
    fused = FusedActor.from_module(actor, layout=layout)
    obs = torch.randn((A, F), device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(idx))
    with torch.no_grad():
        ref = actor(obs)
    action, a_prob, probs = fused.sample(obs, seed=idx, step=idx, want_probs=True)
    if layout == 2:
        torch.testing.assert_close(probs, ref, rtol=2e-3, atol=2e-5)
    else:
        torch.testing.assert_close(probs, ref, rtol=1e-5, atol=2e-6)
    assert torch.equal(a_prob, probs.gather(1, action.long()[:, None]).squeeze(1))
