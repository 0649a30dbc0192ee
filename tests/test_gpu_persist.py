"""The persistent sharded rollout (mdr_env_rollout_persistent, csrc/mdr_persist.hip): houses resident in registers across
steps, the per-step exchange of (cluster power, penalty sum / max) through mailbox granules instead of a kernel boundary
and a collective.  Held bit for bit to the records path (step_begin_records / gather / step_end_records) it replaces:
env/MA_DemandResponse.py:1042-1050 (cluster power), 274-321 (common penalties), main-deploy.py:99-148 (the loop)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

STATE = ("Ta", "Tm", "sso", "flags", "actions", "reward", "obs", "P")


def _cfg(n, mode="individual_L2", signal="sinusoidals"):
    import mdr_amd
    cfg = mdr_amd.default_config()
    env = cfg["default_env_prop"]
    env["cluster_prop"]["nb_agents"] = n
    env["power_grid_prop"]["base_power_mode"] = "constant"
    env["power_grid_prop"]["signal_mode"] = signal
    env["reward_prop"]["temp_penalty_mode"] = mode
    cfg["noise_house_prop"]["noise_mode"] = "big_noise"
    cfg["noise_hvac_prop"]["noise_mode"] = "big_noise"
    return cfg


def _stepwise(env, T):
    """T single bang-bang steps with the accumulators of main-deploy.py:124-152 kept on the side, in step order."""
    E, N = env.nb_envs, env.nb_houses
    rsum = torch.zeros((E, N), dtype=torch.float32, device=env.device)
    terr = torch.zeros(E, dtype=torch.float64, device=env.device)
    serr = torch.zeros(E, dtype=torch.float64, device=env.device)
    trace = []
    for _ in range(T):
        env.step_bangbang()
        rsum = rsum + env.t["reward"]
        d = (env.t["Ta"] - env.t["target"])
        terr += (d * d).double().sum(dim=1)
        s = env.reg_signal() - env.t["P"]
        serr += s * s
        trace.append(env.t["P"].clone())
    return {"reward_sum": rsum, "sq_temp_error_sum": terr, "sq_signal_error_sum": serr, "power_trace": torch.stack(trace)}


@pytest.mark.parametrize("E,N,mode,T", [
    (1, 20000, "individual_L2", 40),      # 20 house workgroups + the reducer; three table windows of 16 steps
    (3, 5000, "mixture", 37),             # several envs, every record field travels
    (2, 4099, "common_L2", 21),           # nb_houses % 4 != 0: one house per lane, 256-house records
    (1, 1500, "common_max", 19),          # two workgroups: also serves envs the fused rollout kernel covers
    (4, 300, "individual_L2", 18),        # one house workgroup per env
    (1, 300000, "mixture", 35),           # 293 records: two reducer workgroups take the steps in turn, partial error sums combined
    (1, 800000, "individual_L2", 33),     # 782 records: four reducers
])
def test_persistent_rollout_equals_single_steps(E, N, mode, T):
    import mdr_amd
    cfg = _cfg(N, mode)
    ref = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=31, table_steps=16,
                                           house_shard=(0, N), exchange_always=True)
    per = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=31, table_steps=16)
    from mdr_amd.sharding import LocalShardGroup
    ref._exchange_impl = _OneShard(ref)
    ref.reset(episode=2)
    per.reset(episode=2)
    want = _stepwise(ref, T)
    got = per.rollout_persistent(T, power_trace=True)
    assert per.steps_taken == T
    for name in STATE:
        assert torch.equal(per.t[name], ref.t[name]), name
    assert torch.equal(got["reward_sum"], want["reward_sum"])
    assert torch.equal(got["power_trace"], want["power_trace"])
    torch.testing.assert_close(got["sq_signal_error_sum"], want["sq_signal_error_sum"], rtol=1e-12, atol=0)
    torch.testing.assert_close(got["sq_temp_error_sum"], want["sq_temp_error_sum"], rtol=1e-6, atol=0)
    # a second call continues where the first ended (tags keep counting, no re-initialisation of the mailbox)
    want2 = _stepwise(ref, 5)
    got2 = per.rollout_persistent(5, power_trace=True)
    for name in STATE:
        assert torch.equal(per.t[name], ref.t[name]), name
    assert torch.equal(got2["reward_sum"], want2["reward_sum"])
    assert per.persist_status() == 0


class _OneShard:
    """Exchange of a world of one without torch.distributed: the records path on the shard's own `partials`."""

    def __init__(self, env):
        self.env = env

    def agree_partial_records(self, env):
        pass

    def sum_max_power(self, env):
        pass

    def sum_base_power(self, env):
        pass

    def gather_partials(self, env):
        return env.t["partials"].unsqueeze(0), 1


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _ipc_worker(rank, world, port, N, mode, E, T):
    """One rank of `world` processes sharing the one GPU: its shard through the records path (gloo all-gather per step) and
    through the persistent rollout with peer mailboxes mapped over hipIpc - the production layout of one process per GPU,
    rehearsed on one device (the kernels of the ranks are resident together and push into each other's boxes)."""
    import os
    import torch.distributed as dist
    import mdr_amd
    from mdr_amd.sharding import house_shard
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      MDR_MAILBOX_CO_RESIDENT=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    cfg = _cfg(N, mode)
    shard = house_shard(N, world, rank)
    rec = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=17, table_steps=16, house_shard=shard)
    per = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=17, table_steps=16, house_shard=shard)
    rec.reset(episode=1)
    per.reset(episode=1)
    want = _stepwise(rec, T)
    got = per.rollout_persistent(T, power_trace=True)
    assert per.persist_status() == 0
    for name in STATE:
        assert torch.equal(per.t[name], rec.t[name]), (rank, name)
    assert torch.equal(per.reg_signal(), rec.reg_signal())
    assert torch.equal(got["reward_sum"], want["reward_sum"])
    assert torch.equal(got["power_trace"], want["power_trace"])
    # every rank's reducer saw every rank's records: the env-wide accumulators are replicated bit for bit
    for key in ("sq_temp_error_sum", "sq_signal_error_sum"):
        mine = got[key].cpu()
        both = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)
        assert all(torch.equal(both[0], b) for b in both), key
    torch.testing.assert_close(got["sq_signal_error_sum"], want["sq_signal_error_sum"], rtol=1e-12, atol=0)
    # a second call: tags keep counting across launches and ranks
    want2 = _stepwise(rec, 7)
    got2 = per.rollout_persistent(7)
    for name in STATE:
        assert torch.equal(per.t[name], rec.t[name]), (rank, name)
    assert torch.equal(got2["reward_sum"], want2["reward_sum"])
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("N,world,mode,E", [(20000, 2, "individual_L2", 1), (30000, 3, "mixture", 2), (9001, 4, "common_L2", 1),
                                            (500000, 4, "individual_L2", 1)])
def test_ranks_on_one_gpu_exchange_through_ipc_mailboxes(N, world, mode, E):
    """>= 2 ranks (one process each, as on an 8-GPU node) resident together on the one GPU, their mailboxes mapped into each other
    over hipIpc: state, P, signal, rewards and reward sums bit-identical to the records path over three table refills."""
    import torch.multiprocessing as mp
    mp.spawn(_ipc_worker, args=(world, _free_port(), N, mode, E, 40), nprocs=world, join=True)


def test_a_missing_peer_ends_in_an_error_word_not_a_hang():
    """World of two, the peer never launches: the reducer's wait is bounded, the error word says who gave up, and the bound
    buffers still hold the state before the launch (nothing is written back)."""
    import mdr_amd
    from mdr_amd import _native as nat
    N = 12000
    env = mdr_amd.BatchedDemandResponseEnv(_cfg(2 * N), nb_envs=1, device="cuda:0", seed=3, house_shard=(0, N))
    env._exchange_impl = _OneShard(env)
    env.reset(episode=0)
    before = {k: env.t[k].clone() for k in ("Ta", "Tm", "sso", "flags", "reward")}
    n = env.persist_records()
    box = env._persist_mailbox(2, n)
    ghost = torch.zeros_like(box)
    mb = nat.MdrMailbox()
    mb.struct_size = C.sizeof(nat.MdrMailbox)
    mb.world, mb.rank, mb.records_per_env, mb.co_resident, mb.spin_limit = 2, 0, n, 1, 2000
    mb.records[0] = mb.records[1] = n
    mb.boxes[0], mb.boxes[1] = box.data_ptr(), ghost.data_ptr()
    env._persist_call(8, mb, False, True)
    word = env.persist_status()
    assert word != 0
    assert (word >> 28) & 0xF in (1, 2)
    assert int(ghost[0].item()) != 0          # the peer is told as well
    for k, v in before.items():
        assert torch.equal(env.t[k], v), k
    with pytest.raises(RuntimeError, match="gave up"):
        env._persist_raise(word)


def test_a_grid_that_cannot_be_resident_is_refused():
    import mdr_amd
    env = mdr_amd.BatchedDemandResponseEnv(_cfg(5000), nb_envs=600, device="cuda:0", seed=3)      # 600 x (5 + 1) workgroups
    env.reset(episode=0)
    with pytest.raises(RuntimeError, match="exceed"):
        env.rollout_persistent(4)
    env.rollout(2)                                                                                 # the handle is untouched


def test_persistent_rollout_at_c5_size_matches_split_rollout():
    """1 env x 1,000,000 houses on one rank (977 house workgroups + the reducer, all resident): same bits as the split path."""
    import mdr_amd
    cfg = _cfg(1_000_000)
    ref = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=1, device="cuda:0", seed=5)
    per = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=1, device="cuda:0", seed=5)
    ref.reset(episode=0)
    per.reset(episode=0)
    ref.rollout(150)
    per.rollout_persistent(150, accumulate=False)
    for name in STATE:
        assert torch.equal(per.t[name], ref.t[name]), name
