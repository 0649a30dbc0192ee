"""Multi-rank paths.

CPU (gloo, world_size 2, runs in the build container): the partition helpers and the oracle's sharded form -
houses split over ranks exchanging only the per-env aggregates - equal the unsharded oracle; env replicas with
env_offset reproduce the corresponding rows of the whole batch (no collective).

GPU (gloo over CUDA tensors, 2 ranks sharing the one GPU of the box - RCCL itself refuses two ranks on one
device; the production code path is identical apart from the backend string): the HIP step_begin / all-reduce /
step_end path equals the single-device run.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import golden_util as gu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _cfg(n, **patches):
    cfg = gu.reference_env_config()
    cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = n
    cfg["default_env_prop"]["power_grid_prop"]["base_power_mode"] = "constant"
    for dotted, v in patches.items():
        node = cfg
        parts = dotted.split(".")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = v
    return cfg


def _spawn(fn, world, *args):
    port = _free_port()
    mp.spawn(fn, args=(world, port) + args, nprocs=world, join=True)


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)


# ------------------------------------------------------------------------------------------- CPU / gloo
def _oracle_sharded_worker(rank, world, port, mode):
    from oracle import mdr_oracle as mo
    from mdr_amd.sharding import house_shard
    _init(rank, world, port)
    N, E, T = 203, 3, 30
    cfg = _cfg(N, **{"noise_house_prop.noise_mode": "big_noise", "noise_hvac_prop.noise_mode": "big_noise",
                     "default_house_prop.deadband": 1, "default_env_prop.reward_prop.temp_penalty_mode": mode,
                     "default_env_prop.power_grid_prop.signal_mode": "sinusoidals"})
    off, cnt = house_shard(N, world, rank)

    def group_reduce(sum_p, sum_pen, max_pen):
        out = []
        for v, op in ((sum_p, dist.ReduceOp.SUM), (sum_pen, dist.ReduceOp.SUM), (max_pen, dist.ReduceOp.MAX)):
            if v is None:
                out.append(None)
                continue
            t = torch.from_numpy(np.ascontiguousarray(v))
            dist.all_reduce(t, op=op)
            out.append(t.numpy())
        return tuple(out)

    shard = mo.OracleEnv(cfg, nb_envs=E, house_offset=off, nb_houses_local=cnt)
    shard.group_reduce = group_reduce
    shard.reset(seed=3, episode=1)
    whole = mo.OracleEnv(cfg, nb_envs=E).reset(seed=3, episode=1)
    rng = np.random.default_rng(0)
    for t in range(T):
        act = rng.random((E, N)) < 0.5
        r_s = shard.step(act[:, off:off + cnt])
        r_w = whole.step(act)
        np.testing.assert_array_equal(shard.P, whole.P)
        np.testing.assert_array_equal(shard.Ta, whole.Ta[:, off:off + cnt])
        np.testing.assert_allclose(r_s, r_w[:, off:off + cnt], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(shard.S, whole.S, rtol=1e-14)
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["individual_L2", "mixture"])
def test_oracle_sharded_houses_gloo_world2(mode):
    _spawn(_oracle_sharded_worker, 2, mode)


def _oracle_replica_worker(rank, world, port):
    from oracle import mdr_oracle as mo
    from mdr_amd.sharding import env_shard
    _init(rank, world, port)
    cfg = _cfg(16, **{"noise_house_prop.noise_mode": "big_noise"})
    E = 7
    off, cnt = env_shard(E, world, rank)
    mine = mo.OracleEnv(cfg, nb_envs=cnt, env_offset=off).reset(seed=11, episode=0)
    whole = mo.OracleEnv(cfg, nb_envs=E).reset(seed=11, episode=0)
    for t in range(10):
        mine.step(mine.bangbang_actions())
        whole.step(whole.bangbang_actions())
    np.testing.assert_array_equal(mine.Ta, whole.Ta[off:off + cnt])
    np.testing.assert_array_equal(mine.S, whole.S[off:off + cnt])
    # throughput aggregation used by bench.py: max over ranks of the elapsed time
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == float(world)
    dist.destroy_process_group()


def test_env_replicas_gloo_world2():
    _spawn(_oracle_replica_worker, 2)


def test_partition_helpers():
    from mdr_amd.sharding import env_shard, house_shard
    for n, w in ((1_000_000, 8), (1001, 3), (1024, 2), (12, 3)):
        parts = [house_shard(n, w, r) for r in range(w)]
        assert parts[0][0] == 0 and sum(c for _, c in parts) == n
        assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(w - 1))
        assert all(o % 4 == 0 for o, _ in parts)
    assert [env_shard(10, 4, r) for r in range(4)] == [(0, 3), (3, 3), (6, 2), (8, 2)]
    with pytest.raises(ValueError):
        house_shard(6, 4, 0)
    with pytest.raises(ValueError):
        env_shard(2, 4, 0)


# ------------------------------------------------------------------------------------------- GPU
def _hip_sharded_worker(rank, world, port, mode, N):
    import mdr_amd
    from mdr_amd.sharding import house_shard
    _init(rank, world, port)
    torch.cuda.set_device(0)
    E, T = 3, 25
    cfg = _cfg(N, **{"noise_house_prop.noise_mode": "big_noise", "noise_hvac_prop.noise_mode": "big_noise",
                     "default_house_prop.deadband": 1, "default_env_prop.reward_prop.temp_penalty_mode": mode,
                     "default_env_prop.power_grid_prop.signal_mode": "perlin",
                     "default_env_prop.power_grid_prop.artificial_signal_ratio_range": 2})
    off, cnt = house_shard(N, world, rank)
    shard = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=21, house_shard=(off, cnt))
    whole = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=21)
    shard.reset(episode=0)
    whole.reset(episode=0)
    assert torch.equal(shard.t["max_power"], whole.t["max_power"])
    assert torch.equal(shard.t["tab_signal"], whole.t["tab_signal"])
    gen = torch.Generator(device="cpu").manual_seed(5)
    for t in range(T):
        act = (torch.rand((E, N), generator=gen) < 0.5).to(torch.uint8).cuda()
        if t % 2 == 0:
            shard.step(act[:, off:off + cnt].contiguous())
            whole.step(act)
        else:
            shard.step_bangbang()
            whole.step_bangbang()
        assert torch.equal(shard.t["P"], whole.t["P"])
        for k in ("Ta", "Tm", "sso", "flags"):
            assert torch.equal(shard.t[k], whole.t[k][:, off:off + cnt]), k
        assert torch.equal(shard.t["obs"], whole.t["obs"][:, :, off:off + cnt])
        if mode == "individual_L2":
            assert torch.equal(shard.t["reward"], whole.t["reward"][:, off:off + cnt])
        else:  # the penalty sum is added in a different order
            torch.testing.assert_close(shard.t["reward"], whole.t["reward"][:, off:off + cnt], rtol=1e-6, atol=1e-6)
    torch.cuda.synchronize()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("mode,N", [("individual_L2", 2048), ("mixture", 1001), ("common_max", 60000)])
def test_hip_sharded_houses_two_ranks_one_gpu(mode, N):
    _spawn(_hip_sharded_worker, 2, mode, N)


def _hip_sharded_interp_worker(rank, world, port, N):
    """Interpolated base power over sharded houses through torch.distributed: one more SUM all-reduce per update."""
    import mdr_amd
    from mdr_amd.sharding import house_shard
    from tests import golden_util as gu
    _init(rank, world, port)
    torch.cuda.set_device(0)
    E, T = 2, 23
    grid = gu.Golden("s12_interp_default_like").interp_grid()
    cfg = _cfg(N, **{"noise_house_prop.noise_mode": "small_noise", "noise_hvac_prop.noise_mode": "big_noise",
                     "default_env_prop.time_step": 60,                           # update every ceil(300 / 60) = 5 steps
                     "default_env_prop.power_grid_prop.base_power_mode": "interpolation",
                     "default_env_prop.power_grid_prop.signal_mode": "sinusoidals"})
    off, cnt = house_shard(N, world, rank)
    shard = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=33, house_shard=(off, cnt), interp_grid=grid)
    whole = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=33, interp_grid=grid)
    shard.reset(episode=0)
    whole.reset(episode=0)
    seen = set()
    for t in range(T):
        torch.testing.assert_close(shard.t["base_power"], whole.t["base_power"], rtol=1e-13, atol=0)
        torch.testing.assert_close(shard.reg_signal(), whole.reg_signal(), rtol=1e-13, atol=0)
        seen.add(float(whole.t["base_power"][0]))
        shard.step_bangbang()
        whole.step_bangbang()
        assert torch.equal(shard.t["P"], whole.t["P"])
        for k in ("Ta", "Tm", "sso", "flags"):
            assert torch.equal(shard.t[k], whole.t[k][:, off:off + cnt]), k
        torch.testing.assert_close(shard.t["obs"], whole.t["obs"][:, :, off:off + cnt], rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(shard.t["reward"], whole.t["reward"][:, off:off + cnt], rtol=1e-6, atol=1e-6)
    assert len(seen) >= 4            # the base power really moved at the updates
    torch.cuda.synchronize()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("N", [40, 1001])      # below / above interp_nb_agents = 100 (all houses / 100 drawn from the whole env)
def test_hip_sharded_interpolated_base_power_two_ranks_one_gpu(N):
    _spawn(_hip_sharded_interp_worker, 2, N)


def _hip_sharded_obs_worker(rank, world, port, N, mode):
    """obs_vector over sharded houses through torch.distributed: ONE all-gather of the exported message records."""
    import mdr_amd
    from mdr_amd.sharding import house_shard
    _init(rank, world, port)
    torch.cuda.set_device(0)
    E = 2
    cfg = _cfg(N, **{"noise_house_prop.noise_mode": "big_noise", "noise_hvac_prop.noise_mode": "big_noise",
                     "default_env_prop.cluster_prop.agents_comm_mode": mode, "default_env_prop.cluster_prop.nb_agents_comm": 6,
                     "default_env_prop.cluster_prop.comm_defect_prob": 0.25,
                     "default_env_prop.message_properties.thermal": True})
    off, cnt = house_shard(N, world, rank)
    shard = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=3, house_shard=(off, cnt))
    whole = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=3)
    shard.reset(episode=0)
    whole.reset(episode=0)
    for t in range(3):
        assert torch.equal(shard.obs_vector("rows"), whole.obs_vector("rows")[:, off:off + cnt])
        assert torch.equal(shard.obs_vector("planes"), whole.obs_vector("planes")[:, :, off:off + cnt])
        shard.step_bangbang()
        whole.step_bangbang()
    torch.cuda.synchronize()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("N,mode", [(1000, "neighbours"), (602, "closed_groups")])
def test_hip_sharded_obs_vector_two_ranks_one_gpu(N, mode):
    _spawn(_hip_sharded_obs_worker, 2, N, mode)
