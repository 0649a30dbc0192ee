"""The production collective backend, executed: RCCL (torch.distributed backend "nccl") with a world of ONE rank on the one-GPU box.

RCCL refuses two ranks on one device, so the two-rank GPU tests of tests/test_sharded.py ride on gloo; this file runs the very
same `sharding.TorchDistExchange` calls - `max_power` SUM all-reduce, the per-step all-gather of the [3][E] aggregate block, the
message-record all-gather of the observation, the `base_power` SUM all-reduce - through RCCL on device tensors, in stream order with
`mdr_env_step_begin` / `mdr_env_step_end_gathered`, and checks the results bit for bit against the unsharded env
(`exchange_always=True` makes a rank that holds the whole env still walk the begin / exchange / end sequence).

The worker runs in a child process under a time limit: a collective that never completes must not hang the test session.
"""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _cfg(n, **patches):
    from tests import golden_util as gu
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


def _worker():
    import torch
    import torch.distributed as dist

    import mdr_amd
    from mdr_amd.sharding import TorchDistExchange
    from tests import golden_util as gu

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == "nccl"

    # bench.py's fence and max-over-ranks timing on the production backend
    dist.barrier()
    t = torch.tensor([1.25], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == 1.25

    # --- step: max_power all-reduce at episode start, ONE all-gather of the aggregate block per step -------------------------
    for N, mode in ((8192, "mixture"), (1000, "common_max"), (20, "individual_L2")):
        E, T = 3, 12
        cfg = _cfg(N, **{"noise_house_prop.noise_mode": "big_noise", "noise_hvac_prop.noise_mode": "big_noise",
                         "default_house_prop.deadband": 1, "default_env_prop.reward_prop.temp_penalty_mode": mode,
                         "default_env_prop.power_grid_prop.signal_mode": "perlin"})
        whole = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device=dev, seed=21)
        rank0 = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device=dev, seed=21, house_shard=(0, N), exchange_always=True)
        assert rank0.sharded and isinstance(rank0._exchange(), TorchDistExchange)
        whole.reset(episode=0)
        rank0.reset(episode=0)
        assert torch.equal(rank0.t["max_power"], whole.t["max_power"])
        gen = torch.Generator(device="cpu").manual_seed(5)
        for s in range(T):
            act = (torch.rand((E, N), generator=gen) < 0.5).to(torch.uint8).to(dev)
            if s % 2 == 0:
                rank0.step(act)
                whole.step(act)
            else:
                rank0.step_bangbang()
                whole.step_bangbang()
            assert torch.equal(rank0.t["P"], whole.t["P"]), (N, s)
            for k in ("Ta", "Tm", "sso", "flags", "obs"):
                assert torch.equal(rank0.t[k], whole.t[k]), (N, k, s)
            if N > 4096:       # both walk the split path: the same partial records in the same order
                assert torch.equal(rank0.t["reward"], whole.t["reward"]), (N, s)
            else:              # env-per-workgroup kernel vs partial records: the penalty sum is added in a different order
                torch.testing.assert_close(rank0.t["reward"], whole.t["reward"], rtol=1e-6, atol=1e-6)
        del whole, rank0

    # --- graph mode: begin - RCCL all-gather - end captured as ONE hipGraph and replayed (no host work per step) ---------------
    for N, E, T in ((8192, 3, 60), (1000, 2, 37), (125000, 1, 2000)):      # the last: the 8-GPU share of C5, 125 table refills
        cfg = _cfg(N, **{"noise_house_prop.noise_mode": "big_noise", "noise_hvac_prop.noise_mode": "big_noise",
                         "default_env_prop.power_grid_prop.signal_mode": "perlin"})
        whole = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device=dev, seed=4)
        rank0 = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device=dev, seed=4, house_shard=(0, N), exchange_always=True,
                                                 graph_mode=True, table_steps=16)      # 16-row tables: the replays stop for refills
        whole.reset(episode=0)
        rank0.reset(episode=0)
        rank0.rollout(T)
        assert getattr(rank0, "_shard_graph", None) is not None, "the captured path did not run"
        whole.rollout(T)
        act = (torch.rand((E, N), device=dev) < 0.5).to(torch.uint8)
        rank0.rollout(9, act)          # another (pointer, source) pair: re-captured
        whole.rollout(9, act)
        rank0.step(act)                # and eager steps keep working in between
        whole.step(act)
        rank0.rollout(5, act)
        whole.rollout(5, act)
        torch.cuda.synchronize()
        assert rank0.steps_taken == whole.steps_taken == T + 15
        for k in ("Ta", "Tm", "sso", "flags", "obs", "P"):
            assert torch.equal(rank0.t[k], whole.t[k]), ("graph", N, k)
        if N > 4096:
            assert torch.equal(rank0.t["reward"], whole.t["reward"]), ("graph", N)
        else:
            torch.testing.assert_close(rank0.t["reward"], whole.t["reward"], rtol=1e-6, atol=1e-6)
        del whole, rank0

    # --- observation: all-gather of the message records (random_sample exports every record) -----------------------------------
    N, E = 600, 2
    cfg = _cfg(N, **{"noise_house_prop.noise_mode": "big_noise", "noise_hvac_prop.noise_mode": "big_noise",
                     "default_env_prop.cluster_prop.agents_comm_mode": "random_sample", "default_env_prop.cluster_prop.nb_agents_comm": 6,
                     "default_env_prop.cluster_prop.comm_defect_prob": 0.25, "default_env_prop.message_properties.thermal": True})
    whole = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device=dev, seed=3)
    rank0 = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device=dev, seed=3, house_shard=(0, N), exchange_always=True)
    whole.reset(episode=0)
    rank0.reset(episode=0)
    for s in range(3):
        assert torch.equal(rank0.obs_vector("rows"), whole.obs_vector("rows"))
        assert torch.equal(rank0.obs_vector("planes"), whole.obs_vector("planes"))
        rank0.step_bangbang()
        whole.step_bangbang()
    del whole, rank0

    # --- interpolated base power: the SUM all-reduce of base_power[E] ------------------------------------------------------------
    grid = gu.Golden("s12_interp_default_like").interp_grid()
    cfg = _cfg(40, **{"default_env_prop.time_step": 60, "default_env_prop.power_grid_prop.base_power_mode": "interpolation"})
    env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=4, device=dev, seed=33, interp_grid=grid)
    env.reset(episode=0)
    before = env.t["base_power"].clone()
    assert float(before.abs().sum()) > 0
    TorchDistExchange().sum_base_power(env)
    TorchDistExchange().sum_max_power(env)
    torch.cuda.synchronize()
    assert torch.equal(env.t["base_power"], before)

    dist.barrier()
    dist.destroy_process_group()
    print("rccl world-1 ok")


@pytest.mark.gpu
def test_exchange_path_over_rccl_world_of_one():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    res = subprocess.run([sys.executable, "-c", "import tests.test_gpu_rccl as t; t._worker()"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=420)
    assert res.returncode == 0, res.stdout[-2000:] + "\n" + res.stderr[-4000:]
    assert "rccl world-1 ok" in res.stdout
