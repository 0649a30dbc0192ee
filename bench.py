#!/usr/bin/env python3
"""Headline benchmark: house-steps/s of the fused env step at 4096 envs x 1024 houses per MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one MADemandResponseEnv.step for every env of the batch (one fused HIP launch) on BASELINE.json
configs[2] (C3): heterogeneous houses (house_big_noise) and HVAC capacities (big_noise), noisy sinusoidal
heat-wave outdoor temperature (one Gaussian per env-step, Philox), solar gain, Perlin regulation signal,
random start date per env, bang-bang actions evaluated in-kernel on the previous observation (closed loop,
no host round trip).  All state is resident in HBM before the timed region.  With N GPUs every rank owns
its own 4096 envs (independent replicas, env_offset = rank * 4096, no data-path collective): weak scaling.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (one
`torch.distributed.run` child, started before this process touches the GPU) and exits with the child's code.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects, plus two
secondary legs measured after the timed region of the headline (BASELINE.json configs[3] and configs[4]):
  `ppo_rollout`  policy-in-the-loop rollout collection on the same per-rank batch (observation -> actor -> sample -> step);
  `c5`           1 env x 1,000,000 houses sharded over the N ranks with the per-step exchange (kernel / collective split);
  `c5_graph`     the same step captured in a hipGraph;
  `c5_persistent` the same env through the persistent rollout: houses resident in registers across steps, the per-step exchange
                 through peer mailboxes instead of a collective (N > 1: measured in one child process per rank, so that a fault on
                 the peer-to-peer path cannot cost the line).
`degraded: true` at the top level whenever a leg carries `error` or an exchange did not run over RCCL.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
_STDOUT_FD = os.dup(1)      # the ONE JSON line goes here whatever fd 1 points at meanwhile (init_group parks it on stderr for RCCL's banner)


def emit(line: dict) -> None:
    sys.stdout.flush()
    os.write(_STDOUT_FD, (json.dumps(line) + "\n").encode())

# algorithmic bytes per house-step of this layout
#   state 13 R + 13 W (Ta, Tm f32; sso i32; flags u8) + parameters 40 R (9 f32 + lockout i32)
#   + action 1 (read, or written when the bang-bang rule runs in-kernel) + reward 4 W + 7 obs planes 28 W
B_ALG = 99
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_COPY_GBS = 6290.0   # the guide's measured float4-copy ceiling (MI355X_MICROARCH.md:36)

E_PER_GPU, N_HOUSES = 4096, 1024
C5_HOUSES = 1_000_000
REFERENCE_CPU_NOTE = ("the reference itself (env/MA_DemandResponse.py, imported in the build container, 8-vCPU Xeon 2.1 GHz, 1 core) "
                      "runs 3.8-5.1e4 house-steps/s (BASELINE.md section 2); its Python cannot travel to the GPU box")


def c3_config(mdr):
    cfg = mdr.default_config()
    env = cfg["default_env_prop"]
    env["cluster_prop"]["nb_agents"] = N_HOUSES
    env["cluster_prop"]["temp_mode"] = "noisy_sinusoidal_heatwave"
    env["power_grid_prop"]["base_power_mode"] = "constant"
    env["power_grid_prop"]["signal_mode"] = "perlin"
    env["start_datetime_mode"] = "random"
    cfg["noise_house_prop"]["noise_mode"] = "house_big_noise"
    cfg["noise_hvac_prop"]["noise_mode"] = "big_noise"
    cfg["default_hvac_prop"]["lockout_noise"] = 0
    cfg["default_house_prop"]["solar_gain_bool"] = True
    return cfg


def cpu_baseline(cfg_full, seconds):
    """The oracle's loop-form port (reference cost structure) timed on one host core, bounded sample."""
    import copy
    from oracle import loop_port
    cfg = copy.deepcopy(cfg_full)
    cfg["default_env_prop"]["power_grid_prop"]["signal_mode"] = "sinusoidals"   # deterministic family the port restates
    rate, steps, el = loop_port.time_baseline(cfg, seconds=seconds)
    out = {"value": rate, "unit": "house-steps/s", "cores": 1, "kind": "port",
           "sample": "oracle/loop_port.py (object-per-house pure-Python restatement, obs dicts + 10-neighbour "
                     "messages as the reference builds them), 1 env x %d houses x %d bang-bang steps, %.1f s; %s"
                     % (N_HOUSES, steps, el, REFERENCE_CPU_NOTE),
           "reference_build_container_value": [3.8e4, 5.1e4]}
    try:   # the reference's only parallelism: independent processes side by side
        procs = max(1, min(16, (os.cpu_count() or 1)))
        mp_rate, procs = loop_port.time_baseline_parallel(cfg, procs, seconds=min(6.0, seconds))
        out["multi_process_value"] = mp_rate
        out["multi_process_cores"] = procs
    except Exception as exc:
        out["multi_process_value"] = None
        out["multi_process_cores"] = "unavailable: %s" % exc
    try:   # the same per-house arithmetic compiled (oracle/mdr_oracle_c.c): what one core does without the interpreter
        from oracle import c_port
        crate, csteps, cel = c_port.time_baseline(cfg, nb_envs=4, seconds=min(4.0, seconds))
        out["c_port_value"] = crate
        out["c_port_sample"] = "oracle/mdr_oracle_c.c (plain C, literal closed-form update, gcc -O2), 4 envs x %d houses x %d steps, %.1f s, 1 core" % (N_HOUSES, csteps, cel)
    except Exception as exc:   # no compiler on the box: the Python port above still stands
        out["c_port_value"] = None
        out["c_port_sample"] = "unavailable: %s" % exc
    try:   # the vectorised NumPy oracle (fp64, [E, N] arrays): SURVEY 8d's third CPU form
        import numpy as np
        from oracle import mdr_oracle as mo
        ora = mo.OracleEnv(cfg, nb_envs=4).reset(seed=2024, episode=0)
        nsteps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < min(3.0, seconds):
            ora.step(ora.bangbang_actions().astype(np.uint8))
            nsteps += 1
        el = time.perf_counter() - t0
        out["numpy_value"] = 4 * N_HOUSES * nsteps / el
        out["numpy_sample"] = "oracle/mdr_oracle.py (vectorised NumPy fp64), 4 envs x %d houses x %d steps, %.1f s, 1 process" % (N_HOUSES, nsteps, el)
    except Exception as exc:
        out["numpy_value"] = None
        out["numpy_sample"] = "unavailable: %s" % exc
    return out


def traffic_from_profiles():
    """HBM bytes per launch of the step kernel from the newest committed PMC passes (profiles/*traffic*.json), or None.
    A constant read from the profile directory, not a measurement of this run (PMC passes need rocprofv3 around the process)."""
    best, src = None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic.json"))):   # the C3 passes of the newest round
        try:
            with open(path) as f:
                best, src = json.load(f), os.path.relpath(path, ROOT)
        except Exception:
            pass
    return (None, None) if best is None else (best.get("hbm_bytes_per_launch"), src)


def out_of_cache_from_profiles():
    """Rate of the step kernel with the re-read set at 8x the Infinity Cache (32768 envs x 1024 houses), from the newest committed
    rocprofv3 + PMC passes (profiles/r*_traffic_32768envs.json): a profile constant, not a measurement of this run."""
    best, src = None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic_32768envs.json"))):
        try:
            with open(path) as f:
                best, src = json.load(f), os.path.relpath(path, ROOT)
        except Exception:
            pass
    return (None, None) if best is None else (best.get("achieved_GBps_from_rocprof_avg"), src)


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as ONE torch.distributed.run child - before this
    process has made any GPU call, and as a child, never an exec - and hand its exit code on.  Rank 0's JSON line goes
    straight to our stdout."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def parse(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the secondary ppo_rollout / c5 legs")
    ap.add_argument("--ppo-steps", type=int, default=20)
    ap.add_argument("--c5-steps", type=int, default=200)
    ap.add_argument("--leg-timeout", type=float, default=float(os.environ.get("MDR_BENCH_LEG_TIMEOUT", "240")),
                    help="seconds the secondary legs may take before the headline line is printed without them")
    ap.add_argument("--stagger", type=int, default=int(os.environ.get("MDR_STAGGER", "2304")))
    ap.add_argument("--persist-child", action="store_true", help=argparse.SUPPRESS)      # one rank of the c5_persistent leg at N > 1
    return ap.parse_args(argv)


class Ranks:
    """Rank bookkeeping + the fence / max-over-ranks of the timing contract."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, self.world))
        # one process per GPU; MDR_BENCH_BACKEND=gloo lets several ranks share one GPU to rehearse the multi-rank code path
        # on a one-GPU box (RCCL refuses two ranks on one device) - never used for reported numbers
        self.backend = os.environ.get("MDR_BENCH_BACKEND", "nccl")
        self.dry = os.environ.get("MDR_BENCH_DRY", "") not in ("", "0")      # launcher rehearsal without a GPU (CPU test)
        self.device = None
        if not self.dry:
            idx = self.local_rank if self.backend == "nccl" else self.local_rank % max(1, torch.cuda.device_count())
            torch.cuda.set_device(idx)
            self.device = torch.device("cuda", idx)
        # a process group also for a world of one when asked (MDR_BENCH_FORCE_DIST=1) or for the C5 leg (see c5_leg)
        self.group_ready = False
        self.backend_note = None
        if self.world > 1 or os.environ.get("MDR_BENCH_FORCE_DIST", "") not in ("", "0"):
            self.init_group()

    def init_group(self):
        if self.group_ready:
            return
        if "MASTER_ADDR" not in os.environ:       # a world of one started without a launcher
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        # RCCL / gloo print a banner on stdout when the first communicator comes up: keep stdout to the ONE JSON line
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if self.backend == "nccl" and not self.dry:
                try:
                    if os.environ.get("MDR_BENCH_BREAK_NCCL"):      # test hook
                        raise RuntimeError("injected failure (MDR_BENCH_BREAK_NCCL)")
                    self.dist.init_process_group("nccl", device_id=self.device)
                    self.dist.barrier()
                except Exception as exc:      # the env replicas of the headline need a fence, not RCCL: keep the scaling curve
                    self.backend_note = "RCCL did not come up (%s: %s): fence over gloo" % (type(exc).__name__, str(exc)[:200])
                    print("bench.py: " + self.backend_note, file=sys.stderr)
                    if self.dist.is_initialized():
                        self.dist.destroy_process_group()
                    self.backend = "gloo"
                    self.dist.init_process_group("gloo")
            else:
                self.dist.init_process_group("gloo")
            self.dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
        self.group_ready = True

    def fence(self):
        if self.group_ready:
            self.dist.barrier()
        if not self.dry:
            self.torch.cuda.synchronize(self.device)

    def max_over_ranks(self, seconds: float) -> float:
        if not self.group_ready:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device=self.device if (self.device is not None and self.backend == "nccl") else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.group_ready:
            self.dist.barrier()
            self.dist.destroy_process_group()


def timed(rk: Ranks, fn, steps: int):
    """fence - clock - fn - fence; returns (wall seconds as max over ranks, this rank's HIP-event ms per step)."""
    torch = rk.torch
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    rk.fence()
    t0 = time.perf_counter()
    ev0.record()               # same stream the kernels are launched on (torch's current stream)
    fn()
    ev1.record()
    rk.fence()
    elapsed = time.perf_counter() - t0
    return rk.max_over_ranks(elapsed), ev0.elapsed_time(ev1) / steps


def ppo_leg(rk: Ranks, env, args):
    """BASELINE.json configs[3] 'PPO rollout collection' on this rank's replica batch: every step = normStateDict vector of
    all agents -> Actor forward + Categorical.sample (one MFMA kernel) -> env step, nothing leaves the GPU
    (train_ppo.py:60-108 for all agents at once).  Random-init Actor of the reference's shape (agents/network.py:14-33)."""
    torch = rk.torch
    from mdr_amd.rollout import ActorMLP, collect_ppo_rollout
    E, N = env.nb_envs, env.nb_houses
    torch.manual_seed(0)
    actor = ActorMLP(env.obs_vector_length()).to(rk.device)
    out = {"metric": "agent-steps/s, policy in the loop (normStateDict -> Actor forward -> Categorical.sample -> env step)",
           "workload": "%d envs x %d houses per GPU (the C3 batch), Actor %d-100-100-2 random init; observation and policy are ONE kernel "
                       "(mdr_env_actor_sample: the 51 features built in LDS from the compact state); transitions stay on the GPU: "
                       "`transitions` = state (204 B/agent, written on the side by the same kernel) + action + a_prob + reward + return, "
                       "`no_states` = the same without keeping the states; during the collection the env steps WITHOUT its seven per-step "
                       "observation planes (mdr_buffers_t.obs = NULL: 71 algorithmic bytes per house-step instead of the headline's 99 - "
                       "nothing in the loop reads them) and brings them up to date once at the end" % (E, N, env.obs_vector_length()),
           "step_kernel_algorithmic_bytes_per_house_step": 71,
           "steps": args.ppo_steps, "n_gpus": rk.world, "scaling": "weak"}
    for prec in ("fp32", "bf16x3"):
        out[prec] = {}
        for key, keep in (("transitions", True), ("no_states", False)):
            # warm-up at the full length: packs the weights and leaves right-sized blocks in torch's caching allocator, so that the
            # timed collection does not pay hipMalloc for its transition buffer (18 GB of states at 20 steps)
            collect_ppo_rollout(env, actor, args.ppo_steps, store_states=keep, policy_precision=prec, seed=rk.rank)
            wall, ev_ms = timed(rk, lambda: collect_ppo_rollout(env, actor, args.ppo_steps, store_states=keep,
                                                                 policy_precision=prec, seed=rk.rank), args.ppo_steps)
            out[prec][key] = {"agent_steps_per_s": E * N * rk.world * args.ppo_steps / wall, "ms_per_step": wall / args.ppo_steps * 1e3,
                              "event_ms_per_step_rank0": ev_ms}
    torch.cuda.empty_cache()
    out["value"] = out["fp32"]["transitions"]["agent_steps_per_s"]
    out["unit"] = "agent-steps/s"
    out["dtype"] = "f32 (exact fp32 MFMA); bf16x3 = split-bf16 operands, fp32 accumulate, probabilities within 2e-5"
    return out


def c5_leg(rk: Ranks, mdr, args, graph=False):
    """BASELINE.json configs[4]: ONE env x 1,000,000 houses, houses sharded over the ranks, one exchange per step
    (all-gather of every rank's per-workgroup records; env/MA_DemandResponse.py:1042-1050, 274-321).  A world of one
    still runs the exchange (through RCCL) so that the collective's cost is on record at every N.
    `graph`: the same step - begin, all-gather, end - captured once in a hipGraph and replayed (graph mode: no host work per step)."""
    torch = rk.torch
    from mdr_amd.sharding import house_shard
    rk.init_group()
    cfg = c3_config(mdr)
    cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = C5_HOUSES
    off, cnt = house_shard(C5_HOUSES, rk.world, rk.rank)
    env = mdr.BatchedDemandResponseEnv(cfg, nb_envs=1, device=rk.device, seed=2024, house_shard=(off, cnt), exchange_always=True,
                                       table_steps=64, graph_mode=graph)
    env.reset(episode=0)
    K = args.c5_steps
    env.rollout(20)
    env.rollout(K)      # untimed: every graph the timed pass replays is captured by now (the eager form takes the same steps: equal checksums)
    wall, ev_ms = timed(rk, lambda: env.rollout(K), K)
    if graph:
        assert env.steps_taken == 20 + 2 * K and bool(torch.isfinite(env.t["Ta"]).all())
        assert rk.backend != "nccl" or getattr(env, "_shard_graph", None) is not None, "the captured path did not run"
        return {"metric": "house-steps/s, 1 env x 1,000,000 houses sharded over the ranks, one all-gather per step, step captured in a hipGraph",
                "value": C5_HOUSES * K / wall, "unit": "house-steps/s", "n_gpus": rk.world, "scaling": "strong", "steps": K,
                "houses_per_rank": cnt, "us_per_step": wall / K * 1e6, "event_us_per_step_rank0": ev_ms * 1e3,
                "captured": getattr(env, "_shard_graph", None) is not None, "checksum_Ta": float(env.t["Ta"].double().sum()),
                "backend": "rccl" if rk.backend == "nccl" else rk.backend,
                "note": "step_begin_records -> all_gather_into_tensor -> step_end_records captured ONCE (device-resident time cursor) and "
                        "replayed until the 64-row time tables need a refill: same work per step as the c5 leg, no host calls per step; "
                        "checksum_Ta equals the c5 leg's (same seed, same number of steps)"}
    # the collective alone (same tensor, same call), and the kernels alone (an unsharded env of this rank's share: the same
    # k_step_partial / k_step_finish launches without the exchange)
    ex = env._exchange()
    for _ in range(5):
        ex.gather_partials(env)
    _, coll_ms = timed(rk, lambda: [ex.gather_partials(env) for _ in range(K)], K)
    cfg_local = c3_config(mdr)
    cfg_local["default_env_prop"]["cluster_prop"]["nb_agents"] = cnt
    local = mdr.BatchedDemandResponseEnv(cfg_local, nb_envs=1, device=rk.device, seed=2024, table_steps=64)
    local.reset(episode=0)
    local.rollout(20)
    _, kern_ms = timed(rk, lambda: local.rollout(K), K)
    assert env.steps_taken == 20 + 2 * K and bool(torch.isfinite(env.t["Ta"]).all())
    return {"metric": "house-steps/s, 1 env x 1,000,000 houses sharded over the ranks, one all-gather per step",
            "value": C5_HOUSES * K / wall, "unit": "house-steps/s", "n_gpus": rk.world, "scaling": "strong", "steps": K,
            "houses_per_rank": cnt, "us_per_step": wall / K * 1e6, "event_us_per_step_rank0": ev_ms * 1e3,
            "checksum_Ta": float(env.t["Ta"].double().sum()),
            "kernel_us_per_step": kern_ms * 1e3, "collective_us_per_step": coll_ms * 1e3,
            "backend": "rccl" if rk.backend == "nccl" else rk.backend,
            "note": "us_per_step is host wall-clock per step (max over ranks) of step_begin_records -> all_gather_into_tensor(24 B per 1024-house workgroup) -> "
                    "step_end_records; kernel = the two step kernels alone on an unsharded env of this rank's %d houses; "
                    "collective = the all-gather alone, back to back" % cnt}


def persist_measure(env, K, fence, max_over_ranks):
    """Warm-up (same step counts as the c5 legs: equal checksums), then K timed steps of the persistent rollout, with and
    without the accumulators of main-deploy.py:124-152."""
    import torch
    out = {}
    env.rollout_persistent(20, check=False)
    env.rollout_persistent(K, check=False)
    for key, acc in (("us_per_step", True), ("us_per_step_no_accumulators", False)):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fence()
        t0 = time.perf_counter()
        ev0.record()
        env.rollout_persistent(K, accumulate=acc, check=False)
        ev1.record()
        fence()
        wall = max_over_ranks(time.perf_counter() - t0)
        out[key] = wall / K * 1e6
        out["event_" + key + "_rank0"] = ev0.elapsed_time(ev1) / K * 1e3
        if acc:
            out["checksum_Ta"] = float(env.t["Ta"].double().sum())      # after 20 + 2 K steps, as the c5 legs
    word = env.persist_status()
    if word:
        raise RuntimeError("a wait inside the persistent kernel gave up: error word 0x%x" % word)
    return out


PERSIST_NOTE = ("mdr_env_rollout_persistent: ONE launch per 64-step table window, each 1024-house workgroup keeps its houses in registers "
                "and pushes its (power sum, penalty sum, max) record as tagged 8-byte granules into every rank's mailbox; one reducer "
                "workgroup per rank re-sums them in the order of step_end_records (bit-identical totals); the houses run up to 7 steps ahead "
                "of the totals. us_per_step includes the per-agent reward sums, squared temperature / signal errors; checksum_Ta equals the "
                "c5 leg's")


def _persist_floor(leg):
    """What a step that STREAMS its state would cost at the HBM figure (the records path: 107 B per house-step - SURVEY 8e / VERDICT r2:
    13.4 us at 1,000,000 houses, 1.7 us at 125,000) beside the measured step: the persistent kernel keeps the state in registers."""
    if "us_per_step" in leg and "houses_per_rank" in leg:
        floor = leg["houses_per_rank"] * 107 / 8.0e12 * 1e6
        leg["streaming_step_floor_us"] = floor
        leg["vs_streaming_step_floor"] = floor / leg["us_per_step"]


def c5_persistent_leg(rk: Ranks, mdr, args):
    """BASELINE.json configs[4] without a kernel boundary or a collective per step ('compare RCCL vs a P2P mailbox')."""
    K = args.c5_steps
    base = {"metric": "house-steps/s, 1 env x 1,000,000 houses sharded over the ranks, persistent kernel + mailbox exchange",
            "unit": "house-steps/s", "n_gpus": rk.world, "scaling": "strong", "steps": K, "note": PERSIST_NOTE}
    if rk.world == 1:
        cfg = c3_config(mdr)
        cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = C5_HOUSES
        env = mdr.BatchedDemandResponseEnv(cfg, nb_envs=1, device=rk.device, seed=2024, table_steps=64)
        env.reset(episode=0)
        m = persist_measure(env, K, rk.fence, rk.max_over_ranks)
        base.update(m)
        base.update({"value": C5_HOUSES * K / (m["us_per_step"] * 1e-6 * K), "houses_per_rank": C5_HOUSES, "exchange": "mailbox on this device (a world of one)"})
        _persist_floor(base)
        return base
    # N > 1: peer mailboxes over hipIpc / xGMI have never run on hardware before the driver's own scaling run - every rank measures
    # in a child process of its own (own process group over gloo, no RCCL: the data path has no collective), so that whatever
    # happens there the parent still prints its line
    port = [free_port() if rk.rank == 0 else 0]
    rk.dist.broadcast_object_list(port, src=0)
    # a rendezvous of their own: without the launcher's agent store (under torch.distributed.run rank 0 would otherwise not serve one)
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port[0]), MDR_BENCH_CHILD_DEVICE=str(rk.device.index))
    if rk.backend != "nccl":      # the one-GPU rehearsal (MDR_BENCH_BACKEND=gloo): every rank's launch must be resident on the same device
        env["MDR_MAILBOX_CO_RESIDENT"] = str(rk.world)
    res = subprocess.run([sys.executable, os.path.abspath(__file__), "--persist-child", "--gpus", str(rk.world), "--c5-steps", str(K)],
                         env=env, capture_output=True, text=True, timeout=max(30.0, args.leg_timeout / 3))
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    if res.returncode != 0 or (rk.rank == 0 and not lines):
        raise RuntimeError("child rc %d: %s" % (res.returncode, (res.stderr or res.stdout)[-400:]))
    if rk.rank == 0:
        base.update(json.loads(lines[-1]))
        _persist_floor(base)
    return base


def persist_child(args):
    """One rank of the c5_persistent leg at N > 1 (started by c5_persistent_leg): its shard of the 1,000,000 houses, peer mailboxes
    mapped over hipIpc, fences over gloo.  Rank 0 prints the leg's numbers as one JSON line."""
    import torch
    import torch.distributed as dist
    import mdr_amd
    from mdr_amd.sharding import house_shard
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", int(os.environ.get("MDR_BENCH_CHILD_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = c3_config(mdr_amd)
    cfg["default_env_prop"]["cluster_prop"]["nb_agents"] = C5_HOUSES
    off, cnt = house_shard(C5_HOUSES, world, rank)
    env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=1, device=dev, seed=2024, house_shard=(off, cnt), table_steps=64)
    env.reset(episode=0)

    def fence():
        dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(seconds):
        t = torch.tensor([seconds], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    m = persist_measure(env, args.c5_steps, fence, max_over_ranks)
    parts = [None] * world
    dist.all_gather_object(parts, m["checksum_Ta"])
    if rank == 0:
        m.update({"value": C5_HOUSES / (m["us_per_step"] * 1e-6), "houses_per_rank": cnt, "checksum_Ta_ranks": parts,
                  "exchange": "peer mailboxes mapped over hipIpc (fine-grained memory, system-scope granules); fences over gloo"})
        emit(m)
    dist.barrier()
    dist.destroy_process_group()


def dry_rank(rk: Ranks, args):
    """MDR_BENCH_DRY=1: rehearse launcher, rendezvous, fence and max-over-ranks without a GPU (tests/test_bench_launch.py).
    No kernel runs and the line says so; it is never a result."""
    if os.environ.get("MDR_BENCH_DRY") == "fail1" and rk.rank == 1:      # a rank that dies must fail the whole launch
        raise SystemExit(3)
    rk.fence()
    t0 = time.perf_counter()
    time.sleep(0.01 * (1 + rk.rank))
    rk.fence()
    elapsed = rk.max_over_ranks(time.perf_counter() - t0)
    if rk.rank == 0:
        emit({"metric": "house-steps/sec at 4096 envs x 1024 houses; achieved HBM GB/s vs roofline", "value": 0.0,
                          "unit": "house-steps/s", "n_gpus": rk.world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": elapsed / max(1, args.steps) * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "dry-run: launcher rehearsal, no kernel ran",
                          "config": {"workload": "none (MDR_BENCH_DRY)"}, "roofline": None, "dry_run": True})
    rk.close()


def run_rank(args):
    rk = Ranks(args)
    if rk.dry:
        return dry_rank(rk, args)
    import mdr_amd
    torch = rk.torch

    cfg = c3_config(mdr_amd)
    e_per_gpu = int(os.environ.get("MDR_BENCH_ENVS", E_PER_GPU))   # rehearsal knob only; the reported config is 4096
    env = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=e_per_gpu, device=rk.device, seed=2024,
                                           env_offset=rk.rank * e_per_gpu, table_steps=64, stagger_bytes=args.stagger)
    env.reset(episode=0)
    env.rollout(args.warmup)
    elapsed, kernel_ms = timed(rk, lambda: env.rollout(args.steps), args.steps)   # kernel_ms: average launch-to-launch duration of the step kernel
    houses = e_per_gpu * N_HOUSES * rk.world
    value = houses * args.steps / elapsed

    # sanity: the rollout really advanced and the state is finite (cheap, outside the timed region)
    assert env.steps_taken == args.warmup + args.steps
    assert bool(torch.isfinite(env.t["Ta"]).all()) and bool(torch.isfinite(env.t["reward"]).all())

    def headline(legs):
        achieved = B_ALG * e_per_gpu * N_HOUSES / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_src = traffic_from_profiles()
        line = {
            "metric": "house-steps/sec at 4096 envs x 1024 houses; achieved HBM GB/s vs roofline",
            "value": value, "unit": "house-steps/s", "n_gpus": rk.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C3: %d envs x %d houses per GPU, house_big_noise + big_noise HVAC, "
                                   "noisy_sinusoidal_heatwave OD temp, solar gain, perlin signal, random start, "
                                   "in-kernel bang-bang closed loop" % (e_per_gpu, N_HOUSES),
                       "envs_per_gpu": e_per_gpu, "houses_per_env": N_HOUSES, "sharding": "independent env replicas, no collective",
                       "seed": 2024, "table_steps": 64},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": "profile constant from %s (PMC passes of an earlier run of this command), not measured in this run" % traffic_src,
                         "kernel": "k_step_fused<4,1,256>", "algorithmic_bytes_per_house_step": B_ALG,
                         "kernel_ms": kernel_ms,
                         "peak_measured_copy": HBM_COPY_GBS, "frac_of_measured_copy": achieved / HBM_COPY_GBS,
                         "cache_note": "C3 re-reads 53 B/house = 222 MB per step, below the 256 MiB Infinity Cache: part of this rate is "
                                       "cache-assisted; the size sweep in profiles/ (r02_size_sweep.jsonl) gives the rate with the re-read set at 2-8x the cache"},
        }
        ooc, ooc_src = out_of_cache_from_profiles()
        if ooc is not None:
            line["roofline"]["frac_out_of_cache"] = ooc / HBM_PEAK_GBS
            line["roofline"]["out_of_cache_source"] = "%s: %.0f GB/s at 32768 envs x 1024 houses (re-read set 8x the Infinity Cache), profile constant" % (ooc_src, ooc)
        if rk.backend_note:
            line["backend_note"] = rk.backend_note
        line.update(legs)
        # rc 0 alone proves nothing (a failing leg never costs the line): say so in ONE place
        reasons = ["%s: %s" % (n, v["error"]) for n, v in legs.items() if isinstance(v, dict) and "error" in v]
        reasons += ["%s ran over %s, not RCCL" % (n, v["backend"]) for n, v in legs.items()
                    if isinstance(v, dict) and v.get("backend") not in (None, "rccl")]
        if rk.backend_note:
            reasons.append(rk.backend_note)
        line["degraded"] = bool(reasons)
        if reasons:
            line["degraded_reasons"] = reasons
        return line

    # The secondary legs never cost the headline: the headline is measured by now, and if a leg has not returned after
    # --leg-timeout seconds (a collective that never completes on some node, say) rank 0 prints the line without it and every
    # rank leaves through os._exit - a hung leg would otherwise take the scaling curve with it.
    legs, lock, printed = {}, threading.Lock(), [False]
    leg_list = (("ppo_rollout", lambda: ppo_leg(rk, env, args)), ("c5", lambda: c5_leg(rk, mdr_amd, args)),
                ("c5_graph", lambda: c5_leg(rk, mdr_amd, args, graph=True)), ("c5_persistent", lambda: c5_persistent_leg(rk, mdr_amd, args)))
    baseline = None
    if rk.rank == 0 and rk.world == 1 and not args.no_cpu_baseline:      # before the legs: a leg that hangs must not cost the line its baseline
        baseline = cpu_baseline(cfg, args.cpu_seconds)

    def give_up():
        with lock:
            if not printed[0] and rk.rank == 0:
                pending = [n for n, _ in leg_list if n not in legs]
                out = dict(legs)
                out.update({n: {"error": "leg did not finish within %.0f s" % args.leg_timeout} for n in pending})
                if baseline is not None:
                    out["cpu_baseline"] = baseline
                emit(headline(out))
            printed[0] = True
        os._exit(0)

    watchdog = threading.Timer(args.leg_timeout, give_up)
    watchdog.daemon = True
    if not args.no_legs:
        watchdog.start()
        for name, leg in leg_list:
            try:
                if os.environ.get("MDR_BENCH_FAIL_LEG") == "%s:%d" % (name, rk.rank):      # test hook: a leg that dies on one rank
                    raise RuntimeError("injected failure (MDR_BENCH_FAIL_LEG)")
                legs[name] = leg()
            except Exception as exc:
                legs[name] = {"error": "%s: %s" % (type(exc).__name__, exc)}
                if rk.world > 1:          # the ranks may no longer agree on what comes next: no further leg; the line goes out, and
                    for other, _ in leg_list:   # if the others sit in a collective the watchdog ends them (and a hung close() here)
                        legs.setdefault(other, {"error": "skipped: an earlier leg failed on this rank"})
                    break

    with lock:
        if rk.rank == 0 and not printed[0]:
            line = headline(legs)
            if baseline is not None:
                line["cpu_baseline"] = baseline
            emit(line)
        printed[0] = True
    rk.close()             # still under the watchdog: a barrier that never completes ends in os._exit(0), the line is out
    watchdog.cancel()


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    if args.persist_child:
        return persist_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(launch_ranks(args, argv))
    run_rank(args)


if __name__ == "__main__":
    main()
