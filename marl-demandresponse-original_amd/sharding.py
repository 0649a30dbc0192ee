"""Partitioning helpers for the two multi-GPU layouts.

* env replicas  - envs are independent (no state is shared between MADemandResponseEnv instances):
                  rank r owns a contiguous block of global env indices, no collective on the data path.
* sharded houses - one env's houses split into contiguous ranges; the ranks exchange only the per-env
                  aggregates (cluster_hvac_power, penalty sum / max, max_power) each step.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple


def env_shard(nb_envs_total: int, world_size: int, rank: int) -> Tuple[int, int]:
    """(env_offset, nb_envs) of `rank`; blocks differ by at most one env."""
    if not (0 <= rank < world_size) or nb_envs_total < world_size:
        raise ValueError("need 0 <= rank < world_size <= nb_envs_total")
    base, extra = divmod(nb_envs_total, world_size)
    count = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return offset, count


def house_shard(nb_houses_total: int, world_size: int, rank: int, granule: int = 4) -> Tuple[int, int]:
    """(house_offset, nb_houses) of `rank`.  Ranges are multiples of `granule` houses (16-byte vector accesses
    need shard bases that are multiples of 4 houses) except that the last rank takes the remainder."""
    if not (0 <= rank < world_size):
        raise ValueError("need 0 <= rank < world_size")
    granules = nb_houses_total // granule
    if granules < world_size:
        raise ValueError("too few houses (%d) to shard over %d ranks" % (nb_houses_total, world_size))
    base, extra = divmod(granules, world_size)
    start = (rank * base + min(rank, extra)) * granule
    count = (base + (1 if rank < extra else 0)) * granule
    if rank == world_size - 1:
        count = nb_houses_total - start
    return start, count


class HaloPlan:
    """Who needs whose messages, for one rank of a sharded-houses layout (halo exchange).

    From the GLOBAL link table [N_total, c] (replicated: it is static) and the ranks' house ranges every rank derives,
    without communication, the same picture:  need[q] = the remote houses rank q's houses listen to;  export[r] = the
    houses of rank r that anyone else needs (sorted);  per step every rank contributes its export records, padded to
    `export_max`, to ONE all-gather, and picks its halo out of the gathered block with (src_rank, src_pos).

    slots      int32 [n_local, c]  record slot of every link: local house -> its index, remote -> n_local + halo index
    export_idx int64 [X_r]         local indices of the houses this rank exports
    src_rank, src_pos int64 [H]    where each halo record sits in the gathered [world, E, export_max, mf] block"""

    all_records = False      # True: every shard exports everything, record slots are global house ids (random_sample)
    local_base = 0           # slot of this rank's first record

    @classmethod
    def everything(cls, ranges: Sequence[Tuple[int, int]], rank: int, nb_comm: int) -> "HaloPlan":
        """agents_comm_mode 'random_sample': the senders are re-drawn among all houses every step, so every record is
        needed everywhere; slots are global house ids, the local records sit at their global position."""
        import numpy as np
        plan = cls.__new__(cls)
        off, cnt = ranges[rank]
        plan.all_records, plan.local_base, plan.ranges = True, int(off), [(int(o), int(c)) for o, c in ranges]
        plan.n_local = int(cnt)
        plan.entries = int(sum(c for _, c in ranges))
        plan.halo = plan.entries - plan.n_local
        plan.export_idx = np.arange(cnt, dtype=np.int64)
        plan.export_max = int(max(c for _, c in ranges))
        plan.src_rank = plan.src_pos = np.zeros(0, dtype=np.int64)
        plan.slots = np.zeros((cnt, nb_comm), dtype=np.int32)        # unused: the kernel draws the senders
        return plan

    def __init__(self, links, ranges: Sequence[Tuple[int, int]], rank: int):
        import numpy as np
        links = np.asarray(links, dtype=np.int64)
        world = len(ranges)
        starts = np.array([r[0] for r in ranges], dtype=np.int64)
        needs = []
        for (off, cnt) in ranges:
            mine = links[off:off + cnt]
            remote = mine[(mine < off) | (mine >= off + cnt)]
            needs.append(np.unique(remote))
        wanted = np.unique(np.concatenate(needs)) if world > 1 else np.zeros(0, dtype=np.int64)
        owner_of = lambda ids: np.searchsorted(starts, ids, side="right") - 1
        exports = [wanted[owner_of(wanted) == r] for r in range(world)]          # sorted global ids each rank exports
        self.export_max = int(max((len(x) for x in exports), default=0))
        off, cnt = ranges[rank]
        self.n_local = int(cnt)
        self.export_idx = exports[rank] - off
        need = needs[rank]
        self.halo = int(len(need))
        self.src_rank = owner_of(need)
        self.src_pos = np.zeros(len(need), dtype=np.int64)
        for r in range(world):
            sel = self.src_rank == r
            self.src_pos[sel] = np.searchsorted(exports[r], need[sel])
        mine = links[off:off + cnt]
        local = (mine >= off) & (mine < off + cnt)
        self.slots = np.where(local, mine - off, cnt + np.searchsorted(need, mine)).astype(np.int32)
        self.entries = self.n_local + self.halo

    def to(self, device):
        import torch
        self.slots_dev = torch.from_numpy(np_contig(self.slots)).to(device)
        self.export_dev = torch.from_numpy(self.export_idx.astype("int64")).to(device)
        self.src_rank_dev = torch.from_numpy(self.src_rank.astype("int64")).to(device)
        self.src_pos_dev = torch.from_numpy(self.src_pos.astype("int64")).to(device)
        return self

    def pick(self, gathered):
        """halo records [E, H, mf] out of the gathered block [world, E, export_max, mf]"""
        return gathered[self.src_rank_dev, :, self.src_pos_dev, :].permute(1, 0, 2)


def np_contig(a):
    import numpy as np
    return np.ascontiguousarray(a)


class TorchDistExchange:
    """The exchanges of the sharded-houses layout over torch.distributed (RCCL on the GPU box, gloo in CPU tests):
    one SUM all-reduce of `max_power` per episode and ONE all-gather of the per-workgroup partial records per step."""

    @property
    def capturable(self) -> bool:
        """May its collectives sit inside a hipGraph capture?  RCCL's may (BatchedDemandResponseEnv.rollout captures begin -
        all-gather - end as one graph); gloo's are host work."""
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized() and dist.get_backend(self.process_group) == "nccl"

    def __init__(self, process_group=None):
        self.process_group = process_group
        self._gathered = None
        self._records = None

    def agree_partial_records(self, env) -> None:
        """Every rank all-gathers equal blocks: raise the record stride of `partials` to the largest shard's (once per env)."""
        import torch
        import torch.distributed as dist
        if getattr(env, "_records_agreed", False):
            return
        m = torch.tensor([env._partial_records], dtype=torch.int64, device=env.device)
        dist.all_reduce(m, op=dist.ReduceOp.MAX, group=self.process_group)
        env._grow_partials(int(m.item()))
        env._records_agreed = True

    def gather_partials(self, env):
        """every rank's `partials` [E][R][3] -> [world][E][R][3] on every rank (R = the agreed record stride)"""
        import torch
        import torch.distributed as dist
        world = dist.get_world_size(self.process_group)
        part = env.t["partials"]
        shape = (world,) + tuple(part.shape)
        if self._records is None or tuple(self._records.shape) != shape:
            self._records = torch.empty(shape, dtype=torch.float64, device=env.device)
        dist.all_gather_into_tensor(self._records.view(world * part.shape[0], part.shape[1], 3), part, group=self.process_group)
        return self._records, world

    def sum_max_power(self, env) -> None:
        import torch.distributed as dist
        dist.all_reduce(env.t["max_power"], op=dist.ReduceOp.SUM, group=self.process_group)

    def sum_base_power(self, env) -> None:
        import torch.distributed as dist
        dist.all_reduce(env.t["base_power"], op=dist.ReduceOp.SUM, group=self.process_group)

    def ranges(self, env):
        import torch.distributed as dist
        world, rank = dist.get_world_size(self.process_group), dist.get_rank(self.process_group)
        ranges = [house_shard(env.nb_agents, world, r) for r in range(world)]
        if ranges[rank] != (env.house_offset, env.nb_houses):
            raise ValueError("obs_vector over sharded houses needs the sharding.house_shard partition")
        return ranges, rank

    def gather_messages(self, env, padded):
        """`padded`: this rank's export records [E, export_max, mf] -> every rank's [world, E, export_max, mf]."""
        import torch
        import torch.distributed as dist
        world = dist.get_world_size(self.process_group)
        out = torch.empty((world,) + tuple(padded.shape), dtype=padded.dtype, device=padded.device)
        if padded.numel() == 0:      # nobody listens across a shard edge (e.g. a world of one): nothing to exchange
            return out
        dist.all_gather_into_tensor(out.view(world * padded.shape[0], padded.shape[1], padded.shape[2]), padded, group=self.process_group)
        return out

    def persist_mailbox(self, env, spin_limit: int = 0):
        """The mailboxes of the persistent rollout (mdr_env_rollout_persistent) across processes: every rank allocates its own
        (zero-filled, fine-grained so that a peer device's stores become visible inside the running kernel), the 64-byte
        inter-process handles travel through one all_gather_object, and every rank maps its peers' boxes.  Done once per env;
        returns (mdr_mailbox_t, address of this rank's box)."""
        import ctypes as C
        import os
        import torch.distributed as dist
        from . import _native as nat
        cached = getattr(env, "_persist_dist", None)
        if cached is not None:
            cached[0].spin_limit = int(spin_limit)
            return cached[0], cached[1]
        lib = env._lib
        world, rank = dist.get_world_size(self.process_group), dist.get_rank(self.process_group)
        if world > nat.MDR_MAX_SHARDS:
            raise ValueError("the mailbox exchange serves at most %d shards" % nat.MDR_MAX_SHARDS)
        ranges, _ = self.ranges(env)
        recs = [env.persist_records(cnt) for _, cnt in ranges]
        stride = max(recs)
        nbytes = int(lib.mdr_mailbox_bytes(env.nb_envs, world, stride))
        own = C.c_void_p()
        import torch
        with torch.cuda.device(env.device):
            fine = 0 if os.environ.get("MDR_MAILBOX_COARSE") == "1" else 1
            nat.check(lib, None, lib.mdr_mailbox_alloc(nbytes, fine, C.byref(own)), "mdr_mailbox_alloc")
            handle = C.create_string_buffer(64)
            boxes = [None] * world
            if world > 1:
                nat.check(lib, None, lib.mdr_mailbox_export(own, handle), "mdr_mailbox_export")
                handles = [None] * world
                dist.all_gather_object(handles, bytes(handle.raw), group=self.process_group)
                for r in range(world):
                    if r == rank:
                        continue
                    peer = C.c_void_p()
                    nat.check(lib, None, lib.mdr_mailbox_open(handles[r], C.byref(peer)), "mdr_mailbox_open")
                    boxes[r] = peer.value
            boxes[rank] = own.value
        mb = nat.MdrMailbox()
        mb.struct_size = C.sizeof(nat.MdrMailbox)
        mb.world, mb.rank, mb.records_per_env, mb.co_resident, mb.spin_limit = world, rank, stride, 1, int(spin_limit)
        # ranks of one device (the one-GPU rehearsal) crowd the same compute units: count them all for the residency check
        mb.co_resident = int(os.environ.get("MDR_MAILBOX_CO_RESIDENT", "1"))
        mb.system_scope = 1 if world > 1 else 0
        for r in range(world):
            mb.records[r] = recs[r]
            mb.boxes[r] = boxes[r]
        if world > 1:
            dist.barrier(group=self.process_group)      # nobody pushes before every mapping exists
        env._persist_dist = (mb, own.value, boxes)
        return mb, own.value

    def gather_totals(self, env):
        import torch
        import torch.distributed as dist
        world = dist.get_world_size(self.process_group)
        if self._gathered is None or self._gathered.shape[0] != world:
            self._gathered = torch.empty((world, 3, env.nb_envs), dtype=torch.float64, device=env.device)
        # concatenation form [world * 3, E] (same memory as [world][3][E]): accepted by RCCL and by gloo alike
        dist.all_gather_into_tensor(self._gathered.view(world * 3, env.nb_envs), env.t["tot"], group=self.process_group)
        return self._gathered, world


class LocalShardGroup:
    """All house shards of the same envs driven from ONE process: `nb_shards` BatchedDemandResponseEnv objects, spread
    round-robin over `devices` (a single device rehearses BASELINE config 5's eight 125,000-house shards on one GPU).
    The per-step exchange is a stack of the shards' partial-record blocks copied to each shard's device - the peer-copy
    alternative to the RCCL all-gather that one-process-per-GPU runs use - and every shard's
    kernels are issued before the exchange so that the shards overlap.  Results are identical to the
    torch.distributed path: both feed mdr_env_step_end_records the same [world][E][records][3] tensor."""

    def __init__(self, config: dict, nb_envs: int, nb_shards: int, devices: Sequence = ("cuda:0",), seed: int = 0, **kw):
        from .batched_env import BatchedDemandResponseEnv
        total = int(config["default_env_prop"]["cluster_prop"]["nb_agents"])
        self.nb_shards, self.nb_envs, self.nb_agents = int(nb_shards), int(nb_envs), total
        self.shards: List = []
        for r in range(self.nb_shards):
            env = BatchedDemandResponseEnv(config, nb_envs=nb_envs, device=devices[r % len(devices)], seed=seed,
                                           house_shard=house_shard(total, self.nb_shards, r), **kw)
            env._exchange_impl = self            # a lone shard.step() would wait for peers that never come: refuse it
            self.shards.append(env)
        records = max(env._partial_records for env in self.shards)      # equal blocks: the largest shard's record count
        for env in self.shards:
            env._grow_partials(records)

    # the shards must move in lockstep, which only the group can guarantee
    def sum_max_power(self, env):
        raise RuntimeError("shards of a LocalShardGroup are stepped through the group, not one by one")

    gather_totals = gather_partials = sum_base_power = sum_max_power

    def agree_partial_records(self, env):
        pass                                             # the constructor gave every shard the largest shard's count

    def _interp_exchange(self):
        due = [env._interp_due() for env in self.shards]
        if not any(due):
            return
        assert all(due), "shards out of lockstep"
        for env in self.shards:
            env._interp_local()
        self._sync_devices()
        total = sum(env.t["base_power"].to(self.shards[0].device) for env in self.shards)
        for env in self.shards:
            env.t["base_power"].copy_(total)
            env._interp_apply()

    def _sync_devices(self):
        import torch
        if len({e.device for e in self.shards}) > 1:    # cross-device reads must see the producers' results
            for env in self.shards:
                torch.cuda.current_stream(env.device).synchronize()

    def reset(self, seed: Optional[int] = None, episode: Optional[int] = None):
        for env in self.shards:
            env._reset_local(seed, episode)
        if self.nb_shards > 1:                           # ClusterHouses.max_power spans the whole env (env 798-802)
            self._sync_devices()
            total = sum(env.t["max_power"].to(self.shards[0].device) for env in self.shards)
            for env in self.shards:
                env.t["max_power"].copy_(total)
        for env in self.shards:
            env._begin_episode_local()
        self._interp_exchange()
        return [env._reset_obs() for env in self.shards]

    def _finish(self):
        import torch
        first = self.shards[0]
        self._sync_devices()
        block = torch.stack([env.t["partials"].to(first.device) for env in self.shards])     # [world][E][records][3]
        for env in self.shards:
            env._gathered_local = block if env.device == first.device else block.to(env.device)
            env._step_end(env._gathered_local, self.nb_shards)
        self._interp_exchange()

    def _begin(self, env, ptr, source):
        if env.sharded:
            env._step_begin(ptr, source)
        else:                                            # nb_shards == 1: the plain fused step
            env._step(ptr, source)

    def step(self, actions: Sequence):
        """`actions[r]`: uint8/bool [E, shard r's houses] on shard r's device."""
        from . import _native as nat
        for env, act in zip(self.shards, actions):
            self._begin(env, env._actions_ptr(act), nat.ACTIONS_EXTERNAL)
        if self.nb_shards > 1:
            self._finish()
        return [(e.t["obs"], e.t["reward"]) for e in self.shards]

    def step_bangbang(self):
        from . import _native as nat
        for env in self.shards:
            self._begin(env, env.t["actions"].data_ptr(), nat.ACTIONS_BANGBANG)
        if self.nb_shards > 1:
            self._finish()
        return [(e.t["obs"], e.t["reward"]) for e in self.shards]

    def step_controller(self, kind: str = "bangbang"):
        """One step of every shard under a rule-based controller evaluated in-kernel (BatchedDemandResponseEnv.set_controller)."""
        from . import _native as nat
        if kind not in nat.CONTROLLERS:
            raise ValueError("unknown controller %r" % (kind,))
        for env in self.shards:
            self._begin(env, env.t["actions"].data_ptr(), nat.CONTROLLERS[kind])
        if self.nb_shards > 1:
            self._finish()
        return [(e.t["obs"], e.t["reward"]) for e in self.shards]

    def ranges(self, env):
        return [(e.house_offset, e.nb_houses) for e in self.shards], self.shards.index(env)

    def gather_messages(self, env, padded):
        raise RuntimeError("shards of a LocalShardGroup are observed through the group, not one by one")

    def obs_vector(self, layout: str = "rows"):
        """utils.normStateDict of every house incl. the neighbour messages that cross shard edges: one tensor per shard
        (`rows` [E, n_r, F] / `planes` [F, E, n_r]).  Same three steps as the torch.distributed path - message records,
        one gather of the exported records, observation from record slots - with the gather done by device copies."""
        import torch
        first = self.shards[0]
        if self.nb_shards == 1:
            return [first.obs_vector(layout)]
        padded = [env._obs_messages() for env in self.shards]
        self._sync_devices()
        gathered = torch.stack([p.to(first.device) for p in padded])
        return [env._obs_from_gathered(layout, gathered if env.device == first.device else gathered.to(env.device))
                for env in self.shards]

    def cluster_hvac_power(self):
        return self.shards[0].t["P"]

    def gather(self, name: str):
        """One [E, N_total] tensor of a per-house array (on the first shard's device)."""
        import torch
        return torch.cat([env.t[name].to(self.shards[0].device) for env in self.shards], dim=1)
