"""Tensor API: E independent demand-response environments x N houses stepped by one HIP launch.

Host-side mirror of ``MADemandResponseEnv`` (env/MA_DemandResponse.py:37-390 of the reference) for a batch of
environments.  PyTorch-ROCm is plumbing only: it owns the device memory (one slab per env batch) and the
stream; all arithmetic happens in libmdr_hip.so through the C ABI of include/mdr.h.

    env = BatchedDemandResponseEnv(config, nb_envs=4096, device="cuda:0", seed=2024)
    obs = env.reset()                                   # float32 [7, E, N] observation planes
    obs, reward, done, info = env.step(actions)         # actions: uint8/bool [E, N] on the device
    env.rollout(1000)                                   # 1000 fused bang-bang steps, no host round trip

When one environment's houses are sharded over several devices (``house_shard=(offset, count)``) the step
is split around an all-reduce of the per-env aggregates (cluster power, penalty sum / max) over
``torch.distributed`` (RCCL on ROCm); independent env replicas need no collective at all.
"""
from __future__ import annotations

import copy
import ctypes as C
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _native as nat
from .config import EnvSpec, flatten_config, from_epoch_seconds

OBS_COLUMNS = ("house_temp", "house_mass_temp", "hvac_turned_on", "hvac_lockout",
               "hvac_seconds_since_off", "reg_signal", "cluster_hvac_power")

_HOUSE_F32 = ("Ta", "Tm", "k01", "s0", "k10", "s1", "inv_Ua", "Q_hvac", "P_max", "target", "deadband",
              "Ua", "Cm", "Ca", "Hm", "capacity", "COP", "latent", "reward")
_HOUSE_I32 = ("sso", "lockout")
_HOUSE_U8 = ("flags", "actions")
_EPISODE_F64 = ("Ta", "Tm", "target", "deadband", "Ua", "Cm", "Ca", "Hm", "capacity", "COP", "latent")


def _align(x: int, a: int) -> int:
    return (x + a - 1) // a * a


class BatchedDemandResponseEnv:
    def __init__(self, config: dict, nb_envs: int = 1, device=None, seed: int = 0, test: bool = False,
                 table_steps: int = 64, env_offset: int = 0,
                 house_shard: Optional[Tuple[int, int]] = None, process_group=None,
                 stagger_bytes: int = 2304, interp_grid=None, regenerate_missing_grid: bool = True,
                 graph_mode: bool = False, exchange_always: bool = False, partial_records: Optional[int] = None,
                 obs_planes: bool = True, prefetch_tables: bool = True):
        if not torch.cuda.is_available():
            raise RuntimeError("BatchedDemandResponseEnv needs a ROCm device (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        self._lib = nat.load()
        self.config = config
        self.test = test
        self.spec: EnvSpec = flatten_config(config, test=test)
        self.device = torch.device(device if device is not None else "cuda:0")
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.nb_envs = int(nb_envs)
        self.nb_agents = self.spec.nb_houses_total
        self.house_offset, self.nb_houses = (0, self.nb_agents) if house_shard is None else map(int, house_shard)
        self._act_shape = torch.Size((self.nb_envs, self.nb_houses))
        # exchange_always: run the begin / exchange / end sequence even when this rank holds the whole env (a world of one):
        # how the production collectives are exercised on a one-GPU box (tests/test_gpu_rccl.py, bench.py's C5 leg at N = 1)
        self.sharded = house_shard is not None and (self.nb_houses != self.nb_agents or bool(exchange_always))
        self._exchange_always = bool(exchange_always)
        self.process_group = process_group
        self.env_offset = int(env_offset)
        self.table_steps = int(table_steps)
        if self.spec.base_power_mode == 1:   # tables are rebuilt at every interpolatePower call (env 1250-1255)
            self.table_steps = max(self.table_steps, -(-self.spec.interp_update_period // self.spec.time_step))
        self.seed = int(seed)
        self.episode = -1
        self._od_table = None
        self._regenerate_missing_grid = bool(regenerate_missing_grid)
        self._stagger = int(stagger_bytes)
        # sharded houses: records per env in `partials` (the largest shard's workgroup count: every rank all-gathers equal blocks);
        # None = this shard's own count, raised to the group's maximum at the first episode (TorchDistExchange.agree_partial_records)
        self._partial_records = None if partial_records is None else int(partial_records)
        self.graph_mode = bool(graph_mode)      # device-resident cursor: steps / observations can be captured in a graph
        # obs_planes=False: mdr_buffers_t.obs stays NULL and the step kernels skip the seven planes (71 instead of 99 B per
        # house-step) - for loops that observe through obs_vector() / FusedActor.sample_env, which read the state itself
        self._obs_planes_alloc = self._obs_planes_on = bool(obs_planes)
        # a second set of time tables: the next window's tables are built on a side stream of the library while this window's steps
        # run (mdr_buffers_t.tab2_*); not in graph mode / interpolation mode, where refills stay in place
        self._prefetch_tables = bool(prefetch_tables) and not self.graph_mode and self.spec.base_power_mode != 1
        self._handle = C.c_void_p()
        self._cfg = self._make_config()
        rc = self._lib.mdr_env_create(C.byref(self._cfg), C.byref(self._handle))
        try:
            nat.check(self._lib, self._handle, rc, "mdr_env_create")
        except Exception:
            self._lib.mdr_env_destroy(self._handle)
            self._handle = C.c_void_p()
            raise
        self._allocate()
        self._bind()
        if self.spec.base_power_mode == 1:
            self._install_interp_grid(interp_grid)
        self.done = torch.zeros((self.nb_envs, self.nb_houses), dtype=torch.bool, device=self.device)

    # ------------------------------------------------------------------ setup
    def _make_config(self) -> nat.MdrConfig:
        s = self.spec
        c = nat.MdrConfig()
        c.struct_size = C.sizeof(nat.MdrConfig)
        c.nb_envs, c.nb_houses, c.nb_houses_total = self.nb_envs, self.nb_houses, s.nb_houses_total
        c.env_offset, c.house_offset = self.env_offset, self.house_offset
        c.time_step, c.table_steps, c.temp_ref = s.time_step, self.table_steps, s.temp_ref
        for name in ("init_air_temp", "init_mass_temp", "target_temp", "deadband", "Ua", "Cm", "Ca", "Hm",
                     "window_area", "shading_coeff", "lockout_duration", "lockout_noise", "COP", "cooling_capacity",
                     "latent_cooling_fraction", "std_start_temp", "std_target_temp", "factor_thermo_low",
                     "factor_thermo_high", "start_epoch", "day_temp", "night_temp", "temp_std", "signal_mode",
                     "avg_power_per_hvac", "steps_amplitude_per_hvac", "steps_period", "perlin_amplitude",
                     "perlin_nb_octaves", "perlin_octaves_step", "perlin_period", "artificial_ratio",
                     "artificial_signal_ratio_range", "alpha_temp", "alpha_sig", "norm_temp_penalty",
                     "norm_sig_penalty", "penalty_mode", "mix_ind_L2", "mix_common_L2", "mix_common_max",
                     "obs_power_norm"):
            setattr(c, name, getattr(s, name))
        c.solar_gain = int(s.solar_gain)
        c.start_random = int(s.start_random)
        c.random_phase_offset = int(s.random_phase_offset)
        if len(s.capacity_list) > nat.MDR_MAX_CAPACITIES:
            raise ValueError("cooling_capacity_list longer than %d" % nat.MDR_MAX_CAPACITIES)
        c.nb_capacities = len(s.capacity_list)
        for i, v in enumerate(s.capacity_list):
            c.capacity_list[i] = v
        if len(s.sin_periods) > nat.MDR_MAX_SINUSOIDS:
            raise ValueError("more than %d sinusoids" % nat.MDR_MAX_SINUSOIDS)
        c.base_power_mode = s.base_power_mode
        c.nb_sinusoids = len(s.sin_periods)
        for i, (p, r) in enumerate(zip(s.sin_periods, s.sin_amplitude_ratios)):
            c.sin_periods[i], c.sin_amplitude_ratios[i] = p, r
        return c

    def _layout(self):
        E, N, K1 = self.nb_envs, self.nb_houses, self.table_steps + 1
        nblk = int(self._lib.mdr_env_partial_records(self._handle))      # one record per workgroup of the split path
        if self._partial_records is not None and self._partial_records < nblk:
            raise ValueError("partial_records smaller than the %d workgroups this shard needs" % nblk)
        nblk = self._partial_records = max(nblk, self._partial_records or 0)
        if not self.sharded:      # the documented size: mdr_env_rollout keeps two sets of records in it (sharded houses: the stride the ranks agreed on)
            nblk = max(nblk, int(self._lib.mdr_partials_per_env(N)))
        items = [(n, torch.float32, (E, N)) for n in _HOUSE_F32]
        items += [(n, torch.int32, (E, N)) for n in _HOUSE_I32]
        items += [(n, torch.uint8, (E, N)) for n in _HOUSE_U8]
        items += [("obs", torch.float32, (nat.MDR_OBS_COLUMNS, E, N) if self._obs_planes_alloc else (0,))]
        items += [("t0", torch.int64, (E,))] + [(n, torch.float64, (E,)) for n in ("phase", "ratio", "max_power", "P", "base_power")]
        items += [("tot", torch.float64, (3, E))]     # local aggregates as ONE block: tot_sum = tot[0:2], tot_max = tot[2]
        items += [("tab_od", torch.float32, (K1, E)), ("tab_solar", torch.float32, (K1, E)), ("tab_signal", torch.float64, (K1, E)),
                  ("tab_abs_noise", torch.float64, (K1, E))]
        K2 = K1 if self._prefetch_tables else 0
        items += [("tab2_od", torch.float32, (K2, E)), ("tab2_solar", torch.float32, (K2, E)), ("tab2_signal", torch.float64, (K2, E)),
                  ("tab2_abs_noise", torch.float64, (K2, E))]
        items += [("partials", torch.float64, (E, nblk, 3))]
        items += [("cursor", torch.int32, (8,))]      # graph mode: {table row, time index, row note 0, arrival counter, row note 1, -} on the device
        # split path (sharded houses, N > 4096): the houses' own penalties between the partial and the finish kernel; else NULL
        items += [("pen_stash", torch.float32, (E, N) if (self.sharded or N > 4096) else (0,))]
        return items

    def _allocate(self):
        """One device slab; every array 256-byte aligned (optionally staggered so that the ~27 concurrently
        streamed arrays do not all start on the same HBM channel)."""
        items = self._layout()
        offsets, off = {}, 0
        for idx, (name, dtype, shape) in enumerate(items):
            off = _align(off, 256) + (self._stagger * (idx % 16))
            off = _align(off, 256)
            nbytes = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
            offsets[name] = (off, nbytes, dtype, shape)
            off += nbytes
        self._offsets = offsets
        self._slab = torch.zeros(_align(off, 256), dtype=torch.uint8, device=self.device)
        self.t: Dict[str, torch.Tensor] = {}
        for name, (o, nbytes, dtype, shape) in offsets.items():
            self.t[name] = self._slab[o:o + nbytes].view(dtype).view(*shape)
        self.t["tot_sum"], self.t["tot_max"] = self.t["tot"][0:2], self.t["tot"][2]

    def _grow_partials(self, records: int) -> None:
        """Raise the record stride of `partials` to the group's maximum (a fresh zero-filled scratch buffer, re-bound)."""
        if records <= self._partial_records:
            return
        self._partial_records = int(records)
        self.t["partials"] = torch.zeros((self.nb_envs, self._partial_records, 3), dtype=torch.float64, device=self.device)
        self._shard_graph = None      # captured steps hold the old buffer
        self._bind()

    def _bind(self):
        b = nat.MdrBuffers()
        b.struct_size = C.sizeof(nat.MdrBuffers)
        for fname, _ in nat.MdrBuffers._fields_:
            if fname in ("struct_size", "reserved0"):
                continue
            if fname == "cursor" and not self.graph_mode:
                continue                                    # NULL: launch arguments carry the table rows
            if fname == "obs" and not self._obs_planes_on:
                continue                                    # NULL: the step kernels do not write the planes
            setattr(b, fname, self.t[fname].data_ptr() if self.t[fname].numel() else None)      # an empty optional buffer is NULL
        self._buffers = b
        nat.check(self._lib, self._handle, self._lib.mdr_env_bind(self._handle, C.byref(b)), "mdr_env_bind")

    def set_obs_planes(self, on: bool) -> None:
        """Switch the seven per-step observation planes on or off (a re-bind: mdr_buffers_t.obs = NULL skips their 28 B per
        house-step).  Switched on again - or on for the first time - they are brought up to date from the current state."""
        on = bool(on)
        if on == self._obs_planes_on:
            return
        if on and self.t["obs"].numel() == 0:
            self.t["obs"] = torch.zeros((nat.MDR_OBS_COLUMNS, self.nb_envs, self.nb_houses), dtype=torch.float32, device=self.device)
        self._obs_planes_on = on
        self._shard_graph = None      # captured steps hold the old binding
        self._bind()
        if on and self.episode >= 0:
            with torch.cuda.device(self.device):
                nat.check(self._lib, self._handle, self._lib.mdr_env_refresh_obs(self._handle, self._stream()), "mdr_env_refresh_obs")

    def _install_interp_grid(self, interp_grid):
        """PowerGrid.__init__ in interpolation mode (env 1130-1165): load the grid and hand it to the library."""
        from .config import DEFAULT_INTERP_AXES, INTERP_KEYS, InterpolationGridMissing, load_interp_grid
        if interp_grid is not None:
            values, axes = interp_grid
        else:
            try:
                values, axes = load_interp_grid(self.spec.interp_paths)
            except InterpolationGridMissing as exc:
                if not self._regenerate_missing_grid:
                    raise
                import warnings
                from .montecarlo import generate_grid
                warnings.warn("interpolation grid %r not found (the reference does not ship it): regenerating the "
                              "4,199,040-point bang-bang grid on the GPU; run tools/regenerate_interp_grid.py once to keep "
                              "it on disk" % self.spec.interp_paths.get("path_datafile"))
                values, axes = generate_grid(device=self.device), DEFAULT_INTERP_AXES
        if tuple(axes.keys()) != INTERP_KEYS:
            raise ValueError("interpolation grid axes must be " + ", ".join(INTERP_KEYS))
        dims = [len(axes[k]) for k in INTERP_KEYS]
        values = np.ascontiguousarray(np.asarray(values, dtype=np.float64).reshape(-1))
        if values.size != int(np.prod(dims)):
            raise ValueError("interpolation grid has %d values, axes imply %d" % (values.size, int(np.prod(dims))))
        if max(dims) > nat.MDR_INTERP_MAX_AXIS:
            raise ValueError("interpolation axes longer than %d" % nat.MDR_INTERP_MAX_AXIS)
        self._interp_values = torch.from_numpy(values).to(self.device)
        g = nat.MdrInterpGrid()
        g.struct_size = C.sizeof(nat.MdrInterpGrid)
        g.update_period, g.nb_agents = self.spec.interp_update_period, self.spec.interp_nb_agents
        g.values = self._interp_values.data_ptr()
        for d, k in enumerate(INTERP_KEYS):
            g.dims[d] = dims[d]
            for i, v in enumerate(axes[k]):
                g.axes[d][i] = float(v)
        self._interp_grid_host = (values, {k: list(axes[k]) for k in INTERP_KEYS})
        nat.check(self._lib, self._handle, self._lib.mdr_env_set_interp_grid(self._handle, C.byref(g)), "mdr_env_set_interp_grid")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def __del__(self):
        try:
            if getattr(self, "_handle", None) and self._handle.value:
                self._lib.mdr_env_destroy(self._handle)
                self._handle = C.c_void_p()
        except Exception:
            pass

    # ------------------------------------------------------------------ episode start
    def _exchange(self):
        """The object that carries the sharded-houses exchanges (sharding.TorchDistExchange unless a
        sharding.LocalShardGroup installed its own)."""
        ex = getattr(self, "_exchange_impl", None)
        if ex is None:
            from .sharding import TorchDistExchange
            ex = self._exchange_impl = TorchDistExchange(self.process_group)
        return ex

    def _begin_episode(self):
        if self.sharded:  # ClusterHouses.max_power spans the whole env (env 798-802, 125)
            self._exchange().agree_partial_records(self)
            self._exchange().sum_max_power(self)
        self._begin_episode_local()
        self._interp_exchange()

    # sharded houses + interpolated base power: interpolatePower draws its houses from the whole env (env 1209-1215)
    def _interp_due(self) -> bool:
        return self.sharded and self.spec.base_power_mode == 1 and bool(self._lib.mdr_env_interp_due(self._handle))

    def _interp_local(self):
        with torch.cuda.device(self.device):
            nat.check(self._lib, self._handle, self._lib.mdr_env_interp_local(self._handle, self._stream()), "mdr_env_interp_local")

    def _interp_apply(self):
        with torch.cuda.device(self.device):
            nat.check(self._lib, self._handle, self._lib.mdr_env_interp_apply(self._handle, self._stream()), "mdr_env_interp_apply")

    def _interp_exchange(self):
        if self._interp_due():
            self._interp_local()
            self._exchange().sum_base_power(self)
            self._interp_apply()

    def _begin_episode_local(self):
        with torch.cuda.device(self.device):
            nat.check(self._lib, self._handle, self._lib.mdr_env_begin_episode(self._handle, self._stream()), "begin_episode")

    def _reset_local(self, seed: Optional[int] = None, episode: Optional[int] = None) -> None:
        if seed is not None:
            self.seed = int(seed)
        self.episode = self.episode + 1 if episode is None else int(episode)
        self._od_table = None        # mdr_env_reset drops a recorded outdoor-temperature sequence: it belonged to a loaded episode
        with torch.cuda.device(self.device):
            rc = self._lib.mdr_env_reset(self._handle, C.c_uint64(self.seed & 0xFFFFFFFFFFFFFFFF),
                                         C.c_uint32(self.episode & 0xFFFFFFFF), self._stream())
            nat.check(self._lib, self._handle, rc, "mdr_env_reset")

    def reset(self, seed: Optional[int] = None, episode: Optional[int] = None) -> torch.Tensor:
        """MADemandResponseEnv.reset (env 135-172): re-samples every house and the start date, on the device."""
        self._reset_local(seed, episode)
        self._begin_episode()
        return self._reset_obs()

    def load_episode(self, params: Dict[str, np.ndarray], od_table=None, seed: Optional[int] = None,
                     episode: int = 0) -> torch.Tensor:
        """Start an episode from given raw parameters (fp64, deg C) instead of sampling: arrays ``Ta Tm target
        deadband Ua Cm Ca Hm capacity COP latent lockout`` of shape [E, N] and ``t0`` (epoch s) ``phase ratio``
        of shape [E].  ``od_table`` ([rows, E], deg C) replaces the outdoor-temperature model (replays)."""
        if seed is not None:
            self.seed = int(seed)
        self.episode = int(episode)
        E, N = self.nb_envs, self.nb_houses
        with torch.cuda.device(self.device):
            keep = {}
            ep = nat.MdrEpisode()
            ep.struct_size = C.sizeof(nat.MdrEpisode)

            def dev(name, dtype, shape):
                a = np.ascontiguousarray(np.broadcast_to(np.asarray(params[name]), shape)).astype(dtype)
                keep[name] = torch.from_numpy(a).to(self.device)
                return keep[name].data_ptr()

            for name in _EPISODE_F64:
                setattr(ep, name, dev(name, np.float64, (E, N)))
            ep.lockout = dev("lockout", np.int64, (E, N))
            ep.t0 = dev("t0", np.int64, (E,))
            ep.phase = dev("phase", np.float64, (E,)) if "phase" in params else None
            ep.ratio = dev("ratio", np.float64, (E,)) if "ratio" in params else None
            if np.any(np.asarray(params["lockout"]) < 0):
                raise ValueError("Lockout duration must be positive")
            self.set_od_table(od_table)
            rc = self._lib.mdr_env_load_episode(self._handle, C.byref(ep), C.c_uint64(self.seed & 0xFFFFFFFFFFFFFFFF),
                                                C.c_uint32(self.episode), self._stream())
            nat.check(self._lib, self._handle, rc, "mdr_env_load_episode")
            self._begin_episode()
            torch.cuda.current_stream(self.device).synchronize()  # `keep` may be freed after this
            return self._reset_obs()

    def set_od_table(self, od_table):
        if od_table is None:
            self._od_table = None
            rc = self._lib.mdr_env_set_od_table(self._handle, None, 0)
        else:
            tab = torch.as_tensor(np.ascontiguousarray(np.asarray(od_table, dtype=np.float64)))
            if tab.dim() != 2 or tab.shape[1] != self.nb_envs:
                raise ValueError("od_table must have shape [rows, nb_envs]")
            self._od_table = tab.to(self.device)
            rc = self._lib.mdr_env_set_od_table(self._handle, self._od_table.data_ptr(), self._od_table.shape[0])
        nat.check(self._lib, self._handle, rc, "mdr_env_set_od_table")

    def _reset_obs(self) -> torch.Tensor:
        """Observation planes after reset, written by mdr_env_begin_episode's k_reset_obs kernel:
        all HVACs off (env 796-801), P = 0, initial signal (env 133)."""
        return self.t["obs"]

    # ------------------------------------------------------------------ stepping
    def _actions_ptr(self, actions):
        if actions is None:
            return self.t["actions"].data_ptr()
        if actions.dtype == torch.bool:
            actions = actions.view(torch.uint8)
        ptr = actions.data_ptr()
        if (actions.dtype != torch.uint8 or actions.device != self.device or not actions.is_contiguous()
                or actions.shape != self._act_shape or ptr % 4 != 0):
            self.t["actions"].copy_(actions.reshape(self.nb_envs, self.nb_houses).to(self.device) != 0)
            return self.t["actions"].data_ptr()
        self._keep_actions = actions
        return ptr

    def _step(self, ptr, source):
        if not self.sharded:
            # the launch-bound regime is host-bound from Python: skip the device context manager when the device is
            # already current (the common case) - it costs more than the ctypes call itself
            if torch.cuda.current_device() == self._dev_index:
                rc = self._lib.mdr_env_step(self._handle, ptr, source, torch.cuda.current_stream().cuda_stream)
                if rc != 0:
                    nat.check(self._lib, self._handle, rc, "mdr_env_step")
                return
            with torch.cuda.device(self.device):
                rc = self._lib.mdr_env_step(self._handle, C.c_void_p(ptr), source, self._stream())
                nat.check(self._lib, self._handle, rc, "mdr_env_step")
            return
        self._step_begin(ptr, source)
        # TWO launches around ONE collective: step_begin leaves one (power sum, penalty sum, penalty max) record per 1024-house
        # workgroup, the ranks all-gather their record blocks, and every workgroup of step_end re-sums its env's records in one
        # fixed order while it writes the rewards (payload 24 B per 1024 houses and env: latency-bound, so the number of
        # collectives and launches is what counts)
        records, world = self._exchange().gather_partials(self)
        self._step_end(records, world)
        self._interp_exchange()       # every ceil(interp_update_period / time_step) steps: one more SUM all-reduce of [E]

    def _step_begin(self, ptr, source):
        with torch.cuda.device(self.device):
            rc = self._lib.mdr_env_step_begin_records(self._handle, C.c_void_p(ptr), source, self._partial_records, self._stream())
            nat.check(self._lib, self._handle, rc, "mdr_env_step_begin_records")

    def _step_end(self, records: torch.Tensor, world: int):
        """`records`: float64 [world][E][partial_records][3] on this device - every shard's `t['partials']`."""
        if tuple(records.shape) != (world, self.nb_envs, self._partial_records, 3) or not records.is_contiguous():
            raise ValueError("records must be a contiguous [world, E, %d, 3] tensor" % self._partial_records)
        with torch.cuda.device(self.device):
            rc = self._lib.mdr_env_step_end_records(self._handle, C.c_void_p(records.data_ptr()), int(world), self._stream())
            nat.check(self._lib, self._handle, rc, "mdr_env_step_end_records")

    def _step_end_begin(self, records: torch.Tensor, world: int, ptr, source) -> bool:
        """Finish of the pending step and begin of the next one in ONE launch (mdr_env_step_end_begin_records).  False, nothing
        launched, where the library asks for the separate calls (the next step leaves the time tables, interpolated base power)."""
        with torch.cuda.device(self.device):
            rc = self._lib.mdr_env_step_end_begin_records(self._handle, C.c_void_p(records.data_ptr()), int(world), C.c_void_p(ptr), source,
                                                          self._stream())
            if rc == nat.MDR_ERR_UNSUPPORTED:
                return False
            nat.check(self._lib, self._handle, rc, "mdr_env_step_end_begin_records")
        return True

    def _steps_sharded(self, n: int, ptr, source) -> None:
        """n consecutive steps of sharded houses: begin, (all-gather, end + begin in one launch) x (n - 1), all-gather, end - one
        launch and one collective per step instead of two launches.  Every step's rewards are written as they are by _step."""
        if n <= 0:
            return
        ex = self._exchange()
        self._step_begin(ptr, source)
        for _ in range(n - 1):
            records, world = ex.gather_partials(self)
            if not self._step_end_begin(records, world, ptr, source):
                self._step_end(records, world)
                self._interp_exchange()
                self._step_begin(ptr, source)
        records, world = ex.gather_partials(self)
        self._step_end(records, world)
        self._interp_exchange()

    def step(self, actions: torch.Tensor):
        """MADemandResponseEnv.step (env 174-210).  Returns (obs [7,E,N], reward [E,N], done [E,N], info)."""
        self._step(self._actions_ptr(actions), nat.ACTIONS_EXTERNAL)
        return self.t["obs"], self.t["reward"], self.done, {"cluster_hvac_power": self.t["P"]}

    def step_bangbang(self):
        """One step with the bang-bang rule evaluated in-kernel; the actions taken land in ``self.t['actions']``."""
        self._step(self.t["actions"].data_ptr(), nat.ACTIONS_BANGBANG)
        return self.t["obs"], self.t["reward"], self.done, {"cluster_hvac_power": self.t["P"]}

    def set_controller(self, kind: str = "bangbang") -> None:
        """The rule-based controller the closed loops apply in-kernel when no actions are handed over (``step_controller``,
        ``rollout``, ``rollout_fused``, ``rollout_persistent``) - the classes of agents/bangbang_controllers.py as main-deploy.py
        drives them: "bangbang" (BangBangController: on iff hotter than the target; the default), "deadband" / "basic"
        (DeadbandBangBangController == BasicController: off below target - deadband / 2, on above target + deadband / 2, otherwise
        what the HVAC is doing), "always_on" (AlwaysOnController)."""
        if kind not in nat.CONTROLLERS:
            raise ValueError("unknown controller %r (one of %s)" % (kind, ", ".join(sorted(nat.CONTROLLERS))))
        rc = self._lib.mdr_env_set_controller(self._handle, nat.CONTROLLERS[kind])
        nat.check(self._lib, self._handle, rc, "mdr_env_set_controller")
        self._controller = nat.CONTROLLERS[kind]

    def greedy_myopic_actions(self) -> torch.Tensor:
        """GreedyMyopic (agents/greedy_myopic_controller.py): per env, the houses ranked hottest-relative-to-target first and switched
        on while the power budget ``reg_signal`` lasts (the reference's rule, lockout quirk included).  Returns ``self.t['actions']``
        (uint8 [E, N]) for the current observation; at most 2048 houses per env, unsharded."""
        with torch.cuda.device(self.device):
            rc = self._lib.mdr_env_greedy_myopic_actions(self._handle, C.c_void_p(self.t["actions"].data_ptr()), self._stream())
            nat.check(self._lib, self._handle, rc, "mdr_env_greedy_myopic_actions")
        return self.t["actions"]

    def step_greedy_myopic(self):
        """One step under the GreedyMyopic controller: its actions (one launch), then the step."""
        return self.step(self.greedy_myopic_actions())

    def step_controller(self):
        """One step with the controller of ``set_controller`` evaluated in-kernel; the actions taken land in ``self.t['actions']``."""
        self._step(self.t["actions"].data_ptr(), getattr(self, "_controller", nat.ACTIONS_BANGBANG))
        return self.t["obs"], self.t["reward"], self.done, {"cluster_hvac_power": self.t["P"]}

    SHARD_GRAPH_UNROLLS = (16, 4)      # steps per captured graph (a graph launch costs ~10 us, a hop between its nodes ~2 us); leftovers before a table refill go 4 at a time, then singly

    def _rollout_sharded_graph(self, nb_steps: int, ptr: int, source: int) -> None:
        """Sharded houses in graph mode: begin, the all-gather of the records, end - SHARD_GRAPH_UNROLLS[0] steps of it - are captured
        in a hipGraph (RCCL collectives are capturable) and replayed; the device cursor walks the time tables, the host only comes
        back when they run out.  The per-step work is unchanged (every step still exchanges), the per-step host cost - three
        Python -> C calls and a torch.distributed dispatch, ~30 us against 8-19 us of kernels - is gone.  Every rank captures and
        replays in lockstep (same cursor, same room)."""
        done = 0
        key = (ptr, source)
        cached = getattr(self, "_shard_graph", None)
        if cached is not None and cached[0] != key:
            cached = self._shard_graph = None
        if cached is None and nb_steps > 0:
            self._step(ptr, source)                 # eager: sizes the exchange buffers, brings the communicator up
            done = 1
            cached = self._shard_graph = (key, {})
        while done < nb_steps:
            self.graph_replayed(0)                  # tables refilled if used up, device cursor current
            n = min(self.graph_room(), nb_steps - done)
            if n < 1:                               # the tables end here: one ordinary step
                self._step(ptr, source)
                done += 1
                continue
            unroll = next((u for u in self.SHARD_GRAPH_UNROLLS if n >= u), 1)
            g = cached[1].get(unroll)
            if g is None:
                side = torch.cuda.Stream(device=self.device)
                side.wait_stream(torch.cuda.current_stream(self.device))
                g = torch.cuda.CUDAGraph()
                # thread-local capture: the communicator's own threads (watchdog, proxies) keep making runtime calls meanwhile
                with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                    self._steps_sharded(unroll, ptr, source)      # begin, (all-gather, end + begin) x (unroll - 1), all-gather, end
                torch.cuda.current_stream(self.device).wait_stream(side)
                cached[1][unroll] = g
            reps = n // unroll
            for _ in range(reps):
                g.replay()
            self.graph_replayed(reps * unroll)
            done += reps * unroll

    def rollout(self, nb_steps: int, actions: Optional[torch.Tensor] = None):
        """nb_steps consecutive launches without returning to Python (the controller of ``set_controller`` - bang-bang by default -
        unless ``actions`` is given)."""
        ctl = getattr(self, "_controller", nat.ACTIONS_BANGBANG)
        if self.sharded:
            source = nat.ACTIONS_EXTERNAL if actions is not None else ctl
            if self.graph_mode and self.spec.base_power_mode != 1 and getattr(self._exchange(), "capturable", False):
                return self._rollout_sharded_graph(int(nb_steps), self._actions_ptr(actions), source)
            self._steps_sharded(int(nb_steps), self._actions_ptr(actions), source)
            return
        src = nat.ACTIONS_EXTERNAL if actions is not None else ctl
        with torch.cuda.device(self.device):
            rc = self._lib.mdr_env_rollout(self._handle, C.c_void_p(self._actions_ptr(actions)), src, int(nb_steps), self._stream())
            nat.check(self._lib, self._handle, rc, "mdr_env_rollout")

    def rollout_fused(self, nb_steps: int, power_trace: bool = False, accumulate: bool = True):
        """`nb_steps` bang-bang steps with the houses held in registers between steps (one launch per time-table
        chunk).  Ends in the same state as ``rollout(nb_steps)`` bit for bit.  Returns a dict with the accumulators
        of main-deploy.py:124-152: ``reward_sum`` [E,N], ``sq_temp_error_sum`` [E], ``sq_signal_error_sum`` [E] and,
        if asked, ``power_trace`` [nb_steps, E].  Shapes without a fused kernel (N > 2048, or N > 512 with N % 4 != 0) are
        stepped one launch at a time inside the library with the same accumulators; sharded houses fall back to
        ``rollout`` and return None."""
        E, N = self.nb_envs, self.nb_houses
        out = nat.MdrRolloutOut()
        out.struct_size = C.sizeof(nat.MdrRolloutOut)
        res = {}
        with torch.cuda.device(self.device):
            if accumulate:
                res["reward_sum"] = torch.zeros((E, N), dtype=torch.float32, device=self.device)
                res["sq_temp_error_sum"] = torch.zeros(E, dtype=torch.float64, device=self.device)
                res["sq_signal_error_sum"] = torch.zeros(E, dtype=torch.float64, device=self.device)
                out.reward_sum = res["reward_sum"].data_ptr()
                out.sq_temp_error_sum = res["sq_temp_error_sum"].data_ptr()
                out.sq_signal_error_sum = res["sq_signal_error_sum"].data_ptr()
            if power_trace:
                res["power_trace"] = torch.zeros((nb_steps, E), dtype=torch.float64, device=self.device)
                out.power_trace = res["power_trace"].data_ptr()
            rc = self._lib.mdr_env_rollout_fused(self._handle, C.c_void_p(self.t["actions"].data_ptr()), int(nb_steps),
                                                 C.byref(out), self._stream())
            if rc == nat.MDR_ERR_UNSUPPORTED:
                self.rollout(nb_steps)
                return None
            nat.check(self._lib, self._handle, rc, "mdr_env_rollout_fused")
        return res

    # ------------------------------------------------------------------ persistent rollout (mailbox exchange, no kernel boundary per step)
    PERSIST_ERRORS = {1: "a house workgroup waited too long for the totals of a step", 2: "a reducer waited too long for a step's records"}

    def persist_records(self, nb_houses: Optional[int] = None) -> int:
        """Records one shard of `nb_houses` houses pushes per env and step (one per 1024-house workgroup)."""
        return int(self._lib.mdr_persist_records(int(self.nb_houses if nb_houses is None else nb_houses)))

    def _persist_mailbox(self, world: int, stride: int) -> torch.Tensor:
        """This shard's mailbox: zero-filled ONCE, from then on written by the persistent launches only."""
        box = getattr(self, "_mailbox", None)
        size = int(self._lib.mdr_mailbox_bytes(self.nb_envs, world, stride)) // 8
        if box is None or box.numel() != size:
            box = self._mailbox = torch.zeros(size, dtype=torch.int64, device=self.device)
        return box

    def persist_status(self) -> int:
        """Word 0 of the mailbox after the stream has drained: 0 = every persistent launch so far ran to its end."""
        addr = getattr(self, "_mailbox_addr", None)
        if addr is None:
            box = getattr(self, "_mailbox", None)
            if box is None:
                return 0
            addr = box.data_ptr()
        word = C.c_uint64()
        with torch.cuda.device(self.device):
            torch.cuda.synchronize(self.device)
            nat.check(self._lib, None, self._lib.mdr_mailbox_peek(C.c_void_p(addr), C.byref(word)), "mdr_mailbox_peek")
        return int(word.value)

    def _persist_raise(self, word: int):
        kind = (word >> 28) & 0xF
        raise RuntimeError("persistent rollout gave up at step tag %d (workgroup %d): %s; the buffers hold the state before the launch, "
                           "the handle's step count does not - rebuild the env" % ((word >> 32) & 0xFFFFFFFF, word & 0x0FFFFFFF,
                                                                                  self.PERSIST_ERRORS.get(kind, "kind %d" % kind)))

    def _persist_call(self, nb_steps: int, mailbox: "nat.MdrMailbox", power_trace: bool, accumulate: bool, stream=None):
        E, N = self.nb_envs, self.nb_houses
        out = nat.MdrRolloutOut()
        out.struct_size = C.sizeof(nat.MdrRolloutOut)
        res = {}
        with torch.cuda.device(self.device):
            if accumulate:
                res["reward_sum"] = torch.zeros((E, N), dtype=torch.float32, device=self.device)
                res["sq_temp_error_sum"] = torch.zeros(E, dtype=torch.float64, device=self.device)
                res["sq_signal_error_sum"] = torch.zeros(E, dtype=torch.float64, device=self.device)
                out.reward_sum = res["reward_sum"].data_ptr()
                out.sq_temp_error_sum = res["sq_temp_error_sum"].data_ptr()
                out.sq_signal_error_sum = res["sq_signal_error_sum"].data_ptr()
            if power_trace:
                res["power_trace"] = torch.zeros((nb_steps, E), dtype=torch.float64, device=self.device)
                out.power_trace = res["power_trace"].data_ptr()
            if stream is not None:      # a side stream (LocalShardGroup): behind the accumulators' zero fills
                stream.wait_stream(torch.cuda.current_stream(self.device))
            st = self._stream() if stream is None else C.c_void_p(stream.cuda_stream)
            rc = self._lib.mdr_env_rollout_persistent(self._handle, C.c_void_p(self.t["actions"].data_ptr()), int(nb_steps), C.byref(out),
                                                      C.byref(mailbox), st)
            nat.check(self._lib, self._handle, rc, "mdr_env_rollout_persistent")
        return res

    def rollout_persistent(self, nb_steps: int, power_trace: bool = False, accumulate: bool = True, check: bool = True,
                           spin_limit: int = 0):
        """`nb_steps` bang-bang steps in ONE launch per time-table window with the houses resident in registers and the per-step
        exchange of (cluster power, penalty sum / max) through a mailbox (mdr_env_rollout_persistent): any cluster size on one
        device - the split path's 1 env x 1,000,000 houses included - and, over torch.distributed, the sharded-houses layout
        with peer mailboxes instead of the per-step all-gather.  Same results and accumulators as `rollout_fused`; `check`
        synchronises and raises if a wait inside the kernel gave up (check=False: poll `persist_status()` yourself)."""
        if self.sharded and hasattr(self._exchange(), "persist_mailbox"):
            mb, self._mailbox_addr = self._exchange().persist_mailbox(self, spin_limit)
        else:
            if self.sharded and not self._exchange_always:
                raise RuntimeError("rollout_persistent over sharded houses needs an exchange that hands out peer mailboxes")
            n = self.persist_records()
            box = self._persist_mailbox(1, n)
            mb = nat.MdrMailbox()
            mb.struct_size = C.sizeof(nat.MdrMailbox)
            mb.world, mb.rank, mb.records_per_env, mb.co_resident, mb.spin_limit = 1, 0, n, 1, int(spin_limit)
            mb.records[0] = n
            mb.boxes[0] = box.data_ptr()
        res = self._persist_call(int(nb_steps), mb, power_trace, accumulate)
        if check:
            word = self.persist_status()
            if word:
                self._persist_raise(word)
        return res

    def pack_env(self, env_index: int = 0) -> np.ndarray:
        """Host copy of what the dict surface shows of one env (mdr_env_pack): float64 [5 N + 7], one launch + one copy."""
        n = self.nb_houses
        buf = self.__dict__.get("_pack_dev")
        if buf is None:
            buf = self._pack_dev = torch.empty(5 * n + 7, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            rc = self._lib.mdr_env_pack(self._handle, int(env_index), C.c_void_p(buf.data_ptr()), self._stream())
            nat.check(self._lib, self._handle, rc, "mdr_env_pack")
        return buf.cpu().numpy()

    # ------------------------------------------------------------------ graph mode (device-resident cursor)
    def graph_room(self) -> int:
        """Steps the time tables still cover: how often a captured step may be replayed before ``graph_replayed``."""
        return int(self._lib.mdr_env_graph_room(self._handle))

    def graph_replayed(self, n: int) -> None:
        """Tell the host side that a captured step was replayed ``n`` times (refills the tables, runs a due update)."""
        with torch.cuda.device(self.device):
            nat.check(self._lib, self._handle, self._lib.mdr_env_graph_replayed(self._handle, int(n), self._stream()), "mdr_env_graph_replayed")

    @property
    def device_time_index(self) -> torch.Tensor:
        """int32 [1] view of the device cursor's time index (graph mode): what FusedActor.sample takes as ``step_dev``."""
        if not self.graph_mode:
            raise RuntimeError("graph_mode is off")
        return self.t["cursor"][1:2]

    # ------------------------------------------------------------------ full normStateDict vector
    def _obs_spec(self, layout: str, with_links: bool = True) -> nat.MdrObsSpec:
        from .comm import nb_comm
        env = self.config["default_env_prop"]
        cluster, sp, mp = env["cluster_prop"], env["state_properties"], env["message_properties"]
        spec = nat.MdrObsSpec()
        spec.struct_size = C.sizeof(nat.MdrObsSpec)
        spec.layout = {"planes": nat.OBS_PLANES, "rows": nat.OBS_ROWS}[layout]
        spec.state_hour, spec.state_day, spec.state_solar_gain = int(sp["hour"]), int(sp["day"]), int(sp["solar_gain"])
        spec.state_thermal, spec.state_hvac = int(sp["thermal"]), int(sp["hvac"])
        spec.message_thermal, spec.message_hvac = int(mp["thermal"]), int(mp["hvac"])
        spec.nb_comm = nb_comm(cluster)
        mode = cluster["agents_comm_mode"]
        if mode == "random_sample":      # senders re-drawn per house and step on the device (env 976-983)
            spec.random_links = 1
        if mode == "no_message":
            spec.nb_comm = 0
        if with_links and (mode not in ("neighbours", "no_message", "random_sample") or getattr(self, "_links_forced", False)):
            links_dev = self._links_device()
            spec.nb_comm = int(links_dev.shape[1])
            spec.links = links_dev.data_ptr() if spec.nb_comm > 0 else None
        spec.comm_defect_prob = float(cluster["comm_defect_prob"])
        house, hvac = self.config["default_house_prop"], self.config["default_hvac_prop"]
        spec.def_Ua, spec.def_Cm, spec.def_Ca, spec.def_Hm = house["Ua"], house["Cm"], house["Ca"], house["Hm"]
        spec.def_COP, spec.def_capacity, spec.def_latent = hvac["COP"], hvac["cooling_capacity"], hvac["latent_cooling_fraction"]
        spec.norm_reg_sig = self.spec.norm_reg_sig
        return spec

    def comm_links_array(self):
        """ClusterHouses.agent_communicators (env 806-902) of the current episode as int32 [nb_agents, c] global sender ids
        (None for 'random_sample', whose senders are re-drawn every step): the table installed with `set_comm_links`, or the
        one the mode implies.  'random_fixed' is re-drawn at every reset (the reference re-draws it in build_environment) as
        a pure function of (seed, episode), so every rank of a sharded run and every view of the env sees the same table."""
        from .comm import links_array
        if getattr(self, "_links_forced", False):
            return self._links_global
        cluster = self.config["default_env_prop"]["cluster_prop"]
        key = (self.seed, self.episode) if cluster["agents_comm_mode"] == "random_fixed" else None
        cached = getattr(self, "_links_cache", None)
        if cached is None or cached[0] != key:
            cached = self._links_cache = (key, links_array(cluster, seed_episode=(self.seed, max(self.episode, 0))))
            self._links_dev_cache = None
            self._halo = None
        return cached[1]

    def _links_device(self) -> torch.Tensor:
        table = self.comm_links_array()
        dev = getattr(self, "_links_dev_cache", None)
        if dev is None or dev[0] is not table:
            dev = self._links_dev_cache = (table, torch.from_numpy(np.ascontiguousarray(table)).to(self.device))
        return dev[1]

    def set_comm_links(self, table) -> None:
        """Install a static [N, c] sender table (ClusterHouses.agent_communicators) instead of the one derived from
        cluster_prop - e.g. the links a 'random_fixed' episode of the reference drew; it stays until replaced (resets keep it)."""
        table = np.ascontiguousarray(np.asarray(table, dtype=np.int32)).reshape(self.nb_agents, -1)
        if table.size and (table.min() < 0 or table.max() >= self.nb_agents):
            raise ValueError("sender ids must be in [0, nb_agents)")
        self._links_global = table
        self._halo = None
        self._links_dev_cache = None
        self._links_forced = True

    def comm_draws(self):
        """The random part of the message gather at the current time index (env 976-1002; mdr_env_comm_draws): for every local
        house and message slot the sender's global house id - int32 [E, N, c] - and whether the link delivers - bool [E, N, c];
        exactly the draws `obs_vector` uses at this step."""
        cluster = self.config["default_env_prop"]["cluster_prop"]
        mode = cluster["agents_comm_mode"]
        spec = self._obs_spec("rows", with_links=False)
        keep_alive = None
        if mode == "random_sample" and not getattr(self, "_links_forced", False):
            spec.random_links = 1
        elif mode != "no_message" and (mode != "neighbours" or getattr(self, "_links_forced", False)):
            spec.random_links = 0
            rows = self.comm_links_array()[self.house_offset:self.house_offset + self.nb_houses]
            keep_alive = torch.from_numpy(np.ascontiguousarray(rows)).to(self.device)
            spec.nb_comm = int(keep_alive.shape[1])
            spec.links = keep_alive.data_ptr() if spec.nb_comm > 0 else None
        c = int(spec.nb_comm)
        senders = torch.empty((self.nb_envs, self.nb_houses, c), dtype=torch.int32, device=self.device)
        keep = torch.empty((self.nb_envs, self.nb_houses, c), dtype=torch.uint8, device=self.device)
        if c > 0:
            with torch.cuda.device(self.device):
                rc = self._lib.mdr_env_comm_draws(self._handle, C.byref(spec), C.c_void_p(senders.data_ptr()), C.c_void_p(keep.data_ptr()),
                                                  self._stream())
                nat.check(self._lib, self._handle, rc, "mdr_env_comm_draws")
                if keep_alive is not None:
                    torch.cuda.current_stream(self.device).synchronize()
        return senders, keep.bool()

    def obs_vector_length(self) -> int:
        spec = self._obs_spec("planes")
        return int(self._lib.mdr_obs_vector_length(C.byref(spec)))

    def obs_vector(self, layout: str = "planes", out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """utils.normStateDict (utils.py:740-880) of every house, messages included, as one tensor:
        ``planes`` -> float32 [F, E, N] (feature-major), ``rows`` -> float32 [E, N, F] (what Actor(num_state) eats).
        Feature order is normStateDict's; F = 11 (+ optional state columns) + nb_comm * 4 (+ optional message columns)."""
        if self.sharded:      # the messages cross shard edges: message records, ONE gather of the exported ones, ext kernel
            if out is not None:
                raise ValueError("obs_vector over sharded houses allocates its own output")
            padded = self._obs_messages()
            return self._obs_from_gathered(layout, self._exchange().gather_messages(self, padded))
        spec = self._obs_spec(layout)
        return self._obs_launch(layout, spec, out, None)

    def _obs_launch(self, layout, spec, out, messages):
        F = int(self._lib.mdr_obs_vector_length(C.byref(spec)))
        E, N = self.nb_envs, self.nb_houses
        shape = (F, E, N) if layout == "planes" else (E, N, F)
        if out is None:
            cache = self.__dict__.setdefault("_obs_vec", {})
            out = cache.get(layout)
            if out is None or tuple(out.shape) != shape:
                if layout == "planes":   # pad the plane stride (+2304 B) so the F planes do not alias one HBM channel
                    stride = E * N + (576 if (E * N * 4) % (1 << 16) == 0 else 0)
                    out = torch.empty(F * stride, dtype=torch.float32, device=self.device).as_strided((F, E, N), (stride, N, 1))
                else:
                    out = torch.empty(shape, dtype=torch.float32, device=self.device)
                cache[layout] = out
        if tuple(out.shape) != shape or out.dtype != torch.float32:
            raise ValueError("out must be a float32 tensor of shape %s" % (shape,))
        if layout == "planes":
            if out.stride(2) != 1 or out.stride(1) != N or out.stride(0) < E * N:
                raise ValueError("planes output needs strides (>= E*N, N, 1)")
            spec.out_plane_stride = out.stride(0)
        elif not out.is_contiguous():
            raise ValueError("rows output must be contiguous")
        with torch.cuda.device(self.device):
            if messages is None:
                rc = self._lib.mdr_env_obs_vector(self._handle, C.byref(spec), C.c_void_p(out.data_ptr()), self._stream())
                nat.check(self._lib, self._handle, rc, "mdr_env_obs_vector")
            else:
                rc = self._lib.mdr_env_obs_vector_ext(self._handle, C.byref(spec), C.c_void_p(messages.data_ptr()),
                                                      int(messages.shape[1]), C.c_void_p(out.data_ptr()), self._stream())
                nat.check(self._lib, self._handle, rc, "mdr_env_obs_vector_ext")
        return out

    # sharded houses: SURVEY 8e "halo exchange" of the neighbour messages
    def _halo_plan(self):
        from .comm import nb_comm
        from .sharding import HaloPlan
        self.comm_links_array()          # a new 'random_fixed' episode drops the cached plan
        plan = getattr(self, "_halo", None)
        if plan is None:
            cluster = self.config["default_env_prop"]["cluster_prop"]
            ranges, rank = self._exchange().ranges(self)
            if cluster["agents_comm_mode"] == "random_sample" and not getattr(self, "_links_forced", False):
                # senders are re-drawn among ALL houses every step: every shard exports all its records (16-44 B per
                # house and env) and the record slots are global house ids
                plan = self._halo = HaloPlan.everything(ranges, rank, nb_comm(cluster)).to(self.device)
            else:
                plan = self._halo = HaloPlan(self.comm_links_array(), ranges, rank).to(self.device)
        return plan

    def _obs_spec_sharded(self, layout, plan):
        spec = self._obs_spec(layout, with_links=False)
        if plan.all_records:             # random_sample: slots are global house ids, drawn in the kernel
            spec.random_links = 1
            return spec
        spec.random_links = 0
        spec.nb_comm = int(plan.slots.shape[1])
        spec.links = plan.slots_dev.data_ptr() if spec.nb_comm > 0 else None
        return spec

    def _obs_messages(self) -> torch.Tensor:
        """Message records of the local houses -> self._msg[:, :n_local]; returns the exported ones padded to the
        group-wide maximum, [E, export_max, mf], ready for the gather."""
        plan = self._halo_plan()
        spec = self._obs_spec_sharded("rows", plan)
        mf = int(self._lib.mdr_obs_message_fields(C.byref(spec)))
        E = self.nb_envs
        msg = getattr(self, "_msg", None)
        if msg is None or tuple(msg.shape) != (E, plan.entries, mf):
            msg = self._msg = torch.zeros((E, plan.entries, mf), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            # local records at slots [local_base, local_base + n_local): 0, or the shard's house offset when slots are global ids
            ptr = msg.data_ptr() + plan.local_base * mf * 4
            rc = self._lib.mdr_env_obs_messages(self._handle, C.byref(spec), C.c_void_p(ptr), plan.entries, self._stream())
            nat.check(self._lib, self._handle, rc, "mdr_env_obs_messages")
        padded = torch.zeros((E, plan.export_max, mf), dtype=torch.float32, device=self.device)
        if len(plan.export_idx):
            padded[:, :len(plan.export_idx)] = msg[:, plan.export_dev + plan.local_base]
        return padded

    def _obs_from_gathered(self, layout, gathered) -> torch.Tensor:
        plan = self._halo_plan()
        if plan.all_records:
            for (off, cnt), block in zip(plan.ranges, gathered):      # every shard's records to their global slots
                if off != plan.local_base:
                    self._msg[:, off:off + cnt] = block[:, :cnt]
        elif plan.halo:
            self._msg[:, plan.n_local:] = plan.pick(gathered)
        return self._obs_launch(layout, self._obs_spec_sharded(layout, plan), None, self._msg)

    # ------------------------------------------------------------------ views of the state
    def cursor(self) -> Tuple[int, int]:
        k, j0 = C.c_int64(), C.c_int64()
        self._lib.mdr_env_cursor(self._handle, C.byref(k), C.byref(j0))
        return k.value, j0.value

    @property
    def steps_taken(self) -> int:
        return self.cursor()[0]

    def _row(self) -> int:
        k, j0 = self.cursor()
        return k - j0

    def house_temp(self) -> torch.Tensor:
        return self.t["Ta"].double() + self.spec.temp_ref

    def house_mass_temp(self) -> torch.Tensor:
        return self.t["Tm"].double() + self.spec.temp_ref

    def target_temp(self) -> torch.Tensor:
        return self.t["target"].double() + self.spec.temp_ref

    def hvac_turned_on(self) -> torch.Tensor:
        return (self.t["flags"] & 1).bool()

    def hvac_lockout(self) -> torch.Tensor:
        return (self.t["flags"] & 2).bool()

    def table(self, name: str) -> torch.Tensor:
        """The time table `name` (tab_od, tab_solar, tab_signal, tab_abs_noise) of the CURRENT window: with table prefetch the two
        sets of table buffers swap roles at every refill (mdr_env_active_tables)."""
        if self._prefetch_tables and int(self._lib.mdr_env_active_tables(self._handle)) == 1:
            return self.t["tab2_" + name[4:]]
        return self.t[name]

    def od_temp(self) -> torch.Tensor:
        return self.table("tab_od")[self._row()].double() + self.spec.temp_ref

    def reg_signal(self) -> torch.Tensor:
        if self.graph_mode:      # row from the device cursor: stays right when the call is captured and replayed
            if not torch.cuda.is_current_stream_capturing():
                self.graph_replayed(0)      # the device cursor is synced lazily (e.g. after reset): make it current
            return torch.index_select(self.t["tab_signal"], 0, self.t["cursor"][:1].long())[0]
        return self.table("tab_signal")[self._row()]

    def solar_gain(self) -> torch.Tensor:
        return self.table("tab_solar")[self._row()].double()

    def datetimes(self):
        k = self.steps_taken
        return [from_epoch_seconds(int(t0) + k * self.spec.time_step) for t0 in self.t["t0"].cpu().tolist()]

    # ------------------------------------------------------------------ snapshot (copy.deepcopy(env) in utils.py:890-1008)
    def state_dict(self) -> dict:
        k, j0 = self.cursor()
        torch.cuda.synchronize(self.device)      # a table prefetch may be writing the other table set on the library's side stream
        slab = self._slab.clone()
        if self._prefetch_tables and int(self._lib.mdr_env_active_tables(self._handle)) == 1:
            for name in ("od", "solar", "signal", "abs_noise"):      # the snapshot keeps the current window in the FIRST table set
                o, nbytes, dtype, shape = self._offsets["tab_" + name]
                slab[o:o + nbytes].view(dtype).view(*shape).copy_(self.t["tab2_" + name])
        return {"slab": slab, "k": k, "j0": j0, "seed": self.seed, "episode": self.episode,
                "od_table": None if self._od_table is None else self._od_table.clone()}

    def load_state_dict(self, sd: dict):
        torch.cuda.synchronize(self.device)      # (a prefetch in flight would write into the slab being replaced)
        self._slab.copy_(sd["slab"])
        self.seed, self.episode = sd["seed"], sd["episode"]
        if sd.get("od_table") is not None:
            self.set_od_table(sd["od_table"].cpu().numpy())
        else:
            self.set_od_table(None)
        rc = self._lib.mdr_env_set_cursor(self._handle, C.c_uint64(self.seed & 0xFFFFFFFFFFFFFFFF),
                                          C.c_uint32(max(self.episode, 0) & 0xFFFFFFFF), sd["k"], sd["j0"])
        nat.check(self._lib, self._handle, rc, "mdr_env_set_cursor")

    def __deepcopy__(self, memo):
        other = BatchedDemandResponseEnv(copy.deepcopy(self.config, memo), nb_envs=self.nb_envs, device=self.device,
                                         seed=self.seed, test=self.test, table_steps=self.table_steps, obs_planes=self._obs_planes_alloc,
                                         prefetch_tables=self._prefetch_tables,
                                         env_offset=self.env_offset,
                                         house_shard=(self.house_offset, self.nb_houses) if self.sharded else None,
                                         exchange_always=self._exchange_always, partial_records=self._partial_records,
                                         process_group=self.process_group, stagger_bytes=self._stagger, graph_mode=self.graph_mode,
                                         interp_grid=getattr(self, "_interp_grid_host", None))
        if getattr(self, "_links_forced", False):
            other.set_comm_links(self._links_global)
        other.episode = self.episode      # 'random_fixed' derives its link table from (seed, episode)
        if hasattr(self, "_controller"):
            other.set_controller(next(k for k, v in nat.CONTROLLERS.items() if v == self._controller))
        if self.episode >= 0:
            other.load_state_dict(self.state_dict())
        return other
