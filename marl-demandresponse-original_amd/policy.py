"""Fused policy forward + action sampling on the matrix cores (include/mdr_policy.h).

``FusedActor`` takes the weights of an ``ActorMLP`` (the reference's Actor, agents/network.py:14-33, two hidden layers,
two actions), lays them out once in MFMA fragment order and then runs, per call, ONE kernel over all agents:
observation rows -> probabilities -> sampled actions (PPO.select_action, agents/ppo.py:68-75, for every agent at once).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np
import torch

from . import _native as nat

MAX_HIDDEN = 127


FRAG32, FRAG16, BF16X3, FRAG16T = 0, 1, 2, 3
FEATURES_NORMSTATE, FEATURES_OBSERVE = 0, 1
OBSERVE_NUM_STATE = 51      # the reference's default observation: 11 own features + 10 neighbours x 4 message fields


def observe_feature_order(num_state: int = OBSERVE_NUM_STATE, msg_floats: int = 40) -> np.ndarray:
    """normStateDict index of staged feature k in MDR_FEATURES_OBSERVE order: the `msg_floats` = 4 * nb_comm message floats
    first, then the own features in normStateDict order (default observation: 10 messages, 11 own features)."""
    own = num_state - msg_floats
    if own < 11 or msg_floats < 0 or msg_floats % 4:
        raise ValueError("an observation of %d features cannot start with %d message floats" % (num_state, msg_floats))
    k = np.arange(num_state)
    return np.where(k < msg_floats, own + k, k - msg_floats)


class MdrActor(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("layout", C.c_int32), ("num_state", C.c_int32), ("hidden1", C.c_int32),
                ("hidden2", C.c_int32), ("greedy", C.c_int32), ("feature_order", C.c_int32), ("observe_msg_floats", C.c_int32),
                ("frag1", C.c_void_p), ("frag2", C.c_void_p), ("wdiff", C.c_void_p)]


def _acc_row(reg: np.ndarray, half: np.ndarray) -> np.ndarray:
    return (reg & 3) + 8 * (reg >> 2) + 4 * half


class FusedActor:
    def __init__(self, w1, b1, w2, b2, w3, b3, device="cuda:0", layout: Optional[int] = None, greedy: bool = False,
                 feature_order: int = FEATURES_NORMSTATE, observe_msg_floats: int = 40):
        """w1 [H1, F], b1 [H1], w2 [H2, H1], b2 [H2], w3 [2, H2], b3 [2] (torch.nn.Linear layout).
        ``feature_order=FEATURES_OBSERVE`` packs W1's columns in the order ``sample_env`` stages the default observation in
        (observe -> act without observation rows: ``mdr_env_actor_sample``); such an actor serves ``sample_env`` only.
        ``layout``: FRAG16 (v_mfma_f32_16x16x4_f32, exact fp32, the default whenever F <= 128 - 16 feature registers per lane up to
        64 features, 32 beyond: the observations with the optional message columns, 81 / 91 / 121 features), FRAG32
        (v_mfma_f32_32x32x2_f32, exact fp32, any F) or BF16X3 (bf16 MFMA on head + tail halves of every operand:
        probabilities within ~1e-5 of the fp32 forward, several times faster; F <= 128 on observation rows, F <= 64 for observe -> act)."""
        self._lib = nat.load()
        w1, b1, w2, b2, w3, b3 = (torch.as_tensor(t, dtype=torch.float32).detach().cpu() for t in (w1, b1, w2, b2, w3, b3))
        self.feature_order = int(feature_order)
        self.observe_msg_floats = int(observe_msg_floats) if self.feature_order == FEATURES_OBSERVE else 0
        if self.feature_order == FEATURES_OBSERVE:
            if w1.shape[1] > 64:
                raise ValueError("observe -> act covers observations of at most 64 features")
            w1 = w1[:, torch.from_numpy(observe_feature_order(w1.shape[1], self.observe_msg_floats))]
        H1, F = w1.shape
        H2 = w2.shape[0]
        if w2.shape[1] != H1 or tuple(w3.shape) != (2, H2):
            raise ValueError("expected Linear(F,H1) - Linear(H1,H2) - Linear(H2,2)")
        if H1 > MAX_HIDDEN or H2 > MAX_HIDDEN:
            raise ValueError("hidden layers of at most %d units" % MAX_HIDDEN)
        if layout is None:      # exact fp32; hidden layers of 97..100 units (the reference's [100, 100]) take the 4x4-tail form
            layout = (FRAG16T if self._tail_shape(H1, H2) else FRAG16) if F <= 128 else FRAG32
        if layout == FRAG16T and not self._tail_shape(H1, H2):
            raise ValueError("FRAG16T needs hidden layers of 97..100 units")
        self.layout = int(layout)
        self.greedy = bool(greedy)      # argmax instead of a draw: the reference's DQNAgent.act on a DQN_network
        self.num_state, self.hidden1, self.hidden2 = int(F), int(H1), int(H2)
        self.device = torch.device(device)
        S1 = int(self._lib.mdr_actor_steps1_order(self.layout, F, self.feature_order))   # observe order, fp32: 13 | 15 | 16 (zero-padded)
        S2 = int(self._lib.mdr_actor_steps2(self.layout, H1))
        if S1 < 0 or S2 < 0:
            raise ValueError("unknown layout %r" % (layout,))
        if self.layout == BF16X3:
            self._pack_bf16x3(w1, b1, w2, b2, w3, b3, S1, S2)
            return
        lane = np.arange(64)
        f16 = self.layout in (FRAG16, FRAG16T)
        kw = 4 if f16 else 2                                                    # k per MFMA step
        bw = 16 if f16 else 32                                                  # rows per block
        nb = 128 // bw                                                          # blocks stored per lane (8 | 4)
        r, g = lane & (bw - 1), lane // bw                                      # row in block, lane group (= k within a step)
        rows = bw * np.arange(nb)[:, None] + r[None, :]                         # [mb, lane] output row of the fragment
        # bias-extended, zero-padded matrices: input feature F / hidden unit H is the constant 1
        w1e = torch.zeros((128, kw * S1))
        w2e = torch.zeros((128, 128))
        w3e = torch.zeros((2, 128))
        pos1 = torch.from_numpy(self._unit_rows(H1))                            # row of w1e / column of w2e that holds hidden-1 unit u
        w1e[pos1, :F], w2e[:H2, pos1], w3e[:, :H2] = w1, w2, w3
        if self.layout == FRAG32:      # biases as a constant-1 input feature / hidden unit
            w1e[:H1, F], w1e[H1, F] = b1, 1.0
            w2e[:H2, H1], w2e[H2, H1] = b2, 1.0
            w3e[:, H2] = b3
        k1 = g[None, :] * S1 + np.arange(S1)[:, None]                           # [s, lane]
        q = np.arange(S2)
        rows1, rows2 = rows, rows.copy()                                        # rows of w1e (unit positions) / w2e (hidden-2 units)
        if self.layout == FRAG16T:      # slot 6 feeds v_mfma_f32_4x4x1: lane -> unit 96 + (lane & 3) (row 127 is zero)
            u = 96 + (lane & 3)
            rows1, rows2 = rows.copy(), rows.copy()
            rows1[6] = np.where(u < H1, 96 + 4 * (lane & 3), 127)      # position of unit u (see _unit_rows)
            rows2[6] = np.where(u < H2, u, 127)
        if f16:
            k2 = 16 * (q >> 2)[:, None] + 4 * g[None, :] + (q & 3)[:, None]     # [q, lane]: the accumulator row the lane holds
            reg = np.arange(4)
            row3 = 16 * np.arange(8)[:, None, None] + 4 * np.arange(4)[None, None, :] + reg[None, :, None]        # [mb, reg, g]
        else:
            k2 = 32 * (q >> 4)[:, None] + _acc_row((q & 15)[:, None], g[None, :])
            reg = np.arange(16)
            row3 = 32 * np.arange(4)[:, None, None] + _acc_row(reg[None, :, None], np.arange(2)[None, None, :])   # [mb, reg, h]
        k2 = np.minimum(k2, 127)
        if self.layout == FRAG16T:      # the tail's head weights count once: lane group 0 holds them
            row3[6] = np.where(np.arange(4)[None, :] == 0, np.where(96 + reg < H2, 96 + reg, 127)[:, None], 127)
        frag1 = w1e[torch.from_numpy(rows1)[None, :, :], torch.from_numpy(k1)[:, None, :]].permute(0, 2, 1)   # [S1, 64, nb]
        frag2 = w2e[torch.from_numpy(rows2)[None, :, :], torch.from_numpy(k2)[:, None, :]].permute(0, 2, 1)   # [S2, 64, nb]
        wdiff = (w3e[0] - w3e[1])[torch.from_numpy(row3)].reshape(-1)
        if f16:      # biases start the accumulators: appended behind the head weights
            wdiff = torch.cat([wdiff, self._bias_block(b1, b2, b3)])
        self._frag1 = frag1.contiguous().to(self.device)
        self._frag2 = frag2.contiguous().to(self.device)
        self._wdiff = wdiff.contiguous().to(self.device)
        assert self._wdiff.numel() == (128 if self.layout == FRAG32 else 388)
        assert self._frag1.numel() * self._lib.mdr_actor_steps1(self.layout, F) == self._lib.mdr_actor_frag1_floats(self.layout, F) * S1
        assert self._frag2.numel() == self._lib.mdr_actor_frag2_floats(self.layout, H1)
        self._desc = MdrActor(C.sizeof(MdrActor), self.layout, F, H1, H2, int(self.greedy), self.feature_order, self.observe_msg_floats, self._frag1.data_ptr(), self._frag2.data_ptr(),
                              self._wdiff.data_ptr())

    @staticmethod
    def _tail_shape(h1: int, h2: int) -> bool:
        return 97 <= h1 <= 100 and 97 <= h2 <= 100

    def _unit_rows(self, hidden1: int) -> np.ndarray:
        """Row that holds hidden-1 unit u.  MDR_ACTOR_FRAG16 stores the units of a partial last 16-row block transposed (unit
        16 b + j at row 16 b + 4 (j % 4) + j // 4): k-step q of layer 2 feeds the rows 16 (q >> 2) + 4 g + (q & 3), so the block's
        first ceil(rem / 4) k-steps then carry all of its units and the others are skipped (100 units: 25 k-steps, not 28)."""
        u = np.arange(hidden1)
        if self.layout not in (FRAG16, FRAG16T):
            return u
        j = u & 15
        return np.where(u >= 16 * (hidden1 // 16), (u & ~15) + 4 * (j & 3) + (j >> 2), u)

    def _bias_block(self, b1, b2, b3) -> torch.Tensor:
        """b1[mb][g][reg] | b2[mb][g][reg] | b3[0] - b3[1] | 0 0 0  (row = 16 mb + 4 g + reg), 260 floats."""
        b1e, b2e = torch.zeros(128), torch.zeros(128)
        b1e[torch.from_numpy(self._unit_rows(self.hidden1))], b2e[:self.hidden2] = b1, b2
        rowb = 16 * np.arange(8)[:, None, None] + 4 * np.arange(4)[None, :, None] + np.arange(4)[None, None, :]      # [mb, g, reg]
        rowb1, rowb2 = rowb.copy(), rowb.copy()
        if self.layout == FRAG16T:      # the tail's biases count once: lane group 0 starts its accumulators with them
            reg = np.arange(4)
            rowb1[6] = np.where(np.arange(4)[:, None] == 0, np.where(96 + reg < self.hidden1, 96 + 4 * reg, 127)[None, :], 127)
            rowb2[6] = np.where(np.arange(4)[:, None] == 0, np.where(96 + reg < self.hidden2, 96 + reg, 127)[None, :], 127)
        rowb1, rowb2 = torch.from_numpy(rowb1), torch.from_numpy(rowb2)
        tail = torch.zeros(4)
        tail[0] = b3[0] - b3[1]
        return torch.cat([b1e[rowb1].reshape(-1), b2e[rowb2].reshape(-1), tail])

    def _pack_bf16x3(self, w1, b1, w2, b2, w3, b3, S1, S2):
        """MDR_ACTOR_BF16X3: every weight as a bf16 head + tail, fragments of 8 k-values per lane (include/mdr_policy.h)."""
        F, H1, H2 = self.num_state, self.hidden1, self.hidden2
        lane = np.arange(64)
        r, g = lane & 15, lane >> 4
        j = np.arange(8)
        rows = 16 * np.arange(8)[:, None] + r[None, :]                          # [mb, lane]
        # biases start the accumulators in this layout: plain zero-padded matrices, no constant-1 feature / unit
        w1e = torch.zeros((128, 32 * S1))
        w1e[:H1, :F] = w1
        w2e = torch.zeros((128, 128))
        w2e[:H2, :H1] = w2
        w3e = torch.zeros((2, 128))
        w3e[:, :H2] = w3
        k1 = (4 * np.arange(S1)[:, None, None] + g[None, :, None]) * 8 + j[None, None, :]                   # [s, lane, j]
        k2 = 16 * (2 * np.arange(S2)[:, None, None] + (j >> 2)[None, None, :]) + 4 * g[None, :, None] + (j & 3)[None, None, :]

        def frags(we, k):                                                       # -> [s, mb, 2, lane, j] bf16 bit patterns
            kk = torch.from_numpy(np.minimum(k, we.shape[1] - 1))
            vals = we[torch.from_numpy(rows)[None, :, :, None], kk[:, None, :, :]]              # [s, mb, lane, j]
            vals = torch.where(torch.from_numpy(k < we.shape[1])[:, None, :, :], vals, torch.zeros(()))
            head = vals.to(torch.bfloat16)
            tail = (vals - head.float()).to(torch.bfloat16)
            return torch.stack([head, tail], dim=2).contiguous()

        reg = np.arange(4)
        row3 = 16 * np.arange(8)[:, None, None] + 4 * np.arange(4)[None, None, :] + reg[None, :, None]       # [mb, reg, g]
        self._frag1 = frags(w1e, k1).view(torch.int16).to(self.device)
        self._frag2 = frags(w2e, k2).view(torch.int16).to(self.device)
        self._wdiff = torch.cat([(w3e[0] - w3e[1])[torch.from_numpy(row3)].reshape(-1),
                                 self._bias_block(b1, b2, b3)]).contiguous().to(self.device)
        assert self._wdiff.numel() == 388
        assert self._frag1.numel() * 2 == 4 * self._lib.mdr_actor_frag1_floats(self.layout, F)
        assert self._frag2.numel() * 2 == 4 * self._lib.mdr_actor_frag2_floats(self.layout, H1)
        self._desc = MdrActor(C.sizeof(MdrActor), self.layout, F, H1, H2, int(self.greedy), self.feature_order, self.observe_msg_floats, self._frag1.data_ptr(), self._frag2.data_ptr(),
                              self._wdiff.data_ptr())

    @classmethod
    def from_module(cls, actor, device=None, layout: Optional[int] = None, greedy: bool = False,
                    feature_order: int = FEATURES_NORMSTATE, observe_msg_floats: int = 40) -> "FusedActor":
        """From an ``ActorMLP`` / the reference's ``Actor`` - or, with ``greedy=True``, its ``DQN_network`` (the same ``fc``
        ModuleList of three Linear layers, agents/network.py:58-77, whose two outputs are Q-values: action = argmax)."""
        fc = list(actor.fc)
        if len(fc) != 3:
            raise ValueError("the fused kernel covers two hidden layers (config.py: layers = [100, 100])")
        dev = device if device is not None else fc[0].weight.device
        return cls(fc[0].weight, fc[0].bias, fc[1].weight, fc[1].bias, fc[2].weight, fc[2].bias, device=dev, layout=layout, greedy=greedy,
                   feature_order=feature_order, observe_msg_floats=observe_msg_floats)

    def sample_env(self, env, seed: int, step: int, want_probs: bool = False, action: Optional[torch.Tensor] = None,
                   a_prob: Optional[torch.Tensor] = None, step_dev: Optional[torch.Tensor] = None,
                   rows_out: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, ...]:
        """Observe -> act in ONE kernel (``mdr_env_actor_sample``): ``utils.normStateDict`` of every agent of ``env`` (a
        ``BatchedDemandResponseEnv`` in its current state) is built in LDS from the compact state and fed straight to the
        matrix-core forward - no observation rows.  Needs an actor packed with ``feature_order=FEATURES_OBSERVE`` (layout FRAG16
        or BF16X3) for this observation shape (optional state columns, up to 13 senders - circular neighbours, a link table or
        random_sample -, link defects; at most 64 features); raises ``NotImplementedError`` otherwise - the optional MESSAGE
        columns - (use ``env.obs_vector('rows')`` + ``sample``).  Same draws and outputs as ``sample`` on the rows (agent = env * N + house).
        ``rows_out`` (float32 [A, 51] contiguous): also receives the observation rows in normStateDict order - bit for bit
        ``env.obs_vector('rows')`` - written on the side by the same kernel (the transition buffer's ``state``)."""
        A = env.nb_envs * env.nb_houses
        action = torch.empty(A, dtype=torch.uint8, device=self.device) if action is None else action
        a_prob = torch.empty(A, dtype=torch.float32, device=self.device) if a_prob is None else a_prob
        probs = torch.empty((A, 2), dtype=torch.float32, device=self.device) if want_probs else None
        if step_dev is not None and (step_dev.dtype != torch.int32 or step_dev.device != self.device):
            raise ValueError("step_dev must be an int32 tensor on the device (env.device_time_index)")
        spec = env._obs_spec("rows")
        if rows_out is not None and (rows_out.dtype != torch.float32 or rows_out.device != self.device or not rows_out.is_contiguous()
                                     or rows_out.numel() != A * self.num_state):
            raise ValueError("rows_out must be a contiguous float32 [A, %d] tensor on the device" % self.num_state)
        tail = (C.c_uint64(seed & (2 ** 64 - 1)), C.c_uint64(step & (2 ** 64 - 1)), C.c_void_p(step_dev.data_ptr()) if step_dev is not None else None,
                C.c_void_p(action.data_ptr()), C.c_void_p(a_prob.data_ptr()), C.c_void_p(probs.data_ptr()) if want_probs else None,
                C.c_void_p(rows_out.data_ptr()) if rows_out is not None else None)
        with torch.cuda.device(self.device):
            stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            if spec.nb_comm > 0 and (spec.links or spec.random_links):
                # senders by a link table / re-drawn every step: the kernel gathers the senders' message records, which the call
                # writes into scratch kept with the env (64 MB at 4.19 M houses; random_sample: + 4 nb_comm bytes per house)
                msg = getattr(env, "_observe_msg", None)
                if msg is None or msg.numel() != A * 4 or msg.device != self.device:
                    msg = env._observe_msg = torch.empty(A * 4, dtype=torch.float32, device=self.device)
                senders = None
                if spec.random_links:
                    senders = getattr(env, "_observe_senders", None)
                    if senders is None or senders.numel() != A * spec.nb_comm or senders.device != self.device:
                        senders = env._observe_senders = torch.empty(A * spec.nb_comm, dtype=torch.int32, device=self.device)
                rc = self._lib.mdr_env_actor_sample_links(env._handle, C.byref(spec), C.byref(self._desc), C.c_void_p(msg.data_ptr()),
                                                          C.c_void_p(senders.data_ptr()) if senders is not None else None, *tail, stream)
            else:
                rc = self._lib.mdr_env_actor_sample(env._handle, C.byref(spec), C.byref(self._desc), *tail, stream)
        if rc == nat.MDR_ERR_UNSUPPORTED:
            raise NotImplementedError("observe -> act: " + self._lib.mdr_last_error(env._handle).decode())
        nat.check(self._lib, env._handle, rc, "mdr_env_actor_sample")
        return (action, a_prob, probs) if want_probs else (action, a_prob)

    def sample(self, obs: torch.Tensor, seed: int, step: int, want_probs: bool = False,
               action: Optional[torch.Tensor] = None, a_prob: Optional[torch.Tensor] = None,
               step_dev: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, ...]:
        """obs float32 on the device: rows [A, F] (contiguous) or feature planes - any [F, ...] tensor whose trailing
        dimensions are contiguous (e.g. ``env.obs_vector("planes")``: [F, E, N] with a padded plane stride), or the
        transposed view [A, F] of one -> (action uint8 [A], a_prob float32 [A][, probs [A, 2]]).  ``step_dev``: device
        int32 added to ``step`` inside the kernel (``env.device_time_index`` when the call is captured in a graph)."""
        F = self.num_state
        if obs.dtype != torch.float32 or obs.device != self.device:
            raise ValueError("obs must be a float32 tensor on %s" % (self.device,))
        if obs.dim() == 2 and obs.shape[1] == F and obs.is_contiguous():
            A, plane = obs.shape[0], 0
        else:
            planes = obs.t() if (obs.dim() == 2 and obs.shape[1] == F and obs.stride(1) >= obs.shape[0] and obs.stride(0) == 1) else obs
            if planes.shape[0] != F or (planes.dim() > 1 and not planes[0].is_contiguous()):
                raise ValueError("obs must be rows [A, %d] or feature planes [%d, ...]" % (F, F))
            A, plane = planes[0].numel(), planes.stride(0)
            if plane < A:
                raise ValueError("feature planes overlap")
        action = torch.empty(A, dtype=torch.uint8, device=self.device) if action is None else action
        a_prob = torch.empty(A, dtype=torch.float32, device=self.device) if a_prob is None else a_prob
        probs = torch.empty((A, 2), dtype=torch.float32, device=self.device) if want_probs else None
        with torch.cuda.device(self.device):
            if step_dev is not None and (step_dev.dtype != torch.int32 or step_dev.device != self.device):
                raise ValueError("step_dev must be an int32 tensor on the device (env.device_time_index)")
            rc = self._lib.mdr_actor_sample(C.byref(self._desc), C.c_void_p(obs.data_ptr()), plane, A, C.c_uint64(seed & (2 ** 64 - 1)),
                                            C.c_uint64(step & (2 ** 64 - 1)), C.c_void_p(step_dev.data_ptr()) if step_dev is not None else None,
                                            C.c_void_p(action.data_ptr()), C.c_void_p(a_prob.data_ptr()),
                                            C.c_void_p(probs.data_ptr()) if want_probs else None,
                                            C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        if rc != 0:
            raise RuntimeError("mdr_actor_sample failed: %s" % self._lib.mdr_status_string(rc).decode())
        return (action, a_prob, probs) if want_probs else (action, a_prob)
