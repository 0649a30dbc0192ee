"""ctypes front end of oracle/mdr_oracle_c.c  --  TEST INFRASTRUCTURE ONLY.

``CPort(oracle_env)`` shares an OracleEnv's episode (parameters, start state) and steps the per-house arithmetic
in C; the per-env time functions (outdoor temperature, solar gain, regulation signal) still come from the NumPy
oracle.  Used by tests/test_oracle_c.py (golden replay) and as bench.py's compiled single-core CPU figure.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libmdr_oracle_c.so")
SANITIZED = os.environ.get("MDR_ORACLE_C_SANITIZED", "") not in ("", "0")   # load the ASan + UBSan build (needs libasan preloaded)


class _Cfg(C.Structure):
    _fields_ = [("nb_envs", C.c_int64), ("nb_houses", C.c_int64), ("dt", C.c_double),
                ("alpha_temp", C.c_double), ("alpha_sig", C.c_double), ("norm_temp", C.c_double), ("norm_sig", C.c_double),
                ("penalty_mode", C.c_int32), ("mix_ind", C.c_double), ("mix_common", C.c_double), ("mix_max", C.c_double)]


class _Buf(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("Ta", "Tm", "sso", "on", "lock", "Ua", "Cm", "Ca", "Hm", "capacity", "COP",
                                          "latent", "target", "deadband", "lockout", "reward", "P")]


def build():
    if SANITIZED:
        subprocess.run(["make", "-s", "-C", HERE, "asan"], check=True)
        return os.path.join(HERE, "libmdr_oracle_c_asan.so")
    if not os.path.isfile(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, "mdr_oracle_c.c")):
        subprocess.run(["make", "-s", "-C", HERE], check=True)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.mdrc_step.argtypes = [C.POINTER(_Cfg), C.POINTER(_Buf), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.mdrc_bangbang.argtypes = [C.POINTER(_Cfg), C.POINTER(_Buf), C.c_void_p]
    return _lib


class CPort:
    def __init__(self, ora):
        """ora: an OracleEnv right after reset()/load_episode() (its own state is advanced alongside for the time functions)."""
        from oracle.mdr_oracle import PENALTY_MODES
        self.ora = ora
        s = ora.spec
        self.cfg = _Cfg(ora.E, ora.N, float(s.dt), s.alpha_temp, s.alpha_sig, ora.norm_temp_pen, ora.norm_sig_pen,
                        PENALTY_MODES.index(s.penalty_mode), *[float(x) for x in s.mix])
        c = lambda a, dt: np.ascontiguousarray(a, dtype=dt)
        self.a = dict(Ta=c(ora.Ta, np.float64), Tm=c(ora.Tm, np.float64), sso=c(ora.sso, np.int64),
                      on=c(ora.on, np.uint8), lock=c(ora.lock, np.uint8),
                      Ua=c(ora.Ua, np.float64), Cm=c(ora.Cm, np.float64), Ca=c(ora.Ca, np.float64), Hm=c(ora.Hm, np.float64),
                      capacity=c(ora.capacity, np.float64), COP=c(ora.COP, np.float64), latent=c(ora.latent, np.float64),
                      target=c(ora.target, np.float64), deadband=c(ora.deadband, np.float64), lockout=c(ora.lockout, np.int64),
                      reward=np.zeros((ora.E, ora.N)), P=np.zeros(ora.E))
        self.buf = _Buf(**{k: v.ctypes.data for k, v in self.a.items()})
        self.actions = np.zeros((ora.E, ora.N), dtype=np.uint8)

    def step_arrays(self, actions, od_old, solar, sig_old):
        act = np.ascontiguousarray(actions, dtype=np.uint8)
        od = np.ascontiguousarray(od_old, dtype=np.float64)
        so = np.ascontiguousarray(solar, dtype=np.float64)
        sg = np.ascontiguousarray(sig_old, dtype=np.float64)
        lib().mdrc_step(C.byref(self.cfg), C.byref(self.buf), act.ctypes.data, od.ctypes.data, so.ctypes.data, sg.ctypes.data)

    def bangbang(self):
        lib().mdrc_bangbang(C.byref(self.cfg), C.byref(self.buf), self.actions.ctypes.data)
        return self.actions


def time_baseline(config, nb_envs=4, seconds=3.0, table=256):
    """house-steps/s of the C port on one core, bang-bang closed loop, per-env time functions precomputed
    (as the HIP path does with its time tables) so that only the per-house arithmetic is timed."""
    from oracle.mdr_oracle import OracleEnv
    ora = OracleEnv(config, nb_envs=nb_envs).reset(seed=1, episode=0)
    port = CPort(ora)
    rng = np.random.default_rng(0)
    od = 30.0 + rng.normal(0, 0.5, (table, nb_envs))
    solar = np.abs(rng.normal(300, 100, (table, nb_envs)))
    sig = ora.S[None, :] * (1 + 0.2 * rng.normal(0, 1, (table, nb_envs)))
    steps, t0 = 0, time.perf_counter()
    while True:
        r = steps % table
        port.step_arrays(port.bangbang(), od[r], solar[r], sig[r])
        steps += 1
        if steps % 8 == 0 and time.perf_counter() - t0 >= seconds:
            break
    el = time.perf_counter() - t0
    return nb_envs * ora.N * steps / el, steps, el
