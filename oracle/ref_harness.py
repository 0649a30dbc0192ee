"""Import the reference environment read-only from /root/reference  --  TEST INFRASTRUCTURE ONLY.

Used in the BUILD CONTAINER only, by tests/golden/make_golden.py (fixture generation) and by
the optional live cross-check in tests/test_oracle_vs_reference_live.py.  /root/reference does not
exist on the GPU box; everything here is skipped there and only the committed fixtures travel.

The reference imports third-party modules that are absent from this image.  None of them carries
arithmetic on the step path except ``perlin_noise``:

  gym, wandb, cvxpy, ray.*    -> empty stand-in modules (only ``MultiAgentEnv`` is used, as an
                                 empty base class: env/MA_DemandResponse.py:13,37,82)
  perlin_noise.PerlinNoise    -> stand-in whose ``noise(x)`` evaluates THIS BUILD's lattice noise
                                 (oracle.mdr_oracle.lattice_noise_1d) with a gradient function the
                                 caller installs via ``set_perlin_gradient``.  That pins the wiring of
                                 utils.Perlin / PowerGrid.step around the noise, not the third-party
                                 lattice values ("parity unpinned" for those, see mdr_oracle.py).

Nothing is copied from the reference; it is imported from where it lies and never written to.
"""
from __future__ import annotations

import os
import sys
import time
import types

REFERENCE_ROOT = "/root/reference"

_perlin_gradient = [None]


def set_perlin_gradient(fn):
    """fn(lattice_index: np.ndarray[int64]) -> gradient in (-1, 1); None -> noise() returns 0."""
    _perlin_gradient[0] = fn


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "env", "MA_DemandResponse.py"))


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _install_standins():
    import numpy as np
    from oracle.mdr_oracle import lattice_noise_1d

    for name in ("gym", "wandb", "cvxpy", "ray", "ray.rllib", "ray.rllib.env", "ray.rllib.utils"):
        if name not in sys.modules:
            _module(name)
    _module("ray.rllib.env.multi_agent_env", MultiAgentEnv=type("MultiAgentEnv", (), {}))
    _module("ray.rllib.utils.annotations", override=lambda cls: (lambda f: f), PublicAPI=lambda f: f)
    _module("ray.rllib.utils.typing", MultiAgentDict=dict, AgentID=int)

    class PerlinNoise:
        def __init__(self, octaves=1, seed=None):
            self.octaves = octaves
            self.seed = seed

        def noise(self, x):
            fn = _perlin_gradient[0]
            if fn is None:
                return 0.0
            return float(lattice_noise_1d(np.float64(x) * self.octaves, fn))

        __call__ = noise

    _module("perlin_noise", PerlinNoise=PerlinNoise)


_loaded = {}


def _greedy_myopic():
    """agents/greedy_myopic_controller.py (needs pandas); its module-level memory is reset so that a fresh set of controller objects
    starts at time step 0 again."""
    import agents.greedy_myopic_controller as gm

    def make(agent_properties, config_dict, num_state=None):
        if agent_properties["id"] == 0:
            gm.global_myopic_memory[0] = gm.global_myopic_memory[1] = None
        return gm.GreedyMyopic(agent_properties, config_dict, num_state)
    return make


def load_reference():
    """Returns a namespace dict: MADemandResponseEnv, config_dict, utils and the rule-based controllers of agents/bangbang_controllers.py."""
    if _loaded:
        return _loaded
    if not available():
        raise RuntimeError("reference not present at " + REFERENCE_ROOT)
    os.environ["TZ"] = "UTC"          # PowerGrid.step uses time.mktime (env/MA_DemandResponse.py:1297)
    time.tzset()
    sys.dont_write_bytecode = True    # the reference tree is read-only
    _install_standins()
    os.chdir(REFERENCE_ROOT)          # the env appends "./monteCarlo" to sys.path (env/MA_DemandResponse.py:27)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from env.MA_DemandResponse import MADemandResponseEnv, HVAC  # noqa: E402
    from agents.bangbang_controllers import (AlwaysOnController, BangBangController, BasicController,   # noqa: E402
                                             DeadbandBangBangController)
    from config import config_dict                                # noqa: E402
    import utils as ref_utils                                     # noqa: E402
    _loaded.update(MADemandResponseEnv=MADemandResponseEnv, HVAC=HVAC, config_dict=config_dict,
                   utils=ref_utils, BangBangController=BangBangController, DeadbandBangBangController=DeadbandBangBangController,
                   BasicController=BasicController, AlwaysOnController=AlwaysOnController, GreedyMyopic=_greedy_myopic())
    return _loaded
