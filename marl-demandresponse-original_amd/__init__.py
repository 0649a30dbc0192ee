"""MI355X-native batched demand-response environment step (import name: ``mdr_amd``).

Only what the hot path needs lives here: ``csrc/`` (HIP kernels + the C ABI of include/mdr.h), the ctypes
binding, the config schema and the host-side mirrors of the reference's env interface.
"""
from .config import default_config, flatten_config, EnvSpec  # noqa: F401
from ._native import LIB_PATH, NativeLibraryMissing, load as load_native  # noqa: F401
from .build import build_native  # noqa: F401


def __getattr__(name):
    # torch is imported lazily so that `import mdr_amd` (config, build) works in tools that only need the schema
    if name in ("BatchedDemandResponseEnv", "OBS_COLUMNS"):
        from . import batched_env
        return getattr(batched_env, name)
    if name in ("sharding", "comm", "rollout", "policy", "metrics", "montecarlo"):      # submodules on first use (mdr_amd.sharding.house_shard ...)
        import importlib
        return importlib.import_module("." + name, __name__)
    if name == "BatchedMetrics":
        from .metrics import BatchedMetrics
        return BatchedMetrics
    if name == "MADemandResponseEnv":
        from .env import MADemandResponseEnv
        return MADemandResponseEnv
    raise AttributeError(name)
