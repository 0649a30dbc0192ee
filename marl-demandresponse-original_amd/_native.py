"""ctypes binding of libmdr_hip.so (include/mdr.h).  Loading fails loudly: there is no CPU fallback.

The structures below mirror ``mdr_config_t`` / ``mdr_buffers_t`` / ``mdr_episode_t`` field for field;
tests/test_abi.py parses include/mdr.h and checks names and order, and the library itself checks the
sizes (``struct_size``).
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MDR_HIP_LIB") or os.path.join(HERE, "csrc", "libmdr_hip.so")   # MDR_HIP_LIB: experiment builds

MDR_ABI_VERSION = 4
MDR_MAX_SINUSOIDS = 8
MDR_MAX_CAPACITIES = 16
MDR_OBS_COLUMNS = 7
MDR_MAX_SHARDS = 8

MDR_OK, MDR_ERR_INVALID, MDR_ERR_UNBOUND, MDR_ERR_HIP, MDR_ERR_UNSUPPORTED = 0, -1, -2, -3, -4
ACTIONS_EXTERNAL, ACTIONS_BANGBANG, ACTIONS_DEADBAND, ACTIONS_ALWAYS_ON = 0, 1, 2, 3
CONTROLLERS = {"bangbang": ACTIONS_BANGBANG, "deadband": ACTIONS_DEADBAND, "basic": ACTIONS_DEADBAND, "always_on": ACTIONS_ALWAYS_ON}

_f32p, _i32p, _u8p, _f64p, _i64p = (C.c_void_p,) * 5  # device pointers travel as plain addresses


class MdrConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("nb_envs", C.c_int32), ("nb_houses", C.c_int32),
        ("nb_houses_total", C.c_int64), ("env_offset", C.c_int64), ("house_offset", C.c_int64),
        ("time_step", C.c_int32), ("table_steps", C.c_int32), ("temp_ref", C.c_double),
        ("init_air_temp", C.c_double), ("init_mass_temp", C.c_double), ("target_temp", C.c_double),
        ("deadband", C.c_double),
        ("Ua", C.c_double), ("Cm", C.c_double), ("Ca", C.c_double), ("Hm", C.c_double),
        ("window_area", C.c_double), ("shading_coeff", C.c_double), ("solar_gain", C.c_int32),
        ("lockout_duration", C.c_int32), ("lockout_noise", C.c_int32),
        ("COP", C.c_double), ("cooling_capacity", C.c_double), ("latent_cooling_fraction", C.c_double),
        ("std_start_temp", C.c_double), ("std_target_temp", C.c_double),
        ("factor_thermo_low", C.c_double), ("factor_thermo_high", C.c_double),
        ("nb_capacities", C.c_int32), ("start_random", C.c_int32),
        ("capacity_list", C.c_double * MDR_MAX_CAPACITIES),
        ("start_epoch", C.c_int64),
        ("day_temp", C.c_double), ("night_temp", C.c_double), ("temp_std", C.c_double),
        ("random_phase_offset", C.c_int32), ("signal_mode", C.c_int32),
        ("avg_power_per_hvac", C.c_double),
        ("nb_sinusoids", C.c_int32), ("perlin_nb_octaves", C.c_int32),
        ("sin_periods", C.c_double * MDR_MAX_SINUSOIDS), ("sin_amplitude_ratios", C.c_double * MDR_MAX_SINUSOIDS),
        ("steps_amplitude_per_hvac", C.c_double), ("steps_period", C.c_double),
        ("perlin_amplitude", C.c_double), ("perlin_octaves_step", C.c_double), ("perlin_period", C.c_double),
        ("artificial_ratio", C.c_double), ("artificial_signal_ratio_range", C.c_double),
        ("alpha_temp", C.c_double), ("alpha_sig", C.c_double),
        ("norm_temp_penalty", C.c_double), ("norm_sig_penalty", C.c_double),
        ("penalty_mode", C.c_int32), ("base_power_mode", C.c_int32),
        ("mix_ind_L2", C.c_double), ("mix_common_L2", C.c_double), ("mix_common_max", C.c_double),
        ("obs_power_norm", C.c_double),
    ]


class MdrBuffers(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("reserved0", C.c_uint32),
        ("Ta", _f32p), ("Tm", _f32p), ("sso", _i32p), ("flags", _u8p),
        ("k01", _f32p), ("s0", _f32p), ("k10", _f32p), ("s1", _f32p),
        ("inv_Ua", _f32p), ("Q_hvac", _f32p), ("P_max", _f32p),
        ("target", _f32p), ("deadband", _f32p), ("lockout", _i32p),
        ("Ua", _f32p), ("Cm", _f32p), ("Ca", _f32p), ("Hm", _f32p),
        ("capacity", _f32p), ("COP", _f32p), ("latent", _f32p),
        ("reward", _f32p), ("obs", _f32p),
        ("t0", _i64p), ("phase", _f64p), ("ratio", _f64p), ("max_power", _f64p),
        ("P", _f64p), ("tot_sum", _f64p), ("tot_max", _f64p),
        ("tab_od", _f32p), ("tab_solar", _f32p), ("tab_signal", _f64p),
        ("partials", _f64p), ("base_power", _f64p), ("cursor", C.c_void_p), ("tab_abs_noise", _f64p),
        ("pen_stash", _f32p),
        ("tab2_od", _f32p), ("tab2_solar", _f32p), ("tab2_signal", _f64p), ("tab2_abs_noise", _f64p),
    ]


class MdrEpisode(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("reserved0", C.c_uint32),
        ("Ta", _f64p), ("Tm", _f64p), ("target", _f64p), ("deadband", _f64p),
        ("Ua", _f64p), ("Cm", _f64p), ("Ca", _f64p), ("Hm", _f64p),
        ("capacity", _f64p), ("COP", _f64p), ("latent", _f64p),
        ("lockout", _i64p), ("t0", _i64p), ("phase", _f64p), ("ratio", _f64p),
    ]


class MdrObsSpec(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("layout", C.c_int32),
        ("state_hour", C.c_int32), ("state_day", C.c_int32), ("state_solar_gain", C.c_int32),
        ("state_thermal", C.c_int32), ("state_hvac", C.c_int32),
        ("message_thermal", C.c_int32), ("message_hvac", C.c_int32),
        ("nb_comm", C.c_int32),
        ("links", _i32p),
        ("random_links", C.c_int32), ("reserved0", C.c_int32),
        ("comm_defect_prob", C.c_double),
        ("out_plane_stride", C.c_int64),
        ("def_Ua", C.c_double), ("def_Cm", C.c_double), ("def_Ca", C.c_double), ("def_Hm", C.c_double),
        ("def_COP", C.c_double), ("def_capacity", C.c_double), ("def_latent", C.c_double), ("norm_reg_sig", C.c_double),
    ]


MDR_INTERP_AXES, MDR_INTERP_MAX_AXIS = 10, 16


class MdrInterpGrid(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("update_period", C.c_int32), ("nb_agents", C.c_int32), ("reserved0", C.c_int32),
        ("values", _f64p),
        ("dims", C.c_int32 * MDR_INTERP_AXES),
        ("axes", (C.c_double * MDR_INTERP_MAX_AXIS) * MDR_INTERP_AXES),
    ]


class MdrRolloutOut(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("reserved0", C.c_uint32),
        ("power_trace", _f64p), ("reward_sum", _f32p), ("sq_temp_error_sum", _f64p), ("sq_signal_error_sum", _f64p),
    ]


class MdrMailbox(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("world", C.c_int32), ("rank", C.c_int32), ("records_per_env", C.c_int32),
        ("records", C.c_int32 * MDR_MAX_SHARDS),
        ("system_scope", C.c_int32), ("co_resident", C.c_int32), ("spin_limit", C.c_uint32), ("reserved0", C.c_uint32),
        ("boxes", C.c_void_p * MDR_MAX_SHARDS),
    ]


OBS_PLANES, OBS_ROWS = 0, 1

EXPORTS = (
    "mdr_abi_version", "mdr_status_string", "mdr_last_error", "mdr_partials_per_env", "mdr_env_partial_records",
    "mdr_env_create", "mdr_env_destroy", "mdr_env_bind", "mdr_env_reset", "mdr_env_load_episode",
    "mdr_env_set_od_table", "mdr_env_set_interp_grid", "mdr_env_begin_episode", "mdr_env_refresh_obs", "mdr_env_step", "mdr_env_rollout", "mdr_env_rollout_fused",
    "mdr_env_step_begin", "mdr_env_step_end", "mdr_env_step_end_gathered", "mdr_env_step_begin_records", "mdr_env_step_end_records", "mdr_env_step_end_begin_records",
    "mdr_env_interp_due", "mdr_env_interp_local", "mdr_env_interp_apply", "mdr_obs_vector_length", "mdr_env_obs_vector",
    "mdr_obs_message_fields", "mdr_env_obs_messages", "mdr_env_obs_vector_ext", "mdr_env_comm_draws",
    "mdr_env_graph_room", "mdr_env_graph_replayed", "mdr_env_pack", "mdr_env_cursor", "mdr_env_set_cursor", "mdr_env_active_tables",
    "mdr_env_set_controller", "mdr_env_greedy_myopic_actions", "mdr_mailbox_bytes", "mdr_persist_records", "mdr_env_rollout_persistent",
    "mdr_mailbox_alloc", "mdr_mailbox_free", "mdr_mailbox_export", "mdr_mailbox_open", "mdr_mailbox_close", "mdr_mailbox_peek",
    # include/mdr_policy.h
    "mdr_actor_steps1", "mdr_actor_steps1_order", "mdr_actor_steps2", "mdr_actor_frag1_floats", "mdr_actor_frag2_floats", "mdr_actor_sample", "mdr_env_actor_sample", "mdr_env_actor_sample_links",
    "mdr_discounted_returns",
)

_lib = None


class NativeLibraryMissing(RuntimeError):
    pass


def load():
    """dlopen the HIP library (once).  Raises NativeLibraryMissing if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise NativeLibraryMissing(
            "HIP library not built: %s is missing. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64, libmdr_hip.so is linked against the
    # ones under /opt/rocm (same SONAMEs).  Whoever is loaded first is used by both; loaded the other way round - this library
    # first, torch afterwards - torch's HSA runtime comes up beside /opt/rocm's and the launches of this library fail with "no
    # ROCm-capable device is detected".  The host side owns its memory through torch anyway: bring torch's runtime in first.
    try:
        import torch  # noqa: F401
    except ImportError:      # a torch-free host (tools/c_abi_client.c is the C example) uses /opt/rocm's runtime alone
        pass
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, u32, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64
    sig = {
        "mdr_abi_version": (C.c_int, []),
        "mdr_status_string": (C.c_char_p, [C.c_int]),
        "mdr_last_error": (C.c_char_p, [vp]),
        "mdr_partials_per_env": (i64, [i32]),
        "mdr_env_partial_records": (i64, [vp]),
        "mdr_env_create": (C.c_int, [C.POINTER(MdrConfig), C.POINTER(vp)]),
        "mdr_env_destroy": (C.c_int, [vp]),
        "mdr_env_bind": (C.c_int, [vp, C.POINTER(MdrBuffers)]),
        "mdr_env_reset": (C.c_int, [vp, u64, u32, vp]),
        "mdr_env_load_episode": (C.c_int, [vp, C.POINTER(MdrEpisode), u64, u32, vp]),
        "mdr_env_set_od_table": (C.c_int, [vp, vp, i64]),
        "mdr_env_set_interp_grid": (C.c_int, [vp, C.POINTER(MdrInterpGrid)]),
        "mdr_env_begin_episode": (C.c_int, [vp, vp]),
        "mdr_env_refresh_obs": (C.c_int, [vp, vp]),
        "mdr_env_step": (C.c_int, [vp, vp, C.c_int, vp]),
        "mdr_env_rollout": (C.c_int, [vp, vp, C.c_int, i32, vp]),
        "mdr_env_rollout_fused": (C.c_int, [vp, vp, i32, C.POINTER(MdrRolloutOut), vp]),
        "mdr_env_set_controller": (C.c_int, [vp, C.c_int]),
        "mdr_env_greedy_myopic_actions": (C.c_int, [vp, vp, vp]),
        "mdr_env_step_begin": (C.c_int, [vp, vp, C.c_int, vp]),
        "mdr_env_step_end": (C.c_int, [vp, vp]),
        "mdr_env_step_end_gathered": (C.c_int, [vp, vp, i32, vp]),
        "mdr_env_step_begin_records": (C.c_int, [vp, vp, C.c_int, i32, vp]),
        "mdr_env_step_end_records": (C.c_int, [vp, vp, i32, vp]),
        "mdr_env_step_end_begin_records": (C.c_int, [vp, vp, i32, vp, C.c_int, vp]),
        "mdr_env_interp_due": (C.c_int, [vp]),
        "mdr_env_interp_local": (C.c_int, [vp, vp]),
        "mdr_env_interp_apply": (C.c_int, [vp, vp]),
        "mdr_obs_vector_length": (i32, [C.POINTER(MdrObsSpec)]),
        "mdr_env_obs_vector": (C.c_int, [vp, C.POINTER(MdrObsSpec), vp, vp]),
        "mdr_obs_message_fields": (i32, [C.POINTER(MdrObsSpec)]),
        "mdr_env_obs_messages": (C.c_int, [vp, C.POINTER(MdrObsSpec), vp, i64, vp]),
        "mdr_env_obs_vector_ext": (C.c_int, [vp, C.POINTER(MdrObsSpec), vp, i64, vp, vp]),
        "mdr_env_comm_draws": (C.c_int, [vp, C.POINTER(MdrObsSpec), vp, vp, vp]),
        "mdr_env_cursor": (C.c_int, [vp, C.POINTER(i64), C.POINTER(i64)]),
        "mdr_env_set_cursor": (C.c_int, [vp, u64, u32, i64, i64]),
        "mdr_env_active_tables": (C.c_int, [vp]),
        "mdr_actor_steps1": (i64, [i32, i32]),
        "mdr_actor_steps1_order": (i64, [i32, i32, i32]),
        "mdr_actor_steps2": (i64, [i32, i32]),
        "mdr_actor_frag1_floats": (i64, [i32, i32]),
        "mdr_actor_frag2_floats": (i64, [i32, i32]),
        "mdr_actor_sample": (C.c_int, [vp, vp, i64, i64, u64, u64, vp, vp, vp, vp, vp]),
        "mdr_env_actor_sample": (C.c_int, [vp, C.POINTER(MdrObsSpec), vp, u64, u64, vp, vp, vp, vp, vp, vp]),
        "mdr_env_actor_sample_links": (C.c_int, [vp, C.POINTER(MdrObsSpec), vp, vp, vp, u64, u64, vp, vp, vp, vp, vp, vp]),
        "mdr_discounted_returns": (C.c_int, [vp, vp, vp, C.c_float, i32, i64, vp, vp]),
        "mdr_env_pack": (C.c_int, [vp, i32, vp, vp]),
        "mdr_env_graph_room": (i64, [vp]),
        "mdr_env_graph_replayed": (C.c_int, [vp, i64, vp]),
        "mdr_mailbox_bytes": (i64, [i32, i32, i32]),
        "mdr_persist_records": (i64, [i32]),
        "mdr_env_rollout_persistent": (C.c_int, [vp, vp, i32, C.POINTER(MdrRolloutOut), C.POINTER(MdrMailbox), vp]),
        "mdr_mailbox_alloc": (C.c_int, [i64, i32, C.POINTER(vp)]),
        "mdr_mailbox_free": (C.c_int, [vp]),
        "mdr_mailbox_export": (C.c_int, [vp, C.c_char_p]),
        "mdr_mailbox_open": (C.c_int, [C.c_char_p, C.POINTER(vp)]),
        "mdr_mailbox_close": (C.c_int, [vp]),
        "mdr_mailbox_peek": (C.c_int, [vp, C.POINTER(u64)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.mdr_abi_version() != MDR_ABI_VERSION:
        raise RuntimeError("libmdr_hip.so ABI version %d, binding expects %d" % (lib.mdr_abi_version(), MDR_ABI_VERSION))
    _lib = lib
    return lib


def check(lib, handle, rc, what):
    """Map a status code to the exception the reference would raise for the same condition."""
    if rc == MDR_OK:
        return
    msg = lib.mdr_last_error(handle).decode() if handle else ""
    text = "%s: %s%s" % (what, lib.mdr_status_string(rc).decode(), (" - " + msg) if msg else "")
    if rc == MDR_ERR_INVALID:
        raise ValueError(text)
    raise RuntimeError(text)
