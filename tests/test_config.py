"""mdr_amd.default_config() restates the env-related part of the reference's config.py: checked key by key
against a values-only snapshot taken from the reference (tests/golden/reference_env_config.json)."""
import copy

import pytest

import mdr_amd
from tests import golden_util as gu


def _walk(a, b, path=""):
    assert type(a) is type(b) or (isinstance(a, (int, float)) and isinstance(b, (int, float))), path
    if isinstance(a, dict):
        assert list(a.keys()) == list(b.keys()) or set(a.keys()) == set(b.keys()), path
        for k in a:
            _walk(a[k], b[k], path + "." + str(k))
    elif isinstance(a, list):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _walk(x, y, "%s[%d]" % (path, i))
    else:
        assert a == b, "%s: %r != %r" % (path, a, b)


def test_default_config_equals_reference_snapshot():
    _walk(mdr_amd.default_config(), gu.reference_env_config())
    _walk(gu.reference_env_config(), mdr_amd.default_config())


def test_flatten_matches_reference_norms_and_modes():
    cfg = mdr_amd.default_config()
    s = mdr_amd.flatten_config(cfg)        # the default base_power_mode is "interpolation" (config.py:326)
    assert s.base_power_mode == 1 and s.interp_update_period == 300 and s.interp_nb_agents == 100
    from mdr_amd.config import InterpolationGridMissing, load_interp_grid
    with pytest.raises(InterpolationGridMissing):   # the grid blob is not shipped by the reference
        load_interp_grid(s.interp_paths)
    cfg["default_env_prop"]["power_grid_prop"]["base_power_mode"] = "cubic"
    with pytest.raises(ValueError):
        mdr_amd.flatten_config(cfg)
    cfg["default_env_prop"]["power_grid_prop"]["base_power_mode"] = "constant"
    s = mdr_amd.flatten_config(cfg)
    assert s.norm_temp_penalty == 1.0 and s.norm_sig_penalty == 3515625.0
    assert s.signal_mode_name == "perlin" and s.perlin_nb_octaves == 5 and s.perlin_period == 400
    assert s.capacity_list == [15000.0] and s.lockout_duration == 40 and s.time_step == 4
    for bad, exc in ((("power_grid_prop", "signal_mode", "triangle"), ValueError),
                     (("reward_prop", "temp_penalty_mode", "huber"), ValueError),
                     (("reward_prop", "sig_penalty_mode", "L1"), ValueError),
                     (("cluster_prop", "agents_comm_mode", "telepathy"), ValueError)):
        c = copy.deepcopy(cfg)
        c["default_env_prop"][bad[0]][bad[1]] = bad[2]
        with pytest.raises(exc):
            mdr_amd.flatten_config(c)
    c = copy.deepcopy(cfg)
    c["default_env_prop"]["start_datetime_mode"] = "sometimes"
    with pytest.raises(ValueError):
        mdr_amd.flatten_config(c)
    with pytest.raises(KeyError):          # the reference's test=True path fails the same way (utils.py:674)
        mdr_amd.flatten_config(cfg, test=True)
