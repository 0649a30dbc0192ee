"""The per-env time tables (outdoor temperature env/MA_DemandResponse.py:1057-1081, solar gain utils.py:1277-1350, regulation
signal env 1236-1316) built by runs of rows per thread - calendar / sinusoid / solar polynomial once per minute, Perlin gradients
once per lattice cell - or by tiles of 64 envs per workgroup (k_fill_tables_tile: windows of at most 8 minutes in batches of
>= 4096 envs) must equal the one-thread-per-entry tables bit for bit.  A batch of a few envs takes the per-entry kernel, a big
batch the runs or the tiles (mdr_kernels.hip launch_tables); env_offset puts the small batch on the same global envs."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfg(signal, temp_mode, dt):
    import mdr_amd
    cfg = mdr_amd.default_config()
    env = cfg["default_env_prop"]
    env["cluster_prop"]["nb_agents"] = 2
    env["cluster_prop"]["temp_mode"] = temp_mode
    env["time_step"] = dt
    env["power_grid_prop"]["base_power_mode"] = "constant"
    env["power_grid_prop"]["signal_mode"] = signal
    env["power_grid_prop"]["artificial_signal_ratio_range"] = 2
    env["start_datetime_mode"] = "random"
    cfg["default_house_prop"]["solar_gain_bool"] = True
    cfg["noise_hvac_prop"]["noise_mode"] = "big_noise"
    return cfg


@pytest.mark.parametrize("signal,temp_mode,dt,E", [
    ("perlin", "noisy_sinusoidal_heatwave", 4, 262144),      # runs of a whole window (65 rows)
    ("perlin", "noisy_sinusoidal_hot", 60, 65536),           # runs of 16 rows, a new minute at every row
    ("sinusoidals", "sinusoidal_hot", 7, 16384),             # runs of 4 rows
    ("regular_steps", "noisy_sinusoidal_cold", 4, 131072),
    ("flat", "constant", 4, 40000),
    ("sinusoidals", "noisy_sinusoidal_heatwave", 4, 100000),   # the tile kernel without gradient slots
    ("regular_steps", "sinusoidal_hot", 5, 65536),
])
def test_tables_by_runs_equal_tables_by_entry(signal, temp_mode, dt, E):
    import mdr_amd
    cfg = _cfg(signal, temp_mode, dt)
    big = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=77, table_steps=64)
    big.reset(episode=3)
    smalls = []
    for off in (0, 1001, E - 8):
        small = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=8, device="cuda:0", seed=77, table_steps=64, env_offset=off)
        small.reset(episode=3)
        smalls.append((off, small))
    for rounds in range(4):       # the first tables, then three refills further into the episode (incl. midnight for dt = 60)
        for off, small in smalls:
            for name in ("tab_od", "tab_solar", "tab_signal", "tab_abs_noise"):
                assert torch.equal(small.table(name), big.table(name)[:, off:off + 8]), (rounds, off, name)
        big.rollout(65)
        for _, small in smalls:
            small.rollout(65)
        assert big.cursor() == smalls[0][1].cursor()


def test_tile_tables_across_midnight_and_at_the_batch_edge():
    """k_fill_tables_tile (a workgroup per 64 envs, every lattice gradient of the window once): envs whose window runs across
    midnight - the seconds-of-day and with them the lattice cells start again, those rows draw their gradients directly - and the
    last, partly filled tile of a batch whose size is not a multiple of 64, against the per-entry kernel."""
    import mdr_amd
    E = 100_003
    cfg = _cfg("perlin", "noisy_sinusoidal_heatwave", 4)
    big = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=5, table_steps=64)
    big.reset(episode=1)
    sod = big.t["t0"] % 86400
    checked = 0
    for rounds in range(2):
        first = sod + big.cursor()[1] * 4                    # seconds of day at row 0 of the current window
        crossing = torch.nonzero((first % 86400) > 86400 - 200).flatten().cpu().tolist()
        assert len(crossing) >= 3
        for off in crossing[:4] + [E - 8, E - 67]:
            off = min(off, E - 8)
            small = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=8, device="cuda:0", seed=5, table_steps=64, env_offset=off)
            small.reset(episode=1)
            if rounds:
                small.rollout(65)
            for name in ("tab_od", "tab_solar", "tab_signal", "tab_abs_noise"):
                assert torch.equal(small.table(name), big.table(name)[:, off:off + 8]), (rounds, off, name)
            assert small.cursor() == big.cursor()
            checked += 1
        big.rollout(65)
    assert checked == 12


def test_tile_tables_take_an_outdoor_temperature_table():
    """mdr_env_set_od_table rows (the reference's recorded draws, env 1057-1081 replaced) through the tile kernel: rows the table
    covers come from it, the rows behind it from the model, as the per-entry kernel has them."""
    import mdr_amd
    E = 8192
    cfg = _cfg("perlin", "noisy_sinusoidal_heatwave", 4)
    g = torch.Generator().manual_seed(3)
    od = 20.0 + 10.0 * torch.rand((100, E), generator=g, dtype=torch.float64)      # covers the first window and part of the second
    big = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=E, device="cuda:0", seed=9, table_steps=64)
    big.set_od_table(od)
    big.reset(episode=0)
    off = 4100
    small = mdr_amd.BatchedDemandResponseEnv(cfg, nb_envs=8, device="cuda:0", seed=9, table_steps=64, env_offset=off)
    small.set_od_table(od[:, off:off + 8])
    small.reset(episode=0)
    for rounds in range(3):
        for name in ("tab_od", "tab_solar", "tab_signal", "tab_abs_noise"):
            assert torch.equal(small.table(name), big.table(name)[:, off:off + 8]), (rounds, name)
        big.rollout(65)
        small.rollout(65)
