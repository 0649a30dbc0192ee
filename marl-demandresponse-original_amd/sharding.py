"""Partitioning helpers for the two multi-GPU layouts.

* env replicas  - envs are independent (no state is shared between MADemandResponseEnv instances):
                  rank r owns a contiguous block of global env indices, no collective on the data path.
* sharded houses - one env's houses split into contiguous ranges; the ranks exchange only the per-env
                  aggregates (cluster_hvac_power, penalty sum / max, max_power) each step.
"""
from __future__ import annotations

from typing import Tuple


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
