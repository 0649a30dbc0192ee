"""Static communication-link tables: ClusterHouses.build_agent_comm_links (env/MA_DemandResponse.py:806-902).

Returns ``{agent_id: [sender ids]}`` for every mode whose links are fixed for an episode; ``random_sample``
re-draws its senders every step (env 976-983) and is returned as ``None`` (the dict adapter samples it per step).
"""
from __future__ import annotations

import random
from typing import Dict, List, Optional

COMM_MODES = ("neighbours", "closed_groups", "random_sample", "random_fixed", "neighbours_2D", "no_message")


def nb_comm(cluster_prop: dict) -> int:
    return int(min(cluster_prop["nb_agents_comm"], cluster_prop["nb_agents"] - 1))   # env 808-810


def random_fixed_table(n: int, c: int, seed: int, episode: int):
    """agents_comm_mode 'random_fixed' (env 849-854): every house keeps `c` distinct random senders among the others for the
    whole episode.  The reference draws them from Python's global `random` stream at every build_environment; here the table
    is a pure function of (seed, episode) - the same on every rank of a sharded run and in every view of one env (flat
    vector, dict messages, halo plan), re-drawn at every reset because the episode index moves.  int32 [n, c]."""
    import numpy as np
    if c <= 0 or n <= 1:
        return np.zeros((n, 0), dtype=np.int32)
    rng = np.random.default_rng([int(seed) & 0xFFFFFFFFFFFFFFFF, int(episode) & 0xFFFFFFFF, 0x6C696E6B])
    own = np.arange(n, dtype=np.int64)[:, None]
    if n - 1 <= 64 or c * c * 8 > n:          # small envs / dense tables: a permutation of the others per house
        picks = rng.permuted(np.broadcast_to(np.arange(n - 1, dtype=np.int64), (n, n - 1)), axis=1)[:, :c]
    else:                                     # large envs: independent draws, rows with a repeated sender re-drawn
        picks = rng.integers(0, n - 1, size=(n, c), dtype=np.int64)
        while True:
            srt = np.sort(picks, axis=1)
            bad = np.nonzero((srt[:, 1:] == srt[:, :-1]).any(axis=1))[0]
            if bad.size == 0:
                break
            picks[bad] = rng.integers(0, n - 1, size=(bad.size, c), dtype=np.int64)
    return (picks + (picks >= own)).astype(np.int32)      # index among the OTHER houses -> house id


def build_comm_links(cluster_prop: dict, seed_episode=None) -> Optional[Dict[int, List[int]]]:
    """`seed_episode` = (seed, episode): where 'random_fixed' takes its table from (`random_fixed_table`); without it the table
    is drawn from Python's global `random` stream as the reference does."""
    n = int(cluster_prop["nb_agents"])
    c = nb_comm(cluster_prop)
    mode = cluster_prop["agents_comm_mode"]
    ids = range(n)
    if mode == "neighbours":       # circular: floor(c/2) before, ceil(c/2) after (env 816-828)
        before, after = c // 2, c - c // 2
        return {i: [(i - before + j) % n for j in range(before)] + [(i + 1 + j) % n for j in range(after)] for i in ids}
    if mode == "closed_groups":    # env 830-844
        links = {}
        for i in ids:
            base = i - (i % (c + 1))
            if base + c <= n:
                group = [base + j for j in range(cluster_prop["nb_agents_comm"] + 1)]
            else:
                group = [n - c - 1 + j for j in range(c + 1)]
            group.remove(i)
            links[i] = group
        return links
    if mode == "random_sample":
        return None
    if mode == "random_fixed":     # env 849-854
        if seed_episode is not None:
            table = random_fixed_table(n, c, *seed_episode)
            return {i: [int(x) for x in table[i]] for i in ids}
        return {i: random.sample([j for j in ids if j != i], k=c) for i in ids}
    if mode == "neighbours_2D":    # env 856-890
        p2 = cluster_prop["agents_comm_parameters"]["neighbours_2D"]
        row, dist = p2["row_size"], p2["distance_comm"]
        if n % row != 0:
            raise ValueError("Neighbours 2D row_size must be a divisor of nb_agents")
        rows = n // row
        if dist >= (row + 1) // 2 or dist >= (rows + 1) // 2:
            raise ValueError("Neighbours 2D distance_comm ({}) must be strictly smaller than (row_size+1) / 2 ({}) "
                             "and (max_y+1) / 2 ({})".format(dist, (row + 1) // 2, (rows + 1) // 2))
        pattern = [(dx, dy) for dx in range(-dist, dist + 1) for dy in range(-dist, dist + 1)
                   if abs(dx) + abs(dy) <= dist and (dx, dy) != (0, 0)]
        return {i: [((i // row + dy) % rows) * row + (i % row + dx) % row for dx, dy in pattern] for i in ids}
    if mode == "no_message":
        return {i: [] for i in ids}
    raise ValueError("Cluster property: unknown agents_comm_mode '{}'.".format(mode))


def links_array(cluster_prop: dict, seed_episode=None):
    """The static link table as int32 [nb_agents, c] (None for random_sample).  'neighbours' is built with array
    arithmetic so that a 1,000,000-house env does not go through a Python dict."""
    import numpy as np
    n = int(cluster_prop["nb_agents"])
    c = nb_comm(cluster_prop)
    mode = cluster_prop["agents_comm_mode"]
    if mode == "random_sample":
        return None
    if mode == "random_fixed" and seed_episode is not None:
        return random_fixed_table(n, c, *seed_episode)
    if mode == "no_message" or c == 0 and mode == "neighbours":
        return np.zeros((n, 0), dtype=np.int32)
    if mode == "neighbours":
        before = c // 2
        offsets = np.concatenate([np.arange(-before, 0), np.arange(1, c - before + 1)])
        return ((np.arange(n, dtype=np.int64)[:, None] + offsets[None, :]) % n).astype(np.int32)
    links = build_comm_links(cluster_prop)
    return np.array([links[i] for i in range(n)], dtype=np.int32).reshape(n, -1)
