"""Restatement of keras-rl's replay memory (SequentialMemory.append / sample, sample_batch_indexes).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED at source level: the reference uses the
un-vendored fork github.com/R-Sweke/keras-rl (call site
/root/reference/cluster_scripts/d5_dp/0.001/Single_Point_Training_Script.py:109
``SequentialMemory(limit=buffer_size, window_length=1)``, pickled whole at :156-157); what is restated here is the
published upstream keras-rl 0.4.2 ``rl/memory.py`` algorithm, written from its description:

  RingBuffer(maxlen)       append drops the oldest element once full; index 0 is always the OLDEST element kept.
  append(o, a, r, term)    one entry per agent step: the observation the agent acted on, the action, and the reward /
                           terminal flag that action earned (DQNAgent.backward appends after env.step).
  sample_batch_indexes     ``high - low >= size``: ``random.sample(range(low, high), size)`` (distinct);
                           else a warning and integers drawn uniformly WITH replacement from [low, high - 1].
  sample(batch_size)       needs nb_entries >= window_length + 2.  idx = sample_batch_indexes(window_length,
                           nb_entries - 1, batch_size) + 1.  For each idx: while terminals[idx - 2]: idx = one fresh
                           draw from [window_length + 1, nb_entries).  state0 = observations[idx - 1] (window of older
                           ones in front, cut at an episode boundary and zero-padded); action / reward / terminal1 are
                           entry idx - 1's; state1 = state0 shifted by one with observations[idx] appended.
  Consequences for window_length = 1: the transition used is entry idx - 1 in [1, nb_entries - 2] -- never the newest
  entry (its successor observation is not stored yet), never entry 0 (the flag before it is unknown), never an entry
  whose predecessor is terminal (that entry holds the terminal observation of the finished episode).

The device sampler (csrc/common.h dq_replay_row) draws rows from a ring shared by N lattices -- first draws distinct (a keyed
permutation of the candidate rows), redraws independent, as above; `device_replay_rows` restates it bit for bit, and the GPU tests
compare its SUPPORT AND VALIDITY with `valid_transitions` below, per lattice.
"""
import random
import warnings

import numpy as np


class RingBuffer:
    def __init__(self, maxlen):
        self.maxlen, self.start, self.length = int(maxlen), 0, 0
        self.data = [None] * self.maxlen

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        if i < 0 or i >= self.length:
            raise KeyError(i)
        return self.data[(self.start + i) % self.maxlen]

    def append(self, v):
        if self.length < self.maxlen:
            self.length += 1
        else:
            self.start = (self.start + 1) % self.maxlen      # full: the oldest element falls out
        self.data[(self.start + self.length - 1) % self.maxlen] = v


def sample_batch_indexes(low, high, size, rng=random, np_rng=np.random):
    """`size` integers from [low, high): distinct when the range is large enough, else with replacement (and a warning)."""
    if high - low >= size:
        return list(rng.sample(range(low, high), size))
    warnings.warn("Not enough entries to sample without replacement")
    return [int(x) for x in np_rng.randint(low, high, size=size)]


class SequentialMemory:
    def __init__(self, limit, window_length=1):
        self.limit, self.window_length = int(limit), int(window_length)
        self.observations, self.actions = RingBuffer(limit), RingBuffer(limit)
        self.rewards, self.terminals = RingBuffer(limit), RingBuffer(limit)

    @property
    def nb_entries(self):
        return len(self.observations)

    def append(self, observation, action, reward, terminal, training=True):
        if training:
            self.observations.append(observation)
            self.actions.append(action)
            self.rewards.append(reward)
            self.terminals.append(terminal)

    def valid_idxs(self):
        """Every idx sample() can end up with (after its redraw loop)."""
        w = self.window_length
        return [i for i in range(w + 1, self.nb_entries) if not self.terminals[i - 2]]

    def sample(self, batch_size, batch_idxs=None, rng=random):
        w = self.window_length
        assert self.nb_entries >= w + 2, "not enough entries in the memory"
        if batch_idxs is None:
            batch_idxs = sample_batch_indexes(w, self.nb_entries - 1, batch_size, rng=rng)
        batch_idxs = [int(i) + 1 for i in batch_idxs]
        assert min(batch_idxs) >= w + 1 and max(batch_idxs) < self.nb_entries and len(batch_idxs) == batch_size
        out = []
        for idx in batch_idxs:
            while self.terminals[idx - 2]:
                idx = sample_batch_indexes(w + 1, self.nb_entries, 1, rng=rng)[0]
            state0 = [self.observations[idx - 1]]
            for off in range(w - 1):
                cur = idx - 2 - off
                if cur < 0 or (cur - 1 >= 0 and self.terminals[cur - 1]):
                    break                                     # never reach into the previous episode
                state0.insert(0, self.observations[cur])
            while len(state0) < w:
                state0.insert(0, np.zeros_like(state0[0]))
            state1 = state0[1:] + [self.observations[idx]]
            out.append(dict(idx=idx, state0=state0, action=self.actions[idx - 1], reward=self.rewards[idx - 1], state1=state1,
                            terminal1=self.terminals[idx - 1]))
        return out


def lattice_memory(terminal_ring, env, n_slots, head_slot, filled_slots, limit=None):
    """The keras-rl memory lattice `env` of a device ring would hold: one entry per slot that has an action recorded (all written
    slots but the newest, which holds only the successor observation), oldest first.  Entries carry their ring slot as the
    'observation' so that sampled experiences can be mapped back to ring rows."""
    nb = filled_slots - 1
    mem = SequentialMemory(limit or max(nb, 1), window_length=1)
    oldest = (head_slot - nb) % n_slots
    for k in range(nb):
        s = (oldest + k) % n_slots
        mem.append(s, 0, 0.0, bool(terminal_ring[s, env]))
    return mem


def valid_transitions(terminal_ring, n_envs, n_slots, head_slot, filled_slots):
    """Set of ring rows (slot * n_envs + env) keras-rl's sample() can return as the experience's entry idx - 1, over all lattices."""
    rows = set()
    for e in range(n_envs):
        mem = lattice_memory(terminal_ring, e, n_slots, head_slot, filled_slots)
        for idx in mem.valid_idxs():
            rows.add(mem.observations[idx - 1] * n_envs + e)
    return rows


# ---- bit-exact restatement of the device sampler (csrc/common.h dq_replay_permute / dq_replay_row) -------------------------------------
def _mix32(h):
    h = np.asarray(h, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    return h


def replay_permute(x, M, keys):
    """pi_t(x) for an array x of values in [0, M): four-round unbalanced Feistel network over the next power of two, cycle-walked."""
    x = np.asarray(x, dtype=np.uint64)
    if M < 2:
        return np.zeros_like(x)
    bits = int(M - 1).bit_length()
    a = bits >> 1
    b = bits - a
    ma, mb = np.uint64((1 << a) - 1), np.uint64((1 << b) - 1)
    k = [np.uint64(int(w)) for w in keys]
    v = x.copy()
    todo = np.ones(v.shape, dtype=bool)
    while todo.any():
        w = v[todo]
        hi, lo = w >> np.uint64(a), w & ma
        hi = hi ^ (_mix32(lo ^ k[0]) & mb)
        lo = lo ^ (_mix32(hi ^ k[1]) & ma)
        hi = hi ^ (_mix32(lo ^ k[2]) & mb)
        lo = lo ^ (_mix32(hi ^ k[3]) & ma)
        v[todo] = (hi << np.uint64(a)) | lo
        todo &= v >= np.uint64(M)
    return v


def device_replay_rows(terminal_ring, n_envs, n_slots, head_slot, filled_slots, batch, seed, t, sample_base=0):
    """Ring rows (slot * n_envs + env) the device draws for samples sample_base .. sample_base + batch - 1 of update t."""
    from . import philox
    term = np.asarray(terminal_ring).reshape(n_slots, n_envs)
    cand = filled_slots - 3
    M = cand * n_envs
    distinct = batch <= M
    ids = (sample_base + np.arange(batch, dtype=np.uint64)) & np.uint64(0xFFFFFFFF)
    t_lo, t_hi = int(t) & philox.MASK, (int(t) >> 32) & philox.MASK
    rows = np.zeros(batch, dtype=np.int64)
    live = np.ones(batch, dtype=bool)
    for attempt in range(64):
        if not live.any():
            break
        sel = np.flatnonzero(live)
        if attempt == 0 and distinct:
            keys = philox.philox4x32((t_lo, t_hi, 0xFFFFFFFF, 0xFFFF | (philox.STREAM_REPLAY << 16)), seed)
            flat = replay_permute(ids[sel] % np.uint64(M), M, keys).astype(np.int64)
            j, e = flat // n_envs, flat % n_envs
        else:
            w = philox.philox4x32_np(t_lo, t_hi, ids[sel].astype(np.uint32), attempt | (philox.STREAM_REPLAY << 16), seed)
            j = ((w[0].astype(np.uint64) * np.uint64(cand)) >> np.uint64(32)).astype(np.int64)
            e = ((w[1].astype(np.uint64) * np.uint64(n_envs)) >> np.uint64(32)).astype(np.int64)
        slot = (head_slot - 2 - j) % n_slots
        rows[sel] = slot * n_envs + e
        live[sel] = term[(slot - 1) % n_slots, e] != 0
    return rows
