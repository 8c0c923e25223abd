"""CPU oracle for the prioritised replay path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A from-scratch numpy restatement of the algorithm in the reference's memory.py
(/root/reference/memory.py, cited per function as memory.py:LINE).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
rainbow_amd package never does.

Parity status: PINNED.  tests/golden/replay_*.npz were produced by running the real
reference classes (tests/golden/make_golden.py, run in the build container where
/root/reference is importable); tests/test_oracle_golden.py replays the same scripted
scenarios through this oracle and requires identical indices / tree floats.

Layout differs from the reference on purpose (struct-of-arrays instead of a numpy
structured array; iterative level loops instead of recursion) — the *values* are what
must match:
  - float32 level-order sum tree with the truncated leaf level (memory.py:17-18),
  - internal node == float32(left + right) of its current children (memory.py:25,39),
  - tree search compares float64 samples against float32 nodes with a strict '>' and
    subtracts in float64 (memory.py:73-75), children clamped on the last internal level
    (memory.py:70-71),
  - numpy NEP-50 dtypes of the sampler arithmetic (memory.py:125-129,151-154).
"""
import numpy as np

FRAME = (84, 84)


def tree_geometry(capacity):
    """memory.py:17-18 -> (levels, tree_start, tree_len)."""
    levels = int(capacity - 1).bit_length()
    tree_start = 2 ** levels - 1
    return levels, tree_start, tree_start + capacity


class SumTreeOracle:
    """SegmentTree (memory.py:12-89) over struct-of-arrays storage."""

    def __init__(self, capacity):
        assert capacity >= 2 and capacity % 2 == 0, "odd capacities crash the reference (memory.py:38-39)"
        self.capacity = int(capacity)
        self.levels, self.tree_start, self.tree_len = tree_geometry(self.capacity)
        self.tree = np.zeros(self.tree_len, dtype=np.float32)            # memory.py:18
        self.timestep = np.zeros(self.capacity, dtype=np.int32)          # blank_trans, memory.py:8,19
        self.frames = np.zeros((self.capacity,) + FRAME, dtype=np.uint8)
        self.action = np.zeros(self.capacity, dtype=np.int32)
        self.reward = np.zeros(self.capacity, dtype=np.float32)
        self.nonterminal = np.zeros(self.capacity, dtype=np.bool_)
        self.index = 0                                                   # memory.py:14
        self.full = False                                                # memory.py:16
        self.max = np.float32(1.0)                                       # memory.py:20

    # -- writes ------------------------------------------------------------------
    def _rebuild_parents_of(self, nodes):
        """One level of memory.py:28-33 / 36-41: every distinct parent := f32(left+right)."""
        parents = np.unique((np.asarray(nodes, dtype=np.int64) - 1) // 2)
        self.tree[parents] = self.tree[2 * parents + 1] + self.tree[2 * parents + 2]
        return parents

    def set_leaves(self, tree_indices, values):
        """SegmentTree.update (memory.py:44-48); duplicates: last write wins (memory.py:45)."""
        tree_indices = np.asarray(tree_indices, dtype=np.int64)
        values = np.asarray(values, dtype=np.float32)
        for node, val in zip(tree_indices, values):      # explicit order == numpy fancy assignment
            self.tree[node] = val
        nodes = tree_indices
        for _ in range(self.levels):
            nodes = self._rebuild_parents_of(nodes)
        self.max = np.float32(max(np.float32(np.max(values)), self.max))  # memory.py:47-48

    def append(self, timestep, frame_u8, action, reward, nonterminal, value):
        """SegmentTree.append (memory.py:56-61)."""
        i = self.index
        self.timestep[i] = timestep
        self.frames[i] = frame_u8
        self.action[i] = action
        self.reward[i] = reward
        self.nonterminal[i] = nonterminal
        node = i + self.tree_start
        self.tree[node] = np.float32(value)              # memory.py:52
        while node != 0:                                 # memory.py:36-41
            node = (node - 1) // 2
            self.tree[node] = self.tree[2 * node + 1] + self.tree[2 * node + 2]
        self.index = (i + 1) % self.capacity             # memory.py:59
        self.full = self.full or self.index == 0         # memory.py:60
        self.max = np.float32(max(np.float32(value), self.max))  # memory.py:54,61

    def bulk_append(self, timesteps, frames_u8, actions, rewards, nonterminals):
        """len(timesteps) sequential appends at the running max priority, vectorised: the tree is a pure
        function of its leaves (memory.py:25,39), so rebuilding the touched ancestors level by level gives the
        floats the one-at-a-time walks (memory.py:36-41) give.  Benchmark-fill helper."""
        n = len(timesteps)
        assert n <= self.capacity
        pos = (self.index + np.arange(n)) % self.capacity
        self.timestep[pos] = timesteps
        self.frames[pos] = frames_u8
        self.action[pos] = actions
        self.reward[pos] = rewards
        self.nonterminal[pos] = nonterminals
        nodes = pos + self.tree_start
        self.tree[nodes] = self.max
        for _ in range(self.levels):
            nodes = self._rebuild_parents_of(nodes)
        self.full = self.full or self.index + n >= self.capacity
        self.index = (self.index + n) % self.capacity

    # -- reads -------------------------------------------------------------------
    def find(self, values):
        """SegmentTree.find (memory.py:64-82) -> (probs f32, data_idx i64, tree_idx i64)."""
        values = np.array(values, dtype=np.float64, copy=True)
        node = np.zeros(values.shape, dtype=np.int64)
        for _ in range(self.levels):
            left = 2 * node + 1
            right = left + 1
            if left.flat[0] >= self.tree_start:          # memory.py:70-71
                left = np.minimum(left, self.tree_len - 1)
                right = np.minimum(right, self.tree_len - 1)
            left_val = self.tree[left]                   # float32
            go_right = values > left_val                 # memory.py:73 (f64 vs f32 -> f64 compare)
            node = np.where(go_right, right, left)       # memory.py:74
            values = values - go_right.astype(np.int32) * left_val  # memory.py:75 (int32*f32 -> f64)
        return self.tree[node], node - self.tree_start, node

    def total(self):
        return self.tree[0]                              # memory.py:88-89


class ReplayOracle:
    """ReplayMemory (memory.py:91-180) with the sampler's RNG injected as unit uniforms."""

    def __init__(self, capacity, history=4, discount=0.99, multi_step=3, priority_weight=0.4,
                 priority_exponent=0.5):
        self.capacity = int(capacity)
        self.history = int(history)
        self.discount = discount
        self.n = int(multi_step)
        self.priority_weight = priority_weight           # memory.py:98
        self.priority_exponent = priority_exponent       # memory.py:99
        self.t = 0                                       # memory.py:100
        # memory.py:101: python doubles rounded to float32
        self.n_step_scaling = np.array([discount ** i for i in range(self.n)], dtype=np.float32)
        self.transitions = SumTreeOracle(self.capacity)

    @staticmethod
    def quantise(state_f32):
        """state[-1].mul(255).to(uint8) (memory.py:106): float32 multiply, truncation."""
        x = np.asarray(state_f32, dtype=np.float32)[-1] * np.float32(255)
        return x.astype(np.uint8)

    def append(self, state_f32, action, reward, terminal):
        """memory.py:105-108."""
        self.append_frame(self.quantise(state_f32), action, reward, terminal)

    def append_frame(self, frame_u8, action, reward, terminal):
        tr = self.transitions
        tr.append(self.t, frame_u8, action, reward, not terminal, tr.max)   # memory.py:107
        self.t = 0 if terminal else self.t + 1                              # memory.py:108

    # -- window with episode-boundary blanking (memory.py:111-121) ------------------
    def window(self, idxs):
        h, n = self.history, self.n
        tr = self.transitions
        offs = np.arange(-h + 1, n + 1, dtype=np.int64)
        ring = (np.asarray(idxs, dtype=np.int64)[:, None] + offs[None, :]) % self.capacity   # memory.py:112,86
        first = tr.timestep[ring] == 0                                                      # memory.py:114
        blank = np.zeros_like(first)
        for t in range(h - 2, -1, -1):                                                      # memory.py:116-117
            blank[:, t] = blank[:, t + 1] | first[:, t + 1]
        for t in range(h, h + n):                                                           # memory.py:118-119
            blank[:, t] = blank[:, t - 1] | first[:, t]
        return ring, blank

    def draw_indices(self, batch, unit_uniforms):
        """The rejection loop of ReplayMemory._get_samples_from_segments (memory.py:124-132) alone: returns
        (probs f32[B], data_idxs i64[B], tree_idxs i64[B], attempts).  Touches the tree and the write head only, so a
        test can run it on a tree downloaded from the device without the 7 GB frame store."""
        tr = self.transitions
        h, n = self.history, self.n
        p_total = tr.total()                                               # np.float32
        seg = np.float32(p_total) / np.float32(batch)                      # memory.py:125 (f32)
        starts = np.arange(batch, dtype=np.int64) * np.float64(seg)        # memory.py:126 (f64)
        unit_uniforms = np.asarray(unit_uniforms, dtype=np.float64).reshape(-1, batch)
        attempts = 0
        for u in unit_uniforms:
            attempts += 1
            samples = (0.0 + np.float64(seg) * u) + starts                 # memory.py:129
            probs, idxs, tree_idxs = tr.find(samples)                      # memory.py:130
            ok = (np.all((tr.index - idxs) % self.capacity > n)
                  and np.all((idxs - tr.index) % self.capacity >= h)
                  and np.all(probs != 0))                                  # memory.py:131
            if ok:
                return probs, idxs, tree_idxs, attempts
        raise RuntimeError("oracle sampler: no valid batch within the supplied attempts")

    def batch_scalars(self, idxs, probs):
        """Everything of the sampled batch except the pixels (memory.py:111-121,140-145,149-154): the window's ring
        slots and blank mask, actions, n-step returns, nonterminals and importance-sampling weights."""
        tr = self.transitions
        h, n = self.history, self.n
        batch = len(idxs)
        p_total = tr.total()
        ring, blank = self.window(idxs)
        rewards = np.where(blank, np.float32(0), tr.reward[ring]).astype(np.float32)
        nonterm = np.where(blank, False, tr.nonterminal[ring])
        actions = np.where(blank, 0, tr.action[ring])
        R = np.zeros(batch, dtype=np.float32)                              # memory.py:142-143
        for k in range(n):
            R = R + rewards[:, h - 1 + k] * self.n_step_scaling[k]
        probs_n = probs / p_total                                          # memory.py:151 (f32)
        cap = self.capacity if tr.full else tr.index                       # memory.py:152
        weights = (np.float32(cap) * probs_n) ** np.float32(-self.priority_weight)  # memory.py:153
        weights = (weights / weights.max()).astype(np.float32)             # memory.py:154
        return dict(ring=ring, blank=blank,
                    actions=actions[:, h - 1].astype(np.int64),           # memory.py:140
                    returns=R, nonterminals=nonterm[:, h + n - 1].astype(np.float32)[:, None],  # memory.py:145
                    weights=weights)

    def sample_with_uniforms(self, batch, unit_uniforms):
        """ReplayMemory.sample (memory.py:124-155).  unit_uniforms[a] are the U[0,1) draws of
        attempt a: np.random.uniform(0.0, seg, B) == 0.0 + seg*u elementwise (memory.py:129).
        Returns a dict with the 7-tuple's fields as numpy arrays (states as uint8 stacks)."""
        tr = self.transitions
        h, n = self.history, self.n
        probs, idxs, tree_idxs, attempts = self.draw_indices(batch, unit_uniforms)
        sc = self.batch_scalars(idxs, probs)
        frames = tr.frames[sc["ring"]]                                     # [B, h+n, 84, 84]
        frames[sc["blank"]] = 0                                            # memory.py:120
        states = frames[:, :h]                                             # memory.py:137
        next_states = frames[:, n:n + h]                                   # memory.py:138
        return dict(tree_idxs=tree_idxs.astype(np.int64), data_idxs=idxs.astype(np.int64), probs=probs,
                    states=states, next_states=next_states, actions=sc["actions"], returns=sc["returns"],
                    nonterminals=sc["nonterminals"], weights=sc["weights"], attempts=attempts)

    def update_priorities(self, tree_idxs, priorities):
        """memory.py:157-159.  NOTE numpy's float32 power is SIMD (SVML) on AVX-512 hosts and
        is only accurate to ~1 ulp, so these values are machine-dependent in the last bit —
        for the reference as well."""
        p = np.power(np.asarray(priorities, dtype=np.float32), np.float32(self.priority_exponent))
        self.transitions.set_leaves(tree_idxs, p)

    def state_at(self, i):
        """ReplayMemory.__next__ (memory.py:167-178): blanked history stack ending at data
        index i (numpy negative indices wrap, equivalent to % capacity)."""
        h = self.history
        tr = self.transitions
        ring = np.arange(i - h + 1, i + 1) % self.capacity
        first = tr.timestep[ring] == 0
        blank = np.zeros(h, dtype=bool)
        for t in range(h - 2, -1, -1):
            blank[t] = blank[t + 1] | first[t + 1]
        frames = tr.frames[ring].copy()
        frames[blank] = 0
        return frames.astype(np.float32) / np.float32(255)
