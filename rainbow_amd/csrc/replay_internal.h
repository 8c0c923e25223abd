// replay_internal.h — pieces of the replay implementation shared inside librainbow_hip.so (not part of the C ABI):
// the kernel-side view of a replay handle and the sum-tree update body, so the learner can run the priority
// write-back (agent.py:100) as one extra workgroup of its own backward launch instead of a separate dependent kernel.
#pragma once
#include "rb_common.h"

struct ReplayView {
  int64_t capacity;
  int32_t history, n, levels;
  int64_t tree_start, tree_len;
  float* tree;
  uint8_t* frames;
  int32_t* timestep;
  int32_t* action;
  float* reward;
  uint8_t* nonterminal;
  rb_replay_header_t* hdr;
  int32_t* dropped;     // pinned host word: write-backs dropped because their indices came from a failed draw (rb_replay_dropped_updates)
};


// floor_mod(a + d, m) for a in [0, m) and a SMALL offset d (window slots, write-head distance): conditional add /
// subtract instead of a 64-bit division (~100 instructions each; the sampler needs ~25 per sample).  The loops run
// at most once unless the ring is shorter than the window (degenerate capacities stay correct).
__device__ __forceinline__ int64_t rb_wrap(int64_t a, int64_t d, int64_t m) {
  int64_t x = a + d;
  while (x < 0) x += m;
  while (x >= m) x -= m;
  return x;
}
__device__ __forceinline__ int64_t rb_floor_mod(int64_t a, int64_t m) {
  int64_t r = a % m;
  return r < 0 ? r + m : r;
}


// ---------------------------------------------------------------------- update --
// SegmentTree.update (memory.py:44-48) for n <= 1024 leaves in ONE workgroup.
// Duplicate indices: numpy fancy assignment is last-write-wins (memory.py:45).
// apply_pow: ReplayMemory.update_priorities' p = loss^w first (memory.py:158).
//
// Latency structure: the L ancestor sums are NOT walked through memory level by level (that is
// L dependent round trips).  Each thread prefetches the sibling of its node on every level in one
// batch of independent loads, then walks to the root in registers.  Where two paths of this batch
// meet, the sibling's FRESH value must be used instead of the prefetched one: every level
// publishes (node -> value) in an LDS hash table (open addressing, 2048 slots for <= 1024 keys,
// three tables in rotation so one barrier per level suffices) and looks its sibling up there —
// O(1) LDS probes per level instead of scanning the batch.  Every parent is still
// fl32(left + right) of its current children (memory.py:25): same floats as the reference.
#define RB_MAX_LEVELS 31
// The top RB_UPD_TOP levels of the tree are not walked path by path: once every path has reached depth RB_UPD_TOP, the
// whole top (2^RB_UPD_TOP nodes of that depth, staged in LDS at kernel start, updated entries overwritten) is rebuilt
// densely — plain LDS adds, no hashing, the last six levels inside one wave without block barriers — and written back.
// Untouched nodes are recomputed to the very value they hold (every node IS fl32(left + right) of its children), so the
// result is bit-identical to the per-path walk; the hashed rounds drop from L to L - RB_UPD_TOP (20 -> 9 for 1M leaves).
#define RB_UPD_TOP 11

// the table size follows the batch (power of two >= 4n, <= 2048 slots): a batch of 32 clears and probes 128 slots
__device__ __forceinline__ int rb_hash_slot(int node, int shift) {
  return (int)(((unsigned)node * 2654435761u) >> shift);
}
__device__ __forceinline__ int rb_hash_insert(int* keys, int node, int shift, int mask) {
  int h = rb_hash_slot(node, shift);
  for (;;) {
    const int prev = atomicCAS(&keys[h], -1, node);
    if (prev == -1 || prev == node) return h;
    h = (h + 1) & mask;
  }
}
__device__ __forceinline__ int rb_hash_find(const int* keys, int node, int shift, int mask) {
  int h = rb_hash_slot(node, shift);
  for (;;) {
    const int k = keys[h];
    if (k == node) return h;
    if (k == -1) return -1;
    h = (h + 1) & mask;
  }
}

// body (all threads of ONE workgroup of >= n threads, multiple of 64); shared by k_update and the learner's fused launch
// LDS comes from the caller (HS hash slots per table, a power of two >= 2 * NMAX; NMAX >= n): the stand-alone kernel
// uses <2048, 1024>, the learner's fused launch <512, 256> so that its other workgroups keep their occupancy.
template <int HS, int NMAX>
struct UpdateLds {
  static constexpr int HEAP = 2 << RB_UPD_TOP;          // level-order heap of the top RB_UPD_TOP + 1 levels
  static constexpr int WORDS = 8 * HS + NMAX + 16 + HEAP;
};
template <int HS, int NMAX>
__device__ __forceinline__ void rb_update_body(ReplayView v, const int64_t* tree_idx, const float* values, int32_t n,
                                               int32_t apply_pow, double omega, float* lds) {
  int (*s_key)[HS] = reinterpret_cast<int (*)[HS]>(lds);                 // [4][HS]; [3] = leaf de-duplication table
  float (*s_tv)[HS] = reinterpret_cast<float (*)[HS]>(lds + 4 * HS);     // [3][HS]
  int* s_pos = reinterpret_cast<int*>(lds + 7 * HS);
  float* s_vi = lds + 8 * HS;
  float* s_red = lds + 8 * HS + NMAX;
  float* s_heap = lds + 8 * HS + NMAX + 16;
  const int i = (int)threadIdx.x;
  // ReplayMemory.update_priorities on the indices of a sampler launch that gave up (no valid batch within max_attempts: the
  // reference would still be spinning in memory.py:128-132): that draw was NOT a legal batch — a never-written leaf or one
  // straddling the write head would receive a non-zero priority and defeat the `prob != 0` validity test of every later
  // draw.  The sampler marks such a draw in its OWN index buffer (every tree index -1), so exactly the write-back that
  // belongs to the failed draw is dropped (block-uniform) and counted; the write-back of an earlier, valid batch still
  // applies whatever the header's status word says by now.
  if (apply_pow && tree_idx[0] < 0) {
    if (threadIdx.x == 0 && v.dropped) rb_atomic_inc_system(v.dropped);
    return;
  }
  // dense top: only when the tree is deeper than the top itself (block-uniform)
  const bool dense = v.levels > RB_UPD_TOP;
  const int path_levels = dense ? v.levels - RB_UPD_TOP : v.levels;
  if (dense) {
    constexpr int BASE = (1 << RB_UPD_TOP) - 1;
    for (int t = i; t < (1 << RB_UPD_TOP); t += (int)blockDim.x) s_heap[BASE + t] = v.tree[BASE + t];
  }
  const bool active = i < n;
  int node = active ? (int)tree_idx[i] : -1;
  // sibling prefetch for every level (stale where another updated path passes; fixed up from LDS)
  float sib[RB_MAX_LEVELS];
  {
    int q = node;
#pragma unroll
    for (int lv = 0; lv < RB_MAX_LEVELS; ++lv) {
      if (active && lv < path_levels) {
        const int sb = (q & 1) ? q + 1 : q - 1;
        sib[lv] = v.tree[sb];
        q = (q - 1) >> 1;
      } else {
        sib[lv] = 0.0f;
      }
    }
  }
  // (sizing the tables by the batch was measured SLOWER on MI355X — 15.3 vs 11.9 us at n=32, more probe collisions in
  // the top hash bits — so all batches use the full 2048 slots)
  constexpr int HBITS = HS == 2048 ? 11 : HS == 1024 ? 10 : HS == 512 ? 9 : HS == 256 ? 8 : -1;
  static_assert(HBITS > 0, "HS must be 256, 512, 1024 or 2048");
  const int hmask = HS - 1, hshift = 32 - HBITS;
  for (int t = i; t <= hmask; t += (int)blockDim.x) {
    s_key[0][t] = -1; s_key[1][t] = -1; s_key[2][t] = -1; s_key[3][t] = -1;
    s_pos[t] = -1;
  }
  float val = 0.0f;
  if (active) {
    val = values[i];
    // loss ** omega (memory.py:158; numpy evaluates it in float32, ~1 ulp): exp(omega * log(x)) in float64, rounded once
    if (apply_pow) val = val > 0.0f ? (float)exp(omega * log((double)val)) : (float)pow((double)val, omega);
  }
  s_vi[i] = val;
  const float vmax = rb_block_max(active ? val : -INFINITY, s_red);  // np.max(values), memory.py:47 (+ barrier)
  int slot = -1;
  if (active) {
    slot = rb_hash_insert(s_key[3], node, hshift, hmask);
    atomicMax(&s_pos[slot], i);                // last occurrence wins (memory.py:45)
  }
  __syncthreads();
  if (active) {
    val = s_vi[s_pos[slot]];
    v.tree[node] = val;
  }
  int prev_slot = -1;
#pragma unroll
  for (int lv = 0; lv < RB_MAX_LEVELS; ++lv) {
    if (lv < path_levels) {                     // block-uniform
      const int c = lv % 3;
      int my = -1;
      if (active) {
        my = rb_hash_insert(s_key[c], node, hshift, hmask);
        s_tv[c][my] = val;                      // paths on the same node carry the same value
      }
      __syncthreads();
      if (active) {
        if (prev_slot >= 0) s_key[(lv + 2) % 3][prev_slot] = -1;   // retire level lv-1's entry (all its lookups are done)
        const int sb = (node & 1) ? node + 1 : node - 1;
        const int f = rb_hash_find(s_key[c], sb, hshift, hmask);
        const float sv = f >= 0 ? s_tv[c][f] : sib[lv];
        const float left = (node & 1) ? val : sv;     // odd index = left child (2p+1)
        const float right = (node & 1) ? sv : val;
        val = __fadd_rn(left, right);                 // memory.py:25
        node = (node - 1) >> 1;
        v.tree[node] = val;
        prev_slot = my;
      }
    }
  }
  if (dense) {
    constexpr int BASE = (1 << RB_UPD_TOP) - 1;
    __syncthreads();                            // staged heap level complete; every path stands at depth RB_UPD_TOP
    if (active) s_heap[node] = val;             // paths on the same node carry the same value
    __syncthreads();
#pragma unroll
    for (int d = RB_UPD_TOP - 1; d >= 6; --d) {
      const int cnt = 1 << d;
      for (int j = i; j < cnt; j += (int)blockDim.x) {
        const int p = cnt - 1 + j;
        s_heap[p] = __fadd_rn(s_heap[2 * p + 1], s_heap[2 * p + 2]);   // memory.py:25
      }
      __syncthreads();
    }
    if (i < 64) {                               // wave 0: the last six levels without block barriers
#pragma unroll
      for (int d = 5; d >= 0; --d) {
        const int cnt = 1 << d;
        if (i < cnt) {
          const int p = cnt - 1 + i;
          s_heap[p] = __fadd_rn(s_heap[2 * p + 1], s_heap[2 * p + 2]);
        }
        rb_wave_sync();
      }
    }
    __syncthreads();
    for (int t = i; t < BASE; t += (int)blockDim.x) v.tree[t] = s_heap[t];
    val = s_heap[0];
  }
  if (threadIdx.x == 0) {
    v.hdr->max = fmaxf(vmax, v.hdr->max);  // memory.py:48
    v.hdr->total = val;                    // the root
  }
}


// ---- the same update for a SORTED batch of at most 64 leaves, by ONE wave and without LDS: what ReplayMemory.sample hands to
// update_priorities (stratified draws: sample i lies in stratum i of the cumulative priorities, so the leaf indices never
// decrease, memory.py:125-130).  Sorted leaves stay sorted on every level, so the only place a path can meet another updated
// path is its NEIGHBOUR among the lanes that are still alive: a left child looks at the next alive lane, a right child at the
// previous one (two ballot-mask bit scans and four lane reads per level instead of a hashed LDS table with a workgroup
// barrier per level).  Where two alive siblings meet, both form the same parent value and the left one retires.  Duplicate
// leaves: the last lane of the run wins (memory.py:45) and the others retire before the walk.  All 64 lanes call.
struct UpdateOperand { int node; float val; int status; int sorted; };
// first loads of the sorted-batch update, all independent (one round trip): leaf index, value, the sampler's status word
__device__ __forceinline__ UpdateOperand rb_update_load(const ReplayView& v, const int64_t* tree_idx, const float* values, int32_t n) {
  const int lane = (int)(threadIdx.x & 63u);
  UpdateOperand op;
  op.node = lane < n ? (int)tree_idx[lane] : -1;
  op.val = lane < n ? values[lane] : 0.0f;
  op.status = __shfl(op.node, 0, 64) < 0 ? 1 : 0;          // the sampler's mark of a failed draw (see rb_update_body)
  const int before = __shfl(op.node, lane > 0 ? lane - 1 : 0, 64);
  op.sorted = __all(lane == 0 || lane >= n || op.node >= before) ? 1 : 0;
  return op;
}
// s_top (optional): an LDS copy of the first n_top nodes of the tree that the caller staged BEFORE this update (the sampler's
// search top, k_update_sample): every node written below n_top is written there as well.
//
// WHO meets WHOM on which level depends on the leaf indices alone, so it is worked out for all levels up front, off the
// value chain: with a = leaf + 1 (heap numbering from 1) the ancestor lv levels up is (a >> lv) - 1, and lanes i, i + 1 become
// siblings on level Lr_i = the highest bit in which a_i and a_(i+1) differ (equal leaves: -1, no right neighbour: never).
// On level lv the lanes fall into groups of equal node — the boundaries are the lanes with Lr >= lv, one ballot — every
// lane of a group carries the group's value, and a group's sibling, if it is in the batch at all, is the neighbouring
// group: to the right for a left child (met iff the boundary lane r of my group has Lr_r == lv), to the left for a right
// child.  What remains per level on the dependent chain is one lane read, one select, one add (a first version that looked
// for its neighbour among the surviving lanes inside the chain ran ~0.25 us per level on this lone wave: 5 us for 20 levels).
template <int MAXL>
__device__ __forceinline__ void rb_update_sorted_levels(const ReplayView& v, unsigned a, float val, float vmax, const float (&sib)[RB_MAX_LEVELS],
                                                        int Lr, bool active, int lane, float* s_top, int n_top) {
  const int levels = v.levels;
  float lvl_val[MAXL];
  unsigned long long store_mask[MAXL];
  unsigned long long bge = __ballot(Lr >= 0 ? 1 : 0);
#pragma unroll
  for (int lv = 0; lv < MAXL; ++lv) {
    const unsigned long long beq = __ballot(Lr == lv ? 1 : 0);
    const unsigned long long bge_up = __ballot(Lr >= lv + 1 ? 1 : 0);
    const bool is_left = ((a >> lv) & 1u) == 0u;            // node (a >> lv) - 1 odd = left child (2p + 1)
    const int r = lane + __builtin_ctzll((bge >> lane) | (1ull << (63 - lane)));          // last lane of my group
    const unsigned long long below = bge & ((1ull << lane) - 1ull);
    const int l = below ? 63 - __builtin_clzll(below) : 0;                                 // last lane of the group before mine
    const bool met = is_left ? (((beq >> r) & 1ull) != 0ull) : (below != 0ull && ((beq >> l) & 1ull) != 0ull);
    const int src = is_left ? (r < 63 ? r + 1 : r) : l;
    const float other = __shfl(val, src, 64);
    const float sv = met ? other : sib[lv];
    val = is_left ? __fadd_rn(val, sv) : __fadd_rn(sv, val);                               // memory.py:25: left + right
    lvl_val[lv] = val;
    store_mask[lv] = bge_up;
    bge = bge_up;
  }
  // one writer per node: the last lane of each group of the level written
#pragma unroll
  for (int lv = 0; lv < MAXL; ++lv) {
    if (lv < levels) {                                      // wave-uniform
      const int node = (int)(a >> (lv + 1)) - 1;
      if (active && ((store_mask[lv] >> lane) & 1ull)) {
        v.tree[node] = lvl_val[lv];
        if (node < n_top) s_top[node] = lvl_val[lv];
        if (lv + 1 == levels) {                             // the root: one lane (every pair has met by now)
          v.hdr->max = fmaxf(vmax, v.hdr->max);             // memory.py:48
          v.hdr->total = lvl_val[lv];
        }
      }
    }
  }
}
__device__ __forceinline__ void rb_update_sorted_wave(ReplayView v, const UpdateOperand& op, int32_t n, int32_t apply_pow, double omega,
                                                      float* s_top, int n_top) {
  const int lane = (int)(threadIdx.x & 63u);
  if (apply_pow && op.status != 0) {                       // (the draw was not a legal batch: see rb_update_body)
    if (lane == 0 && v.dropped) rb_atomic_inc_system(v.dropped);
    return;
  }
  const bool active = lane < n;
  const unsigned a = active ? (unsigned)op.node + 1u : 1u;
  float sib[RB_MAX_LEVELS];
#pragma unroll
  for (int lv = 0; lv < RB_MAX_LEVELS; ++lv) {
    const int q = (int)(a >> lv) - 1;
    const int sb = (q & 1) ? q + 1 : q - 1;
    sib[lv] = (active && lv < v.levels) ? v.tree[sb > 0 ? sb : 0] : 0.0f;
  }
  const unsigned a_next = __shfl(a, lane < 63 ? lane + 1 : lane, 64);
  int Lr = 99;                                             // no right neighbour in the batch: never meets
  if (active && lane + 1 < n) Lr = (a ^ a_next) ? 31 - __builtin_clz(a ^ a_next) : -1;
  float val = op.val;
  if (active && apply_pow) val = val > 0.0f ? (float)exp(omega * log((double)val)) : (float)pow((double)val, omega);   // memory.py:158
  const float vmax = rb_wave_max(active ? val : -INFINITY);                                                              // memory.py:47
  {                                                         // equal leaves: the last lane of the run wins (memory.py:45)
    const unsigned long long b0 = __ballot(Lr >= 0 ? 1 : 0);
    const int r0 = lane + __builtin_ctzll((b0 >> lane) | (1ull << (63 - lane)));
    val = __shfl(val, r0, 64);
    if (active && ((b0 >> lane) & 1ull)) {
      v.tree[op.node] = val;
      if (op.node < n_top) s_top[op.node] = val;
    }
  }
  if (v.levels <= 20) rb_update_sorted_levels<20>(v, a, val, vmax, sib, Lr, active, lane, s_top, n_top);
  else rb_update_sorted_levels<RB_MAX_LEVELS>(v, a, val, vmax, sib, Lr, active, lane, s_top, n_top);
}

// The write-back of one workgroup, whichever way fits the batch: a sorted batch of at most 64 leaves (what the sampler hands
// back) goes through rb_update_sorted_wave on wave 0 — the other waves leave — anything else through the hashed body.  All
// threads of the workgroup call; `lds` as for rb_update_body<HS, NMAX> (its first word doubles as the "sorted" flag before the
// body clears it).  Same tree, bit for bit, either way.
template <int HS, int NMAX>
__device__ __forceinline__ void rb_update_auto(ReplayView v, const int64_t* tree_idx, const float* values, int32_t n, int32_t apply_pow,
                                               double omega, float* lds) {
  if (n <= 64) {                                            // block-uniform
    int* s_sorted = reinterpret_cast<int*>(lds);
    UpdateOperand op;
    op.node = -1; op.val = 0.0f; op.status = 0; op.sorted = 0;
    if (threadIdx.x < 64) {
      op = rb_update_load(v, tree_idx, values, n);           // every first load of the chain in one batch
      if (threadIdx.x == 0) *s_sorted = op.sorted;
    }
    __syncthreads();
    const bool sorted = *s_sorted != 0;
    __syncthreads();                                        // (the hashed body reuses the word)
    if (sorted) {
      if (threadIdx.x < 64) rb_update_sorted_wave(v, op, n, apply_pow, omega, nullptr, 0);
      return;
    }
  }
  rb_update_body<HS, NMAX>(v, tree_idx, values, n, apply_pow, omega, lds);
}

// host side (replay.hip): kernel view + priority exponent of a handle
int rb_replay_internal_view(rb_replay_t* r, ReplayView* view, double* omega);

// ---- the early draw (replay.hip; called by learner.hip train_step, not part of the C ABI).
// ReplayMemory.update_priorities of learn call k and ReplayMemory.sample of call k + 1 (agent.py:100 / :63, memory.py:124-159) depend
// on nothing of call k after its head kernel (the per-sample loss), while call k's backward is another ~90 us of launches that
// never touch the replay.  rb_replay_spec_launch issues that pair on a stream the replay owns, as soon as the head is done:
// the write-back is final (it IS call k's), the draw TENTATIVE — it writes the caller's sample buffers and the replay's OTHER
// window table, but its header effects (Philox counter, status, attempts, failure count) go to a side record.  The next draw on
// the handle with the same arguments ACCEPTS it (k_sample block 0 waits for the record's epoch, commits the header effects and
// returns); any other entry point that touches the replay waits for the stream and discards it; a draw with other arguments
// waits and draws again.  Results are those of the two calls made one after the other.
#define RB_SPEC_ABORTED 2     // SpecResult.status: the pair gave up (its gate expired) — nothing was written back, nothing was drawn
struct SpecResult {
  unsigned long long rng_next;
  int32_t attempts, status;  // status: 0 = a legal batch, 1 = the draw hit its attempt bound, RB_SPEC_ABORTED
  unsigned done;             // epoch of the last tentative draw that completed (release-stored last)
  unsigned abort_epoch;      // epoch of the last pair whose gate (k_spec_gate) expired
};
struct rb_spec_request {
  const int64_t* upd_idx; const float* upd_loss; int32_t upd_n;     // the write-back (loss^w: rb_replay_update_priorities)
  int32_t batch; double priority_weight; int32_t max_attempts;      // the draw (rb_replay_sample with device RNG)
  int64_t* tree_idx; int64_t* actions; float* returns; float* nonterminals; float* weights;
  const unsigned* go_flag; unsigned go_epoch;                       // both kernels wait for *go_flag >= go_epoch first (NULL: no wait)
};
int rb_replay_spec_launch(rb_replay_t* r, const rb_spec_request& q);
// 0 once a cross-stream wait of an early pair has expired on this handle (until rb_replay_reset_failed_samples): the learner then
// keeps the write-back and the draw in its own launches
int rb_replay_spec_allowed(rb_replay_t* r);
// one shot: the NEXT draw on the handle may accept the tentative draw in flight (rb_learner_train_step only — it is the one caller
// that reads rb_replay_current_windows(); every public sample entry point redraws into table 0)
void rb_replay_spec_arm_accept(rb_replay_t* r);
int rb_replay_spec_inflight(rb_replay_t* r);
// the window table the LAST draw on the handle filled (rb_replay_buffers_t.window_dev is table 0; an accepted early draw used the other)
const int32_t* rb_replay_current_windows(rb_replay_t* r);
unsigned long long rb_replay_mutations(rb_replay_t* r);
