// replay.hip — HBM-resident prioritised replay for MI355X (gfx950).
//
// What it replaces: the host-RAM numpy SegmentTree/ReplayMemory of the reference
// (memory.py:12-180).  Layout here is SoA in HBM (frames u8[C][7056] + 4 small
// columns) with the reference's level-order float32 sum-tree (same truncated leaf
// level, memory.py:17-18) so tree indices are interchangeable with the reference's.
//
// Invariant used everywhere: every internal node equals fl32(left + right) of its
// CURRENT children (memory.py:25,39), i.e. the tree is a pure function of its leaves.
// Batched updates therefore rebuild ancestors level by level and land on exactly the
// floats the reference's sequential walks produce.
//
// All of this is HBM-latency / HBM-bandwidth work (no GEMM shape anywhere): the frame
// gather moves (h+n)·7056 B in and 2h·7056 B out per sample with 16-byte lanes, the tree
// search is L dependent 4-byte loads per sample.
#include "noise_body.h"
#include "adam_body.h"
#include "replay_internal.h"

#include <stdlib.h>
#include <string.h>

#include <new>

// -------------------------------------------------------------------------------
struct rb_replay {
  int64_t capacity;
  int32_t history, n;
  int32_t levels;       // L = depth of the leaf level
  int64_t tree_start;   // 2^L - 1
  int64_t tree_len;     // tree_start + capacity
  double omega;         // priority exponent (memory.py:99)
  uint64_t seed;
  float scaling[64];    // gamma^k as float32(double pow)  (memory.py:101)
  // device
  float* tree;
  uint8_t* frames;
  int32_t* timestep;
  int32_t* action;
  float* reward;
  uint8_t* nonterminal;
  rb_replay_header_t* hdr;
  int32_t* win;         // [max_batch][h+n] ring index of each window slot, -1 = blank
  float* scaling_dev;
  int32_t max_batch;
  const float* neg_beta_dev;   // optional device-resident -beta (graph replay: no by-value argument may change)
  int32_t* fail_host;          // pinned, device-mapped, four words: [0] sampler launches that found no valid batch (see k_sample),
                               // [1] priority write-backs dropped because their indices came from such a draw (rb_update_body),
                               // [2] expired cross-stream waits of the early draw (rb_poll_epoch; must stay 0)
  // host mirror of the deterministic part of the header
  int64_t host_index;
  int32_t host_full;
  // the early draw (replay_internal.h rb_replay_spec_launch)
  int32_t* win2;               // the second window table
  int win_sel;                 // table of the last draw (0 = win, 1 = win2)
  hipStream_t spec_stream;
  SpecResult* spec_res;        // device
  unsigned spec_epoch;
  int spec_inflight;
  struct { int32_t batch, max_attempts, win_set; double beta; int64_t* tree_idx; int64_t* actions; float* returns; float* nonterm; float* weights; } spec_args;
  unsigned long long mutations;   // entry points that changed the replay (or drew from it) so far
  int spec_accept_armed;          // the NEXT draw on the handle may accept a tentative draw (set by rb_learner_train_step only, one shot:
                                  // the public sample entry points always redraw into table 0 = rb_replay_buffers_t.window_dev)
  int spec_disabled;              // an expired cross-stream wait was seen (fail_host[2]): no early draw on this handle until
                                  // rb_replay_reset_failed_samples
};
static int32_t* win_of(const rb_replay* r, int set) { return set ? r->win2 : r->win; }
// every entry point that reads or writes the replay outside a draw: wait for an early draw in flight and discard it (its
// write-back part is final and stays, its tentative draw is simply never accepted)
static int spec_join(rb_replay* r) {
  if (!r->spec_inflight) return RB_OK;
#if !defined(RB_HOST_INTERP)
  RB_HIP_TRY(hipStreamSynchronize(r->spec_stream));
#endif
  r->spec_inflight = 0;
  return RB_OK;
}
#define RB_SPEC_JOIN(r)                 \
  do {                                  \
    const int rcj_ = spec_join(r);      \
    if (rcj_ != RB_OK) return rcj_;     \
  } while (0)

static ReplayView view_of(const rb_replay* r) {
  ReplayView v;
  v.capacity = r->capacity; v.history = r->history; v.n = r->n; v.levels = r->levels;
  v.tree_start = r->tree_start; v.tree_len = r->tree_len;
  v.tree = r->tree; v.frames = r->frames; v.timestep = r->timestep; v.action = r->action;
  v.reward = r->reward; v.nonterminal = r->nonterminal; v.hdr = r->hdr;
  v.dropped = r->fail_host ? r->fail_host + 1 : nullptr;
  return v;
}

// ---------------------------------------------------------------- init / header --
__global__ void k_replay_init(rb_replay_header_t* hdr) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    hdr->index = 0;
    hdr->full = 0;
    hdr->max = 1.0f;  // memory.py:20  (1 = 1^w)
    hdr->total = 0.0f;
    hdr->last_attempts = 0;
    hdr->last_status = 0;
    hdr->rng_counter = 0;
  }
}

// ---------------------------------------------------------------------- append --
// One transition (memory.py:105-108 + 56-61).  One 256-thread workgroup: quantise and
// store the frame with 4-byte packed writes, thread 0 walks the L sums to the root.
__global__ __launch_bounds__(256) void k_append_one(ReplayView v, const float* last_frame, int32_t timestep,
                                                     int32_t action, float reward, int32_t nonterminal) {
  const int64_t idx = v.hdr->index;
  const float prio = v.hdr->max;
  __syncthreads();  // every thread has read the header before thread 0 rewrites it
  uint32_t* dst = (uint32_t*)(v.frames + idx * RB_FRAME_BYTES);
  for (int w = (int)threadIdx.x; w < RB_FRAME_BYTES / 4; w += (int)blockDim.x) {
    uint32_t packed = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // state[-1].mul(255).to(uint8): f32 multiply, truncating conversion (memory.py:106)
      const float x = __fmul_rn(last_frame[4 * w + j], 255.0f);
      const uint32_t q = (uint32_t)(int32_t)x & 0xFFu;
      packed |= q << (8 * j);
    }
    dst[w] = packed;
  }
  if (threadIdx.x == 0) {
    v.timestep[idx] = timestep;
    v.action[idx] = action;
    v.reward[idx] = reward;
    v.nonterminal[idx] = nonterminal ? 1 : 0;
    int64_t node = idx + v.tree_start;
    v.tree[node] = prio;  // memory.py:52
    while (node != 0) {   // memory.py:36-41
      const int64_t parent = (node - 1) / 2;
      v.tree[parent] = __fadd_rn(v.tree[2 * parent + 1], v.tree[2 * parent + 2]);
      node = parent;
    }
    const int64_t next = (idx + 1) % v.capacity;
    v.hdr->index = next;                 // memory.py:59
    if (next == 0) v.hdr->full = 1;      // memory.py:60
    v.hdr->total = v.tree[0];
    // memory.py:54,61: max(value, max) with value == max — unchanged
  }
}

// ---------------------------------------------------------------- frame pipeline --
// env.py:27-29 (cv2.resize(gray [H][W] u8, (84, 84), INTER_LINEAR) -> f32 / 255) and env.py:57-69 (element-wise max over the
// last two frames of the action repeat) on the device: raw emulator screens in, the observation the actor and
// ReplayMemory.append consume out — no host-side resize, no 28 KB H2D float frame per environment step.
// The resize is OpenCV's 8-bit fixed-point INTER_LINEAR (11-bit coefficients; oracle/frame_oracle.py has the algebra and
// says why this row is parity-UNPINNED: cv2 is absent here).  One thread per output pixel; the taps of a pixel are four
// bytes per frame, the coefficients two float operations — nothing worth staging.
__device__ __forceinline__ void rb_resize_tap(int d, int dst, int src, bool clamp_f, int* s_out, int* c0, int* c1) {
  const double scale = (double)src / (double)dst;
  float f = (float)__dsub_rn(__dmul_rn((double)d + 0.5, scale), 0.5);          // float((d + 0.5) * scale - 0.5)
  int s = (int)floorf(f);
  f = __fsub_rn(f, (float)s);
  if (clamp_f) {
    if (s < 0) { f = 0.0f; s = 0; }
    if (s >= src - 1) { f = 0.0f; s = src - 1; }
  }
  *s_out = s;
  *c0 = __float2int_rn(__fmul_rn(__fsub_rn(1.0f, f), 2048.0f));               // saturate_cast<short>(cbuf * INTER_RESIZE_COEF_SCALE)
  *c1 = __float2int_rn(__fmul_rn(f, 2048.0f));
}
__device__ __forceinline__ int rb_resize_pixel(const uint8_t* img, int H, int W, int sx, int a0, int a1, int sy, int b0, int b1) {
  const int x1 = sx + 1 < W ? sx + 1 : W - 1;
  const int y0 = sy < 0 ? 0 : (sy > H - 1 ? H - 1 : sy);
  const int y1 = sy + 1 < 0 ? 0 : (sy + 1 > H - 1 ? H - 1 : sy + 1);
  const int h0 = (int)img[(int64_t)y0 * W + sx] * a0 + (int)img[(int64_t)y0 * W + x1] * a1;
  const int h1 = (int)img[(int64_t)y1 * W + sx] * a0 + (int)img[(int64_t)y1 * W + x1] * a1;
  return ((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16)) + 2) >> 2;
}
__global__ __launch_bounds__(256) void k_frame_preprocess(const uint8_t* a, const uint8_t* b, int H, int W, int n_pairs,
                                                           int64_t pair_stride, float* out) {
  const int pair = (int)blockIdx.y;
  const int p = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (p >= 84 * 84 || pair >= n_pairs) return;
  const int dy = p / 84, dx = p - dy * 84;
  int sx, a0, a1, sy, b0, b1;
  rb_resize_tap(dx, 84, W, true, &sx, &a0, &a1);
  rb_resize_tap(dy, 84, H, false, &sy, &b0, &b1);
  int v = rb_resize_pixel(a + pair * pair_stride, H, W, sx, a0, a1, sy, b0, b1);
  if (b) {
    const int w = rb_resize_pixel(b + pair * pair_stride, H, W, sx, a0, a1, sy, b0, b1);
    v = w > v ? w : v;                               // max of the two states == state of the max (x / 255 is monotone)
  }
  out[(int64_t)pair * 84 * 84 + p] = __fdiv_rn((float)(v & 0xFF), 255.0f);    // torch .div_(255)
}

// Bulk append: frames + columns + leaves (any grid), then ancestor rebuild kernels.
__global__ __launch_bounds__(256) void k_append_copy(ReplayView v, int64_t start, const uint8_t* frames,
                                                      const int32_t* timesteps, const int32_t* actions,
                                                      const float* rewards, const uint8_t* nonterminals, int64_t n) {
  constexpr int VEC = RB_FRAME_BYTES / 16;  // 441 uint4 per frame
  const float prio = v.hdr->max;
  for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
    const int64_t idx = (start + i) % v.capacity;
    const uint4* s = (const uint4*)(frames + i * RB_FRAME_BYTES);
    uint4* d = (uint4*)(v.frames + idx * RB_FRAME_BYTES);
    for (int t = (int)threadIdx.x; t < VEC; t += (int)blockDim.x) d[t] = s[t];
    if (threadIdx.x == 0) {
      v.timestep[idx] = timesteps[i];
      v.action[idx] = actions[i];
      v.reward[idx] = rewards[i];
      v.nonterminal[idx] = nonterminals[i] ? 1 : 0;
      v.tree[idx + v.tree_start] = prio;
    }
  }
}

// tree[p] = tree[2p+1] + tree[2p+2] for p in [lo0,hi0] U [lo1,hi1] (inclusive; empty if hi<lo)
__global__ __launch_bounds__(256) void k_rebuild_ranges(float* tree, int64_t lo0, int64_t hi0, int64_t lo1, int64_t hi1) {
  const int64_t n0 = hi0 >= lo0 ? hi0 - lo0 + 1 : 0;
  const int64_t n1 = hi1 >= lo1 ? hi1 - lo1 + 1 : 0;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n0 + n1; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = t < n0 ? lo0 + t : lo1 + (t - n0);
    tree[p] = __fadd_rn(tree[2 * p + 1], tree[2 * p + 2]);
  }
}

// Finishes the rebuild from small ranges up to the root inside one workgroup, then
// publishes the new header.  ranges are NODE ranges at the level to process first.
__global__ __launch_bounds__(1024) void k_rebuild_top(ReplayView v, int64_t lo0, int64_t hi0, int64_t lo1, int64_t hi1,
                                                       int32_t have_ranges, int64_t new_index, int32_t set_full) {
  if (have_ranges) {
    for (;;) {
      const int64_t n0 = hi0 >= lo0 ? hi0 - lo0 + 1 : 0;
      const int64_t n1 = hi1 >= lo1 ? hi1 - lo1 + 1 : 0;
      for (int64_t t = threadIdx.x; t < n0 + n1; t += blockDim.x) {
        const int64_t p = t < n0 ? lo0 + t : lo1 + (t - n0);
        v.tree[p] = __fadd_rn(v.tree[2 * p + 1], v.tree[2 * p + 2]);
      }
      __threadfence_block();
      __syncthreads();
      if (lo0 == 0 || (n0 == 0 && lo1 == 0)) break;  // root done
      if (n0 > 0) { lo0 = (lo0 - 1) / 2; hi0 = (hi0 - 1) / 2; }
      if (n1 > 0) { lo1 = (lo1 - 1) / 2; hi1 = (hi1 - 1) / 2; }
      // merged / overlapping ranges only recompute the same node twice with the same value
    }
  }
  if (threadIdx.x == 0) {
    v.hdr->index = new_index;
    if (set_full) v.hdr->full = 1;
    v.hdr->total = v.tree[0];
  }
}

// ------------------------------------------------------------------------ find --
// SegmentTree._retrieve (memory.py:64-76) for ONE value: float64 value vs float32 nodes,
// strict '>' to go right, float64 subtraction, children clamped on the last internal
// level (memory.py:70-71).  Exactly L steps from the root.
__device__ __forceinline__ int64_t rb_tree_descend(const float* tree, int32_t levels, int64_t tree_start,
                                                   int64_t tree_len, double value) {
  int64_t node = 0;
  for (int32_t lv = 0; lv < levels; ++lv) {
    int64_t left = 2 * node + 1;
    int64_t right = left + 1;
    if (left >= tree_start) {  // children are leaves: bound outliers (memory.py:70-71)
      if (left > tree_len - 1) left = tree_len - 1;
      if (right > tree_len - 1) right = tree_len - 1;
    }
    const float lv_f = tree[left];
    const double lv_d = (double)lv_f;
    const bool go_right = value > lv_d;            // memory.py:73
    node = go_right ? right : left;                // memory.py:74
    if (go_right) value = __dsub_rn(value, lv_d);  // memory.py:75
  }
  return node;
}

__global__ __launch_bounds__(256) void k_find(ReplayView v, const double* values, int32_t n, float* probs,
                                               int64_t* data_idx, int64_t* tree_idx) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n) return;
  const int64_t leaf = rb_tree_descend(v.tree, v.levels, v.tree_start, v.tree_len, values[i]);
  probs[i] = v.tree[leaf];
  data_idx[i] = leaf - v.tree_start;
  tree_idx[i] = leaf;
}

// ---------------------------------------------------------------------- sample --
// Latency-optimised search used by the sampler: identical arithmetic to rb_tree_descend, but
//  (a) the top of the tree (<= 4095 nodes = 16 KB, levels 0..11) is staged once in LDS, and
//  (b) below that, up to five levels are fetched per memory round trip: the descendants of node u at
//      depth j are the 2^j consecutive entries starting at (u+1)*2^j - 1, so 2+4+8+16+32 independent
//      loads replace five dependent ones.  For the 1M-leaf tree: 11 LDS steps + 2 round trips
//      instead of 20 dependent HBM/L2 loads.
// Every load index is clamped to tree_len-1, which IS memory.py:70-71 on the leaf level and a
// no-op above it.
#define RB_TOP_NODES 4095    // levels 0..11 = 16 KB of LDS (16383 nodes = one round trip fewer measured SLOWER: 17.3 vs 15.7 us;
                             // fetching each sample's whole remaining subtree cooperatively into LDS, one trip: 26.9 us)

// D levels of the search with ONE batch of loads.  c holds level j (1..D) at [2^j - 2, 2^(j+1) - 2); `sel` is the path
// taken so far inside the fetched subtree (bit per level).  Register arrays are indexed through select chains only.
// (node indices are 32-bit here: capacity <= 2^30 keeps tree_len below 2^31, and 62 loads with 64-bit address arithmetic
// made a trip instruction-bound — ~1.8 us per trip against ~0.5 us of memory latency)
#define RB_TREE_PAD 64       // floats behind the last node that the search's 16-byte loads may touch (never used as values)
struct rb_f2u { float x, y; };                       // (plain structs: 4-byte alignment, filled with __builtin_memcpy)
struct rb_f4u { float x, y, z, w; };
template <int D>
__device__ __forceinline__ void rb_descend_levels(const float* tree, int32_t& node, double& value, int32_t last, float& nv) {
  float c[(2 << D) - 2];
  // level j = 2^j CONSECUTIVE entries from (node + 1) 2^j - 1 (an odd offset: dword-aligned only), loaded as unaligned 16-byte
  // vectors — 16 load instructions for five levels instead of 62: a wave's 64 samples touch 64 different lines per instruction, so
  // at batch 256 the address unit, not the latency, set the length of a trip (15 us per trip beside the optimiser stream, 6.6 at
  // batch 32).  An entry beyond the last node reads as tree[last] (the clamp above); the vectors themselves start at
  // min(base, last) and may run up to 2^D - 1 entries past the end: the tree buffer is padded for that (RB_TREE_PAD).
#if defined(RB_DESC_SCALAR)      // (variant build for A/B runs: one clamped dword load per entry, the form up to round 5)
#pragma unroll
  for (int j = 1; j <= D; ++j) {
    const uint32_t base = (((uint32_t)node + 1u) << j) - 1u;
#pragma unroll
    for (int t = 0; t < (1 << j); ++t) {
      const uint32_t q = base + (uint32_t)t;
      c[(1 << j) - 2 + t] = tree[q > (uint32_t)last ? (uint32_t)last : q];
    }
  }
#else
  const float t_last = tree[last];
#pragma unroll
  for (int j = 1; j <= D; ++j) {
    const uint32_t base = (((uint32_t)node + 1u) << j) - 1u;
    const float* src = tree + (base > (uint32_t)last ? (uint32_t)last : base);
    if (j == 1) {
      rb_f2u v;
      __builtin_memcpy(&v, src, 8);
      c[0] = base > (uint32_t)last ? t_last : v.x;
      c[1] = base + 1u > (uint32_t)last ? t_last : v.y;
    } else {
#pragma unroll
      for (int t = 0; t < (1 << j); t += 4) {
        rb_f4u v;
        __builtin_memcpy(&v, src + t, 16);
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) c[(1 << j) - 2 + t + u] = base + (uint32_t)(t + u) > (uint32_t)last ? t_last : e[u];
      }
    }
  }
#endif
  int sel = 0;
#pragma unroll
  for (int j = 1; j <= D; ++j) {
    float lf = c[(1 << j) - 2];                       // left child of the current path node: entry 2*sel of level j
#pragma unroll
    for (int t = 1; t < (1 << (j - 1)); ++t) lf = sel == t ? c[(1 << j) - 2 + 2 * t] : lf;
    const double l = (double)lf;
    const bool r = value > l;
    if (r) value = __dsub_rn(value, l);
    const uint32_t nx = 2u * (uint32_t)node + 1u + (r ? 1u : 0u);
    node = (int32_t)(nx > (uint32_t)last ? (uint32_t)last : nx);
    sel = 2 * sel + (r ? 1 : 0);
    if (j == D) {
      nv = c[(1 << j) - 2];
#pragma unroll
      for (int t = 1; t < (1 << j); ++t) nv = sel == t ? c[(1 << j) - 2 + t] : nv;
    }
  }
}

__device__ __forceinline__ int64_t rb_tree_descend_fast(const float* tree, const float* s_top, int n_cached,
                                                        int32_t levels, int64_t tree_len, double value,
                                                        float* node_value) {
  int32_t node = 0;
  int32_t lv = 0;
  float nv = 0.0f;                                  // tree[node] of the node reached (saves the caller a round trip)
  bool have_nv = false;
  const int32_t last = (int32_t)(tree_len - 1);
  for (; lv < levels; ++lv) {                       // LDS phase
    int32_t left = 2 * node + 1, right = left + 1;
    if (left > last) left = last;
    if (right > last) right = last;
    if (right >= n_cached) break;
    const double lv_d = (double)s_top[left];
    const bool go_right = value > lv_d;
    node = go_right ? right : left;
    if (go_right) value = __dsub_rn(value, lv_d);
    nv = s_top[node];
    have_nv = true;
  }
  // global phase: D levels per memory round trip (all 2^(D+1)-2 descendants of the current node are requested at once,
  // level j being the 2^j consecutive entries from (node+1)*2^j - 1), 5 while at least 5 remain: the 9 levels under the
  // LDS top of the 1M-leaf tree take two trips (5 + 4)
  while (lv < levels) {
    const int32_t rem = levels - lv;
    if (rem >= 5) { rb_descend_levels<5>(tree, node, value, last, nv); lv += 5; }
    else if (rem == 4) { rb_descend_levels<4>(tree, node, value, last, nv); lv += 4; }
    else if (rem == 3) { rb_descend_levels<3>(tree, node, value, last, nv); lv += 3; }
    else if (rem == 2) { rb_descend_levels<2>(tree, node, value, last, nv); lv += 2; }
    else { rb_descend_levels<1>(tree, node, value, last, nv); lv += 1; }
    have_nv = true;
  }
  // a child index clamped to the last node may not be the entry that was loaded for the unclamped slot: re-read then
  if (!have_nv || node == last) nv = tree[node];   // rare: explicit branch so the common path carries no load
  *node_value = nv;
  return (int64_t)node;
}

// The same search WITHOUT the LDS top: every level comes from global memory, up to six levels per round trip, the trips
// balanced (20 levels = 5+5+5+5, 17 = 6+6+5).  The nodes of the first trips are the same few KB for every sample and every
// launch (L2-resident, wave-wide broadcast loads); only the last trip reaches rows of the tree that miss.  Measured
// against the LDS-top variant (stage 16 KB, 11 LDS steps, 2 trips): see DESIGN.md §3 sampler row.
template <int DMAX>   // most levels per trip: 6 needs 126 registers for the fetched subtree (the <= 256-thread kernel only)
__device__ __forceinline__ int64_t rb_tree_descend_global(const float* tree, int32_t levels, int64_t tree_len, double value,
                                                          float* node_value) {
  int32_t node = 0;
  int32_t lv = 0;
  float nv = 0.0f;
  const int32_t last = (int32_t)(tree_len - 1);
  while (lv < levels) {
    const int32_t rem = levels - lv;
    const int32_t trips = (rem + DMAX - 1) / DMAX;
    const int32_t d = (rem + trips - 1) / trips;
    switch (d) {
      case 6: if (DMAX >= 6) { rb_descend_levels<(DMAX >= 6 ? 6 : 5)>(tree, node, value, last, nv); break; }
      case 5: rb_descend_levels<5>(tree, node, value, last, nv); break;
      case 4: rb_descend_levels<4>(tree, node, value, last, nv); break;
      case 3: rb_descend_levels<3>(tree, node, value, last, nv); break;
      case 2: rb_descend_levels<2>(tree, node, value, last, nv); break;
      default: rb_descend_levels<1>(tree, node, value, last, nv); break;
    }
    lv += d;
  }
  if (levels == 0 || node == last) nv = tree[node];
  *node_value = nv;
  return (int64_t)node;
}

// Window, scalars and importance weight of ONE sample (memory.py:111-121,140-145,151-153) with every load of the window
// requested in a single batch: NCH chunks of 8 timesteps and 8 rewards (NCH = ceil((h + n) / 8): 1 for n = 3, 3 for the
// data-efficient n = 20 — whose second and third chunk used to be two more dependent round trips each).
template <int NCH>
__device__ __forceinline__ float rb_sample_window(const ReplayView& v, int64_t idx, float prob, float p_total, int32_t full,
                                                  int64_t w_index, float neg_beta_f32, const float* scaling, int32_t* my_win,
                                                  int64_t* action_out, float* return_out, float* nonterminal_out) {
  // ring slots as 32-bit, wrapped with two selects (capacity > window, checked by the host; |offset| < capacity): the
  // 64-bit while-loop form put control flow between the loads, and every load became its own ~0.35 us round trip
  // (5.8 us for the 16 loads of n = 3, 16.8 us for the 48 loads of n = 20 — measured with in-kernel timestamps)
  const int32_t C = (int32_t)v.capacity;
  const int32_t id = (int32_t)idx;
  const int h = v.history, n = v.n;
  const int win_len = h + n;
  auto wrap = [C](int32_t x) { x += x < 0 ? C : 0; x -= x >= C ? C : 0; return x; };
  constexpr int NT = 8 * NCH;
  int ts[NT];
  float rw[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int tc = t < win_len ? t : win_len - 1;
    ts[t] = v.timestep[wrap(id + tc - (h - 1))];
    const int kc = t < n ? t : n - 1;
    rw[t] = v.reward[wrap(id + kc)];
  }
  const int act_now = v.action[id];                                   // slot h-1 is never blanked
  const uint8_t nt_last = v.nonterminal[wrap(id + n)];
  // IS weight while those loads are in flight: probs / p_total ; capacity * probs ; ** -beta (memory.py:151-153).  The
  // reference evaluates the power in float32 (numpy: ~1 ulp, machine dependent); here exp(-beta * log(x)) in float64
  // (relative error ~1e-15, then ONE rounding to float32 — correctly rounded except on near-ties) — a third of the
  // instructions of the general double pow(), which was the longest ALU chain of the kernel.
  float w;
  {
    const float pn = __fdiv_rn(prob, p_total);
    const float cap = (float)(full ? C : w_index);
    const float base = __fmul_rn(cap, pn);
    w = base > 0.0f ? (float)exp((double)neg_beta_f32 * log((double)base)) : (float)pow((double)base, (double)neg_beta_f32);
  }
  unsigned long long first_bits = 0ull;  // h+n <= 64
#pragma unroll
  for (int t = 0; t < NT; ++t)
    if (t < win_len && ts[t] == 0) first_bits |= 1ull << t;
  unsigned long long blank = 0ull;
  for (int t = h - 2; t >= 0; --t) {  // memory.py:116-117
    const bool b = ((blank >> (t + 1)) & 1ull) || ((first_bits >> (t + 1)) & 1ull);
    if (b) blank |= 1ull << t;
  }
  for (int t = h; t < win_len; ++t) {  // memory.py:118-119
    const bool b = ((blank >> (t - 1)) & 1ull) || ((first_bits >> t) & 1ull);
    if (b) blank |= 1ull << t;
  }
  for (int t = 0; t < win_len; ++t) my_win[t] = ((blank >> t) & 1ull) ? -1 : wrap(id + t - (h - 1));
  *action_out = (int64_t)act_now;                                     // memory.py:140
  float R = 0.0f;                                                     // memory.py:142-143, k ascending
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    if (k < n) {
      const float rew = ((blank >> (h - 1 + k)) & 1ull) ? 0.0f : rw[k];
      R = __fadd_rn(R, __fmul_rn(rew, scaling[k]));
    }
  }
  *return_out = R;
  const int t_last = h + n - 1;                                       // memory.py:145
  *nonterminal_out = ((blank >> t_last) & 1ull) ? 0.0f : (nt_last ? 1.0f : 0.0f);
  return w;
}

// ReplayMemory.sample on device (memory.py:124-155).  ONE workgroup, thread i = sample i
// (batch <= 1024).  The rejection loop (memory.py:128-132) runs inside the kernel so the
// steady-state learn step has no host round trip.
#if defined(RB_STAMP)
__device__ long long g_stamp[32];
#define RB_STAMP_AT(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_stamp[i] = wall_clock64(); } while (0)
#else
#define RB_STAMP_AT(i) ((void)0)
#endif
// MAXT = 256 for batches up to 256 (the learn step's shapes): the register budget of a 4-wave workgroup lets a thread hold
// a six-level subtree; MAXT = 1024 (batches up to 1024) keeps to four levels per trip and 128 registers.  NO variant may
// spill: a kernel with a scratch segment slowed every kernel of the step on MI355X (214 -> 283 us per step, measured).
// AU = float4 quadruples per thread of the hosted optimiser workgroups (adam_body.h): they inherit this kernel's register
// allocation, i.e. 2 waves per SIMD under the 256-thread variant's 205 VGPRs (needs AU = 8 to keep enough bytes in flight:
// 42 us per hosted launch against 46 with AU = 4) and 4 under the 1024-thread variant's 127 (AU = 4)
#ifndef RB_HOST_AU_WIDE
#define RB_HOST_AU_WIDE 4      // quadruples per hosted thread under the 1024-thread variant (5 spills under its 128-register cap)
#endif
// learner.hip sizes the pending pass for 4 quadruples per plain thread and 2 (mu, sigma) pairs per pair thread (pair_blk0 and
// the pair grid in clip_adam_impl); the hosting launch rescales the block count by this constant — any other value would split
// plain and pair workgroups differently from what the pass expects (parameters skipped or updated twice)
static_assert(RB_HOST_AU_WIDE == 4, "the hosted optimiser pass is laid out for 4 quadruples per thread (learner.hip clip_adam_impl)");
template <int MAXT>
__device__ __forceinline__ void rb_sample_main(const ReplayView& v, int32_t batch, float neg_beta_arg, const float* neg_beta_ptr,
                                               const double* unit_uniforms, int32_t max_attempts, uint64_t seed, const float* scaling,
                                               int64_t* tree_idx_out, int32_t* win, int64_t* actions_out, float* returns_out,
                                               float* nonterminals_out, float* weights_out, int32_t* fail_count, int32_t lds_top,
                                               int* s_flag, float* s_red, float* s_top, bool top_staged, SpecResult* spec = nullptr,
                                               unsigned spec_epoch = 0u);
__device__ __forceinline__ int rb_poll_epoch(const unsigned* flag, unsigned epoch, int32_t* err_host);
template <int MAXT, int AU>
__global__ __launch_bounds__(MAXT) void k_sample(ReplayView v, int32_t batch, float neg_beta_arg,
                                                  const float* neg_beta_ptr, const double* unit_uniforms, int32_t max_attempts, uint64_t seed,
                                                  const float* scaling, int64_t* tree_idx_out, int32_t* win,
                                                  int64_t* actions_out, float* returns_out, float* nonterminals_out,
                                                  float* weights_out, const NoiseJob* job_dev, float* job_noise, float* job_noise2,
                                                  unsigned long long* job_ctr, int32_t* fail_count, int32_t lds_top,
                                                  int32_t noise_blocks, const ClipAdamArgs* adam_dev, SpecResult* spec,
                                                  unsigned spec_epoch, int32_t spec_mode) {
  if ((int)blockIdx.x > noise_blocks) {
    // co-tenant workgroups behind the noise ones: the previous learn call's optimiser pass (adam_body.h) — independent of
    // this batch's sampling, and 30 us of pure streaming that now runs beside the sampler's serial chain, not before it
    __shared__ float s_adam[18];
#if defined(RB_STAMP)       // slots 6 / 7: start of the first / last hosted workgroup, slot 8: the latest end of any of them
    if (threadIdx.x == 0 && (int)blockIdx.x == noise_blocks + 1) g_stamp[6] = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1) g_stamp[7] = wall_clock64();
#endif
    rb_adam_hosted_block<AU>(adam_dev, (int)blockIdx.x - 1 - noise_blocks, (int)gridDim.x - 1 - noise_blocks, s_adam);
#if defined(RB_STAMP)
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned long long*>(&g_stamp[8]), (unsigned long long)wall_clock64());
#endif
    return;
  }
  if (blockIdx.x > 0) {   // co-tenant workgroups: the learner's noise resample (no dependency on the sampler)
    // the job's SCALARS are read here, from device memory: as a by-value kernel argument its 30 SGPRs were live across the
    // sampler path as well, 17 SGPRs spilled and the kernel carried a private segment (no kernel of the step may: DESIGN.md
    // §6).  Its three POINTERS stay kernel arguments: a pointer loaded from memory is a generic pointer and every access
    // through it a FLAT instruction — this was the only kernel of the library with flat instructions.
    // (field by field, the map by reference: a local copy of the struct would be a dynamically indexed stack object)
    const int nb = (int)blockIdx.x - 1, nblk = job_dev->nblk;
    rb_noise_body(job_noise, job_noise2, nullptr, job_dev->map, job_dev->seed, job_ctr, nb % nblk, nblk, nb / nblk, job_dev->nets);
    return;
  }
  __shared__ int s_flag[16];
  __shared__ float s_red[16];
  __shared__ __attribute__((aligned(16))) float s_top[RB_TOP_NODES + 1];
  // spec_mode (the early draw, replay_internal.h): 1 = THIS is the tentative draw; 2 = an early draw is in flight on another stream
  // and is accepted: wait for it, commit its header effects, done; 3 = in flight but not acceptable (other arguments, or a public
  // entry point): wait, then draw as usual.  FAIL SAFE: when the wait expires, or the pair on the other stream reports that it
  // gave up (SPEC_ABORTED: its gate expired), mode 2 draws here as well — the header was never touched by the tentative draw, so
  // this is exactly the draw a launch without an early draw would have made.
  if (spec_mode >= 2) {
    if (threadIdx.x == 0) {           // ONE lane waits, decides and commits (the decision goes to the others through LDS)
      int accept = rb_poll_epoch(&spec->done, spec_epoch, fail_count ? fail_count + 2 : nullptr);
      if (spec_mode != 2 || spec->status == RB_SPEC_ABORTED) accept = 0;
      if (accept) {
        const int32_t st = spec->status;
        v.hdr->last_attempts = spec->attempts;
        v.hdr->last_status = st;
        v.hdr->rng_counter = spec->rng_next;
        if (st != 0 && fail_count) rb_atomic_inc_system(fail_count);
      }
      s_flag[15] = accept;
    }
    __syncthreads();
    if (s_flag[15]) return;                               // block-uniform
    __syncthreads();                                      // (s_flag is reused by the sampler proper)
  }
  rb_sample_main<MAXT>(v, batch, neg_beta_arg, neg_beta_ptr, unit_uniforms, max_attempts, seed, scaling, tree_idx_out, win, actions_out,
                       returns_out, nonterminals_out, weights_out, fail_count, lds_top, s_flag, s_red, s_top, false,
                       spec_mode == 1 ? spec : nullptr, spec_epoch);
}

// The sampler proper (one workgroup, thread i = sample i): shared by k_sample (block 0) and k_update_sample.  top_staged: the
// caller has already copied the tree top into s_top (and kept it current).
template <int MAXT>
__device__ __forceinline__ void rb_sample_main(const ReplayView& v, int32_t batch, float neg_beta_arg, const float* neg_beta_ptr,
                                               const double* unit_uniforms, int32_t max_attempts, uint64_t seed, const float* scaling,
                                               int64_t* tree_idx_out, int32_t* win, int64_t* actions_out, float* returns_out,
                                               float* nonterminals_out, float* weights_out, int32_t* fail_count, int32_t lds_top,
                                               int* s_flag, float* s_red, float* s_top, bool top_staged, SpecResult* spec,
                                               unsigned spec_epoch) {
  RB_STAMP_AT(0);
  const int i = (int)threadIdx.x;
  const bool active = i < batch;
  const int64_t C = v.capacity;
  const int h = v.history, n = v.n;
  const float neg_beta_f32 = neg_beta_ptr ? *neg_beta_ptr : neg_beta_arg;

  const int n_cached = (int)(v.tree_len < RB_TOP_NODES ? v.tree_len : RB_TOP_NODES);
  if (lds_top && !top_staged) {
    for (int t = 4 * i; t < n_cached; t += 4 * (int)blockDim.x) {        // 16-byte loads (the tree buffer is 16-byte aligned)
      if (t + 3 < n_cached) {
        *reinterpret_cast<float4*>(&s_top[t]) = *reinterpret_cast<const float4*>(&v.tree[t]);
      } else {
        for (int u = t; u < n_cached; ++u) s_top[u] = v.tree[u];
      }
    }
  }
  const int64_t w_index = v.hdr->index;
  const int32_t full = v.hdr->full;
  const uint64_t rng_base = v.hdr->rng_counter;
  const float p_total_g = v.tree[0];
  if (lds_top) __syncthreads();
  RB_STAMP_AT(1);
  const float p_top0 = s_top[0];                                // (an unconditional LDS read: as an operand of the select below the
                                                                //  compiler formed a generic pointer and the kernel's only flat load)
  const float p_total = lds_top ? p_top0 : p_total_g;           // memory.py:149
  // segment_length = p_total / batch_size: float32 / python int -> float32 (NEP 50)
  const float seg_f = __fdiv_rn(p_total, (float)batch);         // memory.py:125
  const double seg = (double)seg_f;
  const double start = __dmul_rn((double)i, seg);               // memory.py:126 (int64 * f32 -> f64)

  int64_t leaf = v.tree_start;
  float prob = 0.0f;
  int attempt = 0;
  int ok = 0;
  for (; attempt < max_attempts; ++attempt) {
    double u;
    if (unit_uniforms) {
      u = active ? unit_uniforms[(int64_t)attempt * batch + i] : 0.0;
    } else {
      const rb_philox_out r = rb_philox(seed, rng_base + (uint64_t)attempt, (uint64_t)i);
      u = rb_u53(r.v[0], r.v[1]);
    }
    // np.random.uniform(0.0, seg, B) = 0.0 + seg*u ; + segment_starts   (memory.py:129)
    const double sample = __dadd_rn(__dadd_rn(0.0, __dmul_rn(seg, u)), start);
    int valid = 1;
    if (active) {
      leaf = lds_top ? rb_tree_descend_fast(v.tree, s_top, n_cached, v.levels, v.tree_len, sample, &prob)
                     : rb_tree_descend_global<(MAXT <= 256 ? 6 : 4)>(v.tree, v.levels, v.tree_len, sample, &prob);   // memory.py:130
      const int64_t idx = leaf - v.tree_start;
      // memory.py:131
      valid = (rb_wrap(w_index, -idx, C) > (int64_t)n) && (rb_wrap(idx, -w_index, C) >= (int64_t)h) &&
              (prob != 0.0f);
    }
    RB_STAMP_AT(2);
    ok = rb_block_all(valid, s_flag);
    if (ok) break;
  }
  RB_STAMP_AT(3);
  const int attempts_used = ok ? attempt + 1 : max_attempts;

  // ---- window (memory.py:111-121), scalars (memory.py:140-145), IS weights (151-154)
  float w = 0.0f;
  if (active) {
    const int64_t idx = leaf - v.tree_start;
    const int win_len = h + n;
    int32_t* my_win = win + (int64_t)i * win_len;
    float nt_f;
    const int nch = (win_len + 7) >> 3;                                 // block-uniform
#define RB_WIN(N) w = rb_sample_window<N>(v, idx, prob, p_total, full, w_index, neg_beta_f32, scaling, my_win, &actions_out[i], &returns_out[i], &nt_f)
    if (nch <= 1) RB_WIN(1);
    else if (nch <= 3 || MAXT > 256) RB_WIN(3);      // (the 1024-thread variant has no registers for longer windows in one batch ...
    else RB_WIN(8);                                  //  ... rb_replay_sample refuses batch > 256 with history + multi_step > 24)
#undef RB_WIN
    nonterminals_out[i] = nt_f;
    // a draw that gave up marks its own index buffer: the write-back of THIS buffer's batch is dropped (rb_update_body), no other
    tree_idx_out[i] = ok ? leaf : (int64_t)-1;
  }
  RB_STAMP_AT(4);
  const float w_max = rb_block_max(active ? w : -INFINITY, s_red);
  // The reference retries until a batch is valid (memory.py:128-132); this loop is bounded.  If the bound is hit (a
  // buffer too small for the batch: some stratum lies inside the write head's exclusion zone) the last draw is NOT a
  // legal batch — windows may straddle the write head and a zero-priority leaf would give w = inf.  Make it harmless:
  // every importance weight is 0, so the step's gradient is exactly zero, and the failure is counted in host-visible
  // memory (rb_replay_failed_samples) so the caller can raise without a device synchronisation.
  if (active) weights_out[i] = ok ? __fdiv_rn(w, w_max) : 0.0f;         // memory.py:154
  if (spec) {
    // a TENTATIVE draw (rb_replay_spec_launch): the header is not touched — what the draw would have done to it goes to the side
    // record, committed by the draw that accepts it (k_sample, spec_mode 2).  The record's epoch is stored LAST, behind an
    // agent-scope release of everything this workgroup wrote: the accepting workgroup polls it from another stream.
    if (threadIdx.x == 0) {
      spec->attempts = attempts_used;
      spec->status = ok ? 0 : 1;
      spec->rng_next = rng_base + (uint64_t)attempts_used;
    }
#if defined(RB_HOST_INTERP)
    __syncthreads();
    if (threadIdx.x == 0) spec->done = spec_epoch;
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(&spec->done, spec_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#endif
    return;
  }
  if (threadIdx.x == 0) {
    v.hdr->last_attempts = attempts_used;
    v.hdr->last_status = ok ? 0 : 1;
    if (!unit_uniforms) v.hdr->rng_counter = rng_base + (uint64_t)attempts_used;
    // (a system-scope atomic on the kernel-argument pointer: a global instruction; the former volatile read-modify-write was
    // compiled to flat loads/stores)
    if (!ok && fail_count) rb_atomic_inc_system(fail_count);
  }
  RB_STAMP_AT(5);
}

// wait for *flag >= epoch (ONE lane polls, relaxed; then one agent-scope acquire; the caller broadcasts).  Returns 1 when the flag
// arrived, 0 when the bound (RB_WAIT_EPOCH_POLLS polls, ~2 ms) expired — counted in *err_host.  The producers are launches submitted
// EARLIER (a head kernel, an early draw), so an expiry means something serialises the two queues against each other (a profiler's
// counter pass does) or the device is wedged; every caller FAILS SAFE: it does the work in its own launch instead (k_sample) or
// drops it and says so (k_spec_gate -> k_update_sample), never proceeds on data that may not be final.
#define RB_WAIT_EPOCH_POLLS (1u << 13)
// (the polling lane's part: returns 1 when the flag arrived; ends with the agent-scope acquire either way)
__device__ __forceinline__ int rb_poll_epoch(const unsigned* flag, unsigned epoch, int32_t* err_host) {
#if defined(RB_HOST_INTERP)
  const int ok = (int)(*flag - epoch) >= 0;             // launches run in submission order there
  if (!ok && err_host) *err_host = *err_host + 1;
  return ok;
#else
  unsigned spins = 0;
  int ok = 1;
  while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0) {
    __builtin_amdgcn_s_sleep(8);
    if (++spins > RB_WAIT_EPOCH_POLLS) { ok = 0; if (err_host) rb_atomic_inc_system(err_host); break; }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return ok;
#endif
}
// Frame-stack gather (memory.py:136-138 minus the /255): block = (sample, stack slot),
// 441 sixteen-byte lanes per 7056-byte frame, zero fill for blanked slots.
__global__ __launch_bounds__(256) void k_gather_stacks(ReplayView v, int32_t batch, const int32_t* win,
                                                        uint8_t* states, uint8_t* next_states) {
  constexpr int VEC = RB_FRAME_BYTES / 16;
  const int h = v.history, n = v.n;
  const int per_sample = 2 * h;
  for (int b = (int)blockIdx.x; b < batch * per_sample; b += (int)gridDim.x) {
    const int i = b / per_sample;
    const int s = b % per_sample;
    const bool is_next = s >= h;
    const int c = is_next ? s - h : s;
    const int slot = is_next ? n + c : c;
    const int32_t ring = win[(int64_t)i * (h + n) + slot];
    uint4* d = (uint4*)((is_next ? next_states : states) + ((int64_t)i * h + c) * RB_FRAME_BYTES);
    if (ring < 0) {
      const uint4 z = make_uint4(0u, 0u, 0u, 0u);
      for (int t = (int)threadIdx.x; t < VEC; t += (int)blockDim.x) d[t] = z;
    } else {
      const uint4* src = (const uint4*)(v.frames + (int64_t)ring * RB_FRAME_BYTES);
      for (int t = (int)threadIdx.x; t < VEC; t += (int)blockDim.x) d[t] = src[t];
    }
  }
}

// ---------------------------------------------------------------------- update --
// (body: replay_internal.h)
__global__ __launch_bounds__(1024) void k_update(ReplayView v, const int64_t* tree_idx, const float* values, int32_t n,
                                                  int32_t apply_pow, double omega) {
  __shared__ float lds[UpdateLds<2048, 1024>::WORDS];
  rb_update_auto<2048, 1024>(v, tree_idx, values, n, apply_pow, omega, lds);
}

// ------------------------------------------------------------ update + sample --
// update_priorities(idx_k, loss_k) followed by sample(k + 1) — the PER loop of memory.py:148-159 / agent.py:62,100 — as ONE
// launch of one workgroup: the two are a dependent pair of single-workgroup latency chains, and as two launches the second
// pays a launch boundary, re-reads the header and stages the 16 KB tree top that the first has just rewritten.  Here the top
// is staged while the update's operands are in flight, the sorted-batch update (rb_update_sorted_wave, one wave) patches that
// LDS copy as it writes the tree, and the search starts from it.  Unsorted or longer batches (<= 256) take the hashed body
// and the top is staged afterwards.  Same arithmetic, same order: tree, header and batch are bit-identical to the two calls.
__global__ __launch_bounds__(256) void k_update_sample(ReplayView v, const int64_t* upd_idx, const float* upd_val, int32_t upd_n,
                                                        int32_t apply_pow, double omega, int32_t batch, float neg_beta_arg,
                                                        const float* neg_beta_ptr, const double* unit_uniforms, int32_t max_attempts,
                                                        uint64_t seed, const float* scaling, int64_t* tree_idx_out, int32_t* win,
                                                        int64_t* actions_out, float* returns_out, float* nonterminals_out,
                                                        float* weights_out, int32_t* fail_count, SpecResult* spec, unsigned spec_epoch) {
  __shared__ int s_flag[16];
  __shared__ float s_red[16];
  __shared__ __attribute__((aligned(16))) float s_top[RB_TOP_NODES + 1];
  __shared__ float lds_upd[UpdateLds<512, 256>::WORDS];
  __shared__ int s_sorted;
  const int i = (int)threadIdx.x;
  if (spec) {
    // the early pair (rb_replay_spec_launch): the gate in front of this launch (same stream) waited for the head kernel of the learn
    // call whose losses are written back here.  If the gate EXPIRED the losses may not be final: give up — no write-back (counted as a
    // dropped one), no draw; the record says so and the accepting sampler launch draws itself (k_sample, spec_mode 2)
    if (i == 0) {
#if defined(RB_HOST_INTERP)
      s_flag[0] = spec->abort_epoch == spec_epoch ? 1 : 0;
#else
      s_flag[0] = __hip_atomic_load(&spec->abort_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == spec_epoch ? 1 : 0;
#endif
    }
    __syncthreads();
    const int aborted = s_flag[0];
    __syncthreads();
    if (aborted) {                                        // block-uniform
      if (i == 0) {
        spec->attempts = 0; spec->status = RB_SPEC_ABORTED;
        if (v.dropped) rb_atomic_inc_system(v.dropped);
#if defined(RB_HOST_INTERP)
        spec->done = spec_epoch;
#else
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&spec->done, spec_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
      }
      return;
    }
  }
  const int n_cached = (int)(v.tree_len < RB_TOP_NODES ? v.tree_len : RB_TOP_NODES);
  UpdateOperand op;
  op.node = -1; op.val = 0.0f; op.status = 0; op.sorted = 0;
  if (upd_n <= 64) {
    if (i < 64) op = rb_update_load(v, upd_idx, upd_val, upd_n);
    if (i == 0) s_sorted = op.sorted;
  } else if (i == 0) {
    s_sorted = 0;
  }
  for (int t = 4 * i; t < n_cached; t += 4 * (int)blockDim.x) {          // (same staging as rb_sample_main)
    if (t + 3 < n_cached) {
      *reinterpret_cast<float4*>(&s_top[t]) = *reinterpret_cast<const float4*>(&v.tree[t]);
    } else {
      for (int u = t; u < n_cached; ++u) s_top[u] = v.tree[u];
    }
  }
  __syncthreads();
  const bool sorted = s_sorted != 0;                                       // block-uniform
  if (sorted) {
    if (i < 64) rb_update_sorted_wave(v, op, upd_n, apply_pow, omega, s_top, n_cached);
  } else {
    rb_update_body<512, 256>(v, upd_idx, upd_val, upd_n, apply_pow, omega, lds_upd);
  }
  __threadfence_block();               // the tree this workgroup wrote, read back by the same workgroup (as in k_rebuild_top)
  __syncthreads();
  rb_sample_main<256>(v, batch, neg_beta_arg, neg_beta_ptr, unit_uniforms, max_attempts, seed, scaling, tree_idx_out, win, actions_out,
                      returns_out, nonterminals_out, weights_out, fail_count, 1, s_flag, s_red, s_top, sorted, spec, spec_epoch);
}

// -------------------------------------------------------------- validation view --
// ReplayMemory.__next__ (memory.py:167-178): history stack ending at data index i.
// NOTE the reference indexes data[i-h+1 .. i] with numpy negative wrap-around, not % C.
__global__ __launch_bounds__(256) void k_state_at(ReplayView v, int64_t data_index, float* out) {
  __shared__ int s_blank[64];
  const int h = v.history;
  if (threadIdx.x == 0) {
    int blank_next = 0;
    s_blank[h - 1] = 0;
    for (int t = h - 2; t >= 0; --t) {
      const int64_t ring_next = rb_floor_mod(data_index - (h - 1) + (t + 1), v.capacity);
      const int b = blank_next || (v.timestep[ring_next] == 0);
      s_blank[t] = b;
      blank_next = b;
    }
  }
  __syncthreads();
  for (int t = 0; t < h; ++t) {
    const int64_t ring = rb_floor_mod(data_index - (h - 1) + t, v.capacity);
    const uint8_t* src = v.frames + ring * RB_FRAME_BYTES;
    float* dst = out + (int64_t)t * RB_FRAME_BYTES;
    const bool blank = s_blank[t] != 0;
    for (int p = (int)threadIdx.x; p < RB_FRAME_BYTES; p += (int)blockDim.x)
      dst[p] = blank ? 0.0f : __fdiv_rn((float)src[p], 255.0f);
  }
}

// The same for n data indices at once (one workgroup per state): the validation pass of test.py:38-39 walks the whole
// validation memory — one launch instead of one launch + host loop per state.
__global__ __launch_bounds__(256) void k_states_at(ReplayView v, const int64_t* data_index, float* out) {
  __shared__ int s_blank[64];
  const int h = v.history;
  const int64_t di = data_index[blockIdx.x];
  float* o = out + (int64_t)blockIdx.x * h * RB_FRAME_BYTES;
  if (threadIdx.x == 0) {
    int blank_next = 0;
    s_blank[h - 1] = 0;
    for (int t = h - 2; t >= 0; --t) {
      const int64_t ring_next = rb_floor_mod(di - (h - 1) + (t + 1), v.capacity);
      const int b = blank_next || (v.timestep[ring_next] == 0);
      s_blank[t] = b;
      blank_next = b;
    }
  }
  __syncthreads();
  for (int t = 0; t < h; ++t) {
    const int64_t ring = rb_floor_mod(di - (h - 1) + t, v.capacity);
    const uint8_t* src = v.frames + ring * RB_FRAME_BYTES;
    float* dst = o + (int64_t)t * RB_FRAME_BYTES;
    const bool blank = s_blank[t] != 0;
    for (int p = (int)threadIdx.x; p < RB_FRAME_BYTES; p += (int)blockDim.x)
      dst[p] = blank ? 0.0f : __fdiv_rn((float)src[p], 255.0f);
  }
}

__global__ __launch_bounds__(256) void k_u8_to_unit(const uint8_t* src, float* dst, int64_t n) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
    dst[t] = __fdiv_rn((float)src[t], 255.0f);  // memory.py:137 .div_(255)
}

// ================================================================ host entry points
// Live handles, so that a raw header restore through rb_copy_to_device (state load, main.py:118) refreshes the host
// mirror of index/full that rb_replay_append_batch plans its ancestor rebuild from.
static rb_replay* g_live[64];
static void live_add(rb_replay* r) { for (auto& p : g_live) if (!p) { p = r; return; } }
static void live_del(rb_replay* r) { for (auto& p : g_live) if (p == r) p = nullptr; }
void rb_replay_note_device_write(void* dst_dev, const void* src_host, size_t nbytes) {
  for (rb_replay* r : g_live) {
    if (!r || !r->hdr) continue;
    const char* h = (const char*)r->hdr;
    const char* d = (const char*)dst_dev;
    if (d <= h && h + sizeof(rb_replay_header_t) <= d + nbytes) {
      rb_replay_header_t hd;
      memcpy(&hd, (const char*)src_host + (h - d), sizeof(hd));
      r->host_index = hd.index;
      r->host_full = hd.full;
    }
  }
}

// One wave, no LDS, a handful of registers: holds the replay's stream back until *flag >= epoch.  The early pair itself must not
// do the waiting: submitted a whole step ahead of the device, a 256-thread / 50 KB workgroup polling on a CU takes that CU away
// from every launch of the step that needs all 256 (the batch-256 conv kernels are one LDS-filling workgroup per CU: each of them
// ran a second round for ONE workgroup — 498 -> 650 us per step, measured); a lone wave fits beside anything.
__global__ __launch_bounds__(64) void k_spec_gate(const unsigned* flag, unsigned epoch, int32_t* err_host, SpecResult* spec, unsigned spec_epoch) {
  // (ONE wave, no LDS, no barrier: with a shared word the gate stopped fitting beside the LDS-filling conv workgroups of batch 256 —
  // every conv launch ran a second round for the one workgroup of the gate's CU: 498 -> 537 us per step, round6_second_trace_b256_spec)
  if (threadIdx.x != 0) return;
  if (!rb_poll_epoch(flag, epoch, err_host)) {
    // the head kernel's launch was not seen to complete within the bound: the pair behind this gate must NOT read its losses.  It is
    // told to give up (k_update_sample: no write-back — counted as dropped — and no draw; the accepting launch draws itself)
#if defined(RB_HOST_INTERP)
    spec->abort_epoch = spec_epoch;
#else
    __hip_atomic_store(&spec->abort_epoch, spec_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
  }
}

int rb_replay_spec_inflight(rb_replay_t* r) { return r ? r->spec_inflight : 0; }
const int32_t* rb_replay_current_windows(rb_replay_t* r) { return win_of(r, r->win_sel); }
unsigned long long rb_replay_mutations(rb_replay_t* r) { return r->mutations; }

// (replay_internal.h) the write-back of learn call k + the tentative draw of call k + 1 on the replay's own stream
int rb_replay_spec_allowed(rb_replay_t* r) {
  if (!r) return 0;
  if (*(volatile int32_t*)(r->fail_host + 2) != 0) r->spec_disabled = 1;   // an expired wait: the handle stays without early draws
  return r->spec_disabled ? 0 : 1;
}
void rb_replay_spec_arm_accept(rb_replay_t* r) { r->spec_accept_armed = 1; }

int rb_replay_spec_launch(rb_replay_t* r, const rb_spec_request& q) {
  RB_REQUIRE(r && q.upd_idx && q.upd_loss && q.tree_idx && q.actions && q.returns && q.nonterminals && q.weights, "rb_replay_spec_launch: NULL argument");
  RB_REQUIRE(q.batch >= 1 && q.batch <= 256 && q.upd_n >= 1 && q.upd_n <= 256, "rb_replay_spec_launch: batch and upd_n must be in [1,256]");
  RB_REQUIRE(r->capacity > (int64_t)r->history + r->n, "rb_replay_spec_launch: capacity must exceed history + multi_step");
  RB_SPEC_JOIN(r);
#if defined(RB_HOST_INTERP)
  hipStream_t s2 = nullptr;
#else
  if (!r->spec_stream) RB_HIP_TRY(hipStreamCreateWithFlags(&r->spec_stream, hipStreamNonBlocking));
  hipStream_t s2 = r->spec_stream;
#endif
  const ReplayView v = view_of(r);
  const int win_set = r->win_sel ^ 1;            // the table the step in flight still reads is left alone
  const float neg_beta = (float)(-q.priority_weight);
  const unsigned epoch = ++r->spec_epoch;
  ++r->mutations;
  if (q.go_flag) {
    RB_LAUNCH(k_spec_gate, dim3(1), dim3(64), s2, q.go_flag, q.go_epoch, r->fail_host + 2, r->spec_res, epoch);
    RB_LAUNCH_CHECK();
  }
  // one launch (latency is no concern on this stream): sorted batches of up to 64 leaves take the one-wave write-back, the rest
  // the hashed body inside the same kernel
  RB_LAUNCH(k_update_sample, dim3(1), dim3(256), s2, v, q.upd_idx, q.upd_loss, q.upd_n, 1, r->omega, q.batch, neg_beta, r->neg_beta_dev,
            (const double*)nullptr, q.max_attempts, r->seed, r->scaling_dev, q.tree_idx, win_of(r, win_set), q.actions, q.returns,
            q.nonterminals, q.weights, r->fail_host, r->spec_res, epoch);
  RB_LAUNCH_CHECK();
  r->spec_args.batch = q.batch; r->spec_args.max_attempts = q.max_attempts; r->spec_args.win_set = win_set;
  r->spec_args.beta = q.priority_weight; r->spec_args.tree_idx = q.tree_idx; r->spec_args.actions = q.actions;
  r->spec_args.returns = q.returns; r->spec_args.nonterm = q.nonterminals; r->spec_args.weights = q.weights;
  r->spec_inflight = 1;
  return RB_OK;
}

int rb_replay_internal_view(rb_replay_t* r, ReplayView* view, double* omega) {
  if (!r || !view || !omega) return RB_ERR_INVALID;
  *view = view_of(r);
  *omega = r->omega;
  return RB_OK;
}

extern "C" {

int rb_replay_create(rb_replay_t** out, int64_t capacity, int32_t history, int32_t multi_step, double discount,
                     double priority_exponent, uint64_t seed) {
  RB_REQUIRE(out != nullptr, "rb_replay_create: out is NULL");
  RB_REQUIRE(capacity >= 2 && (capacity % 2) == 0,
             "rb_replay_create: capacity must be even and >= 2 (the reference's sum-tree walk reads past the "
             "array for odd sizes, memory.py:38-39), got %lld", (long long)capacity);
  RB_REQUIRE(capacity <= ((int64_t)1 << 30), "rb_replay_create: capacity too large");
  RB_REQUIRE(history >= 1 && multi_step >= 1 && history + multi_step <= 64, "rb_replay_create: need history>=1, multi_step>=1, history+multi_step<=64");
  rb_replay* r = new (std::nothrow) rb_replay();
  if (!r) { rb_set_error("rb_replay_create: host OOM"); return RB_ERR_OOM; }
  r->capacity = capacity; r->history = history; r->n = multi_step; r->omega = priority_exponent; r->seed = seed;
  int32_t L = 0;
  while (((int64_t)1 << L) < capacity) ++L;  // (capacity-1).bit_length()   memory.py:17
  r->levels = L;
  r->tree_start = ((int64_t)1 << L) - 1;
  r->tree_len = r->tree_start + capacity;
  for (int k = 0; k < multi_step; ++k) r->scaling[k] = (float)pow(discount, (double)k);  // memory.py:101
  r->max_batch = 1024;
  r->neg_beta_dev = nullptr;
  r->host_index = 0; r->host_full = 0;
  r->tree = nullptr; r->frames = nullptr; r->timestep = nullptr; r->action = nullptr; r->reward = nullptr;
  r->nonterminal = nullptr; r->hdr = nullptr; r->win = nullptr; r->scaling_dev = nullptr; r->fail_host = nullptr;
  r->win2 = nullptr; r->win_sel = 0; r->spec_stream = nullptr; r->spec_res = nullptr; r->spec_epoch = 0; r->spec_inflight = 0; r->spec_accept_armed = 0; r->spec_disabled = 0;
  r->mutations = 0;
#define RB_ALLOC(ptr, bytes)                                                                      \
  do {                                                                                            \
    hipError_t e_ = rb_dev_malloc((void**)&(ptr), (size_t)(bytes));                                   \
    if (e_ != hipSuccess) {                                                                       \
      rb_set_error("rb_replay_create: hipMalloc(%lld B) failed: %s", (long long)(bytes), hipGetErrorString(e_)); \
      rb_replay_destroy(r);                                                                       \
      return RB_ERR_OOM;                                                                          \
    }                                                                                             \
  } while (0)
  RB_ALLOC(r->tree, (r->tree_len + RB_TREE_PAD) * sizeof(float));
  RB_ALLOC(r->frames, capacity * (int64_t)RB_FRAME_BYTES);
  RB_ALLOC(r->timestep, capacity * sizeof(int32_t));
  RB_ALLOC(r->action, capacity * sizeof(int32_t));
  RB_ALLOC(r->reward, capacity * sizeof(float));
  RB_ALLOC(r->nonterminal, capacity);
  RB_ALLOC(r->hdr, sizeof(rb_replay_header_t));
  RB_ALLOC(r->win, (int64_t)r->max_batch * 64 * sizeof(int32_t));
  RB_ALLOC(r->win2, (int64_t)r->max_batch * 64 * sizeof(int32_t));
  RB_ALLOC(r->spec_res, sizeof(SpecResult));
  RB_ALLOC(r->scaling_dev, 64 * sizeof(float));
#undef RB_ALLOC
  {
    hipError_t e_ = hipHostMalloc((void**)&r->fail_host, 4 * sizeof(int32_t), hipHostMallocMapped);
    if (e_ != hipSuccess) {
      rb_set_error("rb_replay_create: hipHostMalloc failed: %s", hipGetErrorString(e_));
      rb_replay_destroy(r);
      return RB_ERR_OOM;
    }
    r->fail_host[0] = 0; r->fail_host[1] = 0; r->fail_host[2] = 0; r->fail_host[3] = 0;
  }
  // blank_trans everywhere (memory.py:8,19), zero tree (memory.py:18)
  RB_HIP_TRY(hipMemset(r->tree, 0, (r->tree_len + RB_TREE_PAD) * sizeof(float)));
  RB_HIP_TRY(hipMemset(r->frames, 0, capacity * (int64_t)RB_FRAME_BYTES));
  RB_HIP_TRY(hipMemset(r->timestep, 0, capacity * sizeof(int32_t)));
  RB_HIP_TRY(hipMemset(r->action, 0, capacity * sizeof(int32_t)));
  RB_HIP_TRY(hipMemset(r->reward, 0, capacity * sizeof(float)));
  RB_HIP_TRY(hipMemset(r->nonterminal, 0, capacity));
  RB_HIP_TRY(hipMemcpy(r->scaling_dev, r->scaling, 64 * sizeof(float), hipMemcpyHostToDevice));
  RB_HIP_TRY(hipMemset(r->spec_res, 0, sizeof(SpecResult)));
  RB_LAUNCH(k_replay_init, dim3(1), dim3(64), nullptr, r->hdr);
  RB_LAUNCH_CHECK();
  RB_HIP_TRY(hipDeviceSynchronize());
  live_add(r);
  *out = r;
  return RB_OK;
}

int rb_replay_destroy(rb_replay_t* r) {
  if (!r) return RB_OK;
  (void)spec_join(r);
#if !defined(RB_HOST_INTERP)
  if (r->spec_stream) (void)hipStreamDestroy(r->spec_stream);
#endif
  live_del(r);
  if (r->win2) rb_dev_free(r->win2);
  if (r->spec_res) rb_dev_free(r->spec_res);
  if (r->tree) rb_dev_free(r->tree);
  if (r->frames) rb_dev_free(r->frames);
  if (r->timestep) rb_dev_free(r->timestep);
  if (r->action) rb_dev_free(r->action);
  if (r->reward) rb_dev_free(r->reward);
  if (r->nonterminal) rb_dev_free(r->nonterminal);
  if (r->hdr) rb_dev_free(r->hdr);
  if (r->win) rb_dev_free(r->win);
  if (r->scaling_dev) rb_dev_free(r->scaling_dev);
  if (r->fail_host) (void)hipHostFree(r->fail_host);
  delete r;
  return RB_OK;
}

int rb_replay_buffers(rb_replay_t* r, rb_replay_buffers_t* o) {
  RB_REQUIRE(r && o, "rb_replay_buffers: NULL argument");
  RB_SPEC_JOIN(r);
  o->sum_tree_dev = r->tree; o->tree_len = r->tree_len; o->tree_start = r->tree_start;
  o->frames_dev = r->frames; o->timestep_dev = r->timestep; o->action_dev = r->action;
  o->reward_dev = r->reward; o->nonterminal_dev = r->nonterminal; o->header_dev = r->hdr;
  o->window_dev = r->win; o->window_len = r->history + r->n;
  return RB_OK;
}

int rb_replay_header(rb_replay_t* r, rb_replay_header_t* o, rb_stream_t stream) {
  RB_REQUIRE(r && o, "rb_replay_header: NULL argument");
  RB_SPEC_JOIN(r);
  RB_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  RB_HIP_TRY(hipMemcpy(o, r->hdr, sizeof(*o), hipMemcpyDeviceToHost));
  r->host_index = o->index;  // resynchronise the mirror (e.g. after a state restore)
  r->host_full = o->full;
  return RB_OK;
}

int rb_replay_append(rb_replay_t* r, const float* state_dev, int32_t timestep, int32_t action, float reward,
                     int32_t nonterminal, rb_stream_t stream) {
  RB_REQUIRE(r && state_dev, "rb_replay_append: NULL argument");
  RB_SPEC_JOIN(r);
  ++r->mutations;
  const float* last = state_dev + (int64_t)(r->history - 1) * RB_FRAME_BYTES;  // state[-1], memory.py:106
  RB_LAUNCH(k_append_one, dim3(1), dim3(256), stream, view_of(r), last, timestep, action, reward, nonterminal);
  RB_LAUNCH_CHECK();
  r->host_index = (r->host_index + 1) % r->capacity;
  if (r->host_index == 0) r->host_full = 1;
  return RB_OK;
}

int rb_replay_append_batch(rb_replay_t* r, const uint8_t* frames_dev, const int32_t* timesteps_dev,
                           const int32_t* actions_dev, const float* rewards_dev, const uint8_t* nonterminals_dev,
                           int64_t n, rb_stream_t stream) {
  RB_REQUIRE(r && frames_dev && timesteps_dev && actions_dev && rewards_dev && nonterminals_dev,
             "rb_replay_append_batch: NULL argument");
  RB_SPEC_JOIN(r);
  ++r->mutations;
  RB_REQUIRE(n >= 0 && n <= r->capacity, "rb_replay_append_batch: n must be in [0, capacity]");
  if (n == 0) return RB_OK;
  const ReplayView v = view_of(r);
  const int64_t start = r->host_index;
  const int grid = (int)(n < 4096 ? n : 4096);
  RB_LAUNCH(k_append_copy, dim3(grid), dim3(256), stream, v, start, frames_dev, timesteps_dev, actions_dev,
            rewards_dev, nonterminals_dev, n);
  RB_LAUNCH_CHECK();
  // leaf ranges (tree indices), possibly wrapping around the ring
  int64_t lo0 = r->tree_start + start, hi0, lo1 = 0, hi1 = -1;
  if (start + n <= r->capacity) {
    hi0 = lo0 + n - 1;
  } else {
    hi0 = r->tree_start + r->capacity - 1;
    lo1 = r->tree_start;
    hi1 = r->tree_start + (start + n - r->capacity) - 1;
  }
  // parents of the leaf ranges
  lo0 = (lo0 - 1) / 2; hi0 = (hi0 - 1) / 2;
  if (hi1 >= lo1) { lo1 = (lo1 - 1) / 2; hi1 = (hi1 - 1) / 2; }
  for (;;) {
    const int64_t cnt = (hi0 - lo0 + 1) + (hi1 >= lo1 ? hi1 - lo1 + 1 : 0);
    if (cnt <= 2048 || lo0 == 0) break;
    const int g = (int)rb_div_up(cnt, 256);
    RB_LAUNCH(k_rebuild_ranges, dim3(g > 2048 ? 2048 : g), dim3(256), stream, r->tree, lo0, hi0, lo1, hi1);
    RB_LAUNCH_CHECK();
    lo0 = (lo0 - 1) / 2; hi0 = (hi0 - 1) / 2;
    if (hi1 >= lo1) { lo1 = (lo1 - 1) / 2; hi1 = (hi1 - 1) / 2; }
  }
  const int64_t new_index = (start + n) % r->capacity;
  const int32_t set_full = (start + n >= r->capacity) ? 1 : 0;
  RB_LAUNCH(k_rebuild_top, dim3(1), dim3(1024), stream, v, lo0, hi0, lo1, hi1, 1, new_index, set_full);
  RB_LAUNCH_CHECK();
  r->host_index = new_index;
  if (set_full) r->host_full = 1;
  return RB_OK;
}

#if defined(RB_STAMP)
int rb_debug_stamps(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp), sizeof(long long) * 32) == hipSuccess ? 0 : -2; }
#endif
int rb_replay_failed_samples(rb_replay_t* r, int64_t* count) {
  RB_REQUIRE(r && count, "rb_replay_failed_samples: NULL argument");
  *count = (int64_t)*(volatile int32_t*)r->fail_host;   // pinned host word the sampler increments: no synchronisation
  return RB_OK;
}

int rb_replay_dropped_updates(rb_replay_t* r, int64_t* count) {
  RB_REQUIRE(r && count, "rb_replay_dropped_updates: NULL argument");
  *count = (int64_t)*(volatile int32_t*)(r->fail_host + 1);   // pinned host word: no synchronisation
  return RB_OK;
}

int rb_replay_expired_waits(rb_replay_t* r, int64_t* count) {
  RB_REQUIRE(r && count, "rb_replay_expired_waits: NULL argument");
  *count = (int64_t)*(volatile int32_t*)(r->fail_host + 2);   // pinned host word: no synchronisation
  return RB_OK;
}

int rb_replay_reset_failed_samples(rb_replay_t* r) {
  RB_REQUIRE(r != nullptr, "rb_replay_reset_failed_samples: NULL handle");
  RB_SPEC_JOIN(r);                             // (an early pair in flight may still count)
  *(volatile int32_t*)r->fail_host = 0;        // (a failed launch still in flight re-increments it when it completes)
  *(volatile int32_t*)(r->fail_host + 1) = 0;
  *(volatile int32_t*)(r->fail_host + 2) = 0;
  r->spec_disabled = 0;
  return RB_OK;
}

int rb_replay_position(rb_replay_t* r, int64_t* index, int32_t* full) {
  RB_REQUIRE(r != nullptr, "rb_replay_position: NULL handle");
  if (index) *index = r->host_index;
  if (full) *full = r->host_full;
  return RB_OK;
}

int rb_replay_set_beta_source(rb_replay_t* r, const float* neg_beta_dev) {
  RB_REQUIRE(r != nullptr, "rb_replay_set_beta_source: NULL handle");
  r->neg_beta_dev = neg_beta_dev;
  return RB_OK;
}

int rb_replay_find(rb_replay_t* r, const double* values_dev, int32_t n, float* probs_dev, int64_t* data_idx_dev,
                   int64_t* tree_idx_dev, rb_stream_t stream) {
  RB_REQUIRE(r && values_dev && probs_dev && data_idx_dev && tree_idx_dev, "rb_replay_find: NULL argument");
  RB_SPEC_JOIN(r);
  if (n <= 0) return RB_OK;
  RB_LAUNCH(k_find, dim3((unsigned)rb_div_up(n, 256)), dim3(256), stream, view_of(r), values_dev, n, probs_dev,
            data_idx_dev, tree_idx_dev);
  RB_LAUNCH_CHECK();
  return RB_OK;
}

// the pending optimiser pass of the learner (adam_body.h) as a launch of its own: what sample_impl falls back to when the
// sampler variant that can host it does not fit the replay's window length
__global__ __launch_bounds__(256) void k_adam_pending(const ClipAdamArgs* ad) {
  __shared__ float s_adam[18];
  rb_adam_hosted_block<4>(ad, (int)blockIdx.x, (int)gridDim.x, s_adam);
}

static int sample_impl(rb_replay_t* r, int32_t batch, double priority_weight, const double* unit_uniforms_dev,
                       int32_t max_attempts, int64_t* tree_idx_dev, uint8_t* states_dev, uint8_t* next_states_dev,
                       int64_t* actions_dev, float* returns_dev, float* nonterminals_dev, float* weights_dev,
                       const rb_noise_job_t* noise_job, rb_stream_t stream) {
  RB_REQUIRE(r && tree_idx_dev && actions_dev && returns_dev && nonterminals_dev && weights_dev,
             "rb_replay_sample: NULL argument");
  RB_REQUIRE(batch >= 1 && batch <= r->max_batch, "rb_replay_sample: batch must be in [1,%d]", r->max_batch);
  RB_REQUIRE(max_attempts >= 1, "rb_replay_sample: max_attempts must be >= 1");
  RB_REQUIRE(r->capacity > (int64_t)r->history + r->n,
             "rb_replay_sample: capacity must exceed history + multi_step (no window can clear the write head otherwise: "
             "the reference's rejection loop, memory.py:128-132, would never end)");
  // an early draw in flight (rb_replay_spec_launch): this launch's sampler workgroup waits for it and either ACCEPTS it — only when
  // the caller is rb_learner_train_step (spec_accept_armed: it reads the table of THIS draw; the public entry points hand out
  // table 0, rb_replay_buffers_t.window_dev, so for them the draw is always made again, into table 0), same batch, beta, attempt
  // bound and output buffers, device RNG, nothing else has touched the replay since (every other entry point cancels it) — or draws
  // again; either way the stream-2 work is complete before anything of this stream goes on.
  const int accept_armed = r->spec_accept_armed;
  r->spec_accept_armed = 0;
  int spec_mode = 0;
  int win_set = 0;
  if (r->spec_inflight) {
    // (a captured launch never accepts: mode 3 bakes "wait for the record of epoch E, then draw" into the graph, and on every
    // replay that record is long complete)
    const bool same = accept_armed && !rb_stream_capturing(stream) && unit_uniforms_dev == nullptr && r->spec_args.batch == batch && r->spec_args.beta == priority_weight &&
                      r->spec_args.max_attempts == max_attempts && r->spec_args.tree_idx == tree_idx_dev &&
                      r->spec_args.actions == actions_dev && r->spec_args.returns == returns_dev &&
                      r->spec_args.nonterm == nonterminals_dev && r->spec_args.weights == weights_dev;
    spec_mode = same ? 2 : 3;
    if (same) win_set = r->spec_args.win_set;
    r->spec_inflight = 0;
  }
  r->win_sel = win_set;
  ++r->mutations;
  int32_t* const win_cur = win_of(r, win_set);
  const ReplayView v = view_of(r);
  int threads = (int)(rb_div_up(batch, 64) * 64);
  if (threads < 256) threads = 256;   // enough lanes to stage the 16 KB tree top into LDS in one sweep
  // weights ** -beta: python float exponent is cast to float32 by numpy (NEP 50 weak scalar)
  const float neg_beta = (float)(-priority_weight);
  NoiseJob job;
  memset(&job, 0, sizeof(job));
  unsigned blocks = 1;
  if (noise_job) {
    memcpy(&job, noise_job, sizeof(job));
    RB_REQUIRE(job.noise && job.ctr && job.nblk > 0 && job.nets >= 1 && job.dev, "rb_replay_sample_fused_noise: empty noise job");
    blocks += (unsigned)(job.nblk * job.nets);
  }
  const int noise_blocks = (int)blocks - 1;
  const ClipAdamArgs* adam_dev = nullptr;
  int host_mode = 0;
  if (noise_job && job.adam_dev && job.adam_blocks > 0) {
    RB_REQUIRE(threads == 256, "rb_replay_sample_fused_noise: the hosted optimiser pass needs a 256-thread sampler launch (batch <= 256)");
    adam_dev = static_cast<const ClipAdamArgs*>(job.adam_dev);
    // the host is the 1024-thread sampler variant (four tree levels per trip, 127 VGPRs -> 4 waves per SIMD for the hosted
    // streaming workgroups; under the 256-thread variant's 205 VGPRs the pass took 46 us instead of 38) with 4 quadruples per
    // hosted thread — job.adam_blocks counts blocks of 4 quadruples per thread.  That variant holds windows of up to 24
    // transitions; a longer window (history + multi_step > 24) gets the pending pass as a launch of its own in front of an
    // un-hosted sampler: same order in the stream, same results.
    if (r->history + r->n <= 24) {
      host_mode = 1;
      blocks += (unsigned)((job.adam_blocks * 4 + RB_HOST_AU_WIDE - 1) / RB_HOST_AU_WIDE);
    } else {
      RB_LAUNCH_T("clip_adam:k_adam_pending", k_adam_pending, dim3((unsigned)job.adam_blocks), dim3(256), stream, adam_dev);
      RB_LAUNCH_CHECK();
      adam_dev = nullptr;
    }
  }
  // (the tree search keeps its top 4095 nodes in LDS: measured against an all-global search, B = 32 / 1M leaves: 11.1 vs
  // 11.6 us, n = 20 / 100k: 14.3 vs 16.1, B = 256: 20.0 vs 24.4 — a trip costs ~1.3 us of issue, more than the staging)
  const int lds_top = 1;
  if (threads <= 256 && host_mode != 1) {
    RB_LAUNCH_T("sample:k_sample", (k_sample<256, 8>), dim3(blocks), dim3(threads), stream, v, batch, neg_beta, r->neg_beta_dev, unit_uniforms_dev, max_attempts, r->seed,
                r->scaling_dev, tree_idx_dev, win_cur, actions_dev, returns_dev, nonterminals_dev, weights_dev, job.dev, job.noise, job.noise2, job.ctr, r->fail_host, lds_top, noise_blocks, adam_dev, r->spec_res, r->spec_epoch, spec_mode);
  } else {
    RB_REQUIRE(r->history + r->n <= 24, "rb_replay_sample: the 1024-thread sampler (batch > 256) supports history + multi_step <= 24");
    RB_LAUNCH_T("sample:k_sample", (k_sample<1024, RB_HOST_AU_WIDE>), dim3(blocks), dim3(threads), stream, v, batch, neg_beta, r->neg_beta_dev, unit_uniforms_dev, max_attempts, r->seed,
                r->scaling_dev, tree_idx_dev, win_cur, actions_dev, returns_dev, nonterminals_dev, weights_dev, job.dev, job.noise, job.noise2, job.ctr, r->fail_host, lds_top, noise_blocks, adam_dev, r->spec_res, r->spec_epoch, spec_mode);
  }
  RB_LAUNCH_CHECK();
  if (states_dev && next_states_dev) {
    RB_LAUNCH(k_gather_stacks, dim3((unsigned)(batch * 2 * r->history)), dim3(256), stream, v, batch, win_cur,
              states_dev, next_states_dev);
    RB_LAUNCH_CHECK();
  }
  return RB_OK;
}

int rb_replay_sample(rb_replay_t* r, int32_t batch, double priority_weight, const double* unit_uniforms_dev,
                     int32_t max_attempts, int64_t* tree_idx_dev, uint8_t* states_dev, uint8_t* next_states_dev,
                     int64_t* actions_dev, float* returns_dev, float* nonterminals_dev, float* weights_dev,
                     rb_stream_t stream) {
  return sample_impl(r, batch, priority_weight, unit_uniforms_dev, max_attempts, tree_idx_dev, states_dev, next_states_dev,
                     actions_dev, returns_dev, nonterminals_dev, weights_dev, nullptr, stream);
}

int rb_replay_sample_fused_noise(rb_replay_t* r, int32_t batch, double priority_weight, const double* unit_uniforms_dev,
                                 int32_t max_attempts, int64_t* tree_idx_dev, uint8_t* states_dev, uint8_t* next_states_dev,
                                 int64_t* actions_dev, float* returns_dev, float* nonterminals_dev, float* weights_dev,
                                 const rb_noise_job_t* noise_job, rb_stream_t stream) {
  RB_REQUIRE(noise_job != nullptr, "rb_replay_sample_fused_noise: noise_job is NULL");
  return sample_impl(r, batch, priority_weight, unit_uniforms_dev, max_attempts, tree_idx_dev, states_dev, next_states_dev,
                     actions_dev, returns_dev, nonterminals_dev, weights_dev, noise_job, stream);
}

static int rb_update_impl(rb_replay_t* r, const int64_t* tree_idx_dev, const float* values_dev, int32_t n,
                          int32_t apply_pow, rb_stream_t stream) {
  RB_REQUIRE(r && tree_idx_dev && values_dev, "rb_replay_update: NULL argument");
  RB_REQUIRE(n >= 1 && n <= 1024, "rb_replay_update: n must be in [1,1024]");
  RB_SPEC_JOIN(r);
  ++r->mutations;
  int threads = (int)(rb_div_up(n, 64) * 64);
  if (threads < 256) threads = 256;     // the dense rebuild of the tree top wants lanes, not just one per leaf
  RB_LAUNCH(k_update, dim3(1), dim3(threads), stream, view_of(r), tree_idx_dev, values_dev, n, apply_pow, r->omega);
  RB_LAUNCH_CHECK();
  return RB_OK;
}

int rb_replay_update_leaves(rb_replay_t* r, const int64_t* tree_idx_dev, const float* values_dev, int32_t n,
                            rb_stream_t stream) {
  return rb_update_impl(r, tree_idx_dev, values_dev, n, 0, stream);
}

int rb_replay_update_priorities(rb_replay_t* r, const int64_t* tree_idx_dev, const float* losses_dev, int32_t n,
                                rb_stream_t stream) {
  return rb_update_impl(r, tree_idx_dev, losses_dev, n, 1, stream);
}

int rb_replay_update_sample(rb_replay_t* r, const int64_t* upd_tree_idx_dev, const float* upd_losses_dev, int32_t upd_n,
                            int32_t batch, double priority_weight, const double* unit_uniforms_dev, int32_t max_attempts,
                            int64_t* tree_idx_dev, uint8_t* states_dev, uint8_t* next_states_dev, int64_t* actions_dev,
                            float* returns_dev, float* nonterminals_dev, float* weights_dev, rb_stream_t stream) {
  RB_REQUIRE(r && upd_tree_idx_dev && upd_losses_dev, "rb_replay_update_sample: NULL argument");
  RB_REQUIRE(upd_n >= 1 && upd_n <= 1024, "rb_replay_update_sample: upd_n must be in [1,1024]");
  // one launch for what a PER loop at the reference's batch sizes produces (a sorted write-back of at most 64 leaves rides on
  // one wave); longer write-backs want the 1024-thread hashed kernel (batch 256: 34 us as two launches, 50 us fused)
  if (upd_n > 64 || batch > 256) {
    const int rc = rb_update_impl(r, upd_tree_idx_dev, upd_losses_dev, upd_n, 1, stream);
    if (rc != RB_OK) return rc;
    return sample_impl(r, batch, priority_weight, unit_uniforms_dev, max_attempts, tree_idx_dev, states_dev, next_states_dev,
                       actions_dev, returns_dev, nonterminals_dev, weights_dev, nullptr, stream);
  }
  RB_REQUIRE(tree_idx_dev && actions_dev && returns_dev && nonterminals_dev && weights_dev, "rb_replay_update_sample: NULL argument");
  RB_REQUIRE(batch >= 1, "rb_replay_update_sample: batch must be in [1,%d]", r->max_batch);
  RB_REQUIRE(max_attempts >= 1, "rb_replay_update_sample: max_attempts must be >= 1");
  RB_REQUIRE(r->capacity > (int64_t)r->history + r->n, "rb_replay_update_sample: capacity must exceed history + multi_step");
  RB_SPEC_JOIN(r);
  ++r->mutations;
  r->win_sel = 0;
  const ReplayView v = view_of(r);
  const float neg_beta = (float)(-priority_weight);
  RB_LAUNCH_T("sample:k_update_sample", k_update_sample, dim3(1), dim3(256), stream, v, upd_tree_idx_dev, upd_losses_dev, upd_n, 1, r->omega,
              batch, neg_beta, r->neg_beta_dev, unit_uniforms_dev, max_attempts, r->seed, r->scaling_dev, tree_idx_dev, r->win, actions_dev,
              returns_dev, nonterminals_dev, weights_dev, r->fail_host, (SpecResult*)nullptr, 0u);
  RB_LAUNCH_CHECK();
  if (states_dev && next_states_dev) {
    RB_LAUNCH(k_gather_stacks, dim3((unsigned)(batch * 2 * r->history)), dim3(256), stream, v, batch, r->win, states_dev, next_states_dev);
    RB_LAUNCH_CHECK();
  }
  return RB_OK;
}

int rb_replay_state_at(rb_replay_t* r, int64_t data_index, float* out_dev, rb_stream_t stream) {
  RB_REQUIRE(r && out_dev, "rb_replay_state_at: NULL argument");
  RB_SPEC_JOIN(r);
  RB_REQUIRE(data_index >= 0 && data_index < r->capacity, "rb_replay_state_at: index out of range");
  RB_LAUNCH(k_state_at, dim3(1), dim3(256), stream, view_of(r), data_index, out_dev);
  RB_LAUNCH_CHECK();
  return RB_OK;
}

int rb_replay_states_at(rb_replay_t* r, const int64_t* data_index_dev, int32_t n, float* out_dev, rb_stream_t stream) {
  RB_REQUIRE(r && data_index_dev && out_dev, "rb_replay_states_at: NULL argument");
  RB_SPEC_JOIN(r);
  RB_REQUIRE(n >= 0, "rb_replay_states_at: n must be >= 0");
  if (n == 0) return RB_OK;
  RB_LAUNCH(k_states_at, dim3((unsigned)n), dim3(256), stream, view_of(r), data_index_dev, out_dev);
  RB_LAUNCH_CHECK();
  return RB_OK;
}

int rb_frame_preprocess(const uint8_t* frame_a_dev, const uint8_t* frame_b_dev, int32_t height, int32_t width, int32_t n,
                        float* out_dev, rb_stream_t stream) {
  RB_REQUIRE(frame_a_dev && out_dev, "rb_frame_preprocess: NULL argument");
  RB_REQUIRE(height >= 2 && width >= 2 && height <= 4096 && width <= 4096, "rb_frame_preprocess: frame size must be in [2, 4096]^2");
  RB_REQUIRE(n >= 0, "rb_frame_preprocess: n must be >= 0");
  if (n == 0) return RB_OK;
  RB_LAUNCH(k_frame_preprocess, dim3((unsigned)rb_div_up(84 * 84, 256), (unsigned)n), dim3(256), stream, frame_a_dev, frame_b_dev,
            height, width, n, (int64_t)height * width, out_dev);
  RB_LAUNCH_CHECK();
  return RB_OK;
}

int rb_u8_to_unit_f32(const uint8_t* src_dev, float* dst_dev, int64_t n, rb_stream_t stream) {
  RB_REQUIRE(src_dev && dst_dev && n >= 0, "rb_u8_to_unit_f32: bad argument");
  if (n == 0) return RB_OK;
  int64_t g = rb_div_up(n, 256);
  if (g > 4096) g = 4096;
  RB_LAUNCH(k_u8_to_unit, dim3((unsigned)g), dim3(256), stream, src_dev, dst_dev, n);
  RB_LAUNCH_CHECK();
  return RB_OK;
}

}  // extern "C"

// (C++ linkage: called by learner.hip flush_update, not part of the C ABI)
int rb_launch_adam_pending(const ClipAdamArgs* args_dev, int blocks, void* stream) {
  RB_LAUNCH_T("clip_adam:k_adam_pending", k_adam_pending, dim3((unsigned)blocks), dim3(256), (hipStream_t)stream, args_dev);
  RB_LAUNCH_CHECK();
  return RB_OK;
}

