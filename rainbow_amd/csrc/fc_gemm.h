// fc_gemm.h — the hidden NoisyLinear layer (model.py:42-46, 56-63) as LDS-tiled f32 MFMA GEMMs for LARGE batches.
//
// noisy_linear.h streams every weight tile through registers once per 16..64 batch rows: right when the batch is 32 (the
// layer is a weight-bandwidth problem there), wrong at batch 256, where M = 768 forward rows make it a 4.9 GFLOP
// contraction: the streamed kernels re-read each weight tile 3.5 times (234 / 194 MB per launch for 67 / 56 MB of
// tensors, PMC) and sit at 0.46 / 0.32 of the f32 MFMA roofline.  Here a workgroup owns a 128 x 128 output tile, walks the
// reduction in slabs of 32 through double-buffered LDS (next slab's global loads in flight under this slab's MFMAs) and
// forms W = mu + sigma * (eps_out * eps_in) — the reference's three roundings, model.py:39,44 — on the way INTO LDS, so
// eps_w is still never materialised.  8 waves = 4 (rows) x 2 (columns); a wave owns 32 x 64 = two 32x32x2 MFMA tiles that
// share the A operand.  f32 MFMA is 1/16 of the bf16 rate: a slab is 256 MFMAs = 1.7 us per CU against 48 KB of loads, so
// the kernel is bound by the MFMA pipe and by nothing else as long as the slab pipeline never drains.
//
// Three contractions, one tile engine.  An operand tile sits in LDS in one of two layouts:
//   KC  s[row][36]   the 32 reduction entries of an output row / column are contiguous (global: reduction-contiguous
//                    rows, e.g. W[n][k..] for the forward) — an MFMA fragment is ONE ds_read_b128 per four MFMAs;
//   MC  s[kk][136]   the 128 output rows / columns of a reduction entry are contiguous (global: output-contiguous rows, e.g.
//                    W[n][k..] for the input gradient, whose reduction runs over n) — four ds_read_b32 per four MFMAs.
// Both are filled with 16-byte global loads and 16-byte LDS stores; neither needs a transposing store.  The k pairing of the
// 32x32x2 MFMA (lane half h takes k = 8g + 4h + j for MFMA j of group g) is the same in both, so any combination works:
//   forward           out[m][n]   = sum_k x[m][k]  W[n][k]      A = x (KC)    B = W (KC)   split-K, last arriver sums
//   input gradient    dx[s][m][k] = sum_n dy[m][n] W[n][k]      A = dy (KC)   B = W (MC)   row splits = the partial slices
//                                                                                          its consumers already sum
//   weight gradient   g[n][k]     = sum_m dy[m][n] x[m][k]      A = dy (MC)   B = x (MC)   sigma / bias / norm epilogue
// The slab pipeline (rb_tg_pipeline): one barrier per slab; the next slab is written to the other LDS buffer in the MIDDLE of
// this slab's MFMAs; the MFMA fragments are pipelined across the barrier (a lone workgroup per CU ran its loop at 2.35 us per
// slab against the pipe's 1.71 with "load fragments, multiply, store, barrier": every wave left the barrier into an LDS round
// trip with nothing to multiply, tools/stamp/gemm_timeline.py).  Epilogues pass the finished tile through LDS once so that every global store is a
// 16-byte store of a row segment (512 contiguous bytes per 32 lanes) instead of 32 dword stores per thread.
// Preconditions (host): K % 32 == 0, leading dimensions and offsets multiples of 4 floats, both streams of the layer read
// the same input columns (the hidden layer; the output layer keeps noisy_linear.h).
#pragma once
#include "noisy_linear.h"

#define RB_TG_T 128
#define RB_TG_KS 32
#define RB_TG_THREADS 512
#define RB_TG_LDK 36            // KC row stride: 16-byte reads of 16 consecutive rows fall into 16 distinct 4-bank groups (36/4 odd)
#define RB_TG_LDM 136           // MC row stride: the two lane halves of a fragment read 4 rows = 544 words = 32 banks apart
#define RB_TG_OP 4608           // floats per operand buffer: max(128 * 36, 32 * 136)
#define RB_TG_LDS (4 * RB_TG_OP)   // two operands, double-buffered: 72 KB (two workgroups per CU)
#define RB_TG_LDE 132           // row stride of the finished tile when the epilogue passes it through LDS (128 x 132 floats <= RB_TG_LDS)

// ---- arrival of the S workgroups that share one output tile: returns true (block-uniform) in the LAST one to arrive, after
// which the partial tiles of all S are visible to agent-coherent loads.  Partials are stored write-through (rb_st4_wt): in
// memory once every wave's stores have drained, no release fence (rb_device.h, rb_chain_signal).  The last arriver puts the
// counter back to zero: no memset between launches.
__device__ __forceinline__ bool rb_tile_arrive(unsigned* counter, unsigned expected, int* s_flag) {
#if defined(RB_HOST_INTERP)
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = *counter;
    *s_flag = old + 1u == expected ? 1 : 0;
    *counter = old + 1u == expected ? 0u : old + 1u;
  }
  __syncthreads();
  return *s_flag != 0;
#else
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = old + 1u == expected ? 1 : 0;
    if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_flag = last;
  }
  __syncthreads();
  return *s_flag != 0;
#endif
}

// ---- MFMA fragments of one reduction group g (8 reduction entries, four MFMAs per output tile) of a slab: the wave's 32 rows of
// A (row block wm) and its two 32-column blocks of B (column blocks 2 wn, 2 wn + 1)
struct TgFrag { float a[4], b0[4], b1[4]; };
template <bool A_KC, bool B_KC>
__device__ __forceinline__ void rb_tg_ldfrag(const float* sa, const float* sb, int g, int wm, int wn, int lane, TgFrag& f) {
  const int idx = lane & 31, half = lane >> 5;
  if (A_KC) {
    const float4 v = *reinterpret_cast<const float4*>(&sa[(32 * wm + idx) * RB_TG_LDK + 8 * g + 4 * half]);
    f.a[0] = v.x; f.a[1] = v.y; f.a[2] = v.z; f.a[3] = v.w;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) f.a[j] = sa[(8 * g + 4 * half + j) * RB_TG_LDM + 32 * wm + idx];
  }
  if (B_KC) {
    const float4 v0 = *reinterpret_cast<const float4*>(&sb[(64 * wn + idx) * RB_TG_LDK + 8 * g + 4 * half]);
    const float4 v1 = *reinterpret_cast<const float4*>(&sb[(64 * wn + 32 + idx) * RB_TG_LDK + 8 * g + 4 * half]);
    f.b0[0] = v0.x; f.b0[1] = v0.y; f.b0[2] = v0.z; f.b0[3] = v0.w;
    f.b1[0] = v1.x; f.b1[1] = v1.y; f.b1[2] = v1.z; f.b1[3] = v1.w;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f.b0[j] = sb[(8 * g + 4 * half + j) * RB_TG_LDM + 64 * wn + idx];
      f.b1[j] = sb[(8 * g + 4 * half + j) * RB_TG_LDM + 64 * wn + 32 + idx];
    }
  }
}
__device__ __forceinline__ void rb_tg_mfma(const TgFrag& f, rb_f32x16 (&acc)[2]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    acc[0] = rb_mfma32(f.a[j], f.b0[j], acc[0]);
    acc[1] = rb_mfma32(f.a[j], f.b1[j], acc[1]);
  }
}

// ---- the slab loop.  load(regs, i) requests slab i's global data into a register set, store(regs, buf) writes a landed set
// into LDS buffer buf (forming the noisy weights on the way).  DEPTH = slabs of global loads in flight: 2 = two register sets
// (the loop is unrolled by two so that they stay registers; one workgroup per CU: nothing else hides the memory latency),
// 1 = one set requested at the top of the iteration that stores it (two workgroups per CU under a 128-register budget).
// The MFMA fragments are software-pipelined ACROSS the slab barrier: group g + 1 is read from LDS while group g multiplies,
// and the last group of slab i multiplies AFTER the barrier, under the reads of slab i + 1's first group — a wave comes out of
// the barrier with eight MFMAs in hand instead of an LDS round trip.  cnt >= 1.  On return LDS is free (every wave is past
// the last barrier and reads nothing after it).
template <bool A_KC, bool B_KC, int DEPTH, class Regs, class LoadF, class StoreF>
__device__ __forceinline__ void rb_tg_pipeline(float* lds, int cnt, int wm, int wn, int lane, rb_f32x16 (&acc)[2], int kid,
                                               LoadF load, StoreF store) {
  (void)kid;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;
  Regs r0, r1;
  load(r0, 0);
  if (DEPTH == 2 && cnt > 1) load(r1, 1);
  store(r0, 0);
  __syncthreads();
  RB_WGT(kid, (int)blockIdx.x, 2);
  TgFrag f0, f1;
  rb_tg_ldfrag<A_KC, B_KC>(lds, lds + RB_TG_OP, 0, wm, wn, lane, f0);
  auto iter = [&](int i, Regs& cur, Regs& nxt) {
    // DEPTH 2: cur holds slab i + 1 (in flight since iteration i - 1), nxt gets slab i + 2.  DEPTH 1: cur gets slab i + 1 now.
    if (DEPTH == 2) { if (i + 2 < cnt) load(nxt, i + 2); } else { if (i + 1 < cnt) load(cur, i + 1); }     // block-uniform
    const float* sa = lds + (i & 1) * 2 * RB_TG_OP;
    rb_tg_ldfrag<A_KC, B_KC>(sa, sa + RB_TG_OP, 1, wm, wn, lane, f1);
    rb_tg_mfma(f0, acc);
    rb_tg_ldfrag<A_KC, B_KC>(sa, sa + RB_TG_OP, 2, wm, wn, lane, f0);
    rb_tg_mfma(f1, acc);
    if (i + 1 < cnt) store(cur, (i + 1) & 1);              // the other buffer: last read in iteration i - 1, behind its barrier
    rb_tg_ldfrag<A_KC, B_KC>(sa, sa + RB_TG_OP, 3, wm, wn, lane, f1);
    rb_tg_mfma(f0, acc);
    __syncthreads();                                       // slab i + 1 is in LDS; nobody reads slab i's buffer any more
    if (i + 1 < cnt) {
      const float* sn = lds + ((i + 1) & 1) * 2 * RB_TG_OP;
      rb_tg_ldfrag<A_KC, B_KC>(sn, sn + RB_TG_OP, 0, wm, wn, lane, f0);
    }
    RB_SCHED_FENCE();
    rb_tg_mfma(f1, acc);
  };
  if (DEPTH == 2) {
    for (int i = 0; i < cnt; i += 2) {
      iter(i, r1, r0);
      if (i + 1 < cnt) iter(i + 1, r0, r1);
    }
  } else {
    for (int i = 0; i < cnt; ++i) iter(i, r0, r0);
  }
  RB_WGT(kid, (int)blockIdx.x, 3);
}

// ---- finished tile -> LDS [128][RB_TG_LDE] (LDS must be free: rb_tg_pipeline's last barrier); ends with a barrier
__device__ __forceinline__ void rb_tg_acc_to_lds(const rb_f32x16 (&acc)[2], float* lds, int wm, int wn, int lane) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e)
      lds[(32 * wm + rb_mfma_row(e, lane)) * RB_TG_LDE + 64 * wn + 32 * t + (lane & 31)] = acc[t][e];
  __syncthreads();
}

template <int V> struct rb_int_c { static constexpr int value = V; };
__device__ __forceinline__ float4 rb_zero4() { return make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
__device__ __forceinline__ float4 rb_sel4(bool keep, float4 v) { return keep ? v : rb_zero4(); }

// ================================================================================ forward ==
struct FcGemmFwdArgs {
  NlFwd2Args f;              // the same problem description as k_nl_fwd3 (x k-blocked; both nets; both row groups)
  int mt[2];                 // 128-row tiles of each net
  int nt;                    // 128-column tiles (over all weight rows of both streams)
  int S;                     // K splits per tile (<= 8)
  float* part;               // S > 1: [tile][S][8][512] float4 partial tiles
  unsigned* ctr;             // S > 1: one arrival counter per tile
};
struct FcFwdRegs { float4 x[2], mu[2], sg[2], ei[2]; };

// grid = (tiles * S): block -> (m-tile slowest, then (n-tile, split)): the m-tiles that share a weight tile and K range have
// the same index mod 8 = the same XCD whenever nt * S is a multiple of 8, so an XCD's L2 fetches those weights once
__global__ __launch_bounds__(RB_TG_THREADS) void k_fc_gemm_fwd(FcGemmFwdArgs g) {
  __shared__ __attribute__((aligned(16))) float lds[RB_TG_LDS];
  __shared__ int s_flag;
  const NlFwd2Args& a = g.f;
  const int lane = rb_lane(), wave = rb_wave(), tid = (int)threadIdx.x;
  const int wm = wave & 3, wn = wave >> 2;
  const int combos = g.nt * g.S;
  int b = (int)blockIdx.x;
  int net = 0;
  if (b >= g.mt[0] * combos) { net = 1; b -= g.mt[0] * combos; }
  const int mtile = b / combos, combo = b % combos;
  const int ntile = combo / g.S, split = combo % g.S;
  const int tile_id = (net ? g.mt[0] * g.nt : 0) + mtile * g.nt + ntile;
  const int M = a.m_cnt[net], m0 = mtile * RB_TG_T, n0 = ntile * RB_TG_T;
  const int N = a.grp[a.n_groups - 1].row_begin + a.grp[a.n_groups - 1].row_cnt;
  const NlWeights w = a.w[net];
  const int K = a.K, nslab = K / RB_TG_KS;
  const int sbase = nslab / g.S, sextra = nslab % g.S;      // host: S <= nslab, so every split has at least one slab
  const int s_begin = split * sbase + (split < sextra ? split : sextra);
  const int s_cnt = sbase + (split < sextra ? 1 : 0);
  // RB_STAMP builds (tools/stamp/gemm_timeline.py): kernel id 12; slot 0 start, 1 last arriver?, 2 first slab staged, 3 loop
  // done, 4 partial stored + arrived, 5 partials summed, 6 end, 7 where it ran
  RB_WGT(12, (int)blockIdx.x, 0);
  RB_WGT_HW(12, (int)blockIdx.x);

  // staging map (both operands KC): thread -> float4 q8 of rows r0 and r0 + 64
  const int q8 = tid & 7, r0 = tid >> 3;
  unsigned xo[2], wo[2];
  bool mv[2], nv[2];
  float eo[2];
  const float* ein_row[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + r0 + 64 * i;
    mv[i] = m < M;
    const int mc = mv[i] ? m : M - 1;
    // k-blocked activations (rb_blocked_index): 16 k of a row are 64 contiguous bytes, the next 16 are rows_total * 64 further
    xo[i] = (unsigned)((((int64_t)(a.grp[0].x_off >> 4) * a.rows_total + a.m_base[net] + mc) * 16 + 4 * (q8 & 3)) * 4) +
            (unsigned)(q8 >> 2) * (unsigned)a.rows_total * 64u;
    const int n = n0 + r0 + 64 * i;
    nv[i] = n < N;
    const int nc = nv[i] ? n : N - 1;
    wo[i] = (unsigned)(((int64_t)nc * K + 4 * q8) * 4);
    eo[i] = w.eout[nc];
    const int grp = (a.n_groups > 1 && nc >= a.grp[1].row_begin) ? 1 : 0;
    ein_row[i] = w.ein + a.grp[grp].ein_off + 4 * q8;
  }
  const rb_buf bmu = rb_make_buf(w.mu), bsg = rb_make_buf(w.sigma), bx = rb_make_buf(a.x);
  const unsigned xslab = (unsigned)a.rows_total * 128u;     // bytes between slabs of the activations (two 16-wide chunks)

  auto load = [&](FcFwdRegs& r, int i) {
    const int s = s_begin + i;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      r.x[u] = rb_ld4_buf(bx, xo[u], (unsigned)s * xslab);
      r.mu[u] = rb_ld4_buf(bmu, wo[u], (unsigned)s * 128u);
      r.sg[u] = rb_ld4_buf(bsg, wo[u], (unsigned)s * 128u);
      r.ei[u] = rb_ld4(ein_row[u] + s * RB_TG_KS);
    }
  };
  auto store = [&](const FcFwdRegs& r, int buf) {
    float* sa = lds + buf * 2 * RB_TG_OP;
    float* sb = sa + RB_TG_OP;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      *reinterpret_cast<float4*>(&sa[(r0 + 64 * u) * RB_TG_LDK + 4 * q8]) = rb_sel4(mv[u], r.x[u]);
      *reinterpret_cast<float4*>(&sb[(r0 + 64 * u) * RB_TG_LDK + 4 * q8]) = rb_sel4(nv[u], rb_noisy4(r.mu[u], r.sg[u], eo[u], r.ei[u]));
    }
  };
  rb_f32x16 acc[2];
  rb_tg_pipeline<true, true, 2, FcFwdRegs>(lds, s_cnt, wm, wn, lane, acc, 12, load, store);

  if (g.S > 1) {
    // partial tile, thread-major (the reader is this kernel): 8 float4 per thread, each store 64 lanes x 16 B contiguous
    float* mine = g.part + ((int64_t)tile_id * g.S + split) * (8 * RB_TG_THREADS * 4);
#pragma unroll
    for (int v = 0; v < 8; ++v) {
      float4 o;
      o.x = acc[v >> 2][4 * (v & 3) + 0]; o.y = acc[v >> 2][4 * (v & 3) + 1];
      o.z = acc[v >> 2][4 * (v & 3) + 2]; o.w = acc[v >> 2][4 * (v & 3) + 3];
      rb_st4_wt(mine, (unsigned)((v * RB_TG_THREADS + tid) * 16), o);
    }
    const bool last = rb_tile_arrive(g.ctr + tile_id, (unsigned)g.S, &s_flag);
    RB_WGT(12, (int)blockIdx.x, 4);
    RB_WGT_ROLE(12, (int)blockIdx.x, last ? 1 : 0);
    if (!last) { RB_WGT(12, (int)blockIdx.x, 6); return; }
    // the last arriver sums the S partial tiles in split order (its own included: the order never depends on who is last)
    // straight into the LDS tile of the epilogue
    const rb_buf bp = rb_make_buf(g.part + (int64_t)tile_id * g.S * (8 * RB_TG_THREADS * 4));
    auto sum_parts = [&](auto smax_c, auto nv_c) {          // SMAX splits at most, NV float4 of every split in flight at a time
      constexpr int SMAX = decltype(smax_c)::value, NV = decltype(nv_c)::value;
#pragma unroll 1
      for (int hv = 0; hv < 8 / NV; ++hv) {
        float4 p[SMAX][NV];
#pragma unroll
        for (int s = 0; s < SMAX; ++s) {
          if (s < g.S) {                                     // block-uniform
#pragma unroll
            for (int v = 0; v < NV; ++v)
              p[s][v] = rb_ld4_buf_sc1(bp, (unsigned)(((NV * hv + v) * RB_TG_THREADS + tid) * 16), (unsigned)s * (8u * RB_TG_THREADS * 16u));
          }
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          float4 sum = p[0][v];
#pragma unroll
          for (int s = 1; s < SMAX; ++s) {
            if (s < g.S) { sum.x += p[s][v].x; sum.y += p[s][v].y; sum.z += p[s][v].z; sum.w += p[s][v].w; }
          }
          // float4 vv of a thread = accumulator registers 4 (vv & 3) .. + 3 of output tile vv >> 2: rows c + 8 (vv & 3) + 4 (lane >> 5)
          const int vv = NV * hv + v;
          float* d = &lds[(32 * wm + 8 * (vv & 3) + 4 * (lane >> 5)) * RB_TG_LDE + 64 * wn + 32 * (vv >> 2) + (lane & 31)];
          d[0] = sum.x; d[RB_TG_LDE] = sum.y; d[2 * RB_TG_LDE] = sum.z; d[3 * RB_TG_LDE] = sum.w;
        }
      }
    };
    // (batch 256 on 256 CUs: S = 5 -> two round trips of 20 loads; up to 8 splits: four of 16)
    if (g.S <= 5) sum_parts(rb_int_c<5>(), rb_int_c<4>());
    else sum_parts(rb_int_c<8>(), rb_int_c<2>());
    __syncthreads();
    RB_WGT(12, (int)blockIdx.x, 5);
  } else {
    rb_tg_acc_to_lds(acc, lds, wm, wn, lane);
  }
  // bias (model.py:44: b = bias_mu + bias_sigma * bias_epsilon, bias_epsilon = eps_out), ReLU; the tile goes through LDS so
  // that the row-major and the k-blocked copy are both written with 16-byte stores: thread -> float4 c4 of rows r, r + 16, ...
  {
    const int c4 = tid & 31, rr = tid >> 5;
    const int n = n0 + 4 * c4;
    if (n < N) {                                             // (N is a multiple of 4: the float4 is all in or all out)
      const float4 bm = rb_ld4(w.bmu + n), bs = rb_ld4(w.bsigma + n), be = rb_ld4(w.eout + n);
      float4 bias;
      bias.x = bm.x + bs.x * be.x; bias.y = bm.y + bs.y * be.y; bias.z = bm.z + bs.z * be.z; bias.w = bm.w + bs.w * be.w;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int r = rr + 16 * it;
        const int m = m0 + r;
        if (m >= M) continue;
        float4 o = *reinterpret_cast<const float4*>(&lds[r * RB_TG_LDE + 4 * c4]);
        o.x += bias.x; o.y += bias.y; o.z += bias.z; o.w += bias.w;
        if (a.relu) { o.x = fmaxf(o.x, 0.0f); o.y = fmaxf(o.y, 0.0f); o.z = fmaxf(o.z, 0.0f); o.w = fmaxf(o.w, 0.0f); }
        const int rowi = a.m_base[net] + m;
        rb_st4(a.out + (int64_t)rowi * a.ld_out + n, o);
        if (a.out_blocked) rb_st4(a.out_blocked + ((int64_t)(n >> 4) * a.rows_total + rowi) * 16 + (n & 15), o);
      }
    }
  }
  RB_WGT(12, (int)blockIdx.x, 6);
}

// =============================================================================== backward ==
// One launch: [0, first) the priority write-back (block 0) and padding, then the input-gradient tiles, then the
// weight-gradient tiles (independent given dY: horizontal fusion as in k_nl_bwd).
struct FcGemmBwdGrid {
  int first;                 // index of the first tile block (8 when the write-back rides along, else 0: keeps XCD = combo mod 8)
  int dx_mt, dx_combos;      // input gradient: 128-row tiles of dy, padded count of (column tile, row split) per m-tile
  int dx_kt, dx_splits;
  int dw_nt, dw_kt;          // weight gradient: 128-row tiles of W x 128-column tiles
};
struct FcDxRegs { float4 y[2], mu[2], sg[2]; float eo[2]; int n[2]; };
struct FcDwRegs { float4 y[2], x[2]; };

__device__ __forceinline__ void rb_fc_gemm_dx(const NlDxArgs& a, int mtile, int ktile, int split, float* lds) {
  const int lane = rb_lane(), wave = rb_wave(), tid = (int)threadIdx.x;
  const int wm = wave & 3, wn = wave >> 2;
  const NlDxProblem pr = a.prob[0];
  const int K = a.K, M = a.M;
  const int m0 = mtile * RB_TG_T, kt = ktile * RB_TG_T;
  const int row_end = pr.row_begin + pr.row_cnt;
  const int rb = pr.row_begin + split * a.rows_per_split;
  int re = rb + a.rows_per_split;
  if (re > row_end) re = row_end;
  if (rb >= re || m0 >= M) return;                          // block-uniform
  const int s_cnt = (re - rb + RB_TG_KS - 1) / RB_TG_KS;

  // A = dy[m][n] (KC: reduction n contiguous): float4 q8 of rows r0, r0 + 64
  const int q8 = tid & 7, r0 = tid >> 3;
  // B = W[n][k] (MC: output k contiguous): float4 c32 of reduction rows nn0, nn0 + 16
  const int c32 = tid & 31, nn0 = tid >> 5;
  int col4 = kt + 4 * c32;
  const bool cv = col4 < K;
  if (!cv) col4 = K - 4;
  const float4 e0 = rb_ld4(a.w.ein + pr.ein_off0 + col4);
  const float4 e1 = rb_ld4(a.w.ein + pr.ein_off1 + col4);
  const float* dyrow[2];
  bool mv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + r0 + 64 * i;
    mv[i] = m < M;
    dyrow[i] = a.dy + (int64_t)(mv[i] ? m : M - 1) * a.ldy;
  }
  auto load = [&](FcDxRegs& r, int s) {
    const int nb = rb + s * RB_TG_KS;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int n4 = nb + 4 * q8;                                  // (re - rb is a multiple of 4: the float4 is all in or all out)
      const bool ok = n4 < re;
      if (!ok) n4 = re - 4;
      r.y[i] = rb_sel4(ok && mv[i], rb_ld4(dyrow[i] + n4));
      const int n = nb + nn0 + 16 * i;
      r.n[i] = n < re ? n : -1;
      const int nc = n < re ? n : re - 1;
      r.mu[i] = rb_ld4(a.w.mu + (int64_t)nc * K + col4);
      r.sg[i] = rb_ld4(a.w.sigma + (int64_t)nc * K + col4);
      r.eo[i] = a.w.eout[nc];
    }
  };
  auto store = [&](const FcDxRegs& r, int buf) {
    float* sa = lds + buf * 2 * RB_TG_OP;
    float* sb = sa + RB_TG_OP;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<float4*>(&sa[(r0 + 64 * i) * RB_TG_LDK + 4 * q8]) = r.y[i];
      const float4 wv = rb_noisy4(r.mu[i], r.sg[i], r.eo[i], r.n[i] >= pr.ein_split_row ? e1 : e0);
      *reinterpret_cast<float4*>(&sb[(nn0 + 16 * i) * RB_TG_LDM + 4 * c32]) = rb_sel4(r.n[i] >= 0 && cv, wv);
    }
  };
  rb_f32x16 acc[2];
  rb_tg_pipeline<true, false, 1, FcDxRegs>(lds, s_cnt, wm, wn, lane, acc, 13, load, store);
  rb_tg_acc_to_lds(acc, lds, wm, wn, lane);
  if (cv) {
    const int rr = tid >> 5;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int r = rr + 16 * it;
      const int m = m0 + r;
      // write-through: 12.8 MB of slices + 25.7 MB of gradient left dirty in the L2s were flushed at the END of the launch
      // (kernel duration 51 us against a 44 us span of its workgroups)
      // (wave-uniform base + per-lane byte offset: a per-lane base makes the buffer store a 64-trip waterfall loop)
      if (m < M) rb_st4_wt(a.out, (unsigned)((((int64_t)split * M + m) * a.ld_out + pr.out_off + col4) * 4), *reinterpret_cast<const float4*>(&lds[r * RB_TG_LDE + 4 * c32]));
    }
  }
}

__device__ __forceinline__ void rb_fc_gemm_dw(const NlDwArgs& a, int ntile, int ktile, int slot_base, float* lds) {
  const int lane = rb_lane(), wave = rb_wave(), tid = (int)threadIdx.x;
  const int wm = wave & 3, wn = wave >> 2;
  const int K = a.K, M = a.M;
  const int N = a.prob[a.n_prob - 1].row_begin + a.prob[a.n_prob - 1].row_cnt;
  const int nt = ntile * RB_TG_T, kt = ktile * RB_TG_T;
  const int x_off = a.prob[0].x_off;
  const int s_cnt = (M + RB_TG_KS - 1) / RB_TG_KS;
  // both operands MC: float4 c32 of reduction rows mm0, mm0 + 16
  const int c32 = tid & 31, mm0 = tid >> 5;
  int ncol4 = nt + 4 * c32, kcol4 = kt + 4 * c32;
  const bool nv = ncol4 < N, kv = kcol4 < K;               // (N and K are multiples of 4: all in or all out)
  if (!nv) ncol4 = N - 4;
  if (!kv) kcol4 = K - 4;
  // epilogue operands, requested with the first slab
  const int p1 = a.n_prob > 1 ? a.prob[1].row_begin : (1 << 30);
  const float4 ei0 = rb_ld4(a.ein + a.prob[0].ein_off + kcol4);
  const float4 ei1 = rb_ld4(a.ein + a.prob[a.n_prob - 1].ein_off + kcol4);
  auto load = [&](FcDwRegs& r, int s) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = s * RB_TG_KS + mm0 + 16 * i;
      const bool ok = m < M;
      const int mc = ok ? m : M - 1;
      r.y[i] = rb_sel4(ok && nv, rb_ld4(a.dy + (int64_t)mc * a.ldy + ncol4));
      r.x[i] = rb_sel4(ok && kv, rb_ld4(a.x + (int64_t)mc * a.ldx + x_off + kcol4));
    }
  };
  // bias gradient g_bmu[n] = sum_m dy[m][n] = the column sums of the A operand, by the first column tile's workgroup: every
  // thread adds up the two rows of each slab it stages (four columns), the 16 threads of a column group meet in LDS afterwards
  const bool bias_wg = ktile == 0;                          // block-uniform
  float4 gb4 = rb_zero4();
  auto store = [&](const FcDwRegs& r, int buf) {
    float* sa = lds + buf * 2 * RB_TG_OP;
    float* sb = sa + RB_TG_OP;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<float4*>(&sa[(mm0 + 16 * i) * RB_TG_LDM + 4 * c32]) = r.y[i];
      *reinterpret_cast<float4*>(&sb[(mm0 + 16 * i) * RB_TG_LDM + 4 * c32]) = r.x[i];
      if (bias_wg) { gb4.x += r.y[i].x; gb4.y += r.y[i].y; gb4.z += r.y[i].z; gb4.w += r.y[i].w; }
    }
  };
  rb_f32x16 acc[2];
  rb_tg_pipeline<false, false, 1, FcDwRegs>(lds, s_cnt, wm, wn, lane, acc, 13, load, store);
  const bool do_bias = bias_wg && tid < RB_TG_T;
  float gb = 0.0f;
  if (bias_wg) {
    *reinterpret_cast<float4*>(&lds[mm0 * RB_TG_T + 4 * c32]) = gb4;
    __syncthreads();
    if (do_bias) {
#pragma unroll
      for (int q = 0; q < 16; ++q) gb += lds[q * RB_TG_T + tid];       // fixed order
    }
    __syncthreads();
  }
  // epilogue: g_mu = acc ; g_sigma = g_mu * (eps_out[n] * eps_in[k]) ; sum of squares of everything this wave wrote
  rb_tg_acc_to_lds(acc, lds, wm, wn, lane);
  float sq = 0.0f;
  if (kv) {
    const int rr = tid >> 5;
    float eo[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int n = nt + rr + 16 * it;
      eo[it] = a.eout[n < N ? n : N - 1];
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int r = rr + 16 * it;
      const int n = nt + r;
      if (n >= N) continue;
      const float4 gm = *reinterpret_cast<const float4*>(&lds[r * RB_TG_LDE + 4 * c32]);
      const float4 ei = n >= p1 ? ei1 : ei0;
      float4 gs;
      gs.x = gm.x * (eo[it] * ei.x); gs.y = gm.y * (eo[it] * ei.y); gs.z = gm.z * (eo[it] * ei.z); gs.w = gm.w * (eo[it] * ei.w);
      rb_st4_wt(a.g_mu, (unsigned)(((int64_t)n * K + kcol4) * 4), gm);
      if (!a.no_sigma) rb_st4_wt(a.g_sigma, (unsigned)(((int64_t)n * K + kcol4) * 4), gs);   // (RB_LEARNER_IMPLICIT_SIGMA: norm only)
      sq = fmaf(gm.x, gm.x, sq); sq = fmaf(gm.y, gm.y, sq); sq = fmaf(gm.z, gm.z, sq); sq = fmaf(gm.w, gm.w, sq);
      sq = fmaf(gs.x, gs.x, sq); sq = fmaf(gs.y, gs.y, sq); sq = fmaf(gs.z, gs.z, sq); sq = fmaf(gs.w, gs.w, sq);
    }
  }
  if (do_bias) {
    const int n = nt + tid;
    if (n < N) {
      const float gbs = gb * a.eout[n];
      a.g_bmu[n] = gb;
      a.g_bsigma[n] = gbs;
      sq = fmaf(gb, gb, sq);
      sq = fmaf(gbs, gbs, sq);
    }
  }
  if (a.sq_part) {
    sq = rb_wave_sum(sq);
    if (lane == 0) a.sq_part[slot_base + wave] = sq;
  }
}

// ---- the same weight gradient for the REPLICA EXCHANGE (rb_learner_finish_grads): the reduction rows are the gathered
// factor blocks of M / rpb ranks (noisy_linear.h NlDwArgs: rpb, bstride, noise_blocks), and every rank's gradient is noisy with ITS
// OWN epsilon:  g_mu = (1 / world) sum_r acc_r,  g_sigma = (1 / world) sum_r acc_r * (eps_out_r[n] * eps_in_r[k]),  acc_r = dY_r^T X_r.
// rb_nl_dw_body_ranks does this one 16 x 64 tile per wave: at config 5's size (M = 8 x 32 rows, [1024 x 3136] weights) every
// 16-row tile re-reads all of X — 0.5 GB of L2 traffic, 58 us per launch (profiles/round6_experiments.txt).  Here a workgroup owns
// a 128 x 128 tile: a rank's 32-row slab of dY and X goes through double-buffered LDS once (both MC), 8 waves multiply it with
// 32x32x2 MFMAs, and at the END of each rank's slabs the wave folds its accumulators into the running g_mu / g_sigma with that rank's
// noise (rank order, the same (eo * ei) product as the one-device bodies).  One barrier per slab, the next slab's global loads in
// flight under the MFMAs.  Every replica runs the same code on the same gathered blocks: identical bits on every replica.
// The bias gradients (column sums of dY per rank) are formed by the first column tile's workgroup in a loop of its own.
__device__ __forceinline__ void rb_fc_gemm_dw_ranks(const NlDwArgs& a, int ntile, int ktile, int slot_base, float* lds) {
  const int lane = rb_lane(), wave = rb_wave(), tid = (int)threadIdx.x;
  const int wm = wave & 3, wn = wave >> 2;
  const int K = a.K;
  const int N = a.prob[a.n_prob - 1].row_begin + a.prob[a.n_prob - 1].row_cnt;
  const int nt = ntile * RB_TG_T, kt = ktile * RB_TG_T;
  const int x_off = a.prob[0].x_off;
  const int rpb = a.rpb, nranks = a.M / a.rpb;
  const int spr = (rpb + RB_TG_KS - 1) / RB_TG_KS;          // slabs per rank
  const int total = nranks * spr;
  const int c32 = tid & 31, mm0 = tid >> 5;
  int ncol4 = nt + 4 * c32, kcol4 = kt + 4 * c32;
  const bool nv = ncol4 < N, kv = kcol4 < K;               // (N and K are multiples of 4: all in or all out)
  if (!nv) ncol4 = N - 4;
  if (!kv) kcol4 = K - 4;
  const int p1 = a.n_prob > 1 ? a.prob[1].row_begin : (1 << 30);
  const int ein0 = a.prob[0].ein_off, ein1 = a.prob[a.n_prob - 1].ein_off;
  auto load = [&](FcDwRegs& r, int idx) {
    const int rk = idx / spr, s = idx - rk * spr;
    const float* dy = a.dy + (int64_t)rk * a.bstride;
    const float* x = a.x + (int64_t)rk * a.bstride;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = s * RB_TG_KS + mm0 + 16 * i;
      const bool ok = m < rpb;
      const int mc = ok ? m : rpb - 1;
      r.y[i] = rb_sel4(ok && nv, rb_ld4(dy + (int64_t)mc * a.ldy + ncol4));
      r.x[i] = rb_sel4(ok && kv, rb_ld4(x + (int64_t)mc * a.ldx + x_off + kcol4));
    }
  };
  auto store = [&](const FcDwRegs& r, int buf) {
    float* sa = lds + buf * 2 * RB_TG_OP;
    float* sb = sa + RB_TG_OP;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<float4*>(&sa[(mm0 + 16 * i) * RB_TG_LDM + 4 * c32]) = r.y[i];
      *reinterpret_cast<float4*>(&sb[(mm0 + 16 * i) * RB_TG_LDM + 4 * c32]) = r.x[i];
    }
  };
  // this lane's cells: rows n(e) = nt + 32 wm + mfma_row(e), columns k(t) = kt + 64 wn + 32 t + (lane & 31)
  int kcol[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) { kcol[t] = kt + 64 * wn + 32 * t + (lane & 31); if (kcol[t] > K - 1) kcol[t] = K - 1; }
  rb_f32x16 gm[2], gs[2], acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) { gm[t][e] = 0.0f; gs[t][e] = 0.0f; acc[t][e] = 0.0f; }
  // One slab of global loads in flight (requested behind the barrier, consumed at the next slab's LDS stores); inside a slab the MFMA
  // fragments are read one group AHEAD of the multiplies (two fragment sets): with "read a group, multiply it" the LDS latency was
  // exposed four times per slab and the loop ran at ~3.7 us per slab instead of the pipe's 1.7 (ablation: 40 us per launch, 23
  // without the loop).  (Two slabs of loads in flight and the next rank's noise a rank ahead were tried: no change, removed.)
  FcDwRegs R;
  load(R, 0);
  float eo[16], ei[2][2];
  for (int idx = 0; idx < total; ++idx) {
    const int rk = idx / spr, s = idx - rk * spr;
    store(R, idx & 1);
    __syncthreads();                                        // slab idx is in LDS; its buffer was last read two slabs ago
    const float* sa = lds + (idx & 1) * 2 * RB_TG_OP;
    TgFrag f0, f1;
    rb_tg_ldfrag<false, false>(sa, sa + RB_TG_OP, 0, wm, wn, lane, f0);
    if (idx + 1 < total) load(R, idx + 1);
    if (s == 0) {                                           // this rank's noise for the fold below: requested with the slab (block-uniform)
      const float* nz = a.noise_blocks + (int64_t)rk * a.bstride;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = nt + 32 * wm + rb_mfma_row(e, lane);
        eo[e] = nz[a.eout_noff + (n < N ? n : N - 1)];
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) { ei[t][0] = nz[a.ein_noff + ein0 + kcol[t]]; ei[t][1] = nz[a.ein_noff + ein1 + kcol[t]]; }
    }
    rb_tg_ldfrag<false, false>(sa, sa + RB_TG_OP, 1, wm, wn, lane, f1);
    rb_tg_mfma(f0, acc);
    rb_tg_ldfrag<false, false>(sa, sa + RB_TG_OP, 2, wm, wn, lane, f0);
    rb_tg_mfma(f1, acc);
    rb_tg_ldfrag<false, false>(sa, sa + RB_TG_OP, 3, wm, wn, lane, f1);
    rb_tg_mfma(f0, acc);
    RB_SCHED_FENCE();
    rb_tg_mfma(f1, acc);
    if (s == spr - 1) {                                     // the rank is complete: fold (rank order), restart the accumulators
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int n = nt + 32 * wm + rb_mfma_row(e, lane);
          gm[t][e] = gm[t][e] + acc[t][e];
          gs[t][e] = gs[t][e] + acc[t][e] * (eo[e] * (n >= p1 ? ei[t][1] : ei[t][0]));
          acc[t][e] = 0.0f;
        }
    }
  }
  __syncthreads();                                          // every wave is done with the operand buffers
  // bias gradients (first column tile only): g_bmu[n] = scale * sum_r colsum_r[n], g_bsigma[n] = scale * sum_r colsum_r[n] * eps_out_r[n].
  // Column sums of a rank's dY by ALL 512 threads — thread (n = tid & 127, group tid >> 7) takes every fourth rank, a rank's rows 32
  // loads at a time — parked in LDS [rank][128]; then thread n folds them in rank order.  (As one thread per column walking all
  // ranks 8 loads at a time this was a chain of 32 round trips: the launch's pole at 47 us.)
  const bool bias_wg = ktile == 0;                          // block-uniform
  const bool do_bias = bias_wg && tid < RB_TG_T && nt + tid < N;
  float gb = 0.0f, gbs = 0.0f;
  if (bias_wg) {
    const int nb = tid & (RB_TG_T - 1), rg = tid >> 7;
    const int n = nt + nb < N ? nt + nb : N - 1;
    for (int rk = rg; rk < nranks; rk += RB_TG_THREADS / RB_TG_T) {
      const float* dy = a.dy + (int64_t)rk * a.bstride;
      float cs = 0.0f;
      for (int m0 = 0; m0 < rpb; m0 += 32) {
        float v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = dy[(int64_t)(m0 + u < rpb ? m0 + u : rpb - 1) * a.ldy + n];
#pragma unroll
        for (int u = 0; u < 32; ++u) cs += (m0 + u < rpb) ? v[u] : 0.0f;
      }
      if (rk < RB_TG_LDS / (2 * RB_TG_T)) {                   // (world <= 64: rb_learner_set_exchange)
        lds[rk * RB_TG_T + nb] = cs;
        lds[(RB_TG_LDS / (2 * RB_TG_T) + rk) * RB_TG_T + nb] = a.noise_blocks[(int64_t)rk * a.bstride + a.eout_noff + n];
      }
    }
    __syncthreads();
    if (do_bias) {
      for (int rk = 0; rk < nranks; ++rk) {
        const float cs = lds[rk * RB_TG_T + tid];
        gb = gb + cs;
        gbs = gbs + cs * lds[(RB_TG_LDS / (2 * RB_TG_T) + rk) * RB_TG_T + tid];
      }
    }
    __syncthreads();                                        // the tiles below reuse the buffer
  }
  // epilogue: the two finished tiles through LDS, 16-byte row-segment stores, sum of squares of everything this wave wrote
  float sq = 0.0f;
  const int rr = tid >> 5;
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    rb_f32x16 (&src)[2] = which == 0 ? gm : gs;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) src[t][e] = src[t][e] * a.scale;
    if (which == 1) __syncthreads();                        // the first tile has been read out of LDS
    rb_tg_acc_to_lds(src, lds, wm, wn, lane);
    if (kv) {
      float* dst = which == 0 ? a.g_mu : a.g_sigma;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int r = rr + 16 * it;
        const int n = nt + r;
        if (n >= N) continue;
        const float4 v = *reinterpret_cast<const float4*>(&lds[r * RB_TG_LDE + 4 * c32]);
        rb_st4_wt(dst, (unsigned)(((int64_t)n * K + kcol4) * 4), v);
        sq = fmaf(v.x, v.x, sq); sq = fmaf(v.y, v.y, sq); sq = fmaf(v.z, v.z, sq); sq = fmaf(v.w, v.w, sq);
      }
    }
  }
  if (do_bias) {
    const int n = nt + tid;
    const float b0 = gb * a.scale, b1 = gbs * a.scale;
    a.g_bmu[n] = b0;
    a.g_bsigma[n] = b1;
    sq = fmaf(b0, b0, sq);
    sq = fmaf(b1, b1, sq);
  }
  if (a.sq_part) {
    sq = rb_wave_sum(sq);
    if (lane == 0) a.sq_part[slot_base + wave] = sq;
  }
}

#if defined(RB_HOST_INTERP)
#define RB_TG_TWO_PER_CU
#else
#define RB_TG_TWO_PER_CU __attribute__((amdgpu_waves_per_eu(4, 4)))     // 128 registers: two 8-wave workgroups per CU
#endif
__global__ __launch_bounds__(RB_TG_THREADS) RB_TG_TWO_PER_CU void k_fc_gemm_bwd(NlDwArgs dw, NlDxArgs dx, FcGemmBwdGrid g, NlPriorityUpdate up) {
  constexpr int LDSW = RB_TG_LDS > UpdateLds<512, RB_TG_THREADS>::WORDS ? RB_TG_LDS : UpdateLds<512, RB_TG_THREADS>::WORDS;
  __shared__ __attribute__((aligned(16))) float lds[LDSW];
  int b = (int)blockIdx.x;
  // RB_STAMP builds: kernel id 13; slot 0 start, 1 role (0 write-back / padding, 1 dW, 2 dX), 2 first slab staged, 3 loop done, 6 end
  const int wgb = (int)blockIdx.x;
  (void)wgb;
  RB_WGT(13, wgb, 0);
  RB_WGT_HW(13, wgb);
  if (b < g.first) {
    // ALL threads of the block: the body strides its table clears and heap levels by blockDim.x and keeps one value slot per
    // thread (NMAX = the block size); n <= 256 samples, 512 hash slots
    if (b == 0 && up.enabled) rb_update_body<512, RB_TG_THREADS>(up.view, up.tree_idx, up.loss, up.n, 1, up.omega, lds);
    RB_WGT_ROLE(13, wgb, 0);
    RB_WGT(13, wgb, 6);
    return;
  }
  b -= g.first;
  const int ndx = g.dx_mt * g.dx_combos;
  if (b < ndx) {
    const int mtile = b / g.dx_combos, combo = b % g.dx_combos;
    if (combo >= g.dx_kt * g.dx_splits) return;             // padding (keeps equal combos of different m-tiles on one XCD)
    rb_fc_gemm_dx(dx, mtile, combo / g.dx_splits, combo % g.dx_splits, lds);
    RB_WGT_ROLE(13, wgb, 2);
  } else {
    b -= ndx;
    rb_fc_gemm_dw(dw, b / g.dw_kt, b % g.dw_kt, 8 * b, lds);
    RB_WGT_ROLE(13, wgb, 1);
  }
  RB_WGT(13, wgb, 6);
}
