// noisy_linear.h — the NoisyLinear layers (model.py:10-46) for skinny batches, streamed
// from HBM into v_mfma_f32_16x16x4_f32.
//
// Why a dedicated family: with M = B..3B rows (32..96) against [1024 x 3136] weights the hidden
// layer is a weight-bandwidth problem — 51 MB of mu/sigma per forward, 26 MB per input-gradient
// pass, 26 MB of gradient written — and each weight is used once per workgroup.  Noisy weights
// are formed in registers, W = mu + sigma * (eps_out * eps_in), with the reference's rounding
// order (model.py:39,44); eps_w is never materialised.
//
// The 16x16x4 MFMA wants lane l to hold weight row / output column l&15 for k-slot l>>4 and
// four MFMAs consume one float4 per lane (the k permutation inside a step is irrelevant to a
// sum).  What decides the speed is how many CACHE LINES a wave's load instruction touches:
//   * input / weight gradients (output columns contiguous): 16 lanes x 16 B = 256 B per row,
//     4 rows per instruction = 8 full lines — loaded straight in the operand layout;
//   * forward (k contiguous): the operand layout would be 16 rows x 64 B = 16 half lines per
//     instruction, which measured 31 us against 22 us for the same bytes read as full lines.
//     k_nl_fwd2 therefore loads 8 rows x 128 B per instruction, forms W in that layout and
//     transposes it to the operand layout through a private LDS tile per wave.
//
// Preconditions (checked by the host; the generic gemm_core path remains the fallback):
// K % 32 == 0, all leading dimensions and offsets multiples of 4 floats.
#pragma once
#include "rb_device.h"
#include "replay_internal.h"

struct NlWeights {
  const float* mu;      // [N][K]
  const float* sigma;   // [N][K]
  const float* eout;    // [N]
  const float* ein;     // [streams][K] (offset per row group)
  const float* bmu;     // [N]
  const float* bsigma;  // [N]
};

// k-blocked activation layout consumed by k_nl_fwd: element (row, k) of a [rows][K] matrix
__host__ __device__ inline int64_t rb_blocked_index(int row, int k, int rows) {
  return ((int64_t)(k >> 4) * rows + row) * 16 + (k & 15);
}

// W = mu + sigma * (eo * ein), component-wise, three separately rounded operations
__device__ __forceinline__ float4 rb_noisy4(float4 mu, float4 sg, float eo, float4 e) {
  float4 w;
  w.x = mu.x + sg.x * (eo * e.x);
  w.y = mu.y + sg.y * (eo * e.y);
  w.z = mu.z + sg.z * (eo * e.z);
  w.w = mu.w + sg.w * (eo * e.w);
  return w;
}

// ============================================================================ forward ==
// out[m][n] = b[n] + sum_k x[m][x_off + k] * W[n][k]   (ReLU optional)
struct NlRowGroup {
  int row_begin, row_cnt;   // weight rows of this stream
  int x_off, ein_off;       // column offset into x, offset into ein
  int tile_begin;           // first 16-row tile index of this group in grid.x
};
#define RB_NL_FWD_WAVES 8
// k_nl_fwd2 — forward without split-K partials: one workgroup owns a 16-row weight tile for the whole K, its 8 waves
// take contiguous K ranges and meet once in a 2 KB-per-wave LDS reduction; bias (+ReLU) is applied there and the
// result is written directly (row-major and, optionally, k-blocked for the next layer) — no partial-sum round trip, no
// separate finish kernel.  (Round-1 history: a K-split-across-workgroups variant with the activations shared through
// LDS and a finish pass lost to this kernel twice: 251 vs 238.5 us per step before the line-wide loads, and, rebuilt
// with them, 222.5 vs 221.0 (its main kernel 19 us bracketed + a 5 us finish pass); non-temporal weight loads measured +13 us on
// the step, prefetching the bias terms before the loop measured nothing; a split-K variant with a finish kernel measured the same 30 us for the
// pair; ablation shows the activation re-reads through the 64 B/clk L1 cost as much as the weight stream itself.)
struct NlFwd2Args {
  const float* x;           // k-blocked activations (rb_blocked_index)
  int m_base[2], m_cnt[2];
  NlWeights w[2];
  int K;
  int n_groups;
  NlRowGroup grp[2];        // tile_begin counts 16-row tiles here
  float* out;               // [rows_total][ld_out]
  float* out_blocked;       // optional k-blocked copy over ld_out columns
  int ld_out, rows_total;
  int relu;
};

// grid = (16-row tiles, 1, 2 * m-chunks of 16 MT rows), block = 512
#define RB_FWD2_MROWS 32
#define RB_FWD2_KMAX 4096          // eps_in slice staged in LDS (host checks K <= this)
#define RB_FWD2_WT_LD 36           // row stride (floats) of a wave's 16 x 32 weight tile: 16-byte reads of 16 rows hit 64 distinct banks
// k_nl_fwd3 — k_nl_fwd2 with the loop overhead taken out (same tiling, same LDS transpose, same results up to the order
// in which the 8 waves' partial sums were formed — the K ranges of the waves are balanced differently):
//   * every load is a buffer load: per-lane byte offset fixed before the loop, the running block offset is wave-uniform and
//     lives in an SGPR — one instruction per load.  k_nl_fwd2 formed `pointer + 64-bit index` per load (5-6 instructions
//     each, ~50 per 32-wide block: as many issue cycles as the block's 16 MFMAs);
//   * a wave's K range is a whole number of 32-wide blocks, balanced 12/13 over the 8 waves (fwd2: 13,...,13,7), and the
//     loop runs floor(n / RING) full ring rounds without any masking plus a tail that only computes — fwd2 padded every
//     wave to a multiple of RING blocks (16 slots for 12.25 blocks of work: 31 % dead MFMAs and loads, zeroed by a mask
//     multiply on every weight).
template <int MT>
__global__ __launch_bounds__(64 * RB_NL_FWD_WAVES) void k_nl_fwd3(NlFwd2Args a) {
  __shared__ float s_red[RB_NL_FWD_WAVES][4 * MT][64];
  __shared__ __attribute__((aligned(16))) float s_ein[RB_FWD2_KMAX];
  __shared__ __attribute__((aligned(16))) float s_wt[RB_NL_FWD_WAVES][16 * RB_FWD2_WT_LD];
  const int lane = rb_lane(), wave = rb_wave();
  const int net = (int)blockIdx.z & 1, mc = (int)blockIdx.z >> 1;
  const int M = a.m_cnt[net];
  const int m0 = mc * (16 * MT);
  if (m0 >= M) return;                                   // block-uniform
  const int g = (a.n_groups > 1 && (int)blockIdx.x >= a.grp[1].tile_begin) ? 1 : 0;
  const NlRowGroup grp = a.grp[g];
  const int row0 = grp.row_begin + ((int)blockIdx.x - grp.tile_begin) * 16;
  const int row_end = grp.row_begin + grp.row_cnt;
  const NlWeights w = a.w[net];
  const int K = a.K;
  const int nblk = K / 32;                               // host guarantees K % 32 == 0
  const int base_n = nblk / RB_NL_FWD_WAVES, extra = nblk % RB_NL_FWD_WAVES;
  const int nsc = rb_wave_uniform(base_n + (wave < extra ? 1 : 0));                     // 32-wide blocks of this wave
  const int b0 = rb_wave_uniform(wave * base_n + (wave < extra ? wave : extra));       // its first block

  const int r = lane & 15, q = lane >> 4;
  const int lr = lane >> 3, lk = lane & 7;               // line-wide weight loads: 8 rows x 128 B per instruction
  const rb_buf bmu = rb_make_buf(w.mu), bsg = rb_make_buf(w.sigma), bx = rb_make_buf(a.x);
  unsigned wo[2];
  float eo2[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int row = row0 + 8 * i + lr;
    if (row > row_end - 1) row = row_end - 1;
    wo[i] = (unsigned)(((int64_t)row * K + 4 * lk) * 4);
    eo2[i] = w.eout[row];
  }
  float* wt = &s_wt[wave][0];                             // [16 rows][RB_FWD2_WT_LD]
  unsigned xo[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int m = m0 + 16 * mt + r;
    if (m > M - 1) m = M - 1;
    xo[mt] = (unsigned)((((int64_t)(grp.x_off >> 4) * a.rows_total + a.m_base[net] + m) * 16 + 4 * q) * 4);
  }
  const unsigned xstep = (unsigned)a.rows_total * 64u;     // bytes between consecutive 16-wide k chunks of the activations
  // the epilogue's bias terms (every epilogue cell of a thread is column row0 + (lane & 15)): requested HERE, with the first
  // operands — in the epilogue they were one more dependent round trip at the end of a 4 us chain (the output layer's launch)
  const int nb_ = row0 + (lane & 15) < row_end ? row0 + (lane & 15) : row_end - 1;
  const float b_mu = w.bmu[nb_], b_sg = w.bsigma[nb_], b_eo = w.eout[nb_];

  rb_f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[mt][e] = 0.0f;

  constexpr int RING = MT >= 4 ? 3 : 4;
  float4 r_mu[RING][2], r_sg[RING][2], r_x[RING][2][MT];
  const int b_last = nsc > 0 ? b0 + nsc - 1 : nblk - 1;  // (a wave without work still issues the ring's loads: keep them in range)
  auto blk_of = [&](int sc) { const int b = b0 + sc; return b < b_last ? b : b_last; };   // wave-uniform, clamped
  {
    const float* ein_g = w.ein + grp.ein_off;
    for (int k4 = (int)threadIdx.x; k4 < (K >> 2); k4 += 64 * RB_NL_FWD_WAVES)
      *reinterpret_cast<float4*>(&s_ein[4 * k4]) = rb_ld4(ein_g + 4 * k4);
  }
  auto load_w = [&](int d, int b) {
    const unsigned so = (unsigned)b * 128u;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      r_mu[d][i] = rb_ld4_buf(bmu, wo[i], so);
      r_sg[d][i] = rb_ld4_buf(bsg, wo[i], so);
    }
  };
  auto load_x = [&](int d, int b) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const unsigned so = (unsigned)(2 * b + h) * xstep;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) r_x[d][h][mt] = rb_ld4_buf(bx, xo[mt], so);
    }
  };
  // EXACT: no ring slot is ever loaded with a block past the wave's range (round 6: the clamped duplicates — the output layer's 2 blocks
  // per wave in a ring of 4, the 3-4 refills of the hidden layer's last round — were a fifth to a half of what a workgroup pulled
  // through its CU: fc_h 15.7 -> 15.0 us, fc_z 6.1 -> 5.1 us; the 64-row form of batch 256 measured 1 us slower with it and keeps the
  // clamped loads, as does an -DRB_FWD_CLAMPED_REFILLS build)
#if defined(RB_FWD_CLAMPED_REFILLS)
  constexpr bool EXACT = false;
#else
  constexpr bool EXACT = MT <= 2;
#endif
#pragma unroll
  for (int d = 0; d < RING; ++d) {
    if (!EXACT) { load_w(d, blk_of(d)); load_x(d, blk_of(d)); }
    else if (d < nsc) { load_w(d, b0 + d); load_x(d, b0 + d); }   // wave-uniform
  }
  __syncthreads();                                       // eps_in visible
  auto compute = [&](int d, int b) {
    const float4 e4 = *reinterpret_cast<const float4*>(&s_ein[b * 32 + 4 * lk]);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      *reinterpret_cast<float4*>(&wt[(8 * i + lr) * RB_FWD2_WT_LD + 4 * lk]) = rb_noisy4(r_mu[d][i], r_sg[d][i], eo2[i], e4);
  };
  auto mfmas = [&](int d) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float4 w4 = *reinterpret_cast<const float4*>(&wt[r * RB_FWD2_WT_LD + 16 * h + 4 * q]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = rb_mfma16(r_x[d][h][mt].x, w4.x, acc[mt]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = rb_mfma16(r_x[d][h][mt].y, w4.y, acc[mt]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = rb_mfma16(r_x[d][h][mt].z, w4.z, acc[mt]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = rb_mfma16(r_x[d][h][mt].w, w4.w, acc[mt]);
    }
  };
  const int full = nsc / RING * RING;
  // EXACT: the LAST full round refills only the slots the tail will consume (nsc - full of RING); else every round refills every slot
  const int steady = !EXACT ? full : full >= RING ? full - RING : 0;
  for (int sc0 = 0; sc0 < steady; sc0 += RING) {
#pragma unroll
    for (int d = 0; d < RING; ++d) {
      const int sc = sc0 + d;
      compute(d, b0 + sc);
      load_w(d, blk_of(sc + RING));                      // refill the weight half of this ring slot
      rb_wave_sync();                                    // the tile is private to the wave: LDS executes its ops in order
      mfmas(d);
      load_x(d, blk_of(sc + RING));                      // ... and its activation half, once the MFMAs have read it
      rb_wave_sync();                                    // tile reads done before the next block overwrites it
      RB_SCHED_FENCE();                                  // keep this slot's refill here, not at the end of the loop
    }
  }
  if (steady < full) {                                   // wave-uniform: the last full round
#pragma unroll
    for (int d = 0; d < RING; ++d) {
      const int sc = steady + d;
      const bool refill = sc + RING < nsc;               // wave-uniform
      compute(d, b0 + sc);
      if (refill) load_w(d, b0 + sc + RING);
      rb_wave_sync();
      mfmas(d);
      if (refill) load_x(d, b0 + sc + RING);
      rb_wave_sync();
      RB_SCHED_FENCE();
    }
  }
#pragma unroll
  for (int d = 0; d < RING; ++d) {                       // tail: the blocks the last refills brought in, no more loads
    if (full + d < nsc) {                                // wave-uniform
      compute(d, b0 + full + d);
      rb_wave_sync();
      mfmas(d);
      rb_wave_sync();
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int e = 0; e < 4; ++e) s_red[wave][mt * 4 + e][lane] = acc[mt][e];
  __syncthreads();
  for (int idx = (int)threadIdx.x; idx < 4 * MT * 64; idx += 64 * RB_NL_FWD_WAVES) {
    const int slot = idx >> 6, l = idx & 63;
    float v = s_red[0][slot][l];
#pragma unroll
    for (int wv = 1; wv < RB_NL_FWD_WAVES; ++wv) v += s_red[wv][slot][l];
    const int mt = slot >> 2, e = slot & 3;
    const int m = m0 + 16 * mt + 4 * (l >> 4) + e;
    const int n = row0 + (l & 15);
    if (m < M && m < m0 + 16 * MT && n < row_end) {
#if defined(RB_NO_BIAS_PRE)                                                 // (A/B build: the loads in the epilogue)
      float o = v + (w.bmu[n] + w.bsigma[n] * w.eout[n]);
#else
      float o = v + (b_mu + b_sg * b_eo);                                 // model.py:44 (n == nb_ for every cell that is stored)
#endif
      if (a.relu) o = fmaxf(o, 0.0f);
      const int rowi = a.m_base[net] + m;
      a.out[(int64_t)rowi * a.ld_out + n] = o;
      if (a.out_blocked) a.out_blocked[((int64_t)(n >> 4) * a.rows_total + rowi) * 16 + (n & 15)] = o;
    }
  }
}

// ====================================================================== input gradient ==
// dx[m][out_off + k] = sum_{n in rows} dy[m][n] * W[n][k]      (adjoint of the forward)
struct NlDxProblem {
  int row_begin, row_cnt;    // reduction range (weight rows)
  int ein_split_row;         // rows >= this use ein_off1 (fc_h: the advantage stream), else ein_off0
  int ein_off0, ein_off1;
  int out_off;               // column offset in dx
};
struct NlDxArgs {
  const float* dy;           // [M][ldy]
  int ldy, M;
  NlWeights w;
  int K;                     // output columns of one problem
  int n_prob;
  NlDxProblem prob[2];
  int rows_per_split;        // reduction rows per block (multiple of 16); grid.y = splits
  float* out;                // split: part[s][M][ld_out]; else dx[M][ld_out]
  int ld_out;
  const float* mask_src;     // non-split only: dx = mask_src > 0 ? v : 0  (ReLU adjoint), same indexing as out
  // Transposed companions.  dyT [rows][ldyT] (dyT[n][m] == dy[m][n]): when set, the dY operand is read from it — a lane
  // group then reads 16 CONSECUTIVE m of one weight row (64 bytes) instead of 16 addresses a whole dy row (4 KB) apart:
  // the gather made the hidden layer's input gradient 45 us for 1.6 GFLOP at batch 256 (0.23 of f32 MFMA).
  // outT [ld_out][M] (non-split only): the result is ALSO stored transposed, for the next layer's dyT.
  const float* dyT;
  int ldyT;
  float* outT;
};

// grid = (64-column tiles, row splits, n_prob * m-chunks of 64), block = 256 (4 waves split the rows)
#define RB_NL_DX_LDS (2 * 64 * 64)   // floats: two wave tiles (the four waves meet pairwise, see the reduction below)
// MTC = 16-row m tiles a workgroup can hold (4: 64-row m-chunks; 2: batches of <= 32 — half the accumulators), ST = row-steps of 4
// weight rows whose loads are in flight together.  <2, 8> (RB_OPTS h_deep, batch <= 32): a wave's 64 rows are TWO dependent load ->
// MFMA trips instead of four (round6_wg_timeline_b32.txt: first trip 4.6 us, the three behind it 6.4 of the workgroup's 14.6) —
// the rows enter the accumulators in the same order: an element's bits do not depend on ST.
template <int MTC, int ST>
__device__ __forceinline__ void rb_nl_dx_body(const NlDxArgs& a, int bx, int by, int bz, float* lds) {
  float (*s_red)[64][64] = reinterpret_cast<float (*)[64][64]>(lds);   // [2][64][64]
  const int lane = rb_lane(), wave = rb_wave();
  const int pi = bz % a.n_prob, mc = bz / a.n_prob;
  const NlDxProblem pr = a.prob[pi];
  const int m0 = mc * 64;
  if (m0 >= a.M) return;
  const int mt_cnt0 = (a.M - m0 >= 64) ? 4 : (a.M - m0 + 15) / 16;
  const int mt_cnt = mt_cnt0 < MTC ? mt_cnt0 : MTC;      // (the caller picks MTC >= the launch's m tiles)
  const int K = a.K;
  const int kt = bx * 64;
  const int row_end = pr.row_begin + pr.row_cnt;
  int rb = pr.row_begin + by * a.rows_per_split;
  int re = rb + a.rows_per_split;
  if (re > row_end) re = row_end;
  if (rb >= re) return;                                  // block-uniform
  const int per_wave = (((re - rb) + 3) / 4 + 3) / 4 * 4;  // rows per wave, multiple of 4
  int wr0 = rb + wave * per_wave, wr1 = wr0 + per_wave;
  if (wr1 > re) wr1 = re;

  const int c = lane & 15, q = lane >> 4;
  int col4 = kt + 4 * c;
  if (col4 > K - 4) col4 = K - 4;                        // clamped lanes are never stored
  const float4 e0 = rb_ld4(a.w.ein + pr.ein_off0 + col4);
  const float4 e1 = rb_ld4(a.w.ein + pr.ein_off1 + col4);

  rb_f32x4 acc[MTC][4];
#pragma unroll
  for (int mt = 0; mt < MTC; ++mt)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[mt][j][e] = 0.0f;

  // (staging this workgroup's dY slab in LDS first — row-contiguous loads instead of 16 dy rows per instruction —
  // measured slower: k_nl_bwd 16.7 -> 18.0 us; dY is 128 KB and L1/L2-resident, the extra phase costs more)
#if defined(RB_STAMP)
  const int kid_ = a.K > 1000 ? 9 : 8;
  RB_WGT(kid_, (int)blockIdx.x, 2);
#endif
  for (int nb = wr0; nb < wr1; nb += 4 * ST) {           // ST row-steps of loads in flight per iteration
    float4 w4[ST];
    float av[ST][MTC];
#pragma unroll
    for (int st = 0; st < ST; ++st) {
      const int n = nb + 4 * st + q;
      const bool nv = n < wr1;
      const int nc = nv ? n : wr1 - 1;
      w4[st] = rb_noisy4(rb_ld4(a.w.mu + (int64_t)nc * K + col4), rb_ld4(a.w.sigma + (int64_t)nc * K + col4),
                         a.w.eout[nc], nc >= pr.ein_split_row ? e1 : e0);
#pragma unroll
      for (int mt = 0; mt < MTC; ++mt) {
        const int m = m0 + 16 * mt + c;
        av[st][mt] = (mt < mt_cnt && nv && m < a.M) ? (a.dyT ? a.dyT[(int64_t)n * a.ldyT + m] : a.dy[(int64_t)m * a.ldy + n]) : 0.0f;
      }
    }
#pragma unroll
    for (int st = 0; st < ST; ++st) {
      if (nb + 4 * st < wr1) {                             // wave-uniform
#pragma unroll
        for (int mt = 0; mt < MTC; ++mt) {
          if (mt < mt_cnt) {
            acc[mt][0] = rb_mfma16(av[st][mt], w4[st].x, acc[mt][0]);
            acc[mt][1] = rb_mfma16(av[st][mt], w4[st].y, acc[mt][1]);
            acc[mt][2] = rb_mfma16(av[st][mt], w4[st].z, acc[mt][2]);
            acc[mt][3] = rb_mfma16(av[st][mt], w4[st].w, acc[mt][3]);
          }
        }
      }
    }
#if defined(RB_STAMP)
    if (nb == wr0) RB_WGT(kid_, (int)blockIdx.x, 3);
#endif
  }
#if defined(RB_STAMP)
  RB_WGT(kid_, (int)blockIdx.x, 4);
#endif
  // cross-wave sum in a fixed order, (w0 + w2) + (w1 + w3), through two 16 KB tiles instead of four (the LDS footprint
  // sets how many workgroups of the fused backward launch a CU holds)
  if (wave >= 2) {
#pragma unroll
    for (int mt = 0; mt < MTC; ++mt)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < 4; ++j) s_red[wave - 2][(mt * 4 + e) * 4 + j][lane] = acc[mt][j][e];
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int mt = 0; mt < MTC; ++mt)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[mt][j][e] += s_red[wave][(mt * 4 + e) * 4 + j][lane];
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int mt = 0; mt < MTC; ++mt)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < 4; ++j) s_red[wave][(mt * 4 + e) * 4 + j][lane] = acc[mt][j][e];
  }
  __syncthreads();
#if defined(RB_STAMP)
  RB_WGT(kid_, (int)blockIdx.x, 5);
#endif
  const int slots = mt_cnt * 4;                          // (mt, e) pairs; j is the float4 lane
  for (int idx = (int)threadIdx.x; idx < slots * 64; idx += 256) {
    const int slot = idx >> 6, l = idx & 63;
    float4 v;
    float* vv = &v.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int s = slot * 4 + j;
      vv[j] = s_red[0][s][l] + s_red[1][s][l];
    }
    const int mt = slot >> 2, e = slot & 3;
    const int m = m0 + 16 * mt + 4 * (l >> 4) + e;
    const int k = kt + 4 * (l & 15);
    if (m < a.M && k < K) {
      const int64_t o = ((int64_t)by * a.M + m) * a.ld_out + pr.out_off + k;
      if (a.mask_src) {
        const float4 ms = rb_ld4(a.mask_src + o);
        v.x = ms.x > 0.0f ? v.x : 0.0f;
        v.y = ms.y > 0.0f ? v.y : 0.0f;
        v.z = ms.z > 0.0f ? v.z : 0.0f;
        v.w = ms.w > 0.0f ? v.w : 0.0f;
      }
      rb_st4(a.out + o, v);
      if (a.outT && a.mask_src) {                      // (non-split launches only: block-uniform)
        float* ot = a.outT + (int64_t)(pr.out_off + k) * a.M + m;
        ot[0] = v.x; ot[a.M] = v.y; ot[2 * (int64_t)a.M] = v.z; ot[3 * (int64_t)a.M] = v.w;
      }
    }
  }
}
// Tall form for SMALL weight matrices at batch <= 32 (the output layer: 357 x 512, 1.46 MB of mu | sigma): the same 64-column
// tiles, EIGHT waves per workgroup.  With four waves the advantage stream's 306 rows are 80 rows per wave = five dependent
// load -> MFMA iterations of ~1.5 us each (a memory round trip here is ~0.8 us, the 32 MFMAs behind it 0.4 us, nothing
// overlaps: 8.7 of the workgroup's 12.1 us, tools/wg_timeline.py; 16-column tiles on four times the workgroups measured
// SLOWER — 17 us — their dword loads cost more issue than the bytes they save per CU).  Eight waves take 40 rows each in ONE batch of loads: one
// round trip, 10 MFMAs per accumulator chain, then one LDS pass sums the eight partial tiles in wave order.
// Non-split launches with M <= 32 only; block = 64 * RB_NL_DXT_WAVES threads.
#define RB_NL_DXT_WAVES 8        // (16 waves = 1024 threads cap the kernel at 128 registers: the weight-gradient body of the same launch spilled)
#define RB_NL_DXT_LDS (RB_NL_DXT_WAVES * 32 * 64)
// CW = columns per lane: 4 = 64-column tiles (16-byte loads), 2 = 32-column tiles (8-byte loads, still one whole 128-byte line per
// weight row and row-step: twice the workgroups at half the bytes each — the advantage stream's 64-column workgroup pulls 157 KB
// of mu | sigma through ONE CU, 8.0 us against the value stream's 5.3; RB_OPTS z_narrow).  Same rows per wave, same wave-order sum:
// an element's value does not depend on CW.
template <int CW>
struct RbVec;
template <> struct RbVec<4> { typedef float4 T; };
template <> struct RbVec<2> { typedef float2 T; };
template <int CW>
__device__ __forceinline__ void rb_ldv(const float* p, float (&v)[CW]) {
  const typename RbVec<CW>::T t = *reinterpret_cast<const typename RbVec<CW>::T*>(p);
  v[0] = t.x; v[1] = t.y;
  if constexpr (CW == 4) { v[2] = t.z; v[3] = t.w; }
}
template <int CW>
__device__ __forceinline__ void rb_stv(float* p, const float (&v)[CW]) {
  typename RbVec<CW>::T t;
  t.x = v[0]; t.y = v[1];
  if constexpr (CW == 4) { t.z = v[2]; t.w = v[3]; }
  *reinterpret_cast<typename RbVec<CW>::T*>(p) = t;
}
template <int CW>
__device__ __forceinline__ void rb_nl_dx_body_tall(const NlDxArgs& a, int bx, int bz, float* lds) {
  float (*s_red)[32][64] = reinterpret_cast<float (*)[32][64]>(lds);   // [waves][(mt * 4 + e) * 4 + j][lane]
  const int lane = rb_lane(), wave = rb_wave();
  const int pi = bz % a.n_prob;
  const NlDxProblem pr = a.prob[pi];
  const int mt_cnt = (a.M + 15) / 16;                     // <= 2
  const int K = a.K;
  const int kt = bx * (16 * CW);
  const int row_end = pr.row_begin + pr.row_cnt;
  constexpr int ST = 10;                                  // row-steps of 4 rows in flight per wave: 40 rows
  const int per_wave = ((pr.row_cnt + RB_NL_DXT_WAVES - 1) / RB_NL_DXT_WAVES + 3) / 4 * 4;   // rows per wave, multiple of 4
  const int wr0 = pr.row_begin + wave * per_wave;
  int wr1 = wr0 + per_wave;
  if (wr1 > row_end) wr1 = row_end;
  const int c = lane & 15, q = lane >> 4;
  int col4 = kt + CW * c;
  if (col4 > K - CW) col4 = K - CW;                      // clamped lanes are never stored
  float e0[CW], e1[CW];
  rb_ldv<CW>(a.w.ein + pr.ein_off0 + col4, e0);
  rb_ldv<CW>(a.w.ein + pr.ein_off1 + col4, e1);
  // the ReLU mask of this thread's output cells: requested with the operands (in the epilogue: one more dependent trip)
  constexpr int TT = 64 * RB_NL_DXT_WAVES, EIT = (8 * 64 + TT - 1) / TT;     // 8 (mt, e) slots x 64 lanes over the threads
  float msk[EIT][CW];
  int64_t oidx[EIT];
#pragma unroll
  for (int it = 0; it < EIT; ++it) {
    const int idx = (int)threadIdx.x + it * TT;            // slot = idx >> 6 over (mt, e), lane = idx & 63
    const int slot = idx >> 6, l = idx & 63;
    const int m = 16 * (slot >> 2) + 4 * (l >> 4) + (slot & 3);
    int k = kt + CW * (l & 15);
    if (k > K - CW) k = K - CW;
    oidx[it] = (int64_t)(m < a.M ? m : a.M - 1) * a.ld_out + pr.out_off + k;
    if (a.mask_src) rb_ldv<CW>(a.mask_src + oidx[it], msk[it]);
    else {
#pragma unroll
      for (int j = 0; j < CW; ++j) msk[it][j] = 1.0f;
    }
  }
  rb_f32x4 acc[2][CW];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int j = 0; j < CW; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[mt][j][e] = 0.0f;
  for (int nb = wr0; nb < wr1; nb += 4 * ST) {            // (one pass for up to 320 rows per problem)
    float w4[ST][CW];
    float av[ST][2];
#pragma unroll
    for (int st = 0; st < ST; ++st) {
      const int n = nb + 4 * st + q;
      const bool nv = n < wr1;
      const int nc = nv ? n : wr1 - 1;
      float mu[CW], sg[CW];
      rb_ldv<CW>(a.w.mu + (int64_t)nc * K + col4, mu);
      rb_ldv<CW>(a.w.sigma + (int64_t)nc * K + col4, sg);
      const float eo = a.w.eout[nc];
      const bool second = nc >= pr.ein_split_row;
#pragma unroll
      for (int j = 0; j < CW; ++j) w4[st][j] = mu[j] + sg[j] * (eo * (second ? e1[j] : e0[j]));     // rb_noisy4's expression
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int m = 16 * mt + c;
        av[st][mt] = (mt < mt_cnt && nv && m < a.M) ? (a.dyT ? a.dyT[(int64_t)n * a.ldyT + m] : a.dy[(int64_t)m * a.ldy + n]) : 0.0f;
      }
    }
#pragma unroll
    for (int st = 0; st < ST; ++st) {
      if (nb + 4 * st < wr1) {                             // wave-uniform
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          if (mt < mt_cnt) {
#pragma unroll
            for (int j = 0; j < CW; ++j) acc[mt][j] = rb_mfma16(av[st][mt], w4[st][j], acc[mt][j]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int j = 0; j < CW; ++j) s_red[wave][(mt * 4 + e) * 4 + j][lane] = acc[mt][j][e];
  __syncthreads();
#pragma unroll
  for (int it = 0; it < EIT; ++it) {
    const int idx = (int)threadIdx.x + it * TT;
    const int slot = idx >> 6, l = idx & 63;
    if (slot >= mt_cnt * 4) continue;
    float vv[CW];
#pragma unroll
    for (int j = 0; j < CW; ++j) {
      float s = s_red[0][slot * 4 + j][l];
#pragma unroll
      for (int w = 1; w < RB_NL_DXT_WAVES; ++w) s += s_red[w][slot * 4 + j][l];     // fixed order w0, w1, ...
      vv[j] = s;
    }
    const int m = 16 * (slot >> 2) + 4 * (l >> 4) + (slot & 3);
    const int k = kt + CW * (l & 15);
    if (m < a.M && k < K) {
      if (a.mask_src) {
#pragma unroll
        for (int j = 0; j < CW; ++j) vv[j] = msk[it][j] > 0.0f ? vv[j] : 0.0f;
      }
      rb_stv<CW>(a.out + oidx[it], vv);
      if (a.outT && a.mask_src) {
        float* ot = a.outT + (int64_t)(pr.out_off + k) * a.M + m;
#pragma unroll
        for (int j = 0; j < CW; ++j) ot[(int64_t)j * a.M] = vv[j];
      }
    }
  }
}

// ===================================================================== weight gradient ==
// g_mu[n][k] = sum_m dy[m][n] * x[m][x_off + k] ; g_sigma = g_mu * (eps_out[n]*eps_in[k]) ;
// g_bmu[n] = sum_m dy[m][n] ; g_bsigma = g_bmu * eps_out[n].   One writer per element.
struct NlDwProblem {
  int row_begin, row_cnt;    // weight rows == columns of dy
  int x_off, ein_off;
  int tile_begin;            // first 16-row tile index in grid.y
};
struct NlDwArgs {
  const float* dy;           // [M][ldy]
  const float* x;            // [M][ldx]
  int ldy, ldx, M, K;
  int n_prob;
  NlDwProblem prob[2];
  float *g_mu, *g_sigma, *g_bmu, *g_bsigma;
  const float *eout, *ein;
  float* sq_part;            // optional: one slot per (block, wave) receiving the sum of squares of what that wave wrote
                             // (feeds clip_grad_norm_ without another pass over the 27 MB gradient)
  int deep;                  // ct == 4: every tile's operands in flight before the first MFMA (rb_nl_dw_body_pipe_all)
  int ct;                    // > 0: pipelined body, `ct` column tiles per wave (M <= 32); 0: one tile per wave;
                             // < 0: the LDS-shared 64 x 64 tile body for large M (rb_nl_dw_body_wide)
  int no_sigma;              // 1: g_sigma is formed for the sum of squares only and NOT stored (the hosted optimiser pass forms it
                             // again from g_mu and the same noise, adam_body.h; pipelined body only)
  int norm_only;             // 1: the weight-gradient tiles are computed for their sum of squares only and NOT stored (the
                             // optimiser pass recomputes each tile while it streams the parameters, k_clip_adam<FUSED>);
                             // the bias gradients are still written
  // Replica exchange: the reduction rows are the all-gathered factor blocks of `world` ranks — row m lives in
  // block m / rpb at row m % rpb, blocks `bstride` floats apart (rpb == 0: one plain matrix) — and the result is the
  // replica MEAN: scale = 1 / world (a power of two for 2/4/8 replicas, i.e. exact; 1.0f otherwise leaves every bit alone)
  int rpb;
  int64_t bstride;
  float scale;
  // ... and every rank's gradient is noisy with ITS OWN epsilon (each replica resamples its own noise, agent.py:49,74):
  // g_sigma = mean_r g_mu_r * (eps_out_r x eps_in_r).  The gathered blocks therefore carry each rank's noise buffer:
  // rank r's eps_out / eps_in of this layer sit at noise_blocks + r * bstride + eout_noff / ein_noff.
  const float* noise_blocks;
  int64_t eout_noff, ein_noff;
};

// grid = (256-column tiles, 16-row tiles), block = 256: wave w owns columns [256*bx + 64*w, +64)
__device__ __forceinline__ void rb_nl_dw_body(const NlDwArgs& a, int bx, int by, int slot_base) {
  const int lane = rb_lane(), wave = rb_wave();
  const int kt = bx * 256 + wave * 64;
  if (kt >= a.K) {                                       // wave-uniform, no barriers below
    if (a.sq_part && lane == 0) a.sq_part[slot_base + wave] = 0.0f;
    return;
  }
  const int g = (a.n_prob > 1 && by >= a.prob[1].tile_begin) ? 1 : 0;
  const NlDwProblem pr = a.prob[g];
  const int row0 = pr.row_begin + (by - pr.tile_begin) * 16;
  const int row_end = pr.row_begin + pr.row_cnt;
  const int c = lane & 15, q = lane >> 4;
  int col4 = kt + 4 * c;
  const bool cv = col4 < a.K;
  if (!cv) col4 = a.K - 4;
  int arow = row0 + c;
  const bool av_ok = arow < row_end;
  if (!av_ok) arow = row_end - 1;

  rb_f32x4 acc[4], accb;
#pragma unroll
  for (int e = 0; e < 4; ++e) { accb[e] = 0.0f; acc[0][e] = 0.0f; acc[1][e] = 0.0f; acc[2][e] = 0.0f; acc[3][e] = 0.0f; }
  const bool do_bias = kt == 0;
  // epilogue operands, requested with the first operand loads (asked for after the loop they cost a second round trip)
  const float4 e4 = rb_ld4(a.ein + pr.ein_off + col4);
  float eo4[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int n = row0 + 4 * q + e;
    eo4[e] = a.eout[n < row_end ? n : row_end - 1];
  }
  // 8 reduction steps (32 samples) of loads per trip, and the NEXT trip's loads requested before this trip's MFMAs: at batch 256 the
  // eight trips were eight dependent round trips (13.3 of the output layer's 14.4 us backward launch, round6_wg_timeline_b256.txt).
  // Same MFMAs in the same order: bit-identical.
  auto issue = [&](int mb, float (&avs)[8], float4 (&xs)[8]) {
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      const int m = mb + 4 * st + q;
      const bool mv = m < a.M;
      const int mcl = mv ? m : a.M - 1;
      avs[st] = a.dy[(int64_t)mcl * a.ldy + arow];
      xs[st] = rb_ld4(a.x + (int64_t)mcl * a.ldx + pr.x_off + col4);
    }
  };
  auto mfmas = [&](int mb, const float (&avs)[8], const float4 (&xs)[8]) {
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      if (mb + 4 * st < a.M) {                             // uniform
        const bool mv = mb + 4 * st + q < a.M;
        const float av = (mv && av_ok) ? avs[st] : 0.0f;
        const float4 x = xs[st];
        acc[0] = rb_mfma16(av, mv ? x.x : 0.0f, acc[0]);
        acc[1] = rb_mfma16(av, mv ? x.y : 0.0f, acc[1]);
        acc[2] = rb_mfma16(av, mv ? x.z : 0.0f, acc[2]);
        acc[3] = rb_mfma16(av, mv ? x.w : 0.0f, acc[3]);
        if (do_bias) accb = rb_mfma16(av, 1.0f, accb);      // wave-uniform
      }
    }
  };
  {
    float avs0[8], avs1[8];
    float4 xs0[8], xs1[8];
#if defined(RB_DW_NOPRE)                                   // (A/B build: one trip at a time)
    for (int mb = 0; mb < a.M; mb += 32) { issue(mb, avs0, xs0); mfmas(mb, avs0, xs0); }
    (void)avs1; (void)xs1;
#else
    issue(0, avs0, xs0);
    for (int mb = 0; mb < a.M; mb += 64) {
      const bool second = mb + 32 < a.M;                   // uniform
      if (second) issue(mb + 32, avs1, xs1);
      mfmas(mb, avs0, xs0);
      if (second) {
        if (mb + 64 < a.M) issue(mb + 64, avs0, xs0);
        mfmas(mb + 32, avs1, xs1);
      }
    }
#endif
  }
  float sq = 0.0f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int n = row0 + 4 * q + e;
    if (n < row_end) {
      const float eo = eo4[e];
      if (cv) {
        float4 gm, gs;
        gm.x = acc[0][e]; gm.y = acc[1][e]; gm.z = acc[2][e]; gm.w = acc[3][e];
        gs.x = gm.x * (eo * e4.x); gs.y = gm.y * (eo * e4.y); gs.z = gm.z * (eo * e4.z); gs.w = gm.w * (eo * e4.w);
        rb_st4(a.g_mu + (int64_t)n * a.K + col4, gm);      // (write-through stores measured no gain here)
        rb_st4(a.g_sigma + (int64_t)n * a.K + col4, gs);
        sq = fmaf(gm.x, gm.x, sq); sq = fmaf(gm.y, gm.y, sq); sq = fmaf(gm.z, gm.z, sq); sq = fmaf(gm.w, gm.w, sq);
        sq = fmaf(gs.x, gs.x, sq); sq = fmaf(gs.y, gs.y, sq); sq = fmaf(gs.z, gs.z, sq); sq = fmaf(gs.w, gs.w, sq);
      }
      if (do_bias && c == 0) {
        const float gb = accb[e], gbs = accb[e] * eo;
        a.g_bmu[n] = gb;
        a.g_bsigma[n] = gbs;
        sq = fmaf(gb, gb, sq); sq = fmaf(gbs, gbs, sq);
      }
    }
  }
  if (a.sq_part) {                                        // wave-uniform
    sq = rb_wave_sum(sq);
    if (lane == 0) a.sq_part[slot_base + wave] = sq;
  }
}

// Replica-exchange variant (rb_learner_finish_grads): the reduction rows are `M / rpb` rank blocks of rpb rows each.  Every
// rank's block is reduced on its own (same MFMA order as the single-device bodies, so acc_r has the bits that rank alone
// would have produced), then folded in rank order:  g_mu += acc_r ;  g_sigma += acc_r * (eps_out_r * eps_in_r) ; the sums
// are scaled by 1 / world at the end — the arithmetic of "every replica computes its gradient, then all-reduce(mean)".
__device__ __forceinline__ void rb_nl_dw_body_ranks(const NlDwArgs& a, int bx, int by, int slot_base) {
  const int lane = rb_lane(), wave = rb_wave();
  const int kt = bx * (int)blockDim.x + wave * 64;       // a wave owns 64 columns: 256 per 4-wave block, 512 per 8-wave block
  if (kt >= a.K) {                                       // wave-uniform, no barriers below
    if (a.sq_part && lane == 0) a.sq_part[slot_base + wave] = 0.0f;
    return;
  }
  const int g = (a.n_prob > 1 && by >= a.prob[1].tile_begin) ? 1 : 0;
  const NlDwProblem pr = a.prob[g];
  const int row0 = pr.row_begin + (by - pr.tile_begin) * 16;
  const int row_end = pr.row_begin + pr.row_cnt;
  const int c = lane & 15, q = lane >> 4;
  int col4 = kt + 4 * c;
  const bool cv = col4 < a.K;
  if (!cv) col4 = a.K - 4;
  int arow = row0 + c;
  const bool av_ok = arow < row_end;
  if (!av_ok) arow = row_end - 1;
  const bool do_bias = kt == 0;
  const int nranks = a.M / a.rpb;
  rb_f32x4 gm[4], gs[4], gbm, gbs;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    gbm[e] = 0.0f; gbs[e] = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { gm[j][e] = 0.0f; gs[j][e] = 0.0f; }
  }
  for (int r = 0; r < nranks; ++r) {
    const float* dy = a.dy + (int64_t)r * a.bstride;
    const float* x = a.x + (int64_t)r * a.bstride;
    const float* nz = a.noise_blocks + (int64_t)r * a.bstride;
    const float4 e4 = rb_ld4(nz + a.ein_noff + pr.ein_off + col4);
    float eo4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = row0 + 4 * q + e;
      eo4[e] = nz[a.eout_noff + (n < row_end ? n : row_end - 1)];
    }
    rb_f32x4 acc[4], accb;
#pragma unroll
    for (int e = 0; e < 4; ++e) { accb[e] = 0.0f; acc[0][e] = 0.0f; acc[1][e] = 0.0f; acc[2][e] = 0.0f; acc[3][e] = 0.0f; }
    for (int mb = 0; mb < a.rpb; mb += 32) {
      float avs[8];
      float4 xs[8];
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        const int m = mb + 4 * st + q;
        const bool mv = m < a.rpb;
        const int mcl = mv ? m : a.rpb - 1;
        avs[st] = (mv && av_ok) ? dy[(int64_t)mcl * a.ldy + arow] : 0.0f;
        xs[st] = rb_ld4(x + (int64_t)mcl * a.ldx + pr.x_off + col4);
        if (!mv) { xs[st].x = 0.0f; xs[st].y = 0.0f; xs[st].z = 0.0f; xs[st].w = 0.0f; }
      }
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        if (mb + 4 * st < a.rpb) {                         // uniform
          acc[0] = rb_mfma16(avs[st], xs[st].x, acc[0]);
          acc[1] = rb_mfma16(avs[st], xs[st].y, acc[1]);
          acc[2] = rb_mfma16(avs[st], xs[st].z, acc[2]);
          acc[3] = rb_mfma16(avs[st], xs[st].w, acc[3]);
          if (do_bias) accb = rb_mfma16(avs[st], 1.0f, accb);   // wave-uniform
        }
      }
    }
    const float ej[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        gm[j][e] = gm[j][e] + acc[j][e];
        gs[j][e] = gs[j][e] + acc[j][e] * (eo4[e] * ej[j]);
      }
      gbm[e] = gbm[e] + accb[e];
      gbs[e] = gbs[e] + accb[e] * eo4[e];
    }
  }
  float sq = 0.0f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int n = row0 + 4 * q + e;
    if (n < row_end) {
      if (cv) {
        float4 m4, s4;
        m4.x = gm[0][e] * a.scale; m4.y = gm[1][e] * a.scale; m4.z = gm[2][e] * a.scale; m4.w = gm[3][e] * a.scale;
        s4.x = gs[0][e] * a.scale; s4.y = gs[1][e] * a.scale; s4.z = gs[2][e] * a.scale; s4.w = gs[3][e] * a.scale;
        rb_st4(a.g_mu + (int64_t)n * a.K + col4, m4);
        rb_st4(a.g_sigma + (int64_t)n * a.K + col4, s4);
        sq = fmaf(m4.x, m4.x, sq); sq = fmaf(m4.y, m4.y, sq); sq = fmaf(m4.z, m4.z, sq); sq = fmaf(m4.w, m4.w, sq);
        sq = fmaf(s4.x, s4.x, sq); sq = fmaf(s4.y, s4.y, sq); sq = fmaf(s4.z, s4.z, sq); sq = fmaf(s4.w, s4.w, sq);
      }
      if (do_bias && c == 0) {
        const float gb = gbm[e] * a.scale, gbsv = gbs[e] * a.scale;
        a.g_bmu[n] = gb;
        a.g_bsigma[n] = gbsv;
        sq = fmaf(gb, gb, sq); sq = fmaf(gbsv, gbsv, sq);
      }
    }
  }
  if (a.sq_part) {                                        // wave-uniform
    sq = rb_wave_sum(sq);
    if (lane == 0) a.sq_part[slot_base + wave] = sq;
  }
}
// Pipelined variant for M <= 32 (one reduction pass per tile): a wave walks `ct` column tiles 256 apart with the same dY
// operand, requesting tile j+1's activations before it multiplies and stores tile j — with one tile per wave every
// wave of the launch loaded, then multiplied, then stored at the same time (a read burst followed by a write burst).
// grid.x = ceil(K / (256 ct)); the sum-of-squares slot of a wave covers all its tiles.
__device__ __forceinline__ void rb_nl_dw_body_pipe(const NlDwArgs& a, int bx, int by, int slot_base) {
  const int lane = rb_lane(), wave = rb_wave();
  const int ct = a.ct;
  const int kt0 = bx * ct * 256 + wave * 64;
  if (kt0 >= a.K) {                                      // wave-uniform, no barriers below
    if (a.sq_part && lane == 0) a.sq_part[slot_base + wave] = 0.0f;
    return;
  }
  const int g = (a.n_prob > 1 && by >= a.prob[1].tile_begin) ? 1 : 0;
  const NlDwProblem pr = a.prob[g];
  const int row0 = pr.row_begin + (by - pr.tile_begin) * 16;
  const int row_end = pr.row_begin + pr.row_cnt;
  const int c = lane & 15, q = lane >> 4;
  int arow = row0 + c;
  const bool av_ok = arow < row_end;
  if (!av_ok) arow = row_end - 1;
  float eo4[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int n = row0 + 4 * q + e;
    eo4[e] = a.eout[n < row_end ? n : row_end - 1];
  }
  float avs[8];
  const float* xrow[8];
  float xmask[8];
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    const int m = 4 * st + q;
    const bool mv = m < a.M;
    const int mcl = mv ? m : a.M - 1;
    avs[st] = (mv && av_ok) ? a.dy[(int64_t)mcl * a.ldy + arow] : 0.0f;
    xrow[st] = a.x + (int64_t)mcl * a.ldx + pr.x_off;
    xmask[st] = mv ? 1.0f : 0.0f;
  }
  auto col_of = [&](int j) { int col4 = kt0 + j * 256 + 4 * c; return col4 < a.K ? col4 : a.K - 4; };
  float4 xs[8], xn[8], e4, e4n;
  {
    const int col4 = col_of(0);
    e4 = rb_ld4(a.ein + pr.ein_off + col4);
#pragma unroll
    for (int st = 0; st < 8; ++st) xs[st] = rb_ld4(xrow[st] + col4);
  }
  float sq = 0.0f;
  for (int j = 0; j < ct; ++j) {
    const int kt = kt0 + j * 256;
    if (kt >= a.K) break;                                // wave-uniform
    {                                                    // next tile's operands (clamped: always a legal address)
      const int coln = col_of(j + 1 < ct ? j + 1 : j);
      e4n = rb_ld4(a.ein + pr.ein_off + coln);
#pragma unroll
      for (int st = 0; st < 8; ++st) xn[st] = rb_ld4(xrow[st] + coln);
    }
    const bool cv = kt + 4 * c < a.K;
    const int col4 = col_of(j);
    const bool do_bias = kt == 0;
    rb_f32x4 acc[4], accb;
#pragma unroll
    for (int e = 0; e < 4; ++e) { accb[e] = 0.0f; acc[0][e] = 0.0f; acc[1][e] = 0.0f; acc[2][e] = 0.0f; acc[3][e] = 0.0f; }
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      if (4 * st < a.M) {                                // uniform
        acc[0] = rb_mfma16(avs[st], xs[st].x * xmask[st], acc[0]);
        acc[1] = rb_mfma16(avs[st], xs[st].y * xmask[st], acc[1]);
        acc[2] = rb_mfma16(avs[st], xs[st].z * xmask[st], acc[2]);
        acc[3] = rb_mfma16(avs[st], xs[st].w * xmask[st], acc[3]);
        if (do_bias) accb = rb_mfma16(avs[st], 1.0f, accb);   // wave-uniform
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = row0 + 4 * q + e;
      if (n < row_end) {
        const float eo = eo4[e];
        if (cv) {
          float4 gm, gs;
          gm.x = acc[0][e]; gm.y = acc[1][e]; gm.z = acc[2][e]; gm.w = acc[3][e];
          gs.x = gm.x * (eo * e4.x); gs.y = gm.y * (eo * e4.y); gs.z = gm.z * (eo * e4.z); gs.w = gm.w * (eo * e4.w);
          if (!a.norm_only) {                            // wave-uniform
            rb_st4(a.g_mu + (int64_t)n * a.K + col4, gm);
            if (!a.no_sigma) rb_st4(a.g_sigma + (int64_t)n * a.K + col4, gs);
          }
          sq = fmaf(gm.x, gm.x, sq); sq = fmaf(gm.y, gm.y, sq); sq = fmaf(gm.z, gm.z, sq); sq = fmaf(gm.w, gm.w, sq);
          sq = fmaf(gs.x, gs.x, sq); sq = fmaf(gs.y, gs.y, sq); sq = fmaf(gs.z, gs.z, sq); sq = fmaf(gs.w, gs.w, sq);
        }
        if (do_bias && c == 0) {
          const float gb = accb[e], gbs = accb[e] * eo;
          a.g_bmu[n] = gb;
          a.g_bsigma[n] = gbs;
          sq = fmaf(gb, gb, sq); sq = fmaf(gbs, gbs, sq);
        }
      }
    }
    e4 = e4n;
#pragma unroll
    for (int st = 0; st < 8; ++st) xs[st] = xn[st];
  }
  if (a.sq_part) {                                        // wave-uniform
    sq = rb_wave_sum(sq);
    if (lane == 0) a.sq_part[slot_base + wave] = sq;
  }
}
// ... with the operands of ALL CT column tiles of a wave requested before the first MFMA (RB_OPTS h_dw_deep): the one-tile-ahead
// form above turns a tile every ~3.3 us (13.2 us for the hidden layer's four: a round trip beside the launch's 39 MB is longer than
// a tile's 40 MFMAs), this one pays the trip once.  Same tiles, same order inside a tile: bit-identical gradients and norm partials.
template <int CT>
__device__ __forceinline__ void rb_nl_dw_body_pipe_all(const NlDwArgs& a, int bx, int by, int slot_base) {
  const int lane = rb_lane(), wave = rb_wave();
  constexpr int ct = CT;
  const int kt0 = bx * ct * 256 + wave * 64;
  if (kt0 >= a.K) {                                      // wave-uniform, no barriers below
    if (a.sq_part && lane == 0) a.sq_part[slot_base + wave] = 0.0f;
    return;
  }
  const int g = (a.n_prob > 1 && by >= a.prob[1].tile_begin) ? 1 : 0;
  const NlDwProblem pr = a.prob[g];
  const int row0 = pr.row_begin + (by - pr.tile_begin) * 16;
  const int row_end = pr.row_begin + pr.row_cnt;
  const int c = lane & 15, q = lane >> 4;
  int arow = row0 + c;
  const bool av_ok = arow < row_end;
  if (!av_ok) arow = row_end - 1;
  float eo4[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int n = row0 + 4 * q + e;
    eo4[e] = a.eout[n < row_end ? n : row_end - 1];
  }
  float avs[8];
  const float* xrow[8];
  float xmask[8];
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    const int m = 4 * st + q;
    const bool mv = m < a.M;
    const int mcl = mv ? m : a.M - 1;
    avs[st] = (mv && av_ok) ? a.dy[(int64_t)mcl * a.ldy + arow] : 0.0f;
    xrow[st] = a.x + (int64_t)mcl * a.ldx + pr.x_off;
    xmask[st] = mv ? 1.0f : 0.0f;
  }
  auto col_of = [&](int j) { int col4 = kt0 + j * 256 + 4 * c; return col4 < a.K ? col4 : a.K - 4; };
  float4 xa[CT][8], ea[CT];
#pragma unroll
  for (int j = 0; j < CT; ++j) {                         // (clamped: always a legal address)
    const int col4 = col_of(j);
    ea[j] = rb_ld4(a.ein + pr.ein_off + col4);
#pragma unroll
    for (int st = 0; st < 8; ++st) xa[j][st] = rb_ld4(xrow[st] + col4);
  }
  float sq = 0.0f;
#pragma unroll
  for (int j = 0; j < CT; ++j) {
    const int kt = kt0 + j * 256;
    if (kt >= a.K) break;                                // wave-uniform
    const float4 e4 = ea[j];
    const float4 (&xs)[8] = xa[j];
    const bool cv = kt + 4 * c < a.K;
    const int col4 = col_of(j);
    const bool do_bias = kt == 0;
    rb_f32x4 acc[4], accb;
#pragma unroll
    for (int e = 0; e < 4; ++e) { accb[e] = 0.0f; acc[0][e] = 0.0f; acc[1][e] = 0.0f; acc[2][e] = 0.0f; acc[3][e] = 0.0f; }
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      if (4 * st < a.M) {                                // uniform
        acc[0] = rb_mfma16(avs[st], xs[st].x * xmask[st], acc[0]);
        acc[1] = rb_mfma16(avs[st], xs[st].y * xmask[st], acc[1]);
        acc[2] = rb_mfma16(avs[st], xs[st].z * xmask[st], acc[2]);
        acc[3] = rb_mfma16(avs[st], xs[st].w * xmask[st], acc[3]);
        if (do_bias) accb = rb_mfma16(avs[st], 1.0f, accb);   // wave-uniform
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = row0 + 4 * q + e;
      if (n < row_end) {
        const float eo = eo4[e];
        if (cv) {
          float4 gm, gs;
          gm.x = acc[0][e]; gm.y = acc[1][e]; gm.z = acc[2][e]; gm.w = acc[3][e];
          gs.x = gm.x * (eo * e4.x); gs.y = gm.y * (eo * e4.y); gs.z = gm.z * (eo * e4.z); gs.w = gm.w * (eo * e4.w);
          if (!a.norm_only) {                            // wave-uniform
            rb_st4(a.g_mu + (int64_t)n * a.K + col4, gm);
            if (!a.no_sigma) rb_st4(a.g_sigma + (int64_t)n * a.K + col4, gs);
          }
          sq = fmaf(gm.x, gm.x, sq); sq = fmaf(gm.y, gm.y, sq); sq = fmaf(gm.z, gm.z, sq); sq = fmaf(gm.w, gm.w, sq);
          sq = fmaf(gs.x, gs.x, sq); sq = fmaf(gs.y, gs.y, sq); sq = fmaf(gs.z, gs.z, sq); sq = fmaf(gs.w, gs.w, sq);
        }
        if (do_bias && c == 0) {
          const float gb = accb[e], gbs = accb[e] * eo;
          a.g_bmu[n] = gb;
          a.g_bsigma[n] = gbs;
          sq = fmaf(gb, gb, sq); sq = fmaf(gbs, gbs, sq);
        }
      }
    }
  }
  if (a.sq_part) {                                        // wave-uniform
    sq = rb_wave_sum(sq);
    if (lane == 0) a.sq_part[slot_base + wave] = sq;
  }
}
// the wide body as a launch of its own: inside k_nl_bwd it would inherit that kernel's register allocation (145 + 64: two
// workgroups per CU), and with 8 short chunks per workgroup it is latency-bound — it wants many resident workgroups

// Horizontal fusion: the weight-gradient and the input-gradient of one layer are independent given dY, so both run in
// ONE launch (one ~5 us kernel boundary less on the critical path).  Blocks [0, dw_x*dw_y) take the dW tiles, the rest
// the dX tiles.
struct NlBwdGrid { int dw_x, dw_y, dx_x, dx_y, dx_z; int dx_narrow; };   // dx_narrow: TALL: 32-column input-gradient tiles; else: 1 = rb_nl_dx_body<2, 8> (batch <= 32), 2 = <4, 8>: deep loads
// Optional third tenant of the output layer's backward launch: the sum-tree priority write-back (agent.py:100,
// memory.py:157-159).  It depends only on (tree indices, per-sample loss), both final before this launch, and is a
// single-workgroup latency chain — as one more block here it costs nothing on the step's critical path.
struct NlPriorityUpdate {
  int enabled;
  ReplayView view;
  const int64_t* tree_idx;
  const float* loss;
  int n;
  double omega;
  // the early draw (replay_internal.h rb_replay_spec_launch; enabled == 0): this launch is the first one behind the head kernel,
  // so its START proves the per-sample losses final — workgroup 0 stores go_epoch for the replay's stream to see (NULL: nothing)
  unsigned* go_flag;
  unsigned go_epoch;
};
#if defined(RB_STAMP)
extern __device__ long long g_span[64];
// [2 * slot] = earliest start, [2 * slot + 1] = latest end of the blocks that ran `slot` since the host last reset the array
#define RB_SPAN_BEGIN(slot) do { if (threadIdx.x == 0) atomicMin((unsigned long long*)&g_span[2 * (slot)], (unsigned long long)wall_clock64()); } while (0)
#define RB_SPAN_END(slot) do { __syncthreads(); if (threadIdx.x == 0) atomicMax((unsigned long long*)&g_span[2 * (slot) + 1], (unsigned long long)wall_clock64()); } while (0)
#else
#define RB_SPAN_BEGIN(slot) ((void)0)
#define RB_SPAN_END(slot) ((void)0)
#endif
// TALL: 512-thread workgroups whose input-gradient blocks run rb_nl_dx_body_tall (the output layer at batch <= 32: 40
// workgroups, LDS and occupancy are no concern); its weight-gradient blocks use the first four waves, the others leave.
template <bool TALL>
__global__ __launch_bounds__(TALL ? 64 * RB_NL_DXT_WAVES : 256) void k_nl_bwd(NlDwArgs dw, NlDxArgs dx, NlBwdGrid g, NlPriorityUpdate up) {
  // ONE LDS buffer for whichever body this workgroup runs (separate static arrays would add up: 132 KB, one workgroup
  // per CU for the whole launch; 32 KB lets five share a CU)
  constexpr int LDS0 = RB_NL_DX_LDS > UpdateLds<512, 256>::WORDS ? RB_NL_DX_LDS : UpdateLds<512, 256>::WORDS;
  constexpr int LDSW = TALL ? (RB_NL_DXT_LDS > LDS0 ? RB_NL_DXT_LDS : LDS0) : LDS0;
  __shared__ __attribute__((aligned(16))) float lds[LDSW];
  // the write-back block goes FIRST: it is the launch's longest single-workgroup chain and must not queue behind the tiles
  int b = (int)blockIdx.x;
  const int sb = dw.K > 1000 ? 3 : 0;                    // RB_STAMP builds: span slots of the hidden / output layer launch
  (void)sb;
  // RB_STAMP builds: per-workgroup timeline (tools/nl_timeline.py): kernel id 8 = output layer's launch, 9 = hidden layer's;
  // slot 0 start, 1 role (0 write-back, 1 dW, 2 dX), 6 end, 7 where it ran
  const int kid = dw.K > 1000 ? 9 : 8, wgb = (int)blockIdx.x;
  (void)kid; (void)wgb;
  RB_WGT(kid, wgb, 0);
  RB_WGT_HW(kid, wgb);
  if (up.go_flag && blockIdx.x == 0 && threadIdx.x == 0) {
#if defined(RB_HOST_INTERP)
    *up.go_flag = up.go_epoch;
#else
    __hip_atomic_store(up.go_flag, up.go_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
  }
  if (up.enabled) {
    if (TALL && threadIdx.x >= 256 && b == 0) return;    // (the write-back body is written for 256 threads)
    if (b == 0) {
      RB_SPAN_BEGIN(sb + 0);
      // (the one-wave sorted write-back of k_update measured no faster HERE — same-box A/B of two builds on all three configs:
      // 160.9 / 103.5 / 498.5 against 161.4 / 103.0 / 497.8 us per step — this block is not the launch's long pole)
      // enabled == 2 (RB_OPTS wb_auto, default): a sorted batch of <= 64 leaves — what the sampler hands back — on ONE wave without LDS
      // tables or workgroup barriers (rb_update_sorted_wave), anything else through the hashed body.  Round 4 measured no gain from it
      // HERE, when the input-gradient tiles were this launch's pole; since round 6's deeper bodies the write-back is (17.0 us hashed,
      // round6_final_wg_timeline_b32.txt).  Same tree, bit for bit, either way.
      if (up.enabled == 2) rb_update_auto<512, 256>(up.view, up.tree_idx, up.loss, up.n, 1, up.omega, lds);   // block-uniform
      else rb_update_body<512, 256>(up.view, up.tree_idx, up.loss, up.n, 1, up.omega, lds);     // n <= 256
      RB_SPAN_END(sb + 0);
      RB_WGT_ROLE(kid, wgb, 0);
      RB_WGT(kid, wgb, 6);
      return;
    }
    b -= 1;
  }
  const int ndw = g.dw_x * g.dw_y;
  if (b < ndw) {
    if (TALL && threadIdx.x >= 256) return;              // the weight-gradient bodies are 4-wave bodies (their barriers count
                                                         // the waves that are still alive)
    RB_SPAN_BEGIN(sb + 1);
    if (dw.ct == 4 && dw.deep) rb_nl_dw_body_pipe_all<4>(dw, b % g.dw_x, b / g.dw_x, 4 * b);     // block-uniform
    else if (dw.ct > 0) rb_nl_dw_body_pipe(dw, b % g.dw_x, b / g.dw_x, 4 * b);
    else rb_nl_dw_body(dw, b % g.dw_x, b / g.dw_x, 4 * b);
    RB_SPAN_END(sb + 1);
    RB_WGT_ROLE(kid, wgb, 1);
  } else {
    const int r = b - ndw;
    RB_SPAN_BEGIN(sb + 2);
    if constexpr (TALL) {
      if (g.dx_narrow) rb_nl_dx_body_tall<2>(dx, r % g.dx_x, r / g.dx_x, lds);      // block-uniform
      else rb_nl_dx_body_tall<4>(dx, r % g.dx_x, r / g.dx_x, lds);
    }
    else if (g.dx_narrow == 2) rb_nl_dx_body<4, 8>(dx, r % g.dx_x, (r / g.dx_x) % g.dx_y, r / (g.dx_x * g.dx_y), lds);   // block-uniform
    else if (g.dx_narrow) rb_nl_dx_body<2, 8>(dx, r % g.dx_x, (r / g.dx_x) % g.dx_y, r / (g.dx_x * g.dx_y), lds);
    else rb_nl_dx_body<4, 4>(dx, r % g.dx_x, (r / g.dx_x) % g.dx_y, r / (g.dx_x * g.dx_y), lds);
    RB_SPAN_END(sb + 2);
    RB_WGT_ROLE(kid, wgb, 2);
  }
  RB_WGT(kid, wgb, 6);
}
