// conv_lds.h — convolution kernels with LDS-resident operands (model.py:56-58,61-62 and their
// adjoints).
//
// At batch 32 the Rainbow conv stack is tiny per image (28 KB of u8 input, <= 51 KB activations,
// <= 147 KB of weights per layer) and the generic implicit GEMM of gemm_core.h spends its time on
// per-element im2col address arithmetic and dependent global loads, not on MFMAs.  Here a
// workgroup stages what it needs ONCE with wide coalesced loads —
//     the input patch of its output positions (u8 frames decoded to exact x/255 on the way in),
//     a 32-channel slab of the weights, transposed to [k][32] so MFMA operand reads are
//     bank-conflict free,
//     a k -> patch-offset table (no div/mod in the inner loop),
// — and then runs a pure LDS -> v_mfma_f32_32x32x2_f32 loop.  The four waves split K; their
// accumulators are reduced through LDS in a fixed order (deterministic) and the epilogue (bias,
// ReLU / ReLU mask) stores rows that are contiguous in the NCHW activation.
#pragma once
#include "learner_problems.h"

#if defined(RB_STAMP)
extern __device__ long long g_cstamp[64];
#ifndef RB_MSTAMP_KS
#define RB_MSTAMP_KS 4
#endif
#define RB_CSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) g_cstamp[i] = wall_clock64(); } while (0)
#define RB_CSTAMP_LAST(i) do { if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1 && blockIdx.z == gridDim.z - 1) g_cstamp[i] = wall_clock64(); } while (0)
// per-workgroup timeline: g_wgt[kernel id][workgroup][slot] = wall_clock64 (100 MHz) at phase boundaries, slot 7 = where
// it ran (XCC id << 16 | HW_ID bits) — tools/wg_timeline.py draws the schedule of a launch from it
#define RB_WGT_KERNELS 14
#define RB_WGT_WGS 2048
extern __device__ long long g_wgt[RB_WGT_KERNELS][RB_WGT_WGS][8];
#define RB_WGT(kid, wg, slot) do { if (threadIdx.x == 0 && (wg) < RB_WGT_WGS) g_wgt[kid][wg][slot] = wall_clock64(); } while (0)
#define RB_WGT_HW(kid, wg) do { if (threadIdx.x == 0 && (wg) < RB_WGT_WGS) g_wgt[kid][wg][7] = ((long long)__builtin_amdgcn_s_getreg(6164) << 16) | (__builtin_amdgcn_s_getreg(63492) & 0xffff); } while (0)
#define RB_WGT_ROLE(kid, wg, role) do { if (threadIdx.x == 0 && (wg) < RB_WGT_WGS) g_wgt[kid][wg][1] = (role); } while (0)
#else
#define RB_WGT_ROLE(kid, wg, role) ((void)0)
#define RB_CSTAMP(i) ((void)0)
#define RB_CSTAMP_LAST(i) ((void)0)
#define RB_WGT(kid, wg, slot) ((void)0)
#define RB_WGT_HW(kid, wg) ((void)0)
#endif
// bank swizzle of the forward kernels' row-major weight slab (rb_conv_fwd_body): column k of row m
__device__ __forceinline__ int rb_wswz(int m, int k) { return (k & ~3) | ((k & 3) ^ ((m >> 3) & 3)); }
struct ConvLdsFwdArgs {
  int cin, cout;
  int n_on;                  // images [0,n_on) use net 0, the rest net 1
  const float* w[2];         // [cout][cin*KK]
  const float* bias[2];
  ImgSrc src;                // FIRST layer input
  const float* in_f;         // later layers: [img][cin][IP]
  float* out;                // [img][cout][P]
  float* out_blocked;        // optional second copy of the flattened output in the k-blocked layout of noisy_linear.h
  int rows_total;            //   ... with this many rows (images)
  int ipb;                   // k_conv_fwd_multi: images per workgroup
  int img_fast;              // k_conv_fwd_lds: grid = (images, cout tiles, position chunks) — the image is the fastest block index
};

// ---- shared staging helpers ------------------------------------------------------------------
// weights slab [32 rows starting at row0][K] (row-major, K % 4 == 0) -> s_w[k][33], rows >= rows_valid zeroed,
// rows k in [K, KPAD) zeroed
__device__ __forceinline__ void rb_stage_weights_t(float* s_w, const float* w, int row0, int rows_valid, int K, int KPAD) {
  const int t = (int)threadIdx.x, T = (int)blockDim.x;
  const int lane = t & 63, wave = t >> 6, nw = T >> 6;
  if (K & 3) {                                       // odd history lengths: scalar staging
    for (int m = wave; m < 32; m += nw)
      for (int k = lane; k < K; k += 64) s_w[k * 33 + m] = m < rows_valid ? w[(int64_t)(row0 + m) * K + k] : 0.0f;
  } else {
    // all of a wave's float4 loads are issued before the first LDS store (one memory round trip, not one per row)
    const int kq = K >> 2;
    constexpr int RMAX = 4, QMAX = 4;                // 32 rows / 8 waves, K <= 1024
    float4 v[RMAX][QMAX];
    if (nw * RMAX >= 32 && kq <= 64 * QMAX) {
#pragma unroll
      for (int r = 0; r < RMAX; ++r) {
        const int m = wave + r * nw;
#pragma unroll
        for (int i = 0; i < QMAX; ++i) {
          const int q = lane + 64 * i;
          v[r][i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
          if (m < rows_valid && q < kq) v[r][i] = rb_ld4(w + (int64_t)(row0 + m) * K + 4 * q);
        }
      }
#pragma unroll
      for (int r = 0; r < RMAX; ++r) {
        const int m = wave + r * nw;
#pragma unroll
        for (int i = 0; i < QMAX; ++i) {
          const int q = lane + 64 * i;
          if (m < 32 && q < kq) {
            s_w[(4 * q + 0) * 33 + m] = v[r][i].x;
            s_w[(4 * q + 1) * 33 + m] = v[r][i].y;
            s_w[(4 * q + 2) * 33 + m] = v[r][i].z;
            s_w[(4 * q + 3) * 33 + m] = v[r][i].w;
          }
        }
      }
    } else {
      for (int m = wave; m < 32; m += nw) {
        const float* src = w + (int64_t)(row0 + m) * K;
        for (int q = lane; q < kq; q += 64) {
          float4 x = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
          if (m < rows_valid) x = rb_ld4(src + 4 * q);
          s_w[(4 * q + 0) * 33 + m] = x.x;
          s_w[(4 * q + 1) * 33 + m] = x.y;
          s_w[(4 * q + 2) * 33 + m] = x.z;
          s_w[(4 * q + 3) * 33 + m] = x.w;
        }
      }
    }
  }
  for (int e = t; e < (KPAD - K) * 32; e += T) s_w[(K + (e >> 5)) * 33 + (e & 31)] = 0.0f;
}

// ================================================================================ forward ==
// grid = (position chunks of 32*NT per image, cout / 32, images); block = 256.
// PR = input rows staged per channel (covers the output rows of one position chunk).
#define RB_CONV_WAVES 8
#define RB_CONV_THREADS (64 * RB_CONV_WAVES)
// PCH = output positions per workgroup (<= 32 NT; a multiple of the row length keeps the patch at PR rows).
// T16 (the t16 variant below): no reduction scratch, no tap table; the channel planes of the patch are padded so that the four
// k-slots of a 16x16x4 operand read (channel groups cin/4 apart) start 16 banks apart.
template <class G, int NT, int PR, int KMAX, int T16 = 0>
struct ConvFwdLdsSize {
  static constexpr int KGRAN = 2 * RB_CONV_WAVES;
  static constexpr int KPAD = (KMAX + KGRAN - 1) / KGRAN * KGRAN;
  static constexpr int RED = RB_CONV_WAVES * 16 * 64;      // reduction scratch (floats) for ONE 32-position tile, overlays the operands
  // the weight slab keeps its GLOBAL orientation in LDS: 32 rows (output channels) of KPAD + 4 floats.  Staging is then
  // 16-byte loads -> 16-byte LDS stores, conflict-free (the former [k][33] transposed image took 16 scalar stores per
  // thread at 8-way bank conflicts: 1-1.5 us of every workgroup, and it serialised the two first-layer workgroups of a CU —
  // tools/wg_timeline.py: 5.3 us input stage for the second one); the MFMA operand read (lane = row) is 4-way conflicted
  // instead, one read per NT MFMAs, hidden under them.
  static constexpr int WS = KPAD + 4;
  // patch rows are stored DE-INTERLEAVED by stride phase: x -> (x % S) * SUB + x / S, SUB = ceil(IH / S).  The lanes of an
  // MFMA operand read are neighbouring output positions, i.e. inputs S apart: in the plain row that is a stride-S access
  // (4-way bank conflicts in the first layer, 2-way in the second; the MFMA loop was LDS-bound enough that the two
  // first-layer workgroups of a CU stretched each other's staging and epilogue from 1.4 / 2.4 to 4.9 / 4.1 us,
  // tools/wg_timeline.py); de-interleaved they are consecutive words.
  static constexpr int SUB = (G::IH + G::S - 1) / G::S;
  static constexpr int RP = G::S * SUB;                    // row pitch (>= IH)
  static_assert((G::OH - 1) + (G::KS - 1) / G::S < SUB, "a tap's positions stay inside their phase's sub-row");
  static constexpr int CQ = (KMAX / G::KK) / 4;            // T16: channels per k-slot
  static constexpr int rb_plane_pad() {
    if (!T16 || CQ == 0) return 0;
    for (int p = 0; p < 64; p += 2)
      if ((CQ * (PR * RP + p)) % 32 == 16) return p;
    return 0;
  }
  static constexpr int PLANE = PR * RP + rb_plane_pad();   // floats per channel in the patch
  static constexpr int OPS = 32 * WS + (KMAX / G::KK) * PLANE;       // weights then patch, contiguous
  static constexpr int WSZ = T16 ? OPS : (OPS > RED ? OPS : RED);
  static constexpr int FLOATS = WSZ + (T16 ? 0 : KPAD);    // + the tap table (ints)
};
// body with explicit block coordinates and caller-provided LDS, so the layers of the stack can share one launch
// F32SRC (first layer only): the input is a.src.f32 (act / evaluate: float states) instead of the u8 frames.  The kind of
// the input loads is a compile-time property so that only ONE staging register array exists (all three alive at once cost
// the first layer its second workgroup per CU).
// T16: the MFMA phase on v_mfma_f32_16x16x4_f32 with NO split of the reduction: the workgroup has one wave per 16-position x
// 16-channel output tile (PT position tiles x 2 channel tiles = NWV waves), every wave runs the WHOLE K for its tile — no
// cross-wave partial sums, no reduction barriers, the epilogue goes from the accumulators to memory (the cross-wave sum +
// epilogue of the 8-way K split was 2.6 / 1.7 us of the second / third layer's 11.4 / 10.3 us workgroups, profiles/
// round3_final_wg_timeline.txt).  Lane (x = l & 15, kq = l >> 4) owns the CONTIGUOUS quarter [kq K/4, (kq + 1) K/4) of the
// reduction: its A operands are whole float4s of weight row x (one ds_read_b128 per four MFMAs), its B operands are patch
// cells whose offsets are compile-time functions of the step (immediates: no tap table).  Needs cin * KK == KMAX, cin % 4 == 0,
// KMAX % 16 == 0 (host-checked).
template <class G, int NT, int PR, int KMAX, bool FIRST, int PCH = 32 * NT, bool F32SRC = false, int T16 = 0>
struct ConvFwdWaves {
  static constexpr int PT = (PCH + 15) / 16;
  // T16 = channel tiles per wave.  The wave count is rounded up to a multiple of 4 — an even share per SIMD: a 10-wave workgroup
  // puts 3 waves on two SIMDs, and the compiler, which sizes the register allocation for the AVERAGE waves per SIMD its LDS
  // footprint allows (next_free_vgpr is raised to the smallest count that still gives that occupancy), then leaves no room
  // for the second workgroup the LDS would admit (profiles/round4_experiments.txt §5); the spare waves help staging and leave
  // SPLIT_LAST (a wave per (position tile, channel tile) unit, 2 PT = 4 n + 2 units: the first layer's 80-position chunks, ten
  // units): waves w, w + 4, w + 8 share a SIMD, so two SIMDs multiplied for three units and two for two — and with two such
  // workgroups per CU those SIMDs' MFMAs were the launch's critical path.  The last two units run instead as four half-units
  // (unit, reduction half) on four waves, one per SIMD; the halves meet through 4 KB of LDS: 2.5 units per SIMD.
#if defined(RB_NO_SPLIT_LAST)    // (variant build for A/B runs)
  static constexpr bool SPLIT_LAST = false;
#else
  static constexpr bool SPLIT_LAST = T16 == 1 && PT >= 3 && ((2 * PT) % 4) == 2 && ((KMAX / 4) % 8) == 0;
#endif
  static constexpr int TILE_WAVES = T16 == 1 ? (SPLIT_LAST ? 2 * PT + 2 : 2 * PT) : PT;
  static constexpr int PARTF = SPLIT_LAST ? 4 * 4 * 64 : 0;   // floats of LDS behind ConvFwdLdsSize::FLOATS for the half-units' partial tiles
  static constexpr int NWV = T16 == 0 ? RB_CONV_WAVES : (TILE_WAVES + 3) / 4 * 4;
};
template <int V> struct rb_conv_int_c { static constexpr int value = V; };
template <class G, int KMAX, int PLANE, int RP, int SUB>
__device__ __forceinline__ constexpr int rb_t16_off(int j) {            // step j of a lane's K quarter -> offset in the patch
  return (j / G::KK) * PLANE + ((j % G::KK) / G::KS) * RP + (((j % G::KK) % G::KS) % G::S) * SUB + ((j % G::KK) % G::KS) / G::S;
}
template <class G, int NT, int PR, int KMAX, bool FIRST, int PCH = 32 * NT, bool F32SRC = false, int T16 = 0>
__device__ __forceinline__ void rb_conv_fwd_body(const ConvLdsFwdArgs& a, int bx, int by, int img, float* smem) {
  typedef ConvFwdLdsSize<G, NT, PR, KMAX, T16> SZ;
  constexpr int NWV = ConvFwdWaves<G, NT, PR, KMAX, FIRST, PCH, F32SRC, T16>::NWV;
  constexpr int THREADS = 64 * NWV;
  constexpr int KPAD = SZ::KPAD;
  constexpr int SUB = SZ::SUB, RP = SZ::RP;
  constexpr int PLANE = SZ::PLANE;                  // floats per channel in the patch
  constexpr int CMAX = KMAX / G::KK;
  static_assert(!T16 || (KMAX % 16 == 0 && CMAX % 4 == 0), "t16: whole float4s per k-slot");
  float* s_all = smem;
  int* s_koff = reinterpret_cast<int*>(smem + SZ::WSZ);    // (not T16: no table)
  constexpr int WS = SZ::WS;
  // patch cell of (channel c, element `off` of the channel's [rows][IH] patch)
  auto pcell = [&](int c, int off) -> int {
    const int r = off / G::IH, x = off - r * G::IH;
    return c * PLANE + r * RP + (x % G::S) * SUB + x / G::S;
  };
  float* s_w = s_all;
  float* s_patch = s_all + 32 * WS;

  const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
  constexpr int SB = G::KS == 8 ? 0 : G::KS == 4 ? 8 : 16;    // stamp slots per layer (RB_STAMP builds only)
  RB_CSTAMP(SB + 0);
  RB_CSTAMP_LAST(SB + 4);
  constexpr int WK = FIRST ? 0 : (G::KS == 3 ? 2 : 1);          // timeline id of this layer (RB_STAMP builds only)
  const int wgi = img * 16 + by * 8 + bx;
  (void)wgi;
  RB_WGT(WK, wgi, 0);
  RB_WGT_HW(WK, wgi);
#if !defined(RB_HOST_INTERP)
  // staging outranks the MFMA phase of a co-resident workgroup: two first-layer workgroups share a CU, and the one that
  // got there second spent 4.4 us converting and storing 7 KB of frames while the first ran its MFMA loop (0.9 us alone;
  // tools/stamp/fine_stage.py) — the wave scheduler favours the older waves.  Dropped again before this workgroup's own MFMAs.
  __builtin_amdgcn_s_setprio(3);
#endif
  const int net = img < a.n_on ? 0 : 1;
  const int cout0 = by * 32;
  const int p0 = bx * PCH;
  const int cin = a.cin;
  const int K = cin * G::KK;
  const int oy0 = p0 / G::OH;
  const int iy0 = oy0 * G::S;
  int rows = G::IH - iy0;
  if (rows > PR) rows = PR;

  // ---- stage: weights (transposed into LDS, or this wave's slice into registers), k -> patch offset table, input patch.
  // Per-workgroup timeline (tools/wg_timeline.py): weights 1.8-2.3 us and input 1.9-3.8 us used to be two memory round
  // trips in sequence (load, store to LDS, load, store to LDS).  Now every global load of BOTH operands is issued before the
  // first LDS store — first-layer frames: the window-table entries first, they gate the frame addresses — and the LDS
  // stores follow in issue order.
  constexpr int KW = KPAD / RB_CONV_WAVES;            // even, compile-time: the MFMA loop is fully unrolled (not T16)
  constexpr int HW = KW / 2;
  const int rows_valid_w = a.cout - cout0 < 32 ? a.cout - cout0 : 32;
  // weights: the fast path of rb_stage_weights_t split into issue (loads) and commit (transposing LDS stores)
  constexpr int WR = (32 + NWV - 1) / NWV, WQ = (KMAX + 255) / 256;      // 32 rows over the waves; K <= KMAX: quads of a row per lane
  const bool w_fast = (K & 3) == 0 && (K >> 2) <= 64 * WQ;
  float4 wv[WR][WQ];
  // input: one batch of loads per thread (every geometry of the two networks fits one batch; more: the loops after it)
  constexpr bool x_u8 = FIRST && !F32SRC;
  constexpr bool x_vec = !x_u8 && (G::IH % 4) == 0;   // then per_c, iy0 * IH and IP are multiples of 4 as well
  // u8 frames, stride-4 geometry (the canonical first layer): DWORD loads — the four bytes of a dword are the four stride
  // phases of one de-interleaved index, so consecutive lanes store consecutive words of each phase's sub-row (conflict-free
  // scalar stores; 16-byte loads put 16-byte-strided lanes on 8 banks)
  constexpr bool x_dw = x_u8 && G::S == 4 && (G::IH % 4) == 0;
  constexpr int XD = x_dw ? (CMAX * PR * G::IH / 4 + THREADS - 1) / THREADS : 1;
  constexpr int XU = (x_u8 && !x_dw) ? 2 : 1, XV = x_vec ? 8 : 1, XS = (!x_u8 && !x_vec) ? 12 : 1;
  const float* xbase = x_u8 ? nullptr : (FIRST ? a.src.f32 + (int64_t)img * cin * G::IP : a.in_f + (int64_t)img * cin * G::IP);
  const int per_c = rows * G::IH;                     // elements per channel of the patch (u8: bytes, a 16-byte multiple for the frame geometries)
  const int v16 = per_c >> 4, total16 = cin * v16;
  const int v4 = per_c >> 2, total4 = cin * v4;
  const int total1 = cin * per_c;
  // (a frame pointer is always one of the kernel's global arguments plus an offset and its validity a flag of its own: with
  // nullptr as the "blank frame" marker the compiler could no longer tell the address space and the first layer's T16
  // instantiation carried six FLAT loads — the library's only ones)
  const uint8_t* fp[x_dw ? XD : XU];
  bool fok[x_dw ? XD : XU];
  uint4 xu[XU];
  unsigned xd[XD];
  const int dpc = per_c >> 2, total_dw = cin * dpc;   // x_dw: dwords per channel of the patch
  float4 xv[XV];
  float xs[XS];
  // zero-copy frames: the window-table entries are REQUESTED here and turned into frame addresses only after the weight
  // loads have been issued (an address formed at once put the table's round trip in front of every other load: 1.5 us
  // from workgroup start to the first weight load, tools/stamp/fine_stage.py)
  int32_t widx[x_dw ? XD : XU];
  if constexpr (x_u8) {
#pragma unroll
    for (int i = 0; i < (x_dw ? XD : XU); ++i) {
      const int e = i * THREADS + t;
      const int c = x_dw ? e / dpc : e / v16;
      widx[i] = -1;
      if ((x_dw ? e < total_dw : e < total16) && a.src.ring) {
        const int sample = img < a.src.B ? img : (img - a.src.B) % a.src.B;
        widx[i] = a.src.win[(int64_t)sample * a.src.win_len + (img < a.src.B ? c : a.src.n_step + c)];
      }
    }
  }
  if (w_fast) {
    const int kq = K >> 2;
#pragma unroll
    for (int r = 0; r < WR; ++r) {
      const int m = wave + r * NWV;
#pragma unroll
      for (int i = 0; i < WQ; ++i) {
        const int q = lane + 64 * i;
        wv[r][i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (m < rows_valid_w && q < kq) wv[r][i] = rb_ld4(a.w[net] + (int64_t)(cout0 + m) * K + 4 * q);
      }
    }
  }
  RB_WGT(WK, wgi, 1);
  RB_WGT(WK, wgi, 2);
  if constexpr (x_u8) {
#pragma unroll
    for (int i = 0; i < (x_dw ? XD : XU); ++i) {
      const int e = i * THREADS + t;
      const int c = x_dw ? e / dpc : e / v16;
      fok[i] = false;
      if (a.src.ring) {
        fp[i] = a.src.ring;
        if (x_dw ? e < total_dw : e < total16) {
          fok[i] = widx[i] >= 0;
          fp[i] = a.src.ring + (int64_t)(widx[i] < 0 ? 0 : widx[i]) * G::IP;                         // rb_frame_ptr, second half
        }
      } else {
        fp[i] = a.src.u8_states;
        if (x_dw ? e < total_dw : e < total16) {
          fok[i] = true;
          fp[i] = img < a.src.B ? a.src.u8_states + ((int64_t)img * cin + c) * G::IP
                                : a.src.u8_next + ((int64_t)((img - a.src.B) % a.src.B) * cin + c) * G::IP;   // rb_frame_ptr, gathered stacks
        }
      }
    }
  }
  // ---- input loads (first batch)
  if constexpr (x_dw) {
#pragma unroll
    for (int i = 0; i < XD; ++i) {
      const int e = i * THREADS + t;
      xd[i] = 0u;
      if (e < total_dw && fok[i]) xd[i] = rb_ldg_u32(fp[i] + iy0 * G::IH + 4 * (e - (e / dpc) * dpc));
    }
  } else if constexpr (x_u8) {
    if constexpr (FIRST) {
#pragma unroll
      for (int i = 0; i < XU; ++i) {
        const int e = i * THREADS + t;
        xu[i] = make_uint4(0u, 0u, 0u, 0u);
        if (e < total16 && fok[i]) xu[i] = *reinterpret_cast<const uint4*>(fp[i] + iy0 * G::IH + (e - (e / v16) * v16) * 16);
      }
    }
  } else if constexpr (x_vec) {
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int e = i * THREADS + t;
      if (e < total4) {
        const int c = e / v4, q = e - c * v4;
        xv[i] = rb_ld4(xbase + (int64_t)c * G::IP + iy0 * G::IH + q * 4);
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < XS; ++i) {
      const int e = i * THREADS + t;
      if (e < total1) {
        const int c = e / per_c, q = e - c * per_c;
        xs[i] = xbase[(int64_t)c * G::IP + iy0 * G::IH + q];
      }
    }
  }
#if defined(RB_STAMP) && defined(RB_STAMP_FINE)
  RB_WGT(WK + 4, wgi, 0);                               // (fine: loads issued)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  RB_WGT(WK + 4, wgi, 1);                               // (fine: thread 0's loads have landed)
#endif
  // ---- LDS: tap table (no memory operand), then the weights, then the input
  if constexpr (T16 == 0) {
    for (int k = t; k < KPAD; k += THREADS) {
      const int kc = k < K ? k : K - 1;
      const int c = kc / G::KK, r = kc % G::KK;
      s_koff[k] = c * PLANE + (r / G::KS) * RP + ((r % G::KS) % G::S) * SUB + (r % G::KS) / G::S;
    }
  }
  {
    if (w_fast) {
      const int kq = K >> 2;
#pragma unroll
      for (int r = 0; r < WR; ++r) {
        const int m = wave + r * NWV;
#pragma unroll
        for (int i = 0; i < WQ; ++i) {
          const int q = lane + 64 * i;
          if (q < kq && m < 32) {                                    // rows >= rows_valid were loaded as zeros
            // bank swizzle (rb_wswz): inside its aligned group of four, column k of row m sits at (k & 3) ^ ((m >> 3) & 3) —
            // m >> 3 == r here, a compile-time permutation of the float4
            const float4 v = wv[r][i];
            float4 o;
            if (T16 || (r & 3) == 0) o = v;                          // (r is an unrolled loop index: folded; T16: no swizzle)
            else if ((r & 3) == 1) o = make_float4(v.y, v.x, v.w, v.z);
            else if ((r & 3) == 2) o = make_float4(v.z, v.w, v.x, v.y);
            else o = make_float4(v.w, v.z, v.y, v.x);
            rb_st4(s_w + m * WS + 4 * q, o);
          }
        }
      }
    } else {                                           // odd history lengths: scalar staging
      for (int m = wave; m < 32; m += NWV)
        for (int k = lane; k < K; k += 64) s_w[m * WS + (T16 ? k : rb_wswz(m, k))] = m < rows_valid_w ? a.w[net][(int64_t)(cout0 + m) * K + k] : 0.0f;
    }
    for (int e = t; e < (KPAD - K) * 32; e += THREADS) s_w[(e & 31) * WS + rb_wswz(e & 31, K + (e >> 5))] = 0.0f;   // columns [K, KPAD)
  }
  if constexpr (x_dw) {
    constexpr int DPR = G::IH / 4;                       // dwords per input row
#pragma unroll
    for (int i = 0; i < XD; ++i) {
      const int e = i * THREADS + t;
      if (e < total_dw) {
        const int c = e / dpc, d = e - c * dpc;
        const int r = d / DPR, xi = d - r * DPR;        // bytes 4 xi .. 4 xi + 3 of row r: phases 0..3 of de-interleaved index xi
        float* cell = s_patch + c * PLANE + r * RP + xi;
#pragma unroll
        for (int b = 0; b < 4; ++b) cell[b * SUB] = rb_unit((uint8_t)((xd[i] >> (8 * b)) & 0xFFu));
      }
    }
  } else if constexpr (x_u8) {
    if constexpr (FIRST) {
#pragma unroll
      for (int i = 0; i < XU; ++i) {
        const int e = i * THREADS + t;
        if (e < total16) {
          const int c = e / v16, q = e - c * v16;
          const unsigned wds[4] = {xu[i].x, xu[i].y, xu[i].z, xu[i].w};
#pragma unroll
          for (int wd = 0; wd < 4; ++wd)
#pragma unroll
            for (int b = 0; b < 4; ++b) s_patch[pcell(c, q * 16 + wd * 4 + b)] = rb_unit((uint8_t)((wds[wd] >> (8 * b)) & 0xFFu));
        }
      }
      for (int e = XU * THREADS + t; e < total16; e += THREADS) {          // beyond one batch (not the frame geometries)
        const int c = e / v16, q = e - c * v16;
        const uint8_t* f = rb_frame_ptr(a.src, img, c, cin, G::IP);
        uint4 raw = make_uint4(0u, 0u, 0u, 0u);
        if (f) raw = *reinterpret_cast<const uint4*>(f + iy0 * G::IH + q * 16);
        const unsigned wds[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int wd = 0; wd < 4; ++wd)
#pragma unroll
          for (int b = 0; b < 4; ++b) s_patch[pcell(c, q * 16 + wd * 4 + b)] = rb_unit((uint8_t)((wds[wd] >> (8 * b)) & 0xFFu));
      }
      for (int e = t; e < cin * (per_c & 15); e += THREADS) {   // (no tail for 84-wide frames; kept for generality)
        const int c = e / (per_c & 15), q = (v16 << 4) + e % (per_c & 15);
        const uint8_t* f = rb_frame_ptr(a.src, img, c, cin, G::IP);
        s_patch[pcell(c, q)] = f ? rb_unit(f[iy0 * G::IH + q]) : 0.0f;
      }
    }
  } else if constexpr (x_vec) {
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int e = i * THREADS + t;
      if (e < total4) {
        const int c = e / v4, q = e - c * v4;
        if constexpr (G::S == 1) { rb_st4(s_patch + c * PLANE + q * 4, xv[i]); }       // RP == IH: the quad stays a quad
        else if constexpr (G::S == 2 && (SUB % 2) == 0) {
          // x, x+2 are neighbours of phase 0 and x+1, x+3 of phase 1: two 8-byte stores, lanes 8 bytes apart (conflict-free)
          const int off = q * 4, r = off / G::IH, x = off - r * G::IH;
          float* cell = s_patch + c * PLANE + r * RP + x / 2;
          *reinterpret_cast<float2*>(cell) = make_float2(xv[i].x, xv[i].z);
          *reinterpret_cast<float2*>(cell + SUB) = make_float2(xv[i].y, xv[i].w);
        } else {
          s_patch[pcell(c, q * 4 + 0)] = xv[i].x; s_patch[pcell(c, q * 4 + 1)] = xv[i].y;
          s_patch[pcell(c, q * 4 + 2)] = xv[i].z; s_patch[pcell(c, q * 4 + 3)] = xv[i].w;
        }
      }
    }
    for (int e = XV * THREADS + t; e < total4; e += THREADS) {               // beyond one batch
      const int c = e / v4, q = e - c * v4;
      const float4 v = rb_ld4(xbase + (int64_t)c * G::IP + iy0 * G::IH + q * 4);
      s_patch[pcell(c, q * 4 + 0)] = v.x; s_patch[pcell(c, q * 4 + 1)] = v.y;
      s_patch[pcell(c, q * 4 + 2)] = v.z; s_patch[pcell(c, q * 4 + 3)] = v.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < XS; ++i) {
      const int e = i * THREADS + t;
      if (e < total1) { const int c = e / per_c, q = e - c * per_c; s_patch[pcell(c, q)] = xs[i]; }
    }
    for (int e = XS * THREADS + t; e < total1; e += THREADS) {               // beyond one batch
      const int c = e / per_c, q = e - c * per_c;
      const float v = xbase[(int64_t)c * G::IP + iy0 * G::IH + q];
      s_patch[pcell(c, q)] = v;
    }
  }
#if defined(RB_STAMP) && defined(RB_STAMP_FINE)
  RB_WGT(WK + 4, wgi, 2);                               // (fine: thread 0's LDS stores issued; then the barrier)
#endif
  __syncthreads();
#if !defined(RB_HOST_INTERP)
  __builtin_amdgcn_s_setprio(0);
#endif
  RB_CSTAMP(SB + 1);
  RB_WGT(WK, wgi, 3);

  if constexpr (T16 != 0) {
    // CTW channel tiles per wave: 1 = a wave per (position tile, channel tile); 2 = a wave per position tile, both channel
    // tiles of the slab from ONE patch operand per step (the first layer: five waves instead of ten per workgroup)
    constexpr int PT = (PCH + 15) / 16, KQ = KMAX / 4, CQ = CMAX / 4, CTW = T16;
    constexpr bool SPLIT_LAST = ConvFwdWaves<G, NT, PR, KMAX, FIRST, PCH, F32SRC, T16>::SPLIT_LAST;
    static_assert(!SPLIT_LAST || ConvFwdWaves<G, NT, PR, KMAX, FIRST, PCH, F32SRC, T16>::TILE_WAVES == NWV, "SPLIT_LAST: every wave reaches the barrier");
    if (wave >= ConvFwdWaves<G, NT, PR, KMAX, FIRST, PCH, F32SRC, T16>::TILE_WAVES) return;   // spare staging waves (no barrier follows)
    const int x = lane & 15, kq = lane >> 4;
    if constexpr (SPLIT_LAST) {
      if (wave >= 2 * PT - 2) {                           // wave-uniform: a half-unit
        float* s_part = smem + SZ::FLOATS;                // [half-unit][4][64]
        const int tu = wave - (2 * PT - 2), unit = 2 * PT - 2 + (tu & 1), kh2 = tu >> 1;
        const int upt = unit % PT, ct = unit / PT;
        int p = p0 + upt * 16 + x;
        const bool pv = p < G::P && p < p0 + PCH;
        if (p > G::P - 1) p = G::P - 1;
        const float* bp = s_patch + kq * CQ * PLANE + (p / G::OH - oy0) * G::S * RP + (p % G::OH);
        const float* ap = s_w + (ct * 16 + x) * WS + kq * KQ;
        rb_f32x4 acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = 0.0f;
        auto half = [&](auto h_c) {                        // the lane's steps [h KQ / 2, (h + 1) KQ / 2) of its quarter: immediates again
          constexpr int H = decltype(h_c)::value;
#pragma unroll
          for (int jq = H * (KQ / 8); jq < (H + 1) * (KQ / 8); ++jq) {
            const float4 w4 = rb_ld4(ap + 4 * jq);
            acc = rb_mfma16(w4.x, bp[rb_t16_off<G, KMAX, PLANE, RP, SUB>(4 * jq + 0)], acc);
            acc = rb_mfma16(w4.y, bp[rb_t16_off<G, KMAX, PLANE, RP, SUB>(4 * jq + 1)], acc);
            acc = rb_mfma16(w4.z, bp[rb_t16_off<G, KMAX, PLANE, RP, SUB>(4 * jq + 2)], acc);
            acc = rb_mfma16(w4.w, bp[rb_t16_off<G, KMAX, PLANE, RP, SUB>(4 * jq + 3)], acc);
          }
        };
        if (kh2 == 0) half(rb_conv_int_c<0>()); else half(rb_conv_int_c<1>());
        float bias1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = cout0 + ct * 16 + 4 * kq + r;
          bias1[r] = a.bias[net][m < a.cout ? m : a.cout - 1];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) s_part[(tu * 4 + r) * 64 + lane] = acc[r];
        __syncthreads();                                  // (the other waves meet it behind their own epilogue, below)
        if (kh2 == 0) {                                   // first half + second half, bias, ReLU
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = cout0 + ct * 16 + 4 * kq + r;
            if (pv && m < a.cout) {
              const float o = fmaxf((s_part[(tu * 4 + r) * 64 + lane] + s_part[((tu + 2) * 4 + r) * 64 + lane]) + bias1[r], 0.0f);
              a.out[((int64_t)img * a.cout + m) * G::P + p] = o;
              if (a.out_blocked) {
                const int k = m * G::P + p;
                a.out_blocked[((int64_t)(k >> 4) * a.rows_total + img) * 16 + (k & 15)] = o;
              }
            }
          }
        }
        RB_WGT(WK, wgi, 4); RB_WGT(WK, wgi, 5); RB_WGT(WK, wgi, 6);
        return;
      }
    }
    const int pt = wave % PT, ct0 = (wave / PT) * CTW;      // wave-uniform: position tile, first channel tile
    int p = p0 + pt * 16 + x;
    const bool pv = p < G::P && p < p0 + PCH;
    if (p > G::P - 1) p = G::P - 1;                   // clamped lanes are never stored
    const float* bp = s_patch + kq * CQ * PLANE + (p / G::OH - oy0) * G::S * RP + (p % G::OH);
    const float* ap = s_w + (ct0 * 16 + x) * WS + kq * KQ;
    float bias4[CTW][4];
#pragma unroll
    for (int u = 0; u < CTW; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = cout0 + (ct0 + u) * 16 + 4 * kq + r;
        bias4[u][r] = a.bias[net][m < a.cout ? m : a.cout - 1];
      }
    rb_f32x4 acc[CTW];
#pragma unroll
    for (int u = 0; u < CTW; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[u][r] = 0.0f;
#pragma unroll
    for (int jq = 0; jq < KQ / 4; ++jq) {
      float4 w4[CTW];
#pragma unroll
      for (int u = 0; u < CTW; ++u) w4[u] = rb_ld4(ap + u * 16 * WS + 4 * jq);
      const float b0 = bp[rb_t16_off<G, KMAX, PLANE, RP, SUB>(4 * jq + 0)], b1 = bp[rb_t16_off<G, KMAX, PLANE, RP, SUB>(4 * jq + 1)];
      const float b2 = bp[rb_t16_off<G, KMAX, PLANE, RP, SUB>(4 * jq + 2)], b3 = bp[rb_t16_off<G, KMAX, PLANE, RP, SUB>(4 * jq + 3)];
#pragma unroll
      for (int u = 0; u < CTW; ++u) acc[u] = rb_mfma16(w4[u].x, b0, acc[u]);
#pragma unroll
      for (int u = 0; u < CTW; ++u) acc[u] = rb_mfma16(w4[u].y, b1, acc[u]);
#pragma unroll
      for (int u = 0; u < CTW; ++u) acc[u] = rb_mfma16(w4[u].z, b2, acc[u]);
#pragma unroll
      for (int u = 0; u < CTW; ++u) acc[u] = rb_mfma16(w4[u].w, b3, acc[u]);
    }
    RB_CSTAMP(SB + 2);
    RB_WGT(WK, wgi, 4);
#pragma unroll
    for (int u = 0; u < CTW; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {                   // D[r]: channel 4 kq + r of the tile, position x
        const int m = cout0 + (ct0 + u) * 16 + 4 * kq + r;
        if (pv && m < a.cout) {
          const float o = fmaxf(acc[u][r] + bias4[u][r], 0.0f);
          a.out[((int64_t)img * a.cout + m) * G::P + p] = o;
          if (a.out_blocked) {
            const int k = m * G::P + p;                                   // x.view(-1, conv_output_size), model.py:71
            a.out_blocked[((int64_t)(k >> 4) * a.rows_total + img) * 16 + (k & 15)] = o;
          }
        }
      }
    RB_CSTAMP(SB + 3);
    RB_CSTAMP_LAST(SB + 5);
    RB_WGT(WK, wgi, 5);
    RB_WGT(WK, wgi, 6);
    if constexpr (SPLIT_LAST) __syncthreads();            // the split tile's waves exchange their halves behind this barrier
    return;
  }
  // ---- MFMA loop: wave w owns k in [w*KW, (w+1)*KW) of the padded reduction (weights beyond K are zero)
  const int kb = wave * KW;
  int noff[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    int p = p0 + nt * 32 + (lane & 31);
    if (p > G::P - 1) p = G::P - 1;                  // clamped lanes are never stored
    noff[nt] = (p / G::OH - oy0) * G::S * RP + (p % G::OH);        // (de-interleaved rows: neighbouring outputs, neighbouring words)
  }
  rb_f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;
  const int kh = lane >> 5, ml = lane & 31;
  float bias_r[(16 * 64) / THREADS];          // the epilogue's bias terms (its rows do not depend on the tile)
#pragma unroll
  for (int it = 0; it < (16 * 64) / THREADS; ++it) {
    const int idx = t + it * THREADS;
    const int m = cout0 + rb_mfma_row(idx >> 6, idx & 63);
    bias_r[it] = a.bias[net][m < a.cout ? m : a.cout - 1];
  }
  // the patch offsets of this wave's k range are read up front: inside the loop they would put an LDS round trip
  // (offset -> operand address) on the critical path of every step (measured 7.1 us of MFMA phase for 5.1 us of MFMAs)
  int kos[HW];
#pragma unroll
  for (int j = 0; j < HW; ++j) kos[j] = s_koff[kb + 2 * j + kh];
  // A operand: row ml, column k = kb + 2 j + kh of the row-major slab.  The row stride is a multiple of 4 (16-byte staging
  // stores), so 32 lanes reading one column would meet in 8 banks, four deep; with the swizzle rows ml, ml + 8, ml + 16, ml + 24
  // keep that column in four different words of its group: conflict-free.  kb % 4 == 0: two lane constants, immediate offsets.
  // (k ranges per wave that are not 4-aligned — the data-efficient first layer, KW = 14 — compute the swizzle per step)
  const int aswz = (ml >> 3) & 3;
  const int a_even = ml * WS + kb + (kh ^ aswz), a_odd = ml * WS + kb + ((2 + kh) ^ aswz);
#pragma unroll
  for (int j = 0; j < HW; ++j) {
    float av;
    if constexpr (KW % 4 == 0) av = s_w[((j & 1) ? a_odd : a_even) + 4 * (j >> 1)];
    else av = s_w[ml * WS + rb_wswz(ml, kb + 2 * j + kh)];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = rb_mfma32(av, s_patch[noff[nt] + kos[j]], acc[nt]);
  }
  RB_CSTAMP(SB + 2);
  RB_WGT(WK, wgi, 4);
  // cross-wave sum, fixed order w0..w7.  Where the operand area is large enough for the partial sums of ALL NT tiles
  // (the later layers: 2-3 x 32 KB inside 97-118 KB) they are exchanged in one pass — two barriers instead of 2 NT; the
  // first layer keeps one 32 KB tile at a time (its LDS footprint decides how many workgroups share a CU).
  constexpr int EIT = (16 * 64) / THREADS;
  constexpr bool ONEPASS = NT * SZ::RED <= SZ::WSZ;
  constexpr int TP = ONEPASS ? NT : 1;                  // tiles per pass
#pragma unroll
  for (int nt0 = 0; nt0 < NT; nt0 += TP) {
    __syncthreads();                                  // operands (first pass) / the previous pass's sums are no longer read
#pragma unroll
    for (int u = 0; u < TP; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) s_all[u * SZ::RED + (wave * 16 + r) * 64 + lane] = acc[nt0 + u][r];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < TP; ++u) {
      const int nt = nt0 + u;
#pragma unroll
      for (int it = 0; it < EIT; ++it) {
        const int idx = t + it * THREADS;
        const int l = idx & 63, r = idx >> 6;
        float v = s_all[u * SZ::RED + (0 * 16 + r) * 64 + l];
#pragma unroll
        for (int wv_ = 1; wv_ < RB_CONV_WAVES; ++wv_) v += s_all[u * SZ::RED + (wv_ * 16 + r) * 64 + l];
        const int m = cout0 + rb_mfma_row(r, l);
        const int p = p0 + nt * 32 + (l & 31);
        if (m < a.cout && p < G::P && p < p0 + PCH) {
          const float o = fmaxf(v + bias_r[it], 0.0f);      // (bias fetched before the MFMA loop: a global load here sat on
                                                            //  the critical path of every tile's epilogue)
          a.out[((int64_t)img * a.cout + m) * G::P + p] = o;
          if (a.out_blocked) {
            const int k = m * G::P + p;                                   // x.view(-1, conv_output_size), model.py:71
            a.out_blocked[((int64_t)(k >> 4) * a.rows_total + img) * 16 + (k & 15)] = o;
          }
        }
      }
    }
  }
  RB_CSTAMP(SB + 3);
  RB_CSTAMP_LAST(SB + 5);
  RB_WGT(WK, wgi, 5);
  RB_WGT(WK, wgi, 6);
}

// (second launch bound = waves per SIMD: a 512-thread workgroup is 2; 4 where the LDS footprint lets two workgroups share a
// CU — the first layer on u8 frames — so that the register allocation does too; the float-input variant of the acting path
// would spill under that cap, and no kernel of this library may carry a scratch segment)
template <class G, int NT, int PR, int KMAX, bool FIRST, int PCH = 32 * NT, bool F32SRC = false>
__global__ __launch_bounds__(RB_CONV_THREADS, (ConvFwdLdsSize<G, NT, PR, KMAX>::FLOATS * 4 <= 80 * 1024 && !F32SRC) ? 4 : 2) void k_conv_fwd_lds(ConvLdsFwdArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[ConvFwdLdsSize<G, NT, PR, KMAX>::FLOATS];
  // img_fast: workgroups are spread over the 8 XCDs by linear block index mod 8; with the image as the fastest index (and an
  // image count that is a multiple of 8) every workgroup of image i, in every layer, runs on XCD i mod 8 — the next layer's
  // input is then in that XCD's own L2 instead of behind the fabric
  if (a.img_fast) rb_conv_fwd_body<G, NT, PR, KMAX, FIRST, PCH, F32SRC>(a, (int)blockIdx.z, (int)blockIdx.y, (int)blockIdx.x, smem);
  else rb_conv_fwd_body<G, NT, PR, KMAX, FIRST, PCH, F32SRC>(a, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, smem);
}

// the t16 variant (rb_conv_fwd_body<..., T16 = CTW>): grid as k_conv_fwd_lds, block = 64 * ConvFwdWaves<..., CTW>::NWV threads
template <class G, int NT, int PR, int KMAX, bool FIRST, int PCH = 32 * NT, int CTW = 1>
__global__ __launch_bounds__((64 * ConvFwdWaves<G, NT, PR, KMAX, FIRST, PCH, false, CTW>::NWV))
void k_conv_fwd_t16(ConvLdsFwdArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[ConvFwdLdsSize<G, NT, PR, KMAX, CTW>::FLOATS + ConvFwdWaves<G, NT, PR, KMAX, FIRST, PCH, false, CTW>::PARTF];
  if (a.img_fast) rb_conv_fwd_body<G, NT, PR, KMAX, FIRST, PCH, false, CTW>(a, (int)blockIdx.z, (int)blockIdx.y, (int)blockIdx.x, smem);
  else rb_conv_fwd_body<G, NT, PR, KMAX, FIRST, PCH, false, CTW>(a, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, smem);
}

// ---- large batches: one weight slab per workgroup, a loop over images -----------------------------------------
// At 768 images the one-image workgroup above stages 34-76 KB of transposed weights for 2.6-5.3 us of MFMAs, 6-15 times
// per CU.  Here a workgroup owns (position chunk, 32-channel slab) and walks a.ipb images: the slab, the tap table and
// the epilogue's bias terms are set up once (again where the image range crosses from the online to the target net);
// per image only the patch is staged, and the NEXT image's patch is already in flight (registers) under this image's
// MFMA loop and reductions.  The reduction scratch has a
// region of its own (it cannot overlay operands that live across images).
// grid = (position chunks, cout / 32, image groups); block = 512.
template <class G, int PR, int KMAX>
struct ConvFwdMultiLds {
  static constexpr int KPAD = (KMAX + 2 * RB_CONV_WAVES - 1) / (2 * RB_CONV_WAVES) * (2 * RB_CONV_WAVES);
  static constexpr int FLOATS = KPAD * 33 + (KMAX / G::KK) * PR * G::IH + RB_CONV_WAVES * 16 * 64;
  static constexpr bool FITS = FLOATS * 4 + KPAD * 4 <= 160 * 1024;
};
template <class G, int NT, int PR, int KMAX, bool FIRST, int PCH = 32 * NT>
__global__ __launch_bounds__(RB_CONV_THREADS) void k_conv_fwd_multi(ConvLdsFwdArgs a) {
  constexpr int KGRAN = 2 * RB_CONV_WAVES;
  constexpr int KPAD = (KMAX + KGRAN - 1) / KGRAN * KGRAN;
  constexpr int PLANE = PR * G::IH;                 // floats per channel in the patch
  constexpr int CMAX = KMAX / G::KK;
  constexpr int RED = RB_CONV_WAVES * 16 * 64;      // reduction scratch for ONE 32-position tile
  constexpr int OPS = KPAD * 33 + CMAX * PLANE;
  __shared__ __attribute__((aligned(16))) float s_all[OPS + RED];
  __shared__ int s_koff[KPAD];
  float* s_w = s_all;
  float* s_patch = s_all + KPAD * 33;
  float* s_red = s_all + OPS;

  const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
#if defined(RB_STAMP)
  const bool mst = G::KS == RB_MSTAMP_KS && t == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0;
#define RB_MSTAMP(i) do { if (mst) g_cstamp[i] = wall_clock64(); } while (0)
#else
#define RB_MSTAMP(i) ((void)0)
#endif
  RB_MSTAMP(48);
  // images [z * ipb, (z + 1) * ipb) of the whole list (net 0's first); a range that straddles the nets re-stages its slab
  // (img_fast: grid = (image groups, cout tiles, position chunks) — a group's workgroups of consecutive layers with the same
  // ipb share an XCD, see k_conv_fwd_lds)
  const int img0 = (a.img_fast ? (int)blockIdx.x : (int)blockIdx.z) * a.ipb;
  const int img_end = img0 + a.ipb < a.rows_total ? img0 + a.ipb : a.rows_total;
  const int cout0 = (int)blockIdx.y * 32;
  const int p0 = (a.img_fast ? (int)blockIdx.z : (int)blockIdx.x) * PCH;
  const int cin = a.cin;
  const int K = cin * G::KK;
  const int oy0 = p0 / G::OH;
  const int iy0 = oy0 * G::S;
  int rows = G::IH - iy0;
  if (rows > PR) rows = PR;
  const int per_c = rows * G::IH;

  // ---- once: tap table (the weight slab: at the first image and where the net changes)
  for (int k = t; k < KPAD; k += RB_CONV_THREADS) {
    const int kc = k < K ? k : K - 1;
    const int c = kc / G::KK, r = kc % G::KK;
    s_koff[k] = c * PLANE + (r / G::KS) * G::IH + (r % G::KS);
  }

  // ---- the patch of one image: loads into registers (issue), LDS stores later (commit)
  constexpr bool VEC = !FIRST && (G::IH % 4 == 0);          // per_c, iy0 * IH and IP are then multiples of 4
  constexpr int NU = FIRST ? 2 : 1, NV = (!FIRST && VEC) ? 8 : 1, NS = (!FIRST && !VEC) ? 12 : 1;
  static_assert(!FIRST || CMAX * PLANE <= 16 * NU * RB_CONV_THREADS, "u8 patch fits one batch of loads");
  static_assert(FIRST || !VEC || CMAX * PLANE <= 4 * NV * RB_CONV_THREADS, "f32 patch fits one batch of float4 loads");
  static_assert(FIRST || VEC || CMAX * PLANE <= NS * RB_CONV_THREADS, "f32 patch fits one batch of scalar loads");
  uint4 pu[NU];
  float4 pv[NV];
  float ps[NS];
  auto issue = [&](int img) {
    if constexpr (FIRST) {
      const int v16 = per_c >> 4, total16 = cin * v16;       // 84-wide frames: 16-byte multiples
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        const int e = i * RB_CONV_THREADS + t;
        pu[i] = make_uint4(0u, 0u, 0u, 0u);
        if (e < total16) {
          const int c = e / v16, q = e - c * v16;
          const uint8_t* fp = rb_frame_ptr(a.src, img, c, cin, G::IP);
          if (fp) pu[i] = *reinterpret_cast<const uint4*>(fp + iy0 * G::IH + q * 16);
        }
      }
    } else if constexpr (VEC) {
      const float* base = a.in_f + (int64_t)img * cin * G::IP;
      const int v4 = per_c >> 2, total = cin * v4;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int e = i * RB_CONV_THREADS + t;
        const int ec = e < total ? e : total - 1;
        const int c = ec / v4, q = ec - c * v4;
        pv[i] = rb_ld4(base + c * G::IP + iy0 * G::IH + q * 4);
      }
    } else {
      const float* base = a.in_f + (int64_t)img * cin * G::IP;
      const int total = cin * per_c;
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int e = i * RB_CONV_THREADS + t;
        const int ec = e < total ? e : total - 1;
        const int c = ec / per_c, q = ec - c * per_c;
        ps[i] = base[c * G::IP + iy0 * G::IH + q];
      }
    }
  };
  auto commit = [&]() {
    if constexpr (FIRST) {
      const int v16 = per_c >> 4, total16 = cin * v16;
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        const int e = i * RB_CONV_THREADS + t;
        if (e < total16) {
          const int c = e / v16, q = e - c * v16;
          float* d = s_patch + c * PLANE + q * 16;
          const unsigned wds[4] = {pu[i].x, pu[i].y, pu[i].z, pu[i].w};
#pragma unroll
          for (int wd = 0; wd < 4; ++wd)
#pragma unroll
            for (int b = 0; b < 4; ++b) d[wd * 4 + b] = rb_unit((uint8_t)((wds[wd] >> (8 * b)) & 0xFFu));
        }
      }
    } else if constexpr (VEC) {
      const int v4 = per_c >> 2, total = cin * v4;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int e = i * RB_CONV_THREADS + t;
        if (e < total) {
          const int c = e / v4, q = e - c * v4;
          float* d = s_patch + c * PLANE + q * 4;
          d[0] = pv[i].x; d[1] = pv[i].y; d[2] = pv[i].z; d[3] = pv[i].w;
        }
      }
    } else {
      const int total = cin * per_c;
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int e = i * RB_CONV_THREADS + t;
        if (e < total) { const int c = e / per_c, q = e - c * per_c; s_patch[c * PLANE + q] = ps[i]; }
      }
    }
  };

  constexpr int KW = KPAD / RB_CONV_WAVES, HW = KW / 2;
  const int kb = wave * KW;
  int noff[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    int p = p0 + nt * 32 + (lane & 31);
    if (p > G::P - 1) p = G::P - 1;                  // clamped lanes are never stored
    noff[nt] = (p / G::OH - oy0) * G::S * G::IH + (p % G::OH) * G::S;
  }
  const int kh = lane >> 5, ml = lane & 31;
  constexpr int EIT = (16 * 64) / RB_CONV_THREADS;
  float bias_r[EIT];
  int kos[HW];

  issue(img0);
  for (int img = img0; img < img_end; ++img) {
    if (img == img0 || img == a.n_on) {               // block-uniform (every wave is past the previous image's MFMA loop)
      const int net = img < a.n_on ? 0 : 1;
      rb_stage_weights_t(s_w, a.w[net], cout0, a.cout - cout0 < 32 ? a.cout - cout0 : 32, K, KPAD);
#pragma unroll
      for (int it = 0; it < EIT; ++it) {
        const int idx = t + it * RB_CONV_THREADS;
        const int m = cout0 + rb_mfma_row(idx >> 6, idx & 63);
        bias_r[it] = a.bias[net][m < a.cout ? m : a.cout - 1];
      }
    }
    commit();
    __syncthreads();            // patch (and slab) complete; the previous image's last reduction has been consumed
    RB_MSTAMP(img == img0 ? 49 : img == img0 + 1 ? 53 : 57);
    if (img + 1 < img_end) issue(img + 1);
    if (img == img0) {
#pragma unroll
      for (int j = 0; j < HW; ++j) kos[j] = s_koff[kb + 2 * j + kh];
    }
    rb_f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;
#pragma unroll
    for (int j = 0; j < HW; ++j) {
      const float av = s_w[(kb + 2 * j + kh) * 33 + ml];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = rb_mfma32(av, s_patch[noff[nt] + kos[j]], acc[nt]);
    }
    RB_MSTAMP(img == img0 ? 50 : img == img0 + 1 ? 54 : 58);
    // cross-wave sum one 32-position tile at a time, fixed order w0..w7 (as the one-image kernel)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (nt > 0) __syncthreads();                    // the previous tile's sums are no longer read
#pragma unroll
      for (int r = 0; r < 16; ++r) s_red[(wave * 16 + r) * 64 + lane] = acc[nt][r];
      __syncthreads();                                // (nt == NT - 1: every wave is also done with this image's patch)
#pragma unroll
      for (int it = 0; it < EIT; ++it) {
        const int idx = t + it * RB_CONV_THREADS;
        const int l = idx & 63, r = idx >> 6;
        float v = s_red[(0 * 16 + r) * 64 + l];
#pragma unroll
        for (int wv = 1; wv < RB_CONV_WAVES; ++wv) v += s_red[(wv * 16 + r) * 64 + l];
        const int m = cout0 + rb_mfma_row(r, l);
        const int p = p0 + nt * 32 + (l & 31);
        if (m < a.cout && p < G::P && p < p0 + PCH) {
          const float o = fmaxf(v + bias_r[it], 0.0f);
          a.out[((int64_t)img * a.cout + m) * G::P + p] = o;
          if (a.out_blocked) {
            const int k = m * G::P + p;                                   // x.view(-1, conv_output_size), model.py:71
            a.out_blocked[((int64_t)(k >> 4) * a.rows_total + img) * 16 + (k & 15)] = o;
          }
        }
      }
      if (nt == 0) RB_MSTAMP(img == img0 ? 51 : img == img0 + 1 ? 55 : 59);
    }
    RB_MSTAMP(img == img0 ? 52 : img == img0 + 1 ? 56 : 60);
  }
  RB_MSTAMP(61);
}

// ---- large batches, later layers, WHOLE-K tiles: k_conv_fwd_multi's image loop around the t16 body ------------------------------
// k_conv_fwd_multi above splits the reduction over its 8 waves and sums the partial tiles through LDS — per image NT rounds of
// (16 stores, barrier, 8-way sum, barrier) during which the MFMA pipe idles: MFMA-busy 0.53-0.57 at batch 256 (profiles/
// round5_sq_counters_*).  Here the workgroup is the t16 body's (rb_conv_fwd_body<..., T16 = 1>): one wave per 16-position x 16-channel
// tile over the WHOLE reduction, the epilogue straight from the accumulators — no partial sums, no reduction scratch, two barriers per
// image (patch complete / patch free).  The row-major 32-channel slab (and the bias terms) are set up once per net, the next image's
// patch is in flight (registers) under this image's MFMA loop.  Same LDS image as the one-image t16 kernel (117 / 95 KB for the
// canonical layers 2 / 3).  Needs cin * KK == KMAX, cin % 4 == 0, KMAX % 16 == 0, cout % 32 == 0 (host-checked).
// grid = (position chunks, cout / 32, image groups) or image-group-fastest (a.img_fast); block = 64 * NWV.
template <class G, int NT, int PR, int KMAX, int PCH = 32 * NT>
__global__ __launch_bounds__((64 * ConvFwdWaves<G, NT, PR, KMAX, false, PCH, false, 1>::NWV))
void k_conv_fwd_multi_t16(ConvLdsFwdArgs a) {
  typedef ConvFwdLdsSize<G, NT, PR, KMAX, 1> SZ;
  typedef ConvFwdWaves<G, NT, PR, KMAX, false, PCH, false, 1> WV;
  constexpr int NWV = WV::NWV, THREADS = 64 * NWV, TILE_WAVES = WV::TILE_WAVES;
  constexpr int WS = SZ::WS, PLANE = SZ::PLANE, SUB = SZ::SUB, RP = SZ::RP, CMAX = KMAX / G::KK;
  constexpr int PT = (PCH + 15) / 16, KQ = KMAX / 4, CQ = CMAX / 4;
  static_assert(KMAX % 16 == 0 && CMAX % 4 == 0, "t16: whole float4s per k-slot");
  static_assert(SZ::KPAD == KMAX, "the slab has no padded columns");
  constexpr bool DB = (32 * WS + 2 * CMAX * PLANE) * 4 <= 150 * 1024;      // room for a second patch buffer
  __shared__ __attribute__((aligned(16))) float smem[SZ::FLOATS + (DB ? CMAX * PLANE : 0)];
  float* s_w = smem;
  float* s_patch = smem + 32 * WS;
  const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
  const int img0 = (a.img_fast ? (int)blockIdx.x : (int)blockIdx.z) * a.ipb;
  const int img_end = img0 + a.ipb < a.rows_total ? img0 + a.ipb : a.rows_total;
  const int cout0 = (int)blockIdx.y * 32;
  const int p0 = (a.img_fast ? (int)blockIdx.z : (int)blockIdx.x) * PCH;
  const int cin = a.cin;                              // == CMAX
  const int oy0 = p0 / G::OH;
  const int iy0 = oy0 * G::S;
  int rows = G::IH - iy0;
  if (rows > PR) rows = PR;
  const int per_c = rows * G::IH;
  auto pcell = [&](int c, int off) -> int {
    const int r = off / G::IH, x = off - r * G::IH;
    return c * PLANE + r * RP + (x % G::S) * SUB + x / G::S;
  };
  // ---- the patch of one image: loads into registers (issue), de-interleaved LDS stores later (commit): rb_conv_fwd_body's f32 paths
  constexpr bool x_vec = (G::IH % 4) == 0;
  constexpr int XV = x_vec ? (CMAX * PR * G::IH / 4 + THREADS - 1) / THREADS : 1;
  constexpr int XS = x_vec ? 1 : (CMAX * PR * G::IH + THREADS - 1) / THREADS;
  const int v4 = per_c >> 2, total4 = cin * v4, total1 = cin * per_c;
  float4 xv[XV];
  float xs[XS];
  auto issue = [&](int img) {
    const float* xbase = a.in_f + (int64_t)img * cin * G::IP;
    if constexpr (x_vec) {
#pragma unroll
      for (int i = 0; i < XV; ++i) {
        const int e = i * THREADS + t;
        const int ec = e < total4 ? e : total4 - 1;
        const int c = ec / v4, q = ec - c * v4;
        xv[i] = rb_ld4(xbase + (int64_t)c * G::IP + iy0 * G::IH + q * 4);
      }
    } else {
#pragma unroll
      for (int i = 0; i < XS; ++i) {
        const int e = i * THREADS + t;
        const int ec = e < total1 ? e : total1 - 1;
        const int c = ec / per_c, q = ec - c * per_c;
        xs[i] = xbase[(int64_t)c * G::IP + iy0 * G::IH + q];
      }
    }
  };
  auto commit = [&](float* dst) {
    if constexpr (x_vec) {
#pragma unroll
      for (int i = 0; i < XV; ++i) {
        const int e = i * THREADS + t;
        if (e < total4) {
          const int c = e / v4, q = e - c * v4;
          if constexpr (G::S == 1) { rb_st4(dst + c * PLANE + q * 4, xv[i]); }
          else if constexpr (G::S == 2 && (SUB % 2) == 0) {
            const int off = q * 4, r = off / G::IH, x = off - r * G::IH;
            float* cell = dst + c * PLANE + r * RP + x / 2;
            *reinterpret_cast<float2*>(cell) = make_float2(xv[i].x, xv[i].z);
            *reinterpret_cast<float2*>(cell + SUB) = make_float2(xv[i].y, xv[i].w);
          } else {
            dst[pcell(c, q * 4 + 0)] = xv[i].x; dst[pcell(c, q * 4 + 1)] = xv[i].y;
            dst[pcell(c, q * 4 + 2)] = xv[i].z; dst[pcell(c, q * 4 + 3)] = xv[i].w;
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < XS; ++i) {
        const int e = i * THREADS + t;
        if (e < total1) { const int c = e / per_c, q = e - c * per_c; dst[pcell(c, q)] = xs[i]; }
      }
    }
  };
  // ---- this wave's tile (rb_conv_fwd_body, T16 section): position tile pt, channel tile ct0; lane (x, kq)
  const bool tile_wave = wave < TILE_WAVES;
  const int pt = wave % PT, ct0 = (wave / PT) % 2;
  const int x = lane & 15, kq = lane >> 4;
  int p = p0 + pt * 16 + x;
  const bool pv = p < G::P && p < p0 + PCH;
  if (p > G::P - 1) p = G::P - 1;                     // clamped lanes are never stored
  const float* bp = s_patch + kq * CQ * PLANE + (p / G::OH - oy0) * G::S * RP + (p % G::OH);
  const float* ap = s_w + (ct0 * 16 + x) * WS + kq * KQ;
  float bias4[4] = {0.0f, 0.0f, 0.0f, 0.0f};

  auto stage_slab = [&](int img) {                    // the slab (row-major, WS apart) and the bias terms of image img's net
    const int net = img < a.n_on ? 0 : 1;
    const int rows_valid_w = a.cout - cout0 < 32 ? a.cout - cout0 : 32;
    for (int e = t; e < 32 * (KMAX / 4); e += THREADS) {
      const int m = e / (KMAX / 4), q = e - m * (KMAX / 4);
      float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (m < rows_valid_w) v = rb_ld4(a.w[net] + (int64_t)(cout0 + m) * KMAX + 4 * q);
      rb_st4(s_w + m * WS + 4 * q, v);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = cout0 + ct0 * 16 + 4 * kq + r;
      bias4[r] = a.bias[net][m < a.cout ? m : a.cout - 1];
    }
  };
  auto tile = [&](int img, const float* bpi) {        // this wave's tile of image img from the patch bpi points into: MFMAs + epilogue
    rb_f32x4 acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int jq = 0; jq < KQ / 4; ++jq) {
      const float4 w4 = rb_ld4(ap + 4 * jq);
      const float b0 = bpi[rb_t16_off<G, KMAX, PLANE, RP, SUB>(4 * jq + 0)], b1 = bpi[rb_t16_off<G, KMAX, PLANE, RP, SUB>(4 * jq + 1)];
      const float b2 = bpi[rb_t16_off<G, KMAX, PLANE, RP, SUB>(4 * jq + 2)], b3 = bpi[rb_t16_off<G, KMAX, PLANE, RP, SUB>(4 * jq + 3)];
      acc = rb_mfma16(w4.x, b0, acc);
      acc = rb_mfma16(w4.y, b1, acc);
      acc = rb_mfma16(w4.z, b2, acc);
      acc = rb_mfma16(w4.w, b3, acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {                     // D[r]: channel 4 kq + r of the tile, position x
      const int m = cout0 + ct0 * 16 + 4 * kq + r;
      if (pv && m < a.cout) {
        const float o = fmaxf(acc[r] + bias4[r], 0.0f);
        a.out[((int64_t)img * a.cout + m) * G::P + p] = o;
        if (a.out_blocked) {
          const int k = m * G::P + p;                                   // x.view(-1, conv_output_size), model.py:71
          a.out_blocked[((int64_t)(k >> 4) * a.rows_total + img) * 16 + (k & 15)] = o;
        }
      }
    }
  };

  issue(img0);
  if constexpr (DB) {
    // TWO patch buffers (they fit beside the slab: the third canonical layer): image i + 1's patch is written to the other buffer
    // at the START of iteration i — its LDS stores overlap the first MFMAs of image i — and image i + 2's loads are requested right
    // behind it; ONE barrier per image (patch i + 1 complete, patch i free).
    stage_slab(img0);
    commit(s_patch);
    __syncthreads();                                  // (the slab's stores and the first patch: once)
    if (img0 + 1 < img_end) issue(img0 + 1);
    int cur = 0;
    for (int img = img0; img < img_end; ++img) {
      if (img != img0 && img == a.n_on) {             // block-uniform: the net changes inside this group (every wave is past the barrier)
        stage_slab(img);
        __syncthreads();
      }
      if (img + 1 < img_end) {
        commit(s_patch + (cur ^ 1) * (CMAX * PLANE));
        if (img + 2 < img_end) issue(img + 2);
      }
      if (tile_wave) tile(img, bp + cur * (CMAX * PLANE));
      __syncthreads();
      cur ^= 1;
    }
  } else {
    // RB_STAMP builds (tools/wg_timeline.py, kernel id = layer): slot 0 start, 1 / 3 the first / second image's patch (and slab)
    // complete, 2 / 4 its tiles done, 6 end
    const int wgt = (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
    constexpr int WKM = G::KS == 4 ? 1 : 2;
    (void)wgt; (void)WKM;
    RB_WGT(WKM, wgt, 0);
    for (int img = img0; img < img_end; ++img) {
      if (img == img0 || img == a.n_on) stage_slab(img);   // block-uniform (every wave is past the previous image's MFMA loop: the barrier below)
      commit(s_patch);
      __syncthreads();                                // patch (and slab) complete
      if (img == img0) RB_WGT(WKM, wgt, 1);
      if (img == img0 + 1) RB_WGT(WKM, wgt, 3);
      if (img + 1 < img_end) issue(img + 1);
      if (tile_wave) tile(img, bp);                   // wave-uniform; the spare waves (NWV is a multiple of 4) only stage
      __syncthreads();                                // every wave is done reading this image's patch (and, at a net change, the slab)
      if (img == img0) RB_WGT(WKM, wgt, 2);
      if (img == img0 + 1) RB_WGT(WKM, wgt, 4);
    }
    RB_WGT(WKM, wgt, 5);
    RB_WGT(WKM, wgt, 6);
  }
}

// ---- large batches, FIRST layer: whole image per workgroup, no split of the reduction ---------------------------
// The first layer's reduction is short (K = 256): splitting it over 8 waves leaves 16 MFMA steps per wave and tile, and
// the cross-wave sum + barriers cost as much as the MFMAs (measured 2.7 us of 5.6 per 80-position chunk).  Here the whole
// image (cin planes, decoded to f32) and the 32-channel slab sit in LDS (113 + 34 KB), every wave owns whole 32-position
// tiles (wave w: tiles w and w + 8) and runs the full reduction for them: no partial sums, no scratch, the epilogue
// goes from the accumulators to memory.  A workgroup walks a.ipb images of one net; the next image's frames are in
// flight under the MFMA loop.
// grid = (1, 1, image groups); block = 512.  Requires cout <= 32, cin * KK == KMAX, u8 frames.
template <class G>
__device__ __forceinline__ constexpr int rb_patch_off(int k) {           // reduction index (c, ky, kx) -> offset in the image planes
  return (k / G::KK) * G::IP + ((k % G::KK) / G::KS) * G::IH + (k % G::KK) % G::KS;
}
template <class G, int KMAX>
struct ConvFwdFullLds {
  static constexpr int KPAD = (KMAX + 1) / 2 * 2;
  static constexpr int CMAX = KMAX / G::KK;
  static constexpr int NTILES = (G::P + 31) / 32;
  // TAIL16: the last 32-position tile holds at most 16 positions and 12 full tiles precede it (the canonical first layer: 400 =
  // 12 * 32 + 16).  As a 13th 32x32 tile it gave ONE SIMD four tiles and the others three (waves w and w + 4 share a SIMD).  It
  // runs instead as four 16x16x4 units — (channel half, reduction half), one on each of waves 4..7, i.e. one per SIMD — whose
  // two reduction halves meet through 4 KB of LDS behind the end-of-image barrier: 3.25 tiles per SIMD instead of 4 / 3 / 3 / 3.
#if defined(RB_NO_TAIL16)      // (variant build for A/B runs)
  static constexpr bool TAIL16 = false;
#else
  static constexpr bool TAIL16 = (G::P % 32) != 0 && (G::P % 32) <= 16 && NTILES == 13 && RB_CONV_WAVES == 8 && (G::KS % 4) == 0 && (CMAX % 2) == 0;
#endif
  static constexpr int TAILF = TAIL16 ? 4 * 4 * 64 : 0;
  static constexpr int FLOATS = KPAD * 33 + CMAX * G::IP + TAILF;
  static constexpr bool FITS = FLOATS * 4 <= 160 * 1024 && NTILES <= 2 * RB_CONV_WAVES && (G::IP % 16) == 0;
};
template <class G, int KMAX>
__global__ __launch_bounds__(RB_CONV_THREADS) void k_conv_fwd_full(ConvLdsFwdArgs a) {
  typedef ConvFwdFullLds<G, KMAX> SZ;
  constexpr int KPAD = SZ::KPAD, CMAX = SZ::CMAX, NTILES = SZ::NTILES;
  __shared__ __attribute__((aligned(16))) float s_all[SZ::FLOATS];
  float* s_w = s_all;
  float* s_patch = s_all + KPAD * 33;
  float* s_tail = s_all + KPAD * 33 + CMAX * G::IP;    // TAIL16: [unit][4][64] partial tiles
  (void)s_tail;
  constexpr bool TAIL16 = SZ::TAIL16;
  constexpr int FT = TAIL16 ? NTILES - 1 : NTILES;     // tiles that run as 32x32 tiles

  const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
#if defined(RB_STAMP)
  const bool fst = G::KS == 8 && t == 0 && blockIdx.z == 0;
#define RB_FSTAMP(i) do { if (fst) g_cstamp[i] = wall_clock64(); } while (0)
#else
#define RB_FSTAMP(i) ((void)0)
#endif
  RB_FSTAMP(48);
  // images [z * ipb, (z + 1) * ipb) of the whole list (net 0's images first): a workgroup whose range straddles the two
  // nets re-stages the weight slab once — uniform groups keep 768 images at exactly 3 per workgroup on 256 CUs
  const int img0 = (int)blockIdx.z * a.ipb;
  const int img_end = img0 + a.ipb < a.rows_total ? img0 + a.ipb : a.rows_total;
  const int cin = a.cin;
  const int K = cin * G::KK;

  // frames of one image: 16-byte loads into registers (issue), decoded to exact x/255 into LDS later (commit)
  constexpr int V16 = G::IP / 16;
  constexpr int NU = (CMAX * V16 + RB_CONV_THREADS - 1) / RB_CONV_THREADS;
  uint4 pu[NU];
  const int total16 = cin * V16;
  auto issue = [&](int img) {
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int e = i * RB_CONV_THREADS + t;
      pu[i] = make_uint4(0u, 0u, 0u, 0u);
      if (e < total16) {
        const int c = e / V16, q = e - c * V16;
        const uint8_t* fp = rb_frame_ptr(a.src, img, c, cin, G::IP);
        if (fp) pu[i] = *reinterpret_cast<const uint4*>(fp + q * 16);
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int e = i * RB_CONV_THREADS + t;
      if (e < total16) {
        float* d = s_patch + e * 16;                   // planes are contiguous: c * IP + q * 16
        const unsigned wds[4] = {pu[i].x, pu[i].y, pu[i].z, pu[i].w};
#pragma unroll
        for (int wd = 0; wd < 4; ++wd)
#pragma unroll
          for (int b = 0; b < 4; ++b) d[wd * 4 + b] = rb_unit((uint8_t)((wds[wd] >> (8 * b)) & 0xFFu));
      }
    }
  };

  const int kh = lane >> 5, ml = lane & 31;
  const bool two = wave + RB_CONV_WAVES < FT;                           // wave-uniform: a second tile
  // TAIL16 unit of waves 4..7: channel half ct, reduction half kh2; lane (x, kq) = (position / channel row, k slot)
  const int tu = wave - 4, tct = tu & 1, tkh = (tu >> 1) & 1, tx = lane & 15, tkq = lane >> 4;
  float tbias[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  (void)tct; (void)tkh; (void)tx; (void)tkq;
  int noff[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    int p = (wave + u * RB_CONV_WAVES) * 32 + ml;
    if (p > G::P - 1) p = G::P - 1;                  // clamped lanes are never stored
    noff[u] = (p / G::OH) * G::S * G::IH + (p % G::OH) * G::S;
  }
  float bias_r[16];

  issue(img0);
  for (int img = img0; img < img_end; ++img) {
    if (img == img0 || img == a.n_on) {               // block-uniform: (re)stage the slab and bias of this image's net
      const int net = img < a.n_on ? 0 : 1;           // (every wave has passed the end-of-image barrier: s_w is idle)
      rb_stage_weights_t(s_w, a.w[net], 0, a.cout < 32 ? a.cout : 32, K, KPAD);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = rb_mfma_row(r, lane);
        bias_r[r] = a.bias[net][m < a.cout ? m : a.cout - 1];
      }
      if constexpr (TAIL16) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = tct * 16 + 4 * tkq + r;
          tbias[r] = a.bias[net][m < a.cout ? m : a.cout - 1];
        }
      }
    }
    RB_FSTAMP(img == img0 ? 57 : 58);
    commit();
    RB_FSTAMP(img == img0 ? 59 : 60);
    __syncthreads();            // image complete (first image: weights and tap table as well)
    RB_FSTAMP(img == img0 ? 49 : img == img0 + 1 ? 53 : 61);
    if (img + 1 < img_end) issue(img + 1);
    rb_f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
    // tap offsets are compile-time functions of the step (no table: a table read per step put two dependent LDS round
    // trips in front of every MFMA — measured 38 us per image for 13 us of MFMAs); K == KMAX (host-checked): straight-line
    // code the compiler can pipeline
    if constexpr (G::KS % 2 == 0) {
      // even kernel sizes: the two taps of a step are neighbours (kx, kx + 1) — the lane's half goes into the base
      // pointers, a step's offsets are immediates, and the channel loop only advances the bases (a fully unrolled
      // 128-step body needed a base register per 1 KB window of ds_read2: 255 VGPRs and a scratch segment)
      const float* pb0 = s_patch + noff[0] + kh;
      const float* pb1 = s_patch + noff[1] + kh;
      const float* wp = s_w + kh * 33 + ml;
      if (two) {
#pragma unroll 1
        for (int c = 0; c < CMAX; ++c) {
#pragma unroll
          for (int jj = 0; jj < G::KK / 2; ++jj) {
            const float av = wp[2 * jj * 33];
            acc0 = rb_mfma32(av, pb0[rb_patch_off<G>(2 * jj)], acc0);
            acc1 = rb_mfma32(av, pb1[rb_patch_off<G>(2 * jj)], acc1);
          }
          pb0 += G::IP; pb1 += G::IP; wp += G::KK * 33;
        }
      } else {
#pragma unroll 1
        for (int c = 0; c < CMAX; ++c) {
#pragma unroll
          for (int jj = 0; jj < G::KK / 2; ++jj) {
            const float av = wp[2 * jj * 33];
            acc0 = rb_mfma32(av, pb0[rb_patch_off<G>(2 * jj)], acc0);
          }
          pb0 += G::IP; wp += G::KK * 33;
        }
      }
    } else {
      // odd sizes: a step's two taps can sit in different rows or planes — select between two constants
      if (two) {
#pragma unroll
        for (int j = 0; j < KPAD / 2; ++j) {
          const int o = kh ? rb_patch_off<G>(2 * j + 1) : rb_patch_off<G>(2 * j);
          const float av = s_w[(2 * j + kh) * 33 + ml];
          acc0 = rb_mfma32(av, s_patch[noff[0] + o], acc0);
          acc1 = rb_mfma32(av, s_patch[noff[1] + o], acc1);
        }
      } else {
#pragma unroll
        for (int j = 0; j < KPAD / 2; ++j) {
          const int o = kh ? rb_patch_off<G>(2 * j + 1) : rb_patch_off<G>(2 * j);
          const float av = s_w[(2 * j + kh) * 33 + ml];
          acc0 = rb_mfma32(av, s_patch[noff[0] + o], acc0);
        }
      }
    }
    if constexpr (TAIL16) {
      if (wave >= 4) {                                  // wave-uniform
        constexpr int CH = CMAX / 2;                    // channels per reduction half
        int p = (NTILES - 1) * 32 + tx;
        if (p > G::P - 1) p = G::P - 1;                 // clamped lanes are never stored
        // k = c * KK + 4 j + kq: KS % 4 == 0, so the slot kq stays inside a kernel row — it goes into the base pointers and a
        // step's offsets are immediates, as in the 32x32 loops above
        const float* pb = s_patch + tkh * CH * G::IP + (p / G::OH) * G::S * G::IH + (p % G::OH) * G::S + tkq;
        const float* wp = s_w + (tkh * CH * G::KK + tkq) * 33 + tct * 16 + tx;
        rb_f32x4 tacc;
#pragma unroll
        for (int r = 0; r < 4; ++r) tacc[r] = 0.0f;
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
          for (int j = 0; j < G::KK / 4; ++j)
            tacc = rb_mfma16(wp[(c * G::KK + 4 * j) * 33], pb[c * G::IP + rb_patch_off<G>(4 * j)], tacc);
#pragma unroll
        for (int r = 0; r < 4; ++r) s_tail[(tu * 4 + r) * 64 + lane] = tacc[r];
      }
    }
    RB_FSTAMP(img == img0 ? 50 : 54);
    // epilogue straight from the accumulators: row r of the tile is output channel rb_mfma_row(r, lane), 32 consecutive
    // positions per half-wave (contiguous in the NCHW activation)
    float* outi = a.out + (int64_t)img * a.cout * G::P;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = rb_mfma_row(r, lane);
      const int p0 = wave * 32 + ml;
      if (m < a.cout && p0 < G::P) outi[m * G::P + p0] = fmaxf(acc0[r] + bias_r[r], 0.0f);
      const int p1 = (wave + RB_CONV_WAVES) * 32 + ml;
      if (two && m < a.cout && p1 < G::P) outi[m * G::P + p1] = fmaxf(acc1[r] + bias_r[r], 0.0f);
    }
    RB_FSTAMP(img == img0 ? 51 : 55);
    __syncthreads();            // every wave is done reading this image before the next one is committed
    if constexpr (TAIL16) {
      // the tail tile: reduction half 0 + half 1 (fixed order), bias, ReLU — waves 4 and 5, one channel half each.  The scratch
      // is written again only behind the next image's commit barrier.
      if (wave == 4 || wave == 5) {
        const int p = (NTILES - 1) * 32 + tx;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = tct * 16 + 4 * tkq + r;
          const float v = s_tail[(tu * 4 + r) * 64 + lane] + s_tail[((tu + 2) * 4 + r) * 64 + lane];
          if (m < a.cout && p < G::P) outi[m * G::P + p] = fmaxf(v + tbias[r], 0.0f);
        }
      }
    }
    RB_FSTAMP(img == img0 ? 52 : 56);
  }
}

// (RB_STAMP: end-of-kernel stamps are written by the host-visible tail below)
// ---- the input-gradient kernels' weight operand, made once per step -------------------------------------------------------
// wT[phase][tile][k'][32]: element (co, c, ky, kx) of a layer's [cout][cin][KS][KS] weights goes to phase (ky % S, kx % S),
// channel tile c / 32, row k' = co * taps(phase) + (ky / S) * ntx(phase) + kx / S, column c % 32.  Runtime geometry (one
// body for every layer); `nblk` workgroups of any size share the elements.
struct ConvWtJob {
  const float* w;
  float* wT;
  int cin, cout, KS, S, kpad;
  int t16;                 // 1: the [phase][c][k'] layout of k_conv_dx_t16_multi (rb_conv_wt16_block), same buffer
};
__device__ __forceinline__ void rb_conv_wt16_block(const ConvWtJob& j, int blk, int nblk);
__device__ __forceinline__ void rb_conv_wt_block(const ConvWtJob& j, int blk, int nblk) {
  if (j.t16) { rb_conv_wt16_block(j, blk, nblk); return; }        // block-uniform
  const int KK = j.KS * j.KS, total = j.cout * j.cin * KK, ntiles = (j.cin + 31) / 32;
  for (int e = blk * (int)blockDim.x + (int)threadIdx.x; e < total; e += nblk * (int)blockDim.x) {
    const int co = e / (j.cin * KK), r = e - co * (j.cin * KK);
    const int c = r / KK, tap = r - c * KK;
    const int ky = tap / j.KS, kx = tap - ky * j.KS;
    const int py = ky % j.S, px = kx % j.S, ty = ky / j.S, tx = kx / j.S;
    const int nty = (j.KS - py + j.S - 1) / j.S, ntx = (j.KS - px + j.S - 1) / j.S;
    const int kp = co * (nty * ntx) + ty * ntx + tx;
    j.wT[(((int64_t)(py * j.S + px) * ntiles + (c >> 5)) * j.kpad + kp) * 32 + (c & 31)] = j.w[e];
  }
}

// ========================================================================= data gradient ==
// dX[img][c][y][x] = relu'(x_act) * sum_{co,ky,kx} W[co][c][ky][kx] * dY[img][co][(y-ky)/S][(x-kx)/S]
// decomposed by phase (y % S, x % S) so only real taps are visited.  The whole dY image sits in LDS.
// grid = (phases S*S * position groups of 32*NT, cin / 32, images B); block = 512.
struct ConvLdsDxArgs {
  int cin, cout;
  const float* w;        // [cout][cin][KS][KS]
  const float* wT;       // the same weights as the kernel wants them: [phase][32-channel tile][KPAD rows k' = (co, tap)][32]
                         // (rb_conv_wt_block below writes it earlier in the step, as tenant workgroups of the head launch)
  const float* dy;       // [B][cout][P]
  const float* x_act;    // [NI][cin][IP] (rows [0,B))
  float* dx;             // [B][cin][IP]
  // LAZY instantiation only (the last conv layer): dY is not materialised — the staging sums the hidden layer's
  // dy_splits (<= 4) row-split partials [s][B][cout * P] and applies relu'(dy_mask) itself
  const float* dy_part;
  const float* dy_mask;  // the layer's own activation, rows [0,B)
  int64_t dy_stride;     // floats between partials
  int dy_splits;
  int ipb, batch;        // MULTI instantiation: images per workgroup (grid z = ceil(batch / ipb)), image count
  int img_fast;          // grid = (image groups, channel tiles, phase x position groups): see k_conv_fwd_lds
};

// MULTI (batches of 64 and more): a workgroup keeps its weight slab and walks a.ipb images — per image only the dY tile
// is staged (31 KB against 76 KB of transposed weights for the third layer); the reduction scratch then has a region of
// its own instead of overlaying the operands.
template <class G, int NT, int COUT, bool LAZY = false, bool MULTI = false>
__global__ __launch_bounds__(RB_CONV_THREADS) void k_conv_dx_lds(ConvLdsDxArgs a) {
  constexpr int TMAX = (G::KS + G::S - 1) / G::S;           // taps per dimension of a phase
  constexpr int KMAX = COUT * TMAX * TMAX;
  constexpr int KPAD = (KMAX + 2 * RB_CONV_WAVES - 1) / (2 * RB_CONV_WAVES) * (2 * RB_CONV_WAVES);
  // dY sits in LDS with a zero halo: every (position, tap) pair then addresses a legal cell and the
  // MFMA loop needs no bounds tests (with them it ran at a quarter of the MFMA rate: 4.1 us for 1 us of MFMAs)
  // (low side: TMAX-1 taps reach before the first output; high side: phase positions up to ceil(IH/S)-1, which also
  // covers input rows no output touches when (OH-1)*S + KS < IH)
  constexpr int PAD = TMAX - 1, PADH = (G::IH + G::S - 1) / G::S - G::OH, PW = G::OH + PAD + PADH, PP = PW * PW;
  constexpr int RED = RB_CONV_WAVES * NT * 16 * 64;
  // row stride of the [k'][c] weight slab.  The MFMA operand read (32 consecutive c of a row) is conflict-free at any stride;
  // the stride-1 staging stores are not: a thread holds 4 consecutive elements of a (c, tap) run, lanes 4 elements apart, and
  // with 33 the bank is (tap + c) mod 32 — 8.8 lanes per bank on average (SQ_LDS_BANK_CONFLICT: 61 % of the kernel's LDS
  // cycles); 38 spreads them to 2.0 per bank
  constexpr int WLD = 36;      // (16-byte aligned rows: the slab is a straight copy of a.wT, 16-byte loads -> 16-byte LDS stores)
  constexpr int OPS = KPAD * WLD + COUT * PP;
  constexpr int WSZ = MULTI ? OPS + RED : (OPS > RED ? OPS : RED);
  __shared__ __attribute__((aligned(16))) float s_all[WSZ];
  float* s_w = s_all;
  float* s_dy = s_all + KPAD * WLD;
  float* s_red = MULTI ? s_all + OPS : s_all;
  __shared__ int s_koff[KPAD];      // co*PP - ty*PW - tx

  const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
  constexpr int WK = G::KS == 3 ? 3 : 4;                   // timeline ids (RB_STAMP builds only)
  const int wgi = (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
  (void)wgi;
  RB_WGT(WK, wgi, 0);
  RB_WGT_HW(WK, wgi);
  const int ipb = MULTI ? a.ipb : 1;
  const int bx_ = a.img_fast ? (int)blockIdx.z : (int)blockIdx.x, bz_ = a.img_fast ? (int)blockIdx.x : (int)blockIdx.z;
  const int img0 = bz_ * ipb;
  const int c0 = (int)blockIdx.y * 32;
  const int phase = bx_ % (G::S * G::S);
  const int n0 = (bx_ / (G::S * G::S)) * (32 * NT);                 // first position of this block inside the phase
  const int py = phase / G::S, px = phase % G::S;
  const int nty = (G::KS - py + G::S - 1) / G::S, ntx = (G::KS - px + G::S - 1) / G::S;
  const int nyy = (G::IH - py + G::S - 1) / G::S, nxx = (G::IH - px + G::S - 1) / G::S;
  const int taps = nty * ntx;
  const int K = a.cout * taps;
  const int npos = nyy * nxx;
  if (n0 >= npos) return;                                            // block-uniform

  // ---- once per workgroup: the tap table and the phase's weight slab transposed to [k'][c]
  for (int k = t; k < KPAD; k += RB_CONV_THREADS) {
    const int kc = k < K ? k : K - 1;
    const int co = kc / taps, r = kc - co * taps;
    const int ty = r / ntx, tx = r - ty * ntx;
    s_koff[k] = co * PP - ty * PW - tx;
  }
  constexpr int KW = KPAD / RB_CONV_WAVES;            // even, compile-time (rows >= K of s_w are zero): full unroll
  const int kb = wave * KW;
  int noff[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    int n = n0 + nt * 32 + (lane & 31);
    if (n > npos - 1) n = npos - 1;
    const int yy = n / nxx, xx = n - yy * nxx;
    noff[nt] = (yy + PAD) * PW + xx + PAD;
  }
  const int kh = lane >> 5, ml = lane & 31;
  // the epilogue's cells (they do not depend on the image): offset inside the image, -1 = nothing to store
  constexpr int EIT = (NT * 16 * 64) / RB_CONV_THREADS;
  int eoff[EIT];
#pragma unroll
  for (int it = 0; it < EIT; ++it) {
    const int idx = t + it * RB_CONV_THREADS;
    const int l = idx & 63, r = (idx >> 6) & 15, nt = idx >> 10;
    const int c = c0 + rb_mfma_row(r, l);
    const int n = n0 + nt * 32 + (l & 31);
    const int yy = n / nxx, xx = n - yy * nxx;
    eoff[it] = (c < a.cin && n < npos) ? c * G::IP + (yy * G::S + py) * G::IH + xx * G::S + px : -1;
  }
  int kos[KW / 2];                                    // tap offsets of this wave's k range, off the per-step critical path

  // dY of an image: global loads into registers (issue), LDS stores later (commit) — with MULTI the next image's loads
  // are in flight under this image's MFMA loop and reduction
  // The staging of these kernels is INSTRUCTION-bound, not memory-bound (fine-grained stamps, tools/stamp/fine_dx.py: 7.5 us from
  // workgroup start to the first MFMA with every load landed at 3.3 us — two waves per SIMD executing ~2000 VALU
  // instructions of index arithmetic each).  So: only the INTERIOR cells of dY are loaded and stored (contiguous in memory:
  // no halo-indexed gather), the zero halo is one block of 16-byte stores at kernel start, all offsets are 32-bit and go
  // through buffer loads (no 64-bit pointer arithmetic per load).
  constexpr int LIT = (COUT * G::P + RB_CONV_THREADS - 1) / RB_CONV_THREADS;
  float pre_m[LAZY ? LIT : 1], pre_p[LAZY ? LIT : 1][4], pre_v[LAZY ? 1 : LIT];
  const int ni = a.cout * G::P, nh = a.cout * PP;
  int cell[LIT];                                        // LDS cell of this thread's i-th interior element (image-independent)
#pragma unroll
  for (int i = 0; i < LIT; ++i) {
    const int e = t + i * RB_CONV_THREADS;
    const int ec = e < ni ? e : ni - 1;
    const int co = ec / G::P, r = ec - co * G::P;
    const int y = r / G::OH, x = r - y * G::OH;
    cell[i] = e < ni ? co * PP + (y + PAD) * PW + x + PAD : -1;
  }
  if (PW > G::OH) {                                     // zero the whole haloed image once: the halo stays zero from image to image
    for (int e = t; e < (COUT * PP) / 4; e += RB_CONV_THREADS) rb_st4(s_dy + 4 * e, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    for (int e = (COUT * PP) / 4 * 4 + t; e < COUT * PP; e += RB_CONV_THREADS) s_dy[e] = 0.0f;
  }
  auto issue = [&](int img) {
    const unsigned ibase = 4u * (unsigned)(img * ni);
    if constexpr (LAZY) {
      // interior cells: the mask and every partial of a thread's cells are requested before the first add (one round trip)
      const rb_buf mk = rb_make_buf(a.dy_mask);
      rb_buf pp[4];
#pragma unroll
      for (int sp = 0; sp < 4; ++sp) pp[sp] = rb_make_buf(a.dy_part + (int64_t)(sp < a.dy_splits ? sp : a.dy_splits - 1) * a.dy_stride);
#pragma unroll
      for (int i = 0; i < LIT; ++i) {
        const int e = t + i * RB_CONV_THREADS;
        const unsigned off = 4u * (unsigned)(e < ni ? e : ni - 1);
        pre_m[i] = rb_ld1_buf(mk, off, ibase);
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) pre_p[i][sp] = rb_ld1_buf(pp[sp], off, ibase);
      }
    } else {
      const rb_buf src = rb_make_buf(a.dy);
#pragma unroll
      for (int i = 0; i < LIT; ++i) {
        const int e = t + i * RB_CONV_THREADS;
        pre_v[i] = rb_ld1_buf(src, 4u * (unsigned)(e < ni ? e : ni - 1), ibase);
      }
    }
  };
  auto commit = [&](bool first) {
    (void)first;
    if constexpr (LAZY) {
#pragma unroll
      for (int i = 0; i < LIT; ++i) {
        if (cell[i] >= 0) {
          float acc = 0.0f;                                                 // k_dfeat_finish's order: ((0 + p0) + p1) + ...
#pragma unroll
          for (int sp = 0; sp < 4; ++sp) acc += sp < a.dy_splits ? pre_p[i][sp] : 0.0f;
          s_dy[cell[i]] = pre_m[i] > 0.0f ? acc : 0.0f;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < LIT; ++i) if (cell[i] >= 0) s_dy[cell[i]] = pre_v[i];
    }
  };

  issue(img0);
#if defined(RB_STAMP) && defined(RB_STAMP_FINE)
  RB_WGT(WK + 3, wgi, 0);
#endif
  // (the weight slab is staged AFTER the first image's dY loads have been issued: its own loads then share their round
  // trip instead of preceding it)
  {
    // The slab [k' = (co, tap of this phase)][32 channels c0 ..] arrives READY-MADE from a.wT — written once per step by
    // tenant workgroups of the head launch (rb_conv_wt_block), rows >= K and channels >= cin zero — as 16-byte loads and
    // 16-byte LDS stores.  Gathering it here from the [co][c][ky][kx] weights cost every workgroup 3.8-4.3 us of its
    // 7.4-7.7 (tools/wg_timeline.py: ~2000 VALU instructions of index arithmetic per thread in front of the first MFMA).
    constexpr int NQ = (KPAD * 8 + RB_CONV_THREADS - 1) / RB_CONV_THREADS;     // float4s of the slab per thread
    const int ntiles = (a.cin + 31) / 32;
    const float* src = a.wT + ((int64_t)(phase * ntiles + (int)blockIdx.y) * KPAD) * 32;
    float4 v[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      int e = t + i * RB_CONV_THREADS;
      if (e > KPAD * 8 - 1) e = KPAD * 8 - 1;
      v[i] = rb_ld4(src + 4 * e);
    }
#if defined(RB_STAMP) && defined(RB_STAMP_FINE)
    RB_WGT(WK + 3, wgi, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RB_WGT(WK + 3, wgi, 2);
#endif
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int e = t + i * RB_CONV_THREADS;
      if (e < KPAD * 8) rb_st4(s_w + (e >> 3) * WLD + 4 * (e & 7), v[i]);
    }
  }

#if defined(RB_STAMP) && defined(RB_STAMP_FINE)
  RB_WGT(WK + 3, wgi, 3);
#endif
  if (PW > G::OH) __syncthreads();                    // the zero fill (other threads' cells) precedes the interior stores
  for (int ii = 0; ii < ipb; ++ii) {
    const int img = img0 + ii;
    if (MULTI && img >= a.batch) break;               // block-uniform
    commit(ii == 0);
#if defined(RB_STAMP) && defined(RB_STAMP_FINE)
    if (ii == 0) RB_WGT(WK + 3, wgi, 4);
#endif
    // the ReLU mask of this image's output cells: requested now, consumed after the MFMA loop (in the epilogue the load
    // sat on the critical path of every store)
    const float* xa = a.x_act + (int64_t)img * a.cin * G::IP;
    float mask[EIT];
#pragma unroll
    for (int it = 0; it < EIT; ++it) mask[it] = xa[eoff[it] >= 0 ? eoff[it] : 0];
    __syncthreads();            // operands complete (and, MULTI, the previous image's reduction scratch has been consumed)
    if (ii == 0) { RB_WGT(WK, wgi, 1); RB_WGT(WK, wgi, 2); RB_WGT(WK, wgi, 3); }
    if (MULTI && ii + 1 < ipb && img + 1 < a.batch) issue(img + 1);
    if (ii == 0) {
#pragma unroll
      for (int j = 0; j < KW / 2; ++j) kos[j] = s_koff[kb + 2 * j + kh];
    }
    rb_f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;
#pragma unroll
    for (int j = 0; j < KW / 2; ++j) {
      const float av = s_w[(kb + 2 * j + kh) * WLD + ml];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = rb_mfma32(av, s_dy[kos[j] + noff[nt]], acc[nt]);
    }
    if (ii == 0) RB_WGT(WK, wgi, 4);
    if (!MULTI) __syncthreads();                      // the scratch overlays the operands
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) s_red[((wave * NT + nt) * 16 + r) * 64 + lane] = acc[nt][r];
    __syncthreads();                                  // (MULTI: every wave is also done reading this image's dY)
    float* dxi = a.dx + (int64_t)img * a.cin * G::IP;
#pragma unroll
    for (int it = 0; it < EIT; ++it) {
      const int idx = t + it * RB_CONV_THREADS;
      const int l = idx & 63, r = (idx >> 6) & 15, nt = idx >> 10;
      float v = s_red[((0 * NT + nt) * 16 + r) * 64 + l];
#pragma unroll
      for (int wv = 1; wv < RB_CONV_WAVES; ++wv) v += s_red[((wv * NT + nt) * 16 + r) * 64 + l];
      if (eoff[it] >= 0) {
        const float o = mask[it] > 0.0f ? v : 0.0f;
        dxi[eoff[it]] = o;
      }
    }
  }
  RB_WGT(WK, wgi, 5);
  RB_WGT(WK, wgi, 6);
}

// ---- the same data gradient on whole-K 16x16x4 tiles, for the image loop of large batches (round 6) -----------------------------
// k_conv_dx_lds<..., MULTI> splits the reduction over its 8 waves and sums the partial tiles through LDS for every image — the pattern
// k_conv_fwd_multi_t16 removed from the forward (MFMA-busy 0.38 / 0.46 at batch 256, half of the launch spent outside the MFMA loop).
// Here a workgroup owns (stride phase, 32 input channels, ALL positions of the phase): one wave per 16-position x 16-channel tile runs the
// WHOLE reduction k' = (co, tap) for its tile — lane (x, kq) takes the contiguous quarter [kq K'/4, (kq + 1) K'/4), its A operands are
// whole float4s of weight row c (the slab arrives ready-made as wT16[phase][c][k'], rb_conv_wt16_block), its B operands are cells of
// the zero-haloed dY image at compile-time offsets co PPL - ty PW - tx — and the epilogue (relu' mask, store) goes from the accumulators
// to memory.  Needs KS % S == 0 (every phase has the same taps), (COUT * taps) % 16 == 0, cin % 32 == 0 (host-checked).
// grid = (phases, cin / 32, image groups) or image-group-fastest (a.img_fast); block = 64 * NWV.
__device__ __forceinline__ void rb_conv_wt16_block(const ConvWtJob& j, int blk, int nblk) {
  const int KK = j.KS * j.KS, total = j.cout * j.cin * KK, T = (j.KS + j.S - 1) / j.S, taps = T * T, kq = j.cout * taps;
  for (int e = blk * (int)blockDim.x + (int)threadIdx.x; e < total; e += nblk * (int)blockDim.x) {
    const int co = e / (j.cin * KK), r = e - co * (j.cin * KK);
    const int c = r / KK, tap = r - c * KK;
    const int ky = tap / j.KS, kx = tap - ky * j.KS;
    const int py = ky % j.S, px = kx % j.S, ty = ky / j.S, tx = kx / j.S;
    j.wT[((int64_t)(py * j.S + px) * j.cin + c) * kq + co * taps + ty * T + tx] = j.w[e];
  }
}
template <class G, int COUT>
struct ConvDxT16 {
  static constexpr int TMAX = (G::KS + G::S - 1) / G::S, TAPS = TMAX * TMAX;
  static constexpr int KP = COUT * TAPS, KQ = KP / 4, CQ = COUT / 4, WS = KP + 4;
  static constexpr int PAD = TMAX - 1, NS = (G::IH + G::S - 1) / G::S, PADH = NS - G::OH, PW = G::OH + PAD + PADH, PP = PW * PW;
  static constexpr int rb_pad() {
    for (int p = 0; p < 64; ++p)
      if ((CQ * (PP + p)) % 32 == 16) return p;
    return 0;
  }
  static constexpr int PPL = PP + rb_pad();                    // dY plane stride: the four k-quarters of an operand read start 16 banks apart
  static constexpr int NPOS = NS * NS, PT = (NPOS + 15) / 16, TILE_WAVES = 2 * PT, NWV = (TILE_WAVES + 3) / 4 * 4;
  static constexpr int FLOATS = 32 * WS + COUT * PPL;
  static constexpr bool OK = (G::KS % G::S) == 0 && (KP % 16) == 0 && (COUT % 4) == 0 && NWV <= 16 && FLOATS * 4 <= 150 * 1024;
};
template <class G, int COUT, bool LAZY>
__global__ __launch_bounds__((64 * ConvDxT16<G, COUT>::NWV)) void k_conv_dx_t16_multi(ConvLdsDxArgs a) {
  typedef ConvDxT16<G, COUT> Z;
  constexpr int THREADS = 64 * Z::NWV, WS = Z::WS, PPL = Z::PPL, PW = Z::PW, PAD = Z::PAD, KQ = Z::KQ, CQ = Z::CQ, TAPS = Z::TAPS, TMAX = Z::TMAX;
  static_assert(Z::OK, "k_conv_dx_t16_multi: geometry");
  __shared__ __attribute__((aligned(16))) float smem[Z::FLOATS];
  float* s_w = smem;
  float* s_dy = smem + 32 * WS;
  const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
  const int bx_ = a.img_fast ? (int)blockIdx.z : (int)blockIdx.x, bz_ = a.img_fast ? (int)blockIdx.x : (int)blockIdx.z;
  const int img0 = bz_ * a.ipb;
  const int img_end = img0 + a.ipb < a.batch ? img0 + a.ipb : a.batch;
  if (img0 >= a.batch) return;                                // block-uniform
  const int c0 = (int)blockIdx.y * 32;
  const int phase = bx_;
  const int py = phase / G::S, px = phase % G::S;
  const int nyy = (G::IH - py + G::S - 1) / G::S, nxx = (G::IH - px + G::S - 1) / G::S;
  const int npos = nyy * nxx;
  // ---- dY of an image: interior cells only, global loads into registers (issue), LDS stores later (commit); the zero halo is
  // written once (k_conv_dx_lds: the staging of these kernels is instruction-bound)
  constexpr int LIT = (COUT * G::P + THREADS - 1) / THREADS;
  float pre_m[LAZY ? LIT : 1], pre_p[LAZY ? LIT : 1][4], pre_v[LAZY ? 1 : LIT];
  const int ni = a.cout * G::P;
  int cell[LIT];
#pragma unroll
  for (int i = 0; i < LIT; ++i) {
    const int e = t + i * THREADS;
    const int ec = e < ni ? e : ni - 1;
    const int co = ec / G::P, r = ec - co * G::P;
    const int y = r / G::OH, x = r - y * G::OH;
    cell[i] = e < ni ? co * PPL + (y + PAD) * PW + x + PAD : -1;
  }
  for (int e = t; e < (COUT * PPL) / 4; e += THREADS) rb_st4(s_dy + 4 * e, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
  for (int e = (COUT * PPL) / 4 * 4 + t; e < COUT * PPL; e += THREADS) s_dy[e] = 0.0f;
  auto issue = [&](int img) {
    const unsigned ibase = 4u * (unsigned)(img * ni);
    if constexpr (LAZY) {
      const rb_buf mk = rb_make_buf(a.dy_mask);
      rb_buf pp[4];
#pragma unroll
      for (int sp = 0; sp < 4; ++sp) pp[sp] = rb_make_buf(a.dy_part + (int64_t)(sp < a.dy_splits ? sp : a.dy_splits - 1) * a.dy_stride);
#pragma unroll
      for (int i = 0; i < LIT; ++i) {
        const int e = t + i * THREADS;
        const unsigned off = 4u * (unsigned)(e < ni ? e : ni - 1);
        pre_m[i] = rb_ld1_buf(mk, off, ibase);
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) pre_p[i][sp] = rb_ld1_buf(pp[sp], off, ibase);
      }
    } else {
      const rb_buf src = rb_make_buf(a.dy);
#pragma unroll
      for (int i = 0; i < LIT; ++i) {
        const int e = t + i * THREADS;
        pre_v[i] = rb_ld1_buf(src, 4u * (unsigned)(e < ni ? e : ni - 1), ibase);
      }
    }
  };
  auto commit = [&]() {
    if constexpr (LAZY) {
#pragma unroll
      for (int i = 0; i < LIT; ++i) {
        if (cell[i] >= 0) {
          float acc = 0.0f;                                                 // k_dfeat_finish's order: ((0 + p0) + p1) + ...
#pragma unroll
          for (int sp = 0; sp < 4; ++sp) acc += sp < a.dy_splits ? pre_p[i][sp] : 0.0f;
          s_dy[cell[i]] = pre_m[i] > 0.0f ? acc : 0.0f;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < LIT; ++i) if (cell[i] >= 0) s_dy[cell[i]] = pre_v[i];
    }
  };
  issue(img0);
  {   // the slab [32 channels c0 ..][K'] of this phase: a straight copy of a.wT (wT16 layout), rows of channels >= cin zero
    const float* src = a.wT + ((int64_t)phase * a.cin + c0) * Z::KP;
    for (int e = t; e < 32 * (Z::KP / 4); e += THREADS) {
      const int m = e / (Z::KP / 4), q = e - m * (Z::KP / 4);
      float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (c0 + m < a.cin) v = rb_ld4(src + (int64_t)m * Z::KP + 4 * q);
      rb_st4(s_w + m * WS + 4 * q, v);
    }
  }
  // ---- this wave's tile: position tile pt, channel tile ct0; lane (x, kq)
  const bool tile_wave = wave < Z::TILE_WAVES;
  const int pt = wave % Z::PT, ct0 = (wave / Z::PT) % 2;
  const int x = lane & 15, kq = lane >> 4;
  int n = pt * 16 + x;
  const bool pv = n < npos;
  if (n > npos - 1) n = npos - 1;
  const int yy = n / nxx, xx = n - yy * nxx;
  const float* bp = s_dy + kq * CQ * PPL + (yy + PAD) * PW + xx + PAD;
  const float* ap = s_w + (ct0 * 16 + x) * WS + kq * KQ;
  int eoff[4];                                          // the lane's four output cells (channel 4 kq + r of its tile): offset in the image, -1 = none
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c = c0 + ct0 * 16 + 4 * kq + r;
    eoff[r] = (pv && c < a.cin) ? c * G::IP + (yy * G::S + py) * G::IH + xx * G::S + px : -1;
  }
  __syncthreads();                                      // zero fill complete before the first interior stores (other threads' cells)
  for (int img = img0; img < img_end; ++img) {
    commit();
    const float* xa = a.x_act + (int64_t)img * a.cin * G::IP;
    float mask[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) mask[r] = xa[eoff[r] >= 0 ? eoff[r] : 0];
    __syncthreads();                                    // dY (and the slab) complete
    if (img + 1 < img_end) issue(img + 1);
    if (tile_wave) {
      rb_f32x4 acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = 0.0f;
#pragma unroll
      for (int jq = 0; jq < KQ / 4; ++jq) {
        const float4 w4 = rb_ld4(ap + 4 * jq);
        float b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          constexpr int dummy = 0; (void)dummy;
          const int j = 4 * jq + u, co = j / TAPS, tap = j % TAPS;
          b[u] = bp[co * PPL - (tap / TMAX) * PW - (tap % TMAX)];
        }
        acc = rb_mfma16(w4.x, b[0], acc);
        acc = rb_mfma16(w4.y, b[1], acc);
        acc = rb_mfma16(w4.z, b[2], acc);
        acc = rb_mfma16(w4.w, b[3], acc);
      }
      float* dxi = a.dx + (int64_t)img * a.cin * G::IP;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (eoff[r] >= 0) dxi[eoff[r]] = mask[r] > 0.0f ? acc[r] : 0.0f;
    }
    __syncthreads();                                    // every wave is done reading this image's dY
  }
}

// ======================================================================= weight gradient ==
// First-layer weight gradient (the one the generic path is worst at: 12800 reduction positions,
// u8 operand): part[slice][co][col] = sum_{pos in chunk} dY[img][co][pos] * X[img][c][oy*S+ky][ox*S+kx],
// slice = (image, chunk of RC output rows); col == K is the bias column (sum of dY).
// The 8 waves own disjoint 32-wide column tiles of the [32 x K] output, so there is NO cross-wave
// reduction: each wave runs the full position loop for its tile(s) out of LDS.
// grid = (chunks per image, cout / 32, images B); block = 512.
struct ConvLdsDwArgs {
  int cin, cout;
  const float* dy;         // [B][cout][P]
  ImgSrc src;              // FIRST: the state stacks (images [0,B))
  const float* x_f;        // else previous activation [NI][cin][IP], rows [0,B)
  float* part;             // [B * chunks][cout][K+1]
  // dy_splits > 0 (last conv layer): dY = relu'(dy_mask) * sum of dy_splits (<= 4) partials, formed while staging
  const float* dy_part;
  const float* dy_mask;
  int64_t dy_stride;
  int dy_splits;
};

template <class G, int RC, int KMAX>
struct ConvDwLdsSize {
  static constexpr int PC = RC * G::OH;               // positions per chunk
  static constexpr int PCP = (PC + 1) / 2 * 2;        // padded to the MFMA k granule
  static constexpr int PR = (RC - 1) * G::S + G::KS;  // input rows per chunk
  static constexpr int PLANE = PR * G::IH;
  static constexpr int CMAX = KMAX / G::KK;
  static constexpr int FLOATS = PCP * 33 + CMAX * PLANE + PCP + 32;   // dY^T, patch, position offsets, per-image bias sums
};

// body with explicit block coordinates and caller-provided LDS, so several layers can share one launch.
// `grp` = slice index along the image axis: the workgroup sums images [grp * ipb, (grp + 1) * ipb) in registers before
// it writes its slice (ipb = 1 at batch 32; batch 256 uses 8, which keeps the slice count — and the reduction pass
// over the slices — at the batch-32 size).
template <class G, int RC, int KMAX, bool FIRST>
__device__ __forceinline__ void rb_conv_dw_body(const ConvLdsDwArgs& a, int chunk, int cotile, int grp, int nchunks,
                                                int ipb, int batch, float* smem) {
  typedef ConvDwLdsSize<G, RC, KMAX> SZ;
  constexpr int PC = SZ::PC, PCP = SZ::PCP, PR = SZ::PR, PLANE = SZ::PLANE;
  constexpr int TPW = ((KMAX + 31) / 32 + RB_CONV_WAVES - 1) / RB_CONV_WAVES;   // 32-wide column tiles per wave
  float* s_a = smem;                                  // dY^T: [pos][co]
  float* s_patch = smem + PCP * 33;
  int* s_poff = reinterpret_cast<int*>(smem + PCP * 33 + SZ::CMAX * PLANE);
  float* s_bias = smem + PCP * 33 + SZ::CMAX * PLANE + PCP;

  const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
  constexpr int SBW = 24 + (G::KS == 8 || (G::KS == 5 && G::IH == 84) ? 0 : G::KS == 4 || G::KS == 5 ? 8 : 16);   // RB_STAMP slots
  const bool stamp_me = chunk == 0 && cotile == 0 && grp == 0 && t == 0;
  (void)stamp_me;
#if defined(RB_STAMP)
  if (stamp_me) g_cstamp[SBW + 0] = wall_clock64();
#endif
  const int wgi = (int)blockIdx.x;
  (void)wgi;
  RB_WGT(5, wgi, 0);
  RB_WGT_HW(5, wgi);
  const int co0 = cotile * 32;
  const int cin = a.cin, K = cin * G::KK;
  const int oy0 = chunk * RC;
  const int p0 = oy0 * G::OH;
  int npos = G::P - p0;
  if (npos > PC) npos = PC;
  const int iy0 = oy0 * G::S;
  int rows = G::IH - iy0;
  if (rows > PR) rows = PR;
  const int ntiles = (K + 31) / 32;
  const int kh = lane >> 5, nl = lane & 31;

  for (int p = t; p < PCP; p += RB_CONV_THREADS) {
    const int pc = p < npos ? p : npos - 1;
    s_poff[p] = (pc / G::OH) * G::S * G::IH + (pc % G::OH) * G::S;
  }
  rb_f32x16 acc[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[j][q] = 0.0f;
  float bias_acc = 0.0f;

  for (int ii = 0; ii < ipb; ++ii) {
    const int img = grp * ipb + ii;
    if (img >= batch) break;                          // block-uniform
    if (ii > 0) __syncthreads();                      // the previous image's operands are no longer being read
    // ---- stage dY^T (zero beyond the chunk) and the input patch of this image.  Every global load of the dY tile is
    // unconditional (clamped address) and issued before the first LDS store: a loop of test-load-store made each of its
    // 4-9 iterations a memory round trip.
    {
      constexpr int NIT = (32 * PCP + RB_CONV_THREADS - 1) / RB_CONV_THREADS;
      const int rows_valid = a.cout - co0 < 32 ? a.cout - co0 : 32;
      float v[NIT];
      int off[NIT];
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const int e = t + i * RB_CONV_THREADS;
        int m = e / PCP;
        int p = e - m * PCP;                              // p fastest: coalesced along positions
        if (m > rows_valid - 1) m = rows_valid - 1;
        if (p > npos - 1) p = npos - 1;
        off[i] = (img * a.cout + co0 + m) * G::P + p0 + p;    // 32-bit: B * cout * P floats < 2^31
      }
      bool lazy = false;
      if constexpr (!FIRST) lazy = a.dy_splits > 0;       // block-uniform
      if (lazy) {
        float mv[NIT], pv[NIT][4];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
          mv[i] = a.dy_mask[off[i]];
#pragma unroll
          for (int sp = 0; sp < 4; ++sp) pv[i][sp] = a.dy_part[(int64_t)(sp < a.dy_splits ? sp : a.dy_splits - 1) * a.dy_stride + off[i]];
        }
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
          float acc = 0.0f;                               // k_dfeat_finish's order
#pragma unroll
          for (int sp = 0; sp < 4; ++sp) acc += sp < a.dy_splits ? pv[i][sp] : 0.0f;
          v[i] = mv[i] > 0.0f ? acc : 0.0f;
        }
      } else {
#pragma unroll
        for (int i = 0; i < NIT; ++i) v[i] = a.dy[off[i]];
      }
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const int e = t + i * RB_CONV_THREADS;
        const int m = e / PCP, p = e - m * PCP;
        if (e < 32 * PCP) s_a[p * 33 + m] = (p < npos && m < rows_valid) ? v[i] : 0.0f;
      }
    }
    if (FIRST) {
      const int per_c = rows * G::IH;
      const int v16 = per_c >> 4;
      const int total16 = cin * v16;
      for (int e0 = 0; e0 < total16; e0 += 2 * RB_CONV_THREADS) {        // both 16-byte loads of a thread are in flight together
        uint4 raw[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int e = e0 + i * RB_CONV_THREADS + t;
          raw[i] = make_uint4(0u, 0u, 0u, 0u);
          if (e < total16) {
            const int c = e / v16, q = e - c * v16;
            const uint8_t* fp = rb_frame_ptr(a.src, img, c, cin, G::IP);
            if (fp) raw[i] = *reinterpret_cast<const uint4*>(fp + iy0 * G::IH + q * 16);
          }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int e = e0 + i * RB_CONV_THREADS + t;
          if (e < total16) {
            const int c = e / v16, q = e - c * v16;
            float* d = s_patch + c * PLANE + q * 16;
            const unsigned wds[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
#pragma unroll
            for (int wd = 0; wd < 4; ++wd)
#pragma unroll
              for (int b = 0; b < 4; ++b) d[wd * 4 + b] = rb_unit((uint8_t)((wds[wd] >> (8 * b)) & 0xFFu));
          }
        }
      }
      const int tail = per_c & 15;
      for (int e = t; e < cin * tail; e += RB_CONV_THREADS) {
        const int c = e / tail, q = (v16 << 4) + e % tail;
        const uint8_t* fp = rb_frame_ptr(a.src, img, c, cin, G::IP);
        s_patch[c * PLANE + q] = fp ? rb_unit(fp[iy0 * G::IH + q]) : 0.0f;
      }
    } else {
      const float* base = a.x_f + (int64_t)img * cin * G::IP;
      const int per_c = rows * G::IH;
      if (PR == G::IH && ((cin * G::IP) & 3) == 0) {
        // the chunk is the whole image (later layers): the patch is one contiguous copy, all of it in flight at once
        constexpr int NV = (SZ::CMAX * PLANE / 4 + RB_CONV_THREADS - 1) / RB_CONV_THREADS;
        const int total4 = (cin * G::IP) >> 2;
        float4 v[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const int e = t + i * RB_CONV_THREADS;
          v[i] = rb_ld4(base + 4 * (e < total4 ? e : total4 - 1));
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const int e = t + i * RB_CONV_THREADS;
          if (e < total4) { float* d = s_patch + 4 * e; d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w; }
        }
      } else {
        for (int e = t; e < cin * per_c; e += RB_CONV_THREADS) {
          const int c = e / per_c, q = e - c * per_c;
          s_patch[c * PLANE + q] = base[(int64_t)c * G::IP + iy0 * G::IH + q];
        }
      }
    }
    __syncthreads();
#if defined(RB_STAMP)
    if (stamp_me && ii == 0) g_cstamp[SBW + 1] = wall_clock64();
#endif
    if (ii == 0) { RB_WGT(5, wgi, 1); RB_WGT(5, wgi, 2); RB_WGT(5, wgi, 3); }

    // bias column: sum over the chunk's positions in a fixed order (then over the images, ascending).  Eight lanes per
    // channel take every eighth position and meet through shuffles (lane = 8 * channel-in-wave + part): one thread per
    // channel walking up to 140 dependent LDS reads (~3.7 us) made wave 0 the last wave of every workgroup to finish.
    {
      const int part = lane & 7, ch = (wave << 3) + (lane >> 3);          // 8 waves x 8 channels = 64 slots >= 32 channels
      float sum = 0.0f;
      if (ch < 32)
        for (int p = part; p < npos; p += 8) sum += s_a[p * 33 + ch];
      sum += __shfl_xor(sum, 1, 64);
      sum += __shfl_xor(sum, 2, 64);
      sum += __shfl_xor(sum, 4, 64);
      if (part == 0 && ch < 32 && co0 + ch < a.cout) s_bias[ch] = sum;
    }
    int pofs[PCP / 2];                                  // position offsets of the whole chunk, read once (not per step)
#pragma unroll
    for (int j = 0; j < PCP / 2; ++j) pofs[j] = s_poff[2 * j + kh];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      const int tile = wave + j * RB_CONV_WAVES;        // wave-uniform
      if (tile < ntiles) {
        int col = tile * 32 + nl;
        if (col > K - 1) col = K - 1;
        const int c = col / G::KK, r = col % G::KK;
        const int koff = c * PLANE + (r / G::KS) * G::IH + (r % G::KS);
#pragma unroll
        for (int jj = 0; jj < PCP / 2; ++jj)
          acc[j] = rb_mfma32(s_a[(2 * jj + kh) * 33 + nl], s_patch[koff + pofs[jj]], acc[j]);
      }
    }
    __syncthreads();                                    // s_bias of this image is complete (and its operands are done with)
    if (t < 32 && co0 + t < a.cout) bias_acc += s_bias[t];
#if defined(RB_STAMP)
    if (stamp_me && ii == 0) g_cstamp[SBW + 2] = wall_clock64();
#endif
    if (ii == 0) RB_WGT(5, wgi, 4);
  }

  float* out = a.part + (((int64_t)grp * nchunks + chunk) * a.cout) * (K + 1);
  if (t < 32 && co0 + t < a.cout) out[(int64_t)(co0 + t) * (K + 1) + K] = bias_acc;
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int tile = wave + j * RB_CONV_WAVES;
    if (tile < ntiles) {
      const bool cv = tile * 32 + nl < K;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int m = co0 + rb_mfma_row(q, lane);
        if (cv && m < a.cout) out[(int64_t)m * (K + 1) + tile * 32 + nl] = acc[j][q];
      }
    }
  }
#if defined(RB_STAMP)
  if (stamp_me) g_cstamp[SBW + 3] = wall_clock64();
#endif
  RB_WGT(5, wgi, 5);
  RB_WGT(5, wgi, 6);
}

// Every conv layer's weight gradient in ONE launch (they only feed the optimiser and are independent of each other
// once all dact[] exist): block ranges [0,n0) layer 0, [n0,n0+n1) layer 1, ...  Saves two kernel boundaries and fills
// the chip with 288 workgroups instead of 160 + 64 + 64 in sequence.
struct ConvDwAllArgs {
  ConvLdsDwArgs layer[3];
  int nblocks[3];          // workgroups of each layer
  int cotiles[3];
  int batch;
  int ipb[3];              // images summed per workgroup, per layer (the layers' workgroups cost differently: learner.hip conv_dw_all)
  int img_fast;            // decode with the image group as the FASTEST index (see k_conv_fwd_lds): needs block ranges and group
                           // counts that are multiples of 8
};
template <class G0, int RC0, class G1, int RC1, int K1, class G2, int RC2, int K2, int NL>
__global__ __launch_bounds__(RB_CONV_THREADS) void k_conv_dw_all(ConvDwAllArgs a) {
  typedef ConvDwLdsSize<G0, RC0, 4 * G0::KK> S0;
  typedef ConvDwLdsSize<G1, RC1, K1> S1;
  typedef ConvDwLdsSize<G2, RC2, K2> S2;
  constexpr int M01 = S0::FLOATS > S1::FLOATS ? S0::FLOATS : S1::FLOATS;
  constexpr int MAXF = (NL > 2 && S2::FLOATS > M01) ? S2::FLOATS : M01;
  __shared__ __attribute__((aligned(16))) float smem[MAXF];
  int b = (int)blockIdx.x;
  if (b < a.nblocks[0]) {                              // decode: chunk fastest, then cout tile, then image
    constexpr int CH = (G0::OH + RC0 - 1) / RC0;
    if (a.img_fast) {
      const int ng = (a.batch + a.ipb[0] - 1) / a.ipb[0], rest = b / ng;
      rb_conv_dw_body<G0, RC0, 4 * G0::KK, true>(a.layer[0], rest % CH, rest / CH, b % ng, CH, a.ipb[0], a.batch, smem);
    } else
    rb_conv_dw_body<G0, RC0, 4 * G0::KK, true>(a.layer[0], b % CH, (b / CH) % a.cotiles[0], b / (CH * a.cotiles[0]), CH, a.ipb[0], a.batch, smem);
    return;
  }
  b -= a.nblocks[0];
  if (b < a.nblocks[1]) {
    constexpr int CH = (G1::OH + RC1 - 1) / RC1;
    if (a.img_fast) {
      const int ng = (a.batch + a.ipb[1] - 1) / a.ipb[1], rest = b / ng;
      rb_conv_dw_body<G1, RC1, K1, false>(a.layer[1], rest % CH, rest / CH, b % ng, CH, a.ipb[1], a.batch, smem);
    } else
    rb_conv_dw_body<G1, RC1, K1, false>(a.layer[1], b % CH, (b / CH) % a.cotiles[1], b / (CH * a.cotiles[1]), CH, a.ipb[1], a.batch, smem);
    return;
  }
  if (NL > 2) {
    b -= a.nblocks[1];
    constexpr int CH = (G2::OH + RC2 - 1) / RC2;
    if (a.img_fast) {
      const int ng = (a.batch + a.ipb[2] - 1) / a.ipb[2], rest = b / ng;
      rb_conv_dw_body<G2, RC2, K2, false>(a.layer[2], rest % CH, rest / CH, b % ng, CH, a.ipb[2], a.batch, smem);
    } else
    rb_conv_dw_body<G2, RC2, K2, false>(a.layer[2], b % CH, (b / CH) % a.cotiles[2], b / (CH * a.cotiles[2]), CH, a.ipb[2], a.batch, smem);
  }
}
