// gemm_core.h — one LDS-tiled fp32 MFMA GEMM skeleton for every contraction of the learn
// step (conv fwd / data-grad / weight-grad as implicit GEMMs, the noisy FC layers).
//
// C[m][n] (+)= sum_k A(m,k) * B(k,n) on v_mfma_f32_32x32x2_f32 (exact f32, 64 FLOP/clk/SIMD).
// A "problem" functor supplies the operand gathers and the epilogue, so im2col addressing,
// u8 frame decoding, factorised-noise weights (mu + sigma*eps_out*eps_in, never
// materialised), ReLU masks and bias columns all fuse into the one kernel:
//
//   struct Prob {
//     static constexpr bool A_KFAST, B_KFAST;   // which index runs fastest across lanes while staging
//     __device__ bool  group(int g, GemmDims& d) const;   // dims + k-range of group g; false = nothing to do
//     __device__ float a(int g, int m, int k) const;       // m < M, k < K guaranteed
//     __device__ float b(int g, int k, int n) const;       // k < K, n < N guaranteed
//     __device__ void  store(int g, int m, int n, float v) const;
//   };
//
// Tile: (32*WM) x (32*WN) x 16, one wave per 32x32 sub-tile, 64*WM*WN threads.
// Staging is register double-buffered (global loads of tile t+1 are in flight while the
// MFMAs of tile t run) with two LDS buffers and ONE barrier per k-step.  LDS rows are
// [k][m] / [k][n] (+1 pad) so the per-lane MFMA operand reads (lane -> consecutive m / n)
// are bank-conflict free.  grid = (m tiles, n tiles, groups).
#pragma once
#include "rb_device.h"

struct GemmDims {
  int M, N, K;        // logical problem of this group
  int k_begin, k_end; // this block's slice of K (split-K)
};

template <int WM, int WN, class Prob>
__global__ __launch_bounds__(64 * WM * WN) void k_gemm(Prob p) {
  constexpr int BM = 32 * WM, BN = 32 * WN, BK = 16, T = 64 * WM * WN;
  constexpr int LDA = BM + 1, LDB = BN + 1;
  constexpr int A_PER = (BM * BK) / T, B_PER = (BK * BN) / T;
  static_assert((BM * BK) % T == 0 && (BK * BN) % T == 0, "tile/thread mismatch");
  __shared__ float As[2][BK * LDA];
  __shared__ float Bs[2][BK * LDB];

  const int g = (int)blockIdx.z;
  GemmDims d;
  if (!p.group(g, d)) return;                    // block-uniform
  const int m0 = (int)blockIdx.x * BM, n0 = (int)blockIdx.y * BN;
  if (m0 >= d.M || n0 >= d.N) return;            // block-uniform

  const int t = (int)threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;

  float ra[A_PER], rb[B_PER];
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int j = 0; j < A_PER; ++j) {
      const int e = t + j * T;
      const int kl = Prob::A_KFAST ? (e % BK) : (e / BM);
      const int ml = Prob::A_KFAST ? (e / BK) : (e % BM);
      const int m = m0 + ml, k = k0 + kl;
      ra[j] = (m < d.M && k < d.k_end) ? p.a(g, m, k) : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < B_PER; ++j) {
      const int e = t + j * T;
      const int kl = Prob::B_KFAST ? (e % BK) : (e / BN);
      const int nl = Prob::B_KFAST ? (e / BK) : (e % BN);
      const int n = n0 + nl, k = k0 + kl;
      rb[j] = (n < d.N && k < d.k_end) ? p.b(g, k, n) : 0.0f;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int j = 0; j < A_PER; ++j) {
      const int e = t + j * T;
      const int kl = Prob::A_KFAST ? (e % BK) : (e / BM);
      const int ml = Prob::A_KFAST ? (e / BK) : (e % BM);
      As[buf][kl * LDA + ml] = ra[j];
    }
#pragma unroll
    for (int j = 0; j < B_PER; ++j) {
      const int e = t + j * T;
      const int kl = Prob::B_KFAST ? (e % BK) : (e / BN);
      const int nl = Prob::B_KFAST ? (e / BK) : (e % BN);
      Bs[buf][kl * LDB + nl] = rb[j];
    }
  };

  rb_f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;

  const int nk = (d.k_end - d.k_begin + BK - 1) / BK;
  if (nk > 0) {
    load_tile(d.k_begin);
    store_tile(0);
  }
  __syncthreads();
  for (int it = 0; it < nk; ++it) {
    const int buf = it & 1;
    if (it + 1 < nk) load_tile(d.k_begin + (it + 1) * BK);
    const float* as = &As[buf][wm * 32 + (lane & 31)];
    const float* bs = &Bs[buf][wn * 32 + (lane & 31)];
    const int kh = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float av = as[(kk + kh) * LDA];
      const float bv = bs[(kk + kh) * LDB];
      acc = rb_mfma32(av, bv, acc);
    }
    if (it + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

  const int n = n0 + wn * 32 + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 32 + rb_mfma_row(r, lane);
    if (m < d.M && n < d.N) p.store(g, m, n, acc[r]);
  }
}

// split-K helper: slice [0,K) into `splits` runs of whole 16-wide k-steps
__host__ __device__ inline void rb_split_k(int K, int splits, int s, int* kb, int* ke) {
  const int steps = (K + 15) / 16;
  const int per = (steps + splits - 1) / splits;
  int b = s * per * 16, e = (s + 1) * per * 16;
  if (b > K) b = K;
  if (e > K) e = K;
  *kb = b;
  *ke = e;
}
