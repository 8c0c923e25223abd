// Agent.act / evaluate_q (agent.py:53-55, 110-112; main.py:153, test.py:26,39): the forward of ONE un-batched
// state.  The training kernels (conv_lds.h, noisy_linear.h) are shaped for 96 images: with one image they run 2-3
// workgroups per layer and cost the same 14 us as with 96 (a pure staging-latency chain).  This path trades MFMA for
// breadth instead: one output channel (conv) or one weight row per wave (noisy linear), hundreds of small
// workgroups, weights read as wave-uniform scalars / one coalesced sweep, so a layer is one memory round trip.
// Same arithmetic as model.py:42-46,69-80 (f32, fused multiply-add accumulation in a fixed order).
#pragma once
#include "noisy_linear.h"
#include "rb_device.h"

#define RB_ACT_LDS 14336          // floats of input patch per workgroup (56 KB)
#define RB_ACT_MAXPOS 128         // output positions per workgroup (two 64-lane passes)
#define RB_ACT_KMAX 1024          // taps of one output channel (cin * KS * KS)

struct ActConvArgs {
  const float* x;      // [cin][IH][IH]
  const float* w;      // [cout][cin*KS*KS]
  const float* bias;   // [cout]
  float* y;            // [cout][OH*OH]   ReLU applied
  int cin, cout, KS, S, IH, OH, RG;   // RG = output rows per workgroup
};

// grid = (cout, ceil(OH / RG)); block = 256: wave w owns the w-th quarter of the reduction, lanes are positions.
__global__ __launch_bounds__(256) void k_act_conv(ActConvArgs a) {
  __shared__ __attribute__((aligned(16))) float s_in[RB_ACT_LDS];
  __shared__ float s_red[4][RB_ACT_MAXPOS];
  __shared__ float s_w[RB_ACT_KMAX];                            // this channel's filter (a scalar load per tap would
  const int t = (int)threadIdx.x, lane = t & 63;                // serialise ~200 cycles of latency per tap: measured 15 us)
  const int wave = rb_wave_uniform(t >> 6);
  const int co = (int)blockIdx.x, oy0 = (int)blockIdx.y * a.RG;
  int rows = a.OH - oy0;
  if (rows > a.RG) rows = a.RG;
  const int npos = rows * a.OH;
  const int iy0 = oy0 * a.S;
  const int plane = ((rows - 1) * a.S + a.KS) * a.IH;          // staged floats per input channel (whole rows)
  const int total = a.cin * plane;
  const int chan = a.IH * a.IH;
  // Staging is ONE memory round trip: every 16-byte load of a thread is issued before the first LDS store (a
  // batch-of-8 dword version took seven dependent round trips: 13.5 us for the 51 KB second-layer input).
  const bool flat = plane == chan;                               // whole image: one contiguous run
  const bool vec = flat ? ((total & 3) == 0) : (((plane | chan | (iy0 * a.IH)) & 3) == 0);
  if (vec) {
    const int plane4 = plane >> 2, total4 = total >> 2;
    for (int e0 = 0; e0 < total4; e0 += 16 * 256) {
      float4 v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        int e = e0 + i * 256 + t;
        if (e > total4 - 1) e = total4 - 1;
        const int c = flat ? 0 : e / plane4;
        const int q = e - c * plane4;
        v[i] = rb_ld4(a.x + (int64_t)c * chan + iy0 * a.IH + 4 * q);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int e = e0 + i * 256 + t;
        if (e < total4) *reinterpret_cast<float4*>(&s_in[4 * e]) = v[i];
      }
    }
  } else {
    for (int e0 = 0; e0 < total; e0 += 8 * 256) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        int e = e0 + i * 256 + t;
        if (e > total - 1) e = total - 1;
        const int c = e / plane, q = e - c * plane;
        v[i] = a.x[(int64_t)c * chan + iy0 * a.IH + q];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int e = e0 + i * 256 + t;
        if (e < total) s_in[e] = v[i];
      }
    }
  }
  const int KK = a.KS * a.KS, K = a.cin * KK;
  for (int k = t; k < K; k += 256) s_w[k] = a.w[(int64_t)co * K + k];
  const int ks = (K + 3) >> 2;
  int k0 = wave * ks, k1 = k0 + ks;
  if (k0 > K) k0 = K;
  if (k1 > K) k1 = K;
  int poff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int p = lane + 64 * j;
    if (p > npos - 1) p = npos - 1;
    poff[j] = (p / a.OH) * a.S * a.IH + (p % a.OH) * a.S;
  }
  __syncthreads();
  float acc[2] = {0.0f, 0.0f};
  int c = k0 / KK, r = k0 - c * KK;
  int ky = r / a.KS, kx = r - ky * a.KS;
#pragma unroll 4
  for (int k = k0; k < k1; ++k) {                               // k, c, ky, kx are wave-uniform (scalar unit)
    const float wv = s_w[k];                                    // LDS broadcast
    const int koff = c * plane + ky * a.IH + kx;
    acc[0] = fmaf(wv, s_in[koff + poff[0]], acc[0]);
    acc[1] = fmaf(wv, s_in[koff + poff[1]], acc[1]);
    if (++kx == a.KS) { kx = 0; if (++ky == a.KS) { ky = 0; ++c; } }
  }
  s_red[wave][lane] = acc[0];
  s_red[wave][lane + 64] = acc[1];
  __syncthreads();
  if (t < npos) {
    const float v = ((s_red[0][t] + s_red[1][t]) + s_red[2][t]) + s_red[3][t] + a.bias[co];
    a.y[(int64_t)co * a.OH * a.OH + oy0 * a.OH + t] = fmaxf(v, 0.0f);
  }
}

struct ActFcArgs {
  const float* x;      // [K] per row group (x_off1 for rows >= split_row)
  NlWeights w;
  int K, n_rows;
  int split_row;       // first row of the second stream (fc_h: advantage stream, fc_z: advantage atoms)
  int x_off1, ein_off1;
  float* out;          // [n_rows]
  int relu;
  int mu_only;         // eval mode (model.py:46): W = mu, b = b_mu — sigma / eps are not even read
};

// grid = ceil(n_rows / 4); block = 256: one weight row per wave, lanes sweep K in float4s (K % 4 == 0).
__global__ __launch_bounds__(256) void k_act_fc(ActFcArgs a) {
  const int t = (int)threadIdx.x, lane = t & 63;
  const int wave = rb_wave_uniform(t >> 6);
  const int row = (int)blockIdx.x * 4 + wave;
  if (row >= a.n_rows) return;                                  // wave-uniform
  const bool g1 = row >= a.split_row;
  const float* mu = a.w.mu + (int64_t)row * a.K;
  const float* sg = a.w.sigma + (int64_t)row * a.K;
  const float* x = a.x + (g1 ? a.x_off1 : 0);
  const float* ein = a.w.ein + (g1 ? a.ein_off1 : 0);
  const float eo = a.mu_only ? 0.0f : a.w.eout[row];
  const int n4 = a.K >> 2;
  float acc = 0.0f;
  for (int b0 = 0; b0 < n4; b0 += 4 * 64) {                     // 4 float4 of every operand in flight per lane
    float4 m[4], s[4], xv[4], e[4];
    float live[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int i = b0 + u * 64 + lane;
      live[u] = i < n4 ? 1.0f : 0.0f;
      if (i > n4 - 1) i = n4 - 1;
      m[u] = rb_ld4(mu + 4 * i);
      xv[u] = rb_ld4(x + 4 * i);
      if (!a.mu_only) { s[u] = rb_ld4(sg + 4 * i); e[u] = rb_ld4(ein + 4 * i); }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float4 w4 = a.mu_only ? m[u] : rb_noisy4(m[u], s[u], eo, e[u]);
      float part = 0.0f;
      part = fmaf(w4.x, xv[u].x, part);
      part = fmaf(w4.y, xv[u].y, part);
      part = fmaf(w4.z, xv[u].z, part);
      part = fmaf(w4.w, xv[u].w, part);
      acc = fmaf(live[u], part, acc);
    }
  }
  acc = rb_wave_sum(acc);
  if (lane == 0) {
    float o = acc + (a.mu_only ? a.w.bmu[row] : (a.w.bmu[row] + a.w.bsigma[row] * eo));   // model.py:44,46
    if (a.relu) o = fmaxf(o, 0.0f);
    a.out[row] = o;
  }
}
