// Agent.act / evaluate_q (agent.py:53-55, 110-112; main.py:153, test.py:26,39): the forward of ONE un-batched
// state.  The training kernels (conv_lds.h, noisy_linear.h) are shaped for 96 images: with one image they run 2-3
// workgroups per layer and cost the same 14 us as with 96 (a pure staging-latency chain).  This path trades MFMA for
// breadth instead: one output channel (conv) or one weight row per wave (noisy linear), hundreds of small
// workgroups, weights read as wave-uniform scalars / one coalesced sweep, so a layer is one memory round trip.
// Same arithmetic as model.py:42-46,69-80 (f32, fused multiply-add accumulation in a fixed order).
#pragma once
#include "noisy_linear.h"
#include "rb_device.h"

#define RB_ACT_LDS 15360          // floats of input patch per workgroup (60 KB)
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

// q = v + a - mean_a(a)          model.py:75
__device__ __forceinline__ float rb_dueling_q(const float* lg, const float* mean_a, int Z, int a, int z) {
  return (lg[z] + lg[Z + a * Z + z]) - mean_a[z];
}
// Agent.act / evaluate_q head (agent.py:53-55, 110-112) for ONE image: `lg` = its logits row (global or LDS), s_mean [Z]
// and s_ev [A] workgroup scratch.  All threads of a 256-thread workgroup call.  err (optional): the word an expired in-launch
// wait of launch number `err_epoch` sets to that number -> action -1 for THAT launch only (a later launch compares with its own
// number: the failure state needs no reset and cannot stick).
__device__ __forceinline__ void rb_head_act_body(int Z, int A, const float* lg, const float* support, float* s_mean, float* s_ev,
                                                 int32_t* action_out, float* q_out, const unsigned* err, unsigned err_epoch = 0u) {
  const int t = (int)threadIdx.x;
  const int lane = rb_lane(), wave = rb_wave(), nw = (int)(blockDim.x >> 6);
  for (int z = t; z < Z; z += (int)blockDim.x) {
    float acc = 0.0f;
    for (int a = 0; a < A; ++a) acc += lg[Z + a * Z + z];
    s_mean[z] = acc / (float)A;
  }
  __syncthreads();
  for (int a = wave; a < A; a += nw) {
    float mx = -INFINITY;
    for (int z = lane; z < Z; z += 64) mx = fmaxf(mx, rb_dueling_q(lg, s_mean, Z, a, z));
    mx = rb_wave_max(mx);
    float se = 0.0f, sv = 0.0f;
    for (int z = lane; z < Z; z += 64) {
      const float e = expf(rb_dueling_q(lg, s_mean, Z, a, z) - mx);
      se += e;
      sv += support[z] * e;
    }
    se = rb_wave_sum(se);
    sv = rb_wave_sum(sv);
    if (lane == 0) s_ev[a] = sv / se;
  }
  __syncthreads();
  if (t == 0) {
    int best = 0;
    float bv = s_ev[0];
    for (int a = 1; a < A; ++a)
      if (s_ev[a] > bv) { bv = s_ev[a]; best = a; }
#if defined(RB_HOST_INTERP)
    if (err && err_epoch != 0u && *err == err_epoch) best = -1;
#else
    if (err && err_epoch != 0u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == err_epoch) best = -1;   // a bounded in-launch wait of THIS launch expired: no action
#endif
#if !defined(RB_HOST_INTERP) && !defined(RB_ACT_NO_PACK)      // (RB_ACT_NO_PACK: variant build for A/B runs)
    // (action, q) as ONE aligned 8-byte word where the caller laid them out that way (rainbow_amd/agent.py _forward_single: a pinned
    // pair): a single system-scope store — both are there when the host sees the action change, and the launch ends one host-memory
    // round trip earlier than with q, a system-scope fence, then the action (round 6: ~1 us of act()'s 39)
    if (action_out && q_out && reinterpret_cast<const void*>(q_out) == reinterpret_cast<const void*>(action_out + 1) &&
        (reinterpret_cast<uintptr_t>(action_out) & 7u) == 0) {
      const unsigned long long pack = (unsigned long long)(unsigned)best | ((unsigned long long)__builtin_bit_cast(unsigned, bv) << 32);
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(action_out), pack, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
#endif
    // q BEFORE the action, a system-scope fence in between: a host that polls the (pinned) action word for the value it
    // preset to change may read q right after
    if (q_out) *q_out = bv;
#if !defined(RB_HOST_INTERP)
    __threadfence_system();
#endif
    if (action_out) *action_out = best;
  }
}

// ============================================================================ one launch ==
// The six launches above (conv x L, fc_h, fc_z, head: 59 us for 22 MFLOP and 27 MB, half of main.py's loop period once a
// learn step is 170 us) as ONE persistent launch.  G workgroups (one per CU) walk the phases together; a phase boundary is
// an all-to-all hand-off: every workgroup stores its outputs write-through, arrives on the boundary's counter, polls it and
// reads the previous layer with agent-coherent loads (rb_device.h rb_chain_*: the form validated for in-launch hand-offs).
// What does NOT depend on the previous layer is requested before the wait: the hidden layer's 25.7 MB of mu | sigma — the
// bulk of the forward's bytes — stream into REGISTERS (one weight row per wave) while the conv phases run, so that phase is
// one small activation load + FMAs.  Same arithmetic and summation order as the per-layer kernels above (bit-identical).
// Counters are monotonic (target = launch number x G): no reset between launches.  Every spin is bounded; on expiry the
// error word is set and the head writes action -1.
// phase_lo / phase_hi: the launch runs phases [lo, hi) of {conv 0, 1, 2, fc_h, fc_z, head}.  One launch with all of them is
// the product path; single-phase launches (no in-launch dependency) are what the host interpreter runs, and the fallback.
struct ActFusedArgs {
  ActConvArgs conv[3];
  int nconv;
  ActFcArgs h, z;
  int Z, A;
  const float* logits;     // = z.out
  const float* support;
  int32_t* action_out;
  float* q_out;
  unsigned* ctr;           // [6][8 shards x 32 words] arrival counters, one set per phase (rb_device.h rb_fan_*)
  unsigned epoch;
  unsigned* err;
  int phase_lo, phase_hi;
};

// one (output channel, row group) of a conv layer: k_act_conv's body with coherent input loads and write-through stores
__device__ __forceinline__ void rb_act_conv_unit(const ActConvArgs& a, int co, int oy0, float* s_in, float (*s_red)[RB_ACT_MAXPOS], float* s_w) {
  const int t = (int)threadIdx.x, lane = t & 63;
  const int wave = rb_wave_uniform(t >> 6);
  int rows = a.OH - oy0;
  if (rows > a.RG) rows = a.RG;
  const int npos = rows * a.OH;
  const int iy0 = oy0 * a.S;
  const int plane = ((rows - 1) * a.S + a.KS) * a.IH;
  const int total = a.cin * plane;
  const int chan = a.IH * a.IH;
  const rb_buf bx = rb_make_buf(a.x);
  const bool flat = plane == chan;
  const bool vec = flat ? ((total & 3) == 0) : (((plane | chan | (iy0 * a.IH)) & 3) == 0);
  const int KK = a.KS * a.KS, K = a.cin * KK;
  float wreg[RB_ACT_KMAX / 256];                                  // this thread's taps of the filter: requested with the input
#pragma unroll
  for (int i = 0; i < RB_ACT_KMAX / 256; ++i) {
    const int k = t + 256 * i;
    wreg[i] = a.w[(int64_t)co * K + (k < K ? k : K - 1)];
  }
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
        v[i] = rb_ld4_buf_sc1(bx, 4u * (unsigned)(c * chan + iy0 * a.IH + 4 * q), 0u);
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
        v[i] = rb_ld1_buf_sc1(bx, 4u * (unsigned)(c * chan + iy0 * a.IH + q), 0u);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int e = e0 + i * 256 + t;
        if (e < total) s_in[e] = v[i];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < RB_ACT_KMAX / 256; ++i) {
    const int k = t + 256 * i;
    if (k < K) s_w[k] = wreg[i];
  }
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
  for (int k = k0; k < k1; ++k) {
    const float wv = s_w[k];
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
    rb_st1_wt(a.y, 4u * (unsigned)(co * a.OH * a.OH + oy0 * a.OH + t), fmaxf(v, 0.0f));
  }
  __syncthreads();                                               // s_in / s_red / s_w are reused by the next unit or phase
}

// One 16-position x 16-channel output tile of a conv layer on v_mfma_f32_16x16x4_f32, for the canonical geometries (compile-time
// KS / S / IH / OH / CIN).  The scalar unit above is an LDS-latency chain — per tap one broadcast read, two operand reads and
// two FMAs on ONE wave per SIMD: ~10 us per unit (tools/stamp/act_timeline.py), three quarters of the one-launch path.  Here
// the four waves split K (wave w, k-slot kq: the contiguous sixteenth [(4 w + kq) K / 16, +K / 16) — channel- or row-aligned for
// these geometries, so a step's patch offset is lane base + compile-time constant), 16-36 MFMAs per wave, one LDS sum in
// wave order, bias + ReLU, write-through store.  Stages only the input rows its 16 positions touch.
template <int KS, int S, int IH, int OH, int CIN>
__device__ __forceinline__ void rb_act_conv_tile(const ActConvArgs& a, int pt, int ct, float* smem) {
  constexpr int KK = KS * KS, K = CIN * KK, WS = K + 4, P = OH * OH, KL = K / 16;     // KL = steps per (wave, k-slot)
  static_assert(K % 64 == 0, "whole float4s per k-slot");
  static_assert(((KL % KK) == 0) || ((KK % KL) == 0 && (KL % KS) == 0), "k-slot ranges are channel- or kernel-row-aligned");
  constexpr int RSPAN = (15 + OH - 1) / OH + 1;                 // output rows 16 consecutive positions can touch
  constexpr int NR = (RSPAN - 1) * S + KS;                      // input rows staged (clipped to the image)
  constexpr int PLANE = NR * IH;
  static_assert(16 * WS + CIN * PLANE + 4 * 4 * 64 <= RB_ACT_LDS, "tile operands fit the act path's LDS");
  float* s_wt = smem;                                           // [16][WS]
  float* s_p = smem + 16 * WS;                                  // [CIN][NR][IH]
  float* s_rd = smem + 16 * WS + CIN * PLANE;                   // [4 waves][4][64]
  const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
  const int p0 = pt * 16, co0 = ct * 16;
  const int oy_a = p0 / OH;
  int iy_a = oy_a * S;
  if (iy_a + NR > IH) iy_a = IH - NR;                           // keep the staged window inside the image (NR <= IH)
  const rb_buf bx = rb_make_buf(a.x);
  // ---- every global load of both operands is issued before the first LDS store
  constexpr int WQ = (16 * (K / 4) + 255) / 256;
  float4 wv[WQ];
#pragma unroll
  for (int i = 0; i < WQ; ++i) {
    int e = t + 256 * i;
    if (e > 16 * (K / 4) - 1) e = 16 * (K / 4) - 1;
    const int m = e / (K / 4), q = e - m * (K / 4);
    wv[i] = rb_ld4(a.w + (int64_t)(co0 + m) * K + 4 * q);
  }
  constexpr bool VEC = (IH % 4) == 0;
  constexpr int XN = VEC ? (CIN * PLANE / 4 + 255) / 256 : (CIN * PLANE + 255) / 256;
  float4 xv[VEC ? XN : 1];
  float xs[VEC ? 1 : XN];
#pragma unroll
  for (int i = 0; i < XN; ++i) {
    int e = t + 256 * i;
    if constexpr (VEC) {
      if (e > CIN * PLANE / 4 - 1) e = CIN * PLANE / 4 - 1;
      const int c = e / (PLANE / 4), q = e - c * (PLANE / 4);
      xv[i] = rb_ld4_buf_sc1(bx, 4u * (unsigned)(c * IH * IH + iy_a * IH + 4 * q), 0u);
    } else {
      if (e > CIN * PLANE - 1) e = CIN * PLANE - 1;
      const int c = e / PLANE, q = e - c * PLANE;
      xs[i] = rb_ld1_buf_sc1(bx, 4u * (unsigned)(c * IH * IH + iy_a * IH + q), 0u);
    }
  }
  const int x = lane & 15, kq = lane >> 4;
  float bias4[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bias4[r] = a.bias[co0 + 4 * kq + r];
#pragma unroll
  for (int i = 0; i < WQ; ++i) {
    const int e = t + 256 * i;
    if (e < 16 * (K / 4)) { const int m = e / (K / 4), q = e - m * (K / 4); rb_st4(s_wt + m * WS + 4 * q, wv[i]); }
  }
#pragma unroll
  for (int i = 0; i < XN; ++i) {
    const int e = t + 256 * i;
    if constexpr (VEC) { if (e < CIN * PLANE / 4) rb_st4(s_p + 4 * e, xv[i]); }
    else { if (e < CIN * PLANE) s_p[e] = xs[i]; }
  }
  __syncthreads();
  int p = p0 + x;
  const bool pv = p < P;
  if (p > P - 1) p = P - 1;
  const int k0 = (4 * wave + kq) * KL;                           // first k of this lane's range
  const int c0 = k0 / KK, r0 = k0 % KK;                          // (r0 is a multiple of KS, or 0: the static_assert above)
  const float* bp = s_p + c0 * PLANE + ((p / OH) * S - iy_a + r0 / KS) * IH + (p % OH) * S;
  const float* ap = s_wt + x * WS + k0;
  rb_f32x4 acc;
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = 0.0f;
#pragma unroll
  for (int jq = 0; jq < KL / 4; ++jq) {
    const float4 w4 = rb_ld4(ap + 4 * jq);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      constexpr int dummy = 0; (void)dummy;
      const int j = 4 * jq + s;                                  // compile-time after unrolling
      const int off = (j / KK) * PLANE + ((j % KK) / KS) * IH + (j % KK) % KS;
      const float wq = s == 0 ? w4.x : s == 1 ? w4.y : s == 2 ? w4.z : w4.w;
      acc = rb_mfma16(wq, bp[off], acc);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) s_rd[(wave * 4 + r) * 64 + lane] = acc[r];
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {                                // D[r]: channel 4 kq + r of the tile, position x
      const float v = ((s_rd[(0 * 4 + r) * 64 + lane] + s_rd[(1 * 4 + r) * 64 + lane]) + s_rd[(2 * 4 + r) * 64 + lane]) + s_rd[(3 * 4 + r) * 64 + lane];
      if (pv) rb_st1_wt(a.y, 4u * (unsigned)((co0 + 4 * kq + r) * P + p), fmaxf(v + bias4[r], 0.0f));
    }
  }
  __syncthreads();                                               // the LDS is reused by the next unit or phase
}
// the canonical geometries run on tiles; anything else (the data-efficient stack, other history lengths) on the scalar units
__device__ __forceinline__ int rb_act_conv_tiles(const ActConvArgs& c) {
  const bool ok = (c.KS == 8 && c.S == 4 && c.IH == 84 && c.OH == 20 && c.cin == 4) || (c.KS == 4 && c.S == 2 && c.IH == 20 && c.OH == 9 && c.cin == 32) ||
                  (c.KS == 3 && c.S == 1 && c.IH == 9 && c.OH == 7 && c.cin == 64);
  return (ok && c.cout % 16 == 0) ? ((c.OH * c.OH + 15) / 16) * (c.cout / 16) : 0;
}
__device__ __forceinline__ void rb_act_conv_tile_any(const ActConvArgs& c, int u, float* smem) {
  const int pts = (c.OH * c.OH + 15) / 16;
  if (c.KS == 8) rb_act_conv_tile<8, 4, 84, 20, 4>(c, u % pts, u / pts, smem);
  else if (c.KS == 4) rb_act_conv_tile<4, 2, 20, 9, 32>(c, u % pts, u / pts, smem);
  else rb_act_conv_tile<3, 1, 9, 7, 64>(c, u % pts, u / pts, smem);
}

// one weight row of a noisy layer (k_act_fc's wave body).  PQ > 0: the row's mu | sigma quads are already in `pm` / `ps`
// (requested before the previous phase's wait); the activations and eps_in are loaded here (coherent).
template <int PQ>
__device__ __forceinline__ void rb_act_fc_row(const ActFcArgs& a, int row, const float4* pm, const float4* ps) {
  const int lane = rb_lane();
  const bool g1 = row >= a.split_row;
  const float* mu = a.w.mu + (int64_t)row * a.K;
  const float* sg = a.w.sigma + (int64_t)row * a.K;
  const rb_buf bx = rb_make_buf(a.x + (g1 ? a.x_off1 : 0));
  const float* ein = a.w.ein + (g1 ? a.ein_off1 : 0);
  const float eo = a.mu_only ? 0.0f : a.w.eout[row];
  const int n4 = a.K >> 2;
  float acc = 0.0f;
  // the same 4-quad chunks and order as k_act_fc.  PQ > 0: the chunk count is a compile-time number (the host picked PQ with
  // K / 4 <= 64 PQ), so that the prefetched quads are indexed by constants and stay in registers
  auto chunk = [&](int b0, const float4* cm, const float4* cs, int have) {
    float4 m[4], s[4], xv[4], e[4];
    float live[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int i = b0 + u * 64 + lane;
      live[u] = i < n4 ? 1.0f : 0.0f;
      if (i > n4 - 1) i = n4 - 1;
      xv[u] = rb_ld4_buf_sc1(bx, 16u * (unsigned)i, 0u);
      if (u < have) { m[u] = cm[u]; s[u] = cs[u]; }
      else { m[u] = rb_ld4(mu + 4 * i); s[u] = a.mu_only ? m[u] : rb_ld4(sg + 4 * i); }
      e[u] = a.mu_only ? m[u] : rb_ld4(ein + 4 * i);
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
  };
  if constexpr (PQ > 0) {
#pragma unroll
    for (int ch = 0; ch < (PQ + 3) / 4; ++ch)
      if (ch * 256 < n4) chunk(ch * 256, pm + 4 * ch, ps + 4 * ch, PQ - 4 * ch < 4 ? PQ - 4 * ch : 4);
  } else {
    for (int b0 = 0; b0 < n4; b0 += 4 * 64) chunk(b0, nullptr, nullptr, 0);
  }
  acc = rb_wave_sum(acc);
  if (lane == 0) {
    float o = acc + (a.mu_only ? a.w.bmu[row] : (a.w.bmu[row] + a.w.bsigma[row] * eo));   // model.py:44,46
    if (a.relu) o = fmaxf(o, 0.0f);
    rb_st1_wt(a.out, 4u * (unsigned)row, o);
  }
}
template <int PQ>
__device__ __forceinline__ void rb_act_fc_prefetch(const ActFcArgs& a, int row, float4* pm, float4* ps) {
  const int lane = rb_lane();
  const int n4 = a.K >> 2;
  if (row >= a.n_rows) row = a.n_rows - 1;                       // (a wave without a row still issues legal loads)
#pragma unroll
  for (int j = 0; j < PQ; ++j) {
    int i = j * 64 + lane;                                       // chunk j / 4, u = j % 4: index b0 + u * 64 + lane
    if (i > n4 - 1) i = n4 - 1;
    pm[j] = rb_ld4(a.w.mu + (int64_t)row * a.K + 4 * i);
    ps[j] = a.mu_only ? pm[j] : rb_ld4(a.w.sigma + (int64_t)row * a.K + 4 * i);
  }
}

// HQ: quads of a hidden-layer weight row per lane kept in registers across the conv phases (ceil(F / 256): 13 canonical, 3
// data-efficient; 0 = no prefetch, any shape).  grid = G workgroups (all resident: G <= CUs), block = 256.
template <int HQ>
__global__ __launch_bounds__(256) void k_act_fused(ActFusedArgs a) {
  __shared__ __attribute__((aligned(16))) float s_in[RB_ACT_LDS];
  __shared__ float s_red[4][RB_ACT_MAXPOS];
  __shared__ float s_w[RB_ACT_KMAX];
  const int wg = (int)blockIdx.x, G = (int)gridDim.x;
  const int t = (int)threadIdx.x;
  const int wave = rb_wave_uniform(t >> 6);
  const bool fused = a.phase_hi - a.phase_lo > 1;
  float4 hm[HQ > 0 ? HQ : 1], hs[HQ > 0 ? HQ : 1];
  // the hidden layer's row of this wave, requested first thing by every workgroup (requesting it AFTER the first layer's unit
  // in the workgroups that have one measured slower: 35.5 against 33.3 us for the launch)
  if (HQ > 0 && fused && a.phase_lo <= 3 && a.phase_hi > 3) rb_act_fc_prefetch<HQ>(a.h, 4 * wg + wave, hm, hs);
  int prev = -1, prev_units = 0;
  RB_WGT(10, wg, 0);                                             // RB_STAMP builds: slot 0 start, slot 1 + phase = end of that phase
  RB_WGT(11, wg, 0);                                             // ... kernel id 11: slot 1 + phase = the phase's wait is over
  // units of work of a phase (workgroup wg takes units wg, wg + G, ...: it has work iff wg < units)
  auto units_of = [&](int phase) {
    if (phase < 3) {
      const ActConvArgs& c = a.conv[phase];
      const int tiles = rb_act_conv_tiles(c);
      return tiles > 0 ? tiles : c.cout * ((c.OH + c.RG - 1) / c.RG);
    }
    return phase == 3 ? (a.h.n_rows + 3) / 4 : phase == 4 ? (a.z.n_rows + 3) / 4 : 1;
  };
  for (int phase = a.phase_lo; phase < a.phase_hi; ++phase) {
    if (phase < 3 && phase >= a.nconv) continue;                 // (two conv layers: phase 2 does not exist)
    // Only the workgroups WITH work in a phase arrive at its boundary, and only those with work in the next one wait for it
    // (round 6: the conv phases have 50 / 24 / 16 units, the output layer 90 — with every one of the 256 workgroups arriving and
    // polling at every boundary, ~200 idle pollers sat on the counter lines the few working workgroups had to get their arrivals
    // through).  Transitive: a producer of boundary p waited for boundary p - 1 before it produced.
    const int units = units_of(phase);
    const bool has_work = wg < units;                            // block-uniform
#if defined(RB_ACT_ALL_ARRIVE)      // (variant build for A/B runs: every workgroup arrives and waits at every boundary)
    if (prev >= 0) rb_fan_wait(a.ctr + prev * (RB_FAN_SHARDS * RB_FAN_STRIDE), a.epoch * (unsigned)(G / RB_FAN_SHARDS), a.err, a.epoch);
#else
    if (prev >= 0 && has_work)
      rb_fan_wait_first(a.ctr + prev * (RB_FAN_SHARDS * RB_FAN_STRIDE), a.epoch, prev_units < G ? prev_units : G, a.err, a.epoch);
#endif
    RB_WGT(11, wg, 1 + phase);
    if (phase < 3) {
      const ActConvArgs& c = a.conv[phase];
      const int tiles = rb_act_conv_tiles(c);
      if (tiles > 0) {
        for (int u = wg; u < tiles; u += G) rb_act_conv_tile_any(c, u, s_in);
      } else {
        const int groups = (c.OH + c.RG - 1) / c.RG, units = c.cout * groups;
        for (int u = wg; u < units; u += G) rb_act_conv_unit(c, u % c.cout, (u / c.cout) * c.RG, s_in, s_red, s_w);
      }
    } else if (phase == 3) {
      const int units = (a.h.n_rows + 3) / 4;
      for (int u = wg; u < units; u += G) {
        const int row = 4 * u + wave;
        if (row < a.h.n_rows) {
          if (HQ > 0 && fused && u == wg) rb_act_fc_row<HQ>(a.h, row, hm, hs);
          else rb_act_fc_row<0>(a.h, row, nullptr, nullptr);
        }
      }
    } else if (phase == 4) {
      const int units = (a.z.n_rows + 3) / 4;
      for (int u = wg; u < units; u += G) {
        const int row = 4 * u + wave;
        if (row < a.z.n_rows) rb_act_fc_row<0>(a.z, row, nullptr, nullptr);
      }
    } else if (wg == 0) {
      // head (k_head_act's arithmetic) on the logits staged in LDS through coherent loads
      const int NZ = a.Z + a.A * a.Z;
      const rb_buf bl = rb_make_buf(a.logits);
      float* lg = s_in;                                            // NZ <= RB_HEAD_MAX_NZ <= RB_ACT_LDS
      for (int i = t; i < NZ; i += 256) lg[i] = rb_ld1_buf_sc1(bl, 4u * (unsigned)i, 0u);
      float* s_mean = s_w;                                         // Z <= 256 <= RB_ACT_KMAX
      float* s_ev = &s_red[0][0];                                  // A <= 64
      __syncthreads();
      rb_head_act_body(a.Z, a.A, lg, a.support, s_mean, s_ev, a.action_out, a.q_out, a.err, a.epoch);
    }
    RB_WGT(10, wg, 1 + phase);
#if defined(RB_ACT_ALL_ARRIVE)
    if (phase + 1 < a.phase_hi) rb_fan_signal(a.ctr + phase * (RB_FAN_SHARDS * RB_FAN_STRIDE), wg);
#else
    if (phase + 1 < a.phase_hi && has_work) rb_fan_signal(a.ctr + phase * (RB_FAN_SHARDS * RB_FAN_STRIDE), wg);
#endif
    prev = phase; prev_units = units;
  }
}
