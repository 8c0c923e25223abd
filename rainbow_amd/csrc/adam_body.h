// adam_body.h — clip_grad_norm_ + Adam (agent.py:97-98) as a device body that a foreign launch can host.
// k_clip_adam (learner.hip) runs it as a launch of its own; k_sample (replay.hip) can carry it as extra workgroups: the
// optimiser pass of learn call k has no data dependency on the sampler of call k + 1 (it reads the gradient, its norm
// partials and the moments; the sampler reads the sum-tree, whose write-back happened in call k's backward), so when two
// learn calls follow each other the 6.4 M-parameter streaming pass runs beside the sampler's serial, latency-bound
// workgroup instead of in front of it.  Same arithmetic, same summation order: the parameters are bit-identical.
#pragma once
#include "rb_common.h"

// clip_grad_norm_ + Adam in ONE pass over the flat buffers (agent.py:97-98).  Every block re-sums the partial list in the
// same fixed order (so all blocks agree on the clip coefficient) while its first parameter/gradient/moment loads are
// already in flight, then applies torch.optim.Adam's single-tensor update:
//   g' = g * clamp(max_norm / (norm + 1e-6), max=1)                       (clip_grad_norm_)
//   m  = lerp(m, g', 1-b1);  v = v*b2 + (1-b2)*g'*g'
//   p += -(lr / (1-b1^t)) * m / (sqrt(v) / sqrt(1-b2^t) + eps)
// The scaled gradient is stored back only when the clip actually bites (the reference leaves .grad scaled).
struct ClipAdamArgs {
  float* p; float* g; float* m; float* v;
  int64_t n;
  const float* part; int nparts;
  float max_norm; float* norm_out;
  float w1, b2, w2, neg_step_size, bc2_sqrt, eps;
  // hipGraph replay / hosted pass: the step number lives on the device (rb_learner_set_step_counter; incremented by the
  // head kernel of every learn call), and the bias corrections 1 - beta^t are formed here, in double like the host path,
  // by one thread per block — by-value scalars would freeze at capture time
  const long long* step_dev;
  double lr, beta1, beta2;
  // FUSED: elements [skip_lo, skip_lo + skip_len) (the hidden layer's mu | sigma weight arrays) are not touched by the
  // elementwise part: the tile part updates them from gradients it recomputes on the fly
  int64_t skip_lo4, skip_len4;        // in float4 units
  // non-NULL and non-zero on the device: the batch behind this gradient was not a legal one (the sampler gave up; its
  // importance weights are zero and so is the gradient) — the whole update is skipped instead of letting Adam's momentum
  // move the parameters on a step the reference would never have taken
  const int32_t* batch_status;
};
// (IEEE sqrt and divisions, as torch computes them: hardware rcp / approximate sqrt measured 1.5 us faster per launch
// and stay far inside the test tolerance, but the update would no longer be the reference's formula rounding for rounding)
__device__ __forceinline__ void rb_adam_elem(float& p, float& g, float& m, float& v, float coef, const ClipAdamArgs& a) {
  g = g * coef;
  m = fmaf(a.w1, g - m, m);
  v = v * a.b2 + a.w2 * g * g;
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
  p = p + a.neg_step_size * (m / denom);
}
__device__ __forceinline__ void rb_adam_quad(float4& P, float4& G, float4& M, float4& V, float coef, const ClipAdamArgs& a) {
  rb_adam_elem(P.x, G.x, M.x, V.x, coef, a);
  rb_adam_elem(P.y, G.y, M.y, V.y, coef, a);
  rb_adam_elem(P.z, G.z, M.z, V.z, coef, a);
  rb_adam_elem(P.w, G.w, M.w, V.w, coef, a);
}

// The plain pass (no skipped range) as hosted workgroups: block `eb` of `nblk`, any block size that is a multiple of 64.
// The arguments live in DEVICE memory (`ad`; the host rewrites them only when a pointer or a hyper-parameter changes):
// by value they would occupy ~40 SGPRs of the hosting kernel on every path.  Pointers that come out of memory are generic
// pointers — every access goes through a buffer descriptor instead (no flat instructions; 32-bit byte offsets: the caller
// guarantees 4 n < 2^31).  Requires a.step_dev (the step number cannot be a launch-time scalar here).
template <int UNROLL>
__device__ __forceinline__ void rb_adam_hosted_block(const ClipAdamArgs* ad, int eb, int nblk, float* s_red16 /* [18] */) {
  ClipAdamArgs a = *ad;
  const unsigned T = blockDim.x;
  const unsigned n4 = (unsigned)(a.n >> 2);
  const unsigned base = (unsigned)eb * (T * UNROLL) + threadIdx.x;
  const rb_buf bp = rb_make_buf(a.p), bg = rb_make_buf(a.g), bm = rb_make_buf(a.m), bv = rb_make_buf(a.v);
  // the step number first (one lane): its bias corrections — two double pow — are formed while the block's parameter loads
  // are in flight, not behind the norm's barrier
  unsigned st_lo = 0, st_hi = 0;
  if (threadIdx.x == 0) {
    const rb_buf bs = rb_make_buf(a.step_dev);
    st_lo = __builtin_bit_cast(unsigned, rb_ld1_buf(bs, 0, 0)); st_hi = __builtin_bit_cast(unsigned, rb_ld1_buf(bs, 4, 0));
  }
  float4 P[UNROLL], G[UNROLL], M[UNROLL], V[UNROLL];
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) {
    unsigned i = base + u * T;
    if (i >= n4) i = n4 > 0 ? n4 - 1 : 0;            // clamped load (always legal), masked store
    P[u] = rb_ld4_buf(bp, 16 * i, 0); G[u] = rb_ld4_buf(bg, 16 * i, 0);
    M[u] = rb_ld4_buf(bm, 16 * i, 0); V[u] = rb_ld4_buf(bv, 16 * i, 0);
  }
  if (a.batch_status && __builtin_bit_cast(int, rb_ld1_buf(rb_make_buf(a.batch_status), 0, 0)) != 0) {   // block-uniform
    if (eb == 0 && threadIdx.x == 0 && a.norm_out) rb_st1_wt(a.norm_out, 0, 0.0f);
    return;
  }
  float acc = 0.0f;
  {
    // 16 partials in flight per trip, added in index order (a loop of single loads is one L2 round trip per iteration)
    const rb_buf bpart = rb_make_buf(a.part);
    for (int i0 = (int)threadIdx.x; i0 < a.nparts; i0 += 16 * (int)T) {
      float pv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int i = i0 + u * (int)T;
        pv[u] = rb_ld1_buf(bpart, 4u * (unsigned)(i < a.nparts ? i : a.nparts - 1), 0);
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (i0 + u * (int)T < a.nparts) acc += pv[u];
    }
  }
  if (threadIdx.x == 0) {
    const double t = (double)(long long)(((unsigned long long)st_hi << 32) | st_lo);
    const double bc1 = 1.0 - pow(a.beta1, t), bc2 = 1.0 - pow(a.beta2, t);
    s_red16[16] = (float)(-(a.lr / bc1));
    s_red16[17] = (float)sqrt(bc2);
  }
  acc = rb_block_sum(acc, s_red16);
  const float total = sqrtf(acc);
  float coef = a.max_norm / (total + 1e-6f);
  if (coef > 1.0f) coef = 1.0f;                                    // clamp(max=1.0)
  if (eb == 0 && threadIdx.x == 0 && a.norm_out) rb_st1_wt(a.norm_out, 0, total);
  a.neg_step_size = s_red16[16];                                   // (written before rb_block_sum's barriers)
  a.bc2_sqrt = s_red16[17];
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) {
    const unsigned i = base + u * T;
    if (i >= n4) continue;
    rb_adam_quad(P[u], G[u], M[u], V[u], coef, a);
    rb_st4_wt(a.p, 16 * i, P[u]); rb_st4_wt(a.m, 16 * i, M[u]); rb_st4_wt(a.v, 16 * i, V[u]);
    if (coef < 1.0f) rb_st4_wt(a.g, 16 * i, G[u]);
  }
  // tail (n % 4 elements): last block's first threads
  if (eb == nblk - 1) {
    const int64_t t = ((a.n >> 2) << 2) + threadIdx.x;
    if (t < a.n) {
      const unsigned o = (unsigned)(4 * t);
      float p = rb_ld1_buf(bp, o, 0), g = rb_ld1_buf(bg, o, 0), m = rb_ld1_buf(bm, o, 0), v = rb_ld1_buf(bv, o, 0);
      rb_adam_elem(p, g, m, v, coef, a);
      rb_st1_wt(a.p, o, p); rb_st1_wt(a.m, o, m); rb_st1_wt(a.v, o, v);
      if (coef < 1.0f) rb_st1_wt(a.g, o, g);
    }
  }
}
