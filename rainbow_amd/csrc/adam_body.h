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
  // IMPLICIT sigma gradient (RB_LEARNER_IMPLICIT_SIGMA, hosted pass only): the hidden layer's sigma-weight gradient is
  // g_mu * (eps_out[n] * eps_in[k]) element for element (model.py:44's product rule; the backward forms exactly this), so the
  // backward does not store it and this pass does not load it: quads [pair_mu4, pair_mu4 + pair_len4) (mu) and the pair_len4
  // quads behind them (sigma) are updated TOGETHER by the pair workgroups [pair_blk0, nblk) — 7 loads instead of 8, and
  // 12.85 MB less written by the backward at the canonical shape.  eps come from a snapshot the backward took (the launch that
  // hosts this pass also resamples the noise).  The scaled gradients are stored back when the clip bites, sigma's included.
  int64_t pair_mu4, pair_len4;
  unsigned hole_lo4, hole4;    // the same range for the plain workgroups, which walk the quads OUTSIDE it: virtual quad j is quad j
                               // below hole_lo4 and quad j + hole4 from there on (hole4 = 0: no pairing)
  int pair_blk0;               // first pair workgroup (= number of plain workgroups)
  int pair_f4, pair_split_row; // quads per weight row; rows >= split_row take eps_in from the second stream's vector
  const float* pair_eout;      // [rows]
  const float* pair_ein;       // [2][4 * pair_f4]
  int32_t* pair_clipped;       // written by the pair pass: 1 = the clip bit and the SCALED gradients (sigma's included) were stored
                               // back, exactly as the reference leaves .grad — a later materialisation must not redo them from
                               // the scaled g_mu (the product would round in another order); 0 = nothing was stored
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
// everything between a hosted workgroup's first loads and its update: batch status, the norm from the partial list (fixed
// order), the bias corrections from the device step counter.  Returns false when the update is to be skipped.
__device__ __forceinline__ bool rb_adam_hosted_prologue(ClipAdamArgs& a, int eb, unsigned st_lo, unsigned st_hi, float* s_red16, float* coef_out) {
  const unsigned T = blockDim.x;
  if (a.batch_status && __builtin_bit_cast(int, rb_ld1_buf(rb_make_buf(a.batch_status), 0, 0)) != 0) {   // block-uniform
    if (eb == 0 && threadIdx.x == 0 && a.norm_out) rb_st1_wt(a.norm_out, 0, 0.0f);
    return false;
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
  *coef_out = coef;
  return true;
}

// pair workgroup `pb` (0-based among the pair workgroups): 2 (mu, sigma) quad pairs per thread = 14 data quads in flight.
// The noise products eps_out * eps_in of a thread's pairs are requested FIRST, parked in LDS (8 KB per 256-thread workgroup)
// while the data loads and the prologue are in flight, and read back for the update: as registers they pushed the hosting
// sampler kernel over its 128-register budget (5 spilled VGPRs, a scratch segment: +10 us per step for EVERY kernel), and with
// one pair per thread the pass streamed too thinly to gain anything (38.8 against 36.3 us for the hosting launch).
#define RB_ADAM_PAIR_T 256            // hosted workgroups are 256 threads (replay.hip sample_impl)
__device__ __forceinline__ void rb_adam_hosted_pairs(ClipAdamArgs& a, int eb, int pb, float* s_red16) {
  constexpr int PU = 2;
  __shared__ float4 s_prod[PU * RB_ADAM_PAIR_T];
  const unsigned T = blockDim.x;
  const unsigned len4 = (unsigned)a.pair_len4, mu4 = (unsigned)a.pair_mu4;
  const unsigned base = (unsigned)pb * (T * PU) + threadIdx.x;
  unsigned st_lo = 0, st_hi = 0;
  if (threadIdx.x == 0) {
    const rb_buf bs = rb_make_buf(a.step_dev);
    st_lo = __builtin_bit_cast(unsigned, rb_ld1_buf(bs, 0, 0)); st_hi = __builtin_bit_cast(unsigned, rb_ld1_buf(bs, 4, 0));
  }
  {
    const rb_buf beo = rb_make_buf(a.pair_eout), bei = rb_make_buf(a.pair_ein);
    float4 E[PU];
    float eo[PU];
#pragma unroll
    for (int u = 0; u < PU; ++u) {
      unsigned j = base + u * T;
      if (j >= len4) j = len4 - 1;
      const unsigned row = j / (unsigned)a.pair_f4, cq = j - row * (unsigned)a.pair_f4;
      eo[u] = rb_ld1_buf(beo, 4 * row, 0);
      E[u] = rb_ld4_buf(bei, 16 * (cq + ((int)row >= a.pair_split_row ? (unsigned)a.pair_f4 : 0u)), 0);
    }
#pragma unroll
    for (int u = 0; u < PU; ++u) {                       // eps_out * eps_in: the inner product of the backward's g_mu * (eo * e)
      float4 pr;
      pr.x = eo[u] * E[u].x; pr.y = eo[u] * E[u].y; pr.z = eo[u] * E[u].z; pr.w = eo[u] * E[u].w;
      s_prod[u * RB_ADAM_PAIR_T + threadIdx.x] = pr;
    }
  }
  const rb_buf bp = rb_make_buf(a.p), bg = rb_make_buf(a.g), bm = rb_make_buf(a.m), bv = rb_make_buf(a.v);
  float4 P[PU], G[PU], M[PU], V[PU], P2[PU], M2[PU], V2[PU];
#pragma unroll
  for (int u = 0; u < PU; ++u) {
    unsigned j = base + u * T;
    if (j >= len4) j = len4 - 1;                         // clamped loads (always legal), masked stores
    const unsigned i = mu4 + j, i2 = i + len4;
    P[u] = rb_ld4_buf(bp, 16 * i, 0); G[u] = rb_ld4_buf(bg, 16 * i, 0);
    M[u] = rb_ld4_buf(bm, 16 * i, 0); V[u] = rb_ld4_buf(bv, 16 * i, 0);
    P2[u] = rb_ld4_buf(bp, 16 * i2, 0); M2[u] = rb_ld4_buf(bm, 16 * i2, 0); V2[u] = rb_ld4_buf(bv, 16 * i2, 0);
  }
  float coef;
  if (!rb_adam_hosted_prologue(a, eb, st_lo, st_hi, s_red16, &coef)) return;
  if (pb == 0 && threadIdx.x == 0 && a.pair_clipped) rb_st1_wt(reinterpret_cast<float*>(a.pair_clipped), 0, __builtin_bit_cast(float, coef < 1.0f ? 1 : 0));
#pragma unroll
  for (int u = 0; u < PU; ++u) {
    const unsigned j = base + u * T;
    if (j >= len4) continue;
    const unsigned i = mu4 + j, i2 = i + len4;
    const float4 pr = s_prod[u * RB_ADAM_PAIR_T + threadIdx.x];   // (each thread reads back its own entries: no barrier needed)
    float4 G2;                                           // the backward's own expression: g_sigma = g_mu * (eps_out * eps_in)
    G2.x = G[u].x * pr.x; G2.y = G[u].y * pr.y; G2.z = G[u].z * pr.z; G2.w = G[u].w * pr.w;
    rb_adam_quad(P[u], G[u], M[u], V[u], coef, a);
    rb_adam_quad(P2[u], G2, M2[u], V2[u], coef, a);
    rb_st4_wt(a.p, 16 * i, P[u]); rb_st4_wt(a.m, 16 * i, M[u]); rb_st4_wt(a.v, 16 * i, V[u]);
    rb_st4_wt(a.p, 16 * i2, P2[u]); rb_st4_wt(a.m, 16 * i2, M2[u]); rb_st4_wt(a.v, 16 * i2, V2[u]);
    if (coef < 1.0f) { rb_st4_wt(a.g, 16 * i, G[u]); rb_st4_wt(a.g, 16 * i2, G2); }
  }
}

template <int UNROLL>
__device__ __forceinline__ void rb_adam_hosted_block(const ClipAdamArgs* ad, int eb, int nblk, float* s_red16 /* [18] */) {
  // (the branch is decided from two words; each role then reads the argument fields IT uses: the whole struct live on both
  // paths cost the hosting sampler kernel 18 spilled VGPRs and a scratch segment, and the whole step 10 us)
  const int nplain = ad->hole4 > 0 ? ad->pair_blk0 : nblk;
  if (eb >= nplain) {                                    // block-uniform: a (mu, sigma) pair workgroup
    ClipAdamArgs ap = *ad;
    rb_adam_hosted_pairs(ap, eb, eb - nplain, s_red16);
    return;
  }
  ClipAdamArgs a = *ad;
  const unsigned T = blockDim.x;
  const unsigned n4 = (unsigned)(a.n >> 2);
  const unsigned hole_lo = a.hole4 > 0 ? a.hole_lo4 : n4, hole = a.hole4;
  const unsigned nv4 = n4 - hole;
  auto real_of = [&](unsigned j) { return j < hole_lo ? j : j + hole; };
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
    unsigned j = base + u * T;
    if (j >= nv4) j = nv4 > 0 ? nv4 - 1 : 0;         // clamped load (always legal), masked store
    const unsigned i = real_of(j);
    P[u] = rb_ld4_buf(bp, 16 * i, 0); G[u] = rb_ld4_buf(bg, 16 * i, 0);
    M[u] = rb_ld4_buf(bm, 16 * i, 0); V[u] = rb_ld4_buf(bv, 16 * i, 0);
  }
  float coef;
  if (!rb_adam_hosted_prologue(a, eb, st_lo, st_hi, s_red16, &coef)) return;
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) {
    const unsigned j = base + u * T;
    if (j >= nv4) continue;
    const unsigned i = real_of(j);
    rb_adam_quad(P[u], G[u], M[u], V[u], coef, a);
    rb_st4_wt(a.p, 16 * i, P[u]); rb_st4_wt(a.m, 16 * i, M[u]); rb_st4_wt(a.v, 16 * i, V[u]);
    if (coef < 1.0f) rb_st4_wt(a.g, 16 * i, G[u]);
  }
  // tail (n % 4 elements): the last plain block's first threads
  if (eb == nplain - 1) {
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

// The pending pass as a launch of its own through the HOSTED body (replay.hip k_adam_pending; arguments in device memory): what
// runs a pass with (mu, sigma) pairing outside a sampler launch.  Returns a hipError_t as int.
int rb_launch_adam_pending(const ClipAdamArgs* args_dev, int blocks, void* stream);
