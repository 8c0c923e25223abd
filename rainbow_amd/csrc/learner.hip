// learner.hip — the Rainbow learn step on MI355X (gfx950): three forwards, double-Q select,
// C51 projection, importance-weighted cross-entropy and the full backward, as hand-written
// HIP over flat float32 parameter / gradient / noise buffers borrowed from the caller.
//
// Reference being replaced: model.py (NoisyLinear, DQN) and agent.py:61-98.  The reference
// issues ~1750 framework ops per learn(); here the step is a fixed chain of 14 launches after the sampler
// (DESIGN.md §3 has the table with what bounds each):
//   conv fwd x L (conv_lds.h: operands in LDS, 8 waves split K)
//   -> fc_h, fc_z forward (noisy_linear.h k_nl_fwd3: weights streamed once, whole K per block, no partials)
//   -> head (dueling + softmaxes + double-Q + projection + loss + dlogits, one workgroup per sample, atom bins in LDS)
//   -> fc_z backward (dW || dX in one launch) -> fc_h backward (dW || dX || the sum-tree priority write-back)
//   -> dfeat finish -> conv dX x (L-1) -> every conv dW in one launch -> slice reduction
//   -> clip + Adam in one pass (norm from per-producer partials; optional fused fc_h weight gradient).
// All contractions run on v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 (exact f32: parity with the float32 reference
// is the contract).  gemm_core.h + learner_problems.h remain the generic fallback for shapes the fast kernels refuse.
#include "conv_lds.h"
#include "noise_body.h"
#include "adam_body.h"
#include "learner_problems.h"
#include "noisy_linear.h"
#include "fc_gemm.h"
#include "act_path.h"
#include "rb_common.h"

#include <stdlib.h>

#include <math.h>
#include <string.h>

#include <new>

#if defined(RB_STAMP)
__device__ long long g_span[64];
extern "C" int rb_debug_spans(long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_span), sizeof(long long) * 64) != hipSuccess) return -2;
  if (reset) {
    long long init[64];
    for (int i = 0; i < 64; ++i) init[i] = (i & 1) ? 0 : 0x7fffffffffffffffLL;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_span), init, sizeof(init)) != hipSuccess) return -2;
  }
  return 0;
}
__device__ long long g_cstamp[64];
extern "C" int rb_debug_cstamps(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cstamp), sizeof(long long) * 64) == hipSuccess ? 0 : -2; }
__device__ long long g_wgt[RB_WGT_KERNELS][RB_WGT_WGS][8];
extern "C" int rb_debug_wgtrace(long long* out, int clear) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wgt), sizeof(long long) * RB_WGT_KERNELS * RB_WGT_WGS * 8) != hipSuccess) return -2;
  if (clear) { void* p = nullptr; if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_wgt)) != hipSuccess || hipMemset(p, 0, sizeof(long long) * RB_WGT_KERNELS * RB_WGT_WGS * 8) != hipSuccess) return -2; }
  return 0;
}
#endif
#define RB_HEAD_MAX_NZ 1408   // 3 logit rows of this many floats live in the head kernel's LDS (18 actions x 51 atoms = 969)
typedef ConvGeom<8, 4, 84, 20> GeomC1;   // model.py:56
typedef ConvGeom<4, 2, 20, 9> GeomC2;    // model.py:57
typedef ConvGeom<3, 1, 9, 7> GeomC3;     // model.py:58
typedef ConvGeom<5, 5, 84, 16> GeomD1;   // model.py:61
typedef ConvGeom<5, 5, 16, 3> GeomD2;    // model.py:62
// geometries of the LAST conv layer of their network (its output is the feature vector of the hidden layer)
#define RB_LAST_CONV_GEOM(G) (G::KS == 3 || (G::KS == 5 && G::IH == 16))

struct ConvLayer {
  int cin, cout, ks, s, ih, oh;
  int K() const { return cin * ks * ks; }
  int P() const { return oh * oh; }
  int IP() const { return ih * ih; }
};

struct Layout {
  int B, Z, A, H, F, NZ, hist, nconv;
  ConvLayer conv[3];
  // offsets (floats) inside the flat parameter buffer
  int64_t conv_w[3], conv_b[3];
  int64_t h_mu, h_sigma, h_bmu, h_bsigma, z_mu, z_sigma, z_bmu, z_bsigma;
  int64_t n_params;
  // offsets inside the flat noise buffer
  int64_t h_ein, h_eout, z_ein, z_eout, n_noise;
};

static int64_t align64(int64_t x) { return (x + 63) / 64 * 64; }

static int make_layout(const rb_learner_config_t* c, Layout* L) {
  RB_REQUIRE(c != nullptr, "learner config is NULL");
  RB_REQUIRE(c->batch >= 1 && c->batch <= 1024, "batch must be in [1,1024]");
  RB_REQUIRE(c->atoms >= 2 && c->atoms <= 256, "atoms must be in [2,256]");
  RB_REQUIRE(c->actions >= 1 && c->actions <= 64, "actions must be in [1,64]");
  RB_REQUIRE(c->atoms * (c->actions + 1) <= RB_HEAD_MAX_NZ, "atoms*(actions+1) must be <= %d (head kernel LDS rows)", RB_HEAD_MAX_NZ);
  RB_REQUIRE(c->history >= 1 && c->history <= 16, "history must be in [1,16]");
  RB_REQUIRE(c->hidden >= 1 && c->hidden <= 8192, "hidden must be in [1,8192]");
  RB_REQUIRE(c->architecture == 0 || c->architecture == 1, "architecture must be 0 (canonical) or 1 (data-efficient)");
  RB_REQUIRE(c->multi_step >= 1, "multi_step must be >= 1");
  RB_REQUIRE(c->v_max > c->v_min, "v_max must exceed v_min");
  memset(L, 0, sizeof(*L));
  L->B = c->batch; L->Z = c->atoms; L->A = c->actions; L->H = c->hidden; L->hist = c->history;
  L->NZ = L->Z + L->A * L->Z;
  if (c->architecture == 0) {
    L->nconv = 3;
    L->conv[0] = ConvLayer{c->history, 32, 8, 4, 84, 20};
    L->conv[1] = ConvLayer{32, 64, 4, 2, 20, 9};
    L->conv[2] = ConvLayer{64, 64, 3, 1, 9, 7};
    L->F = 3136;  // model.py:59
  } else {
    L->nconv = 2;
    L->conv[0] = ConvLayer{c->history, 32, 5, 5, 84, 16};
    L->conv[1] = ConvLayer{32, 64, 5, 5, 16, 3};
    L->F = 576;   // model.py:63
  }
  int64_t off = 0;
  for (int l = 0; l < L->nconv; ++l) {
    L->conv_w[l] = off; off = align64(off + (int64_t)L->conv[l].cout * L->conv[l].K());
    L->conv_b[l] = off; off = align64(off + L->conv[l].cout);
  }
  const int64_t H2 = 2 * L->H;
  L->h_mu = off; off = align64(off + H2 * L->F);
  L->h_sigma = off; off = align64(off + H2 * L->F);
  L->h_bmu = off; off = align64(off + H2);
  L->h_bsigma = off; off = align64(off + H2);
  L->z_mu = off; off = align64(off + (int64_t)L->NZ * L->H);
  L->z_sigma = off; off = align64(off + (int64_t)L->NZ * L->H);
  L->z_bmu = off; off = align64(off + L->NZ);
  L->z_bsigma = off; off = align64(off + L->NZ);
  L->n_params = off;
  int64_t n = 0;
  L->h_ein = n; n = align64(n + 2 * (int64_t)L->F);
  L->h_eout = n; n = align64(n + H2);
  L->z_ein = n; n = align64(n + H2);
  L->z_eout = n; n = align64(n + L->NZ);
  L->n_noise = n;
  return RB_OK;
}

static NetPtrs net_ptrs(const Layout& L, const float* params, const float* noise) {
  NetPtrs p;
  for (int l = 0; l < 3; ++l) {
    p.conv_w[l] = l < L.nconv ? params + L.conv_w[l] : nullptr;
    p.conv_b[l] = l < L.nconv ? params + L.conv_b[l] : nullptr;
  }
  p.h_mu = params + L.h_mu; p.h_sigma = params + L.h_sigma; p.h_bmu = params + L.h_bmu; p.h_bsigma = params + L.h_bsigma;
  p.z_mu = params + L.z_mu; p.z_sigma = params + L.z_sigma; p.z_bmu = params + L.z_bmu; p.z_bsigma = params + L.z_bsigma;
  p.h_ein = noise + L.h_ein; p.h_eout = noise + L.h_eout; p.z_ein = noise + L.z_ein; p.z_eout = noise + L.z_eout;
  return p;
}

// ---------------------------------------------------------------------- handle --
struct rb_learner {
  rb_learner_config_t cfg;
  Layout L;
  float *p_online, *p_target, *grads, *n_online, *n_target;   // borrowed
  uint64_t seed;
  uint64_t noise_epoch;
  // owned workspace
  float* act[3];        // [NI][cout][P]; act[nconv-1] doubles as feat [NI][F]
  float* dact[3];       // [B][cout][P]
  float* hpart;         // [hs][NI][2H]
  float* h;             // [NI][2H]
  float *feat_b, *h_b;  // k-blocked copies of feat [NI][F] and h [NI][2H] for the streamed forward kernels
  float* logits;        // [NI][NZ]
  float* dlogits;       // [B][NZ]
  float* dlogitsT;      // [NZ][B]: the same, transposed (the output layer's input gradient reads its dY operand from it)
  float* dh;            // [B][2H]
  float* dhT;           // [2H][B]: the same, transposed (the hidden layer's input gradient reads its dY operand from it)
  float* dfeat_part;    // [xs][B][F]
  int lazy_dfeat;       // this step: the last conv layer's backward kernels sum the partials themselves (no k_dfeat_finish)
  int lazy_splits;
  // test hooks read ONCE, when the handle is created (RB_OPTS, rb_opts below): they force the large-batch code paths and the
  // fallback block order onto small fixtures — conv_multi (-1 = by image count), conv_full, dx_ipb (0 = by batch), img_fast
  int opt_conv_multi, opt_conv_multi_t16, opt_conv_full, opt_dx_ipb, opt_dx_t16, opt_img_fast, opt_finish_tiled, opt_dw_ipb[3], opt_dw_balance, opt_wt_blocks;
  int opt_h_dw_deep;    // ... and its weight gradient with all four column tiles' operands in flight (rb_nl_dw_body_pipe_all)
  int opt_z_deep;       // the output layer's input gradient at batch > 32 with 8 row-steps of loads in flight (rb_nl_dx_body<4, 8>)
  int opt_wb_auto;      // the priority write-back inside the hidden layer's backward launch through rb_update_auto (sorted batch: one wave)
  int opt_h_deep;       // the hidden layer's input gradient at batch <= 32 with 8 row-steps of loads in flight (rb_nl_dx_body<2, 8>)
  int opt_z_ct, opt_h_ct;   // column tiles per wave of the pipelined weight-gradient body (output / hidden layer)
  int opt_z_narrow;     // ... on 32-column tiles (rb_nl_dx_body_tall<2>)
  int opt_z_tall;       // the output layer's input gradient with 16 waves per workgroup (noisy_linear.h rb_nl_dx_body_tall)
  int opt_t16;          // bit l: conv layer l's forward on the whole-K 16x16x4 kernel (conv_lds.h k_conv_fwd_t16) at small batches
  int opt_implicit_small;
  int opt_fc_gemm;      // hidden layer on the LDS-tiled GEMMs of fc_gemm.h: -1 = from 128 rows per net on (default), 1 = always, 0 = never
  float* gemm_part;     // split-K partial tiles of k_fc_gemm_fwd: one 64 KB tile per workgroup slot (n_cu of them)
  unsigned* gemm_ctr;   // its per-tile arrival counters (self-resetting)
  float* dw_part[3];    // [ws_l][cout][K+1]
  float* conv_wT[3];    // layers >= 1: the input-gradient kernels' weight operand [S*S phases][cin / 32 tiles][kpad][32], rewritten
                        // every step by tenant workgroups of the head launch (conv_lds.h rb_conv_wt_block); pad rows stay zero
  float* log_ps_a;      // [B][Z]
  float* pns_a;         // [B][Z]
  float* m;             // [B][Z]
  int32_t* a_star;      // [B]
  float* support;       // [Z]
  float* zero_noise;    // [n_noise] zeros (eval mode, model.py:46)
  float* norm_part;     // sum-of-squares partials: [0,1024) k_sumsq; fused producers use [0, norm_slots)
  int norm_conv_base;   // first slot of the conv reduction blocks
  int norm_slots;       // > 0: the last learn() left the gradient's sum of squares in norm_part (no k_sumsq pass needed)
  unsigned long long* noise_ctr;   // [0] Philox epoch of the noise generator, [1] block ticket
  NoiseJob* job_dev;               // [3] device copies of the noise jobs (rb_learner_noise_job), uploaded on request
  unsigned* act_ctr;    // arrival counters of the one-launch act path (act_path.h k_act_fused; monotonic, sharded) + its error word
  unsigned act_epoch;   // launches of k_act_fused so far
  int n_cu;             // compute units of the device (the one-launch act path runs one workgroup per CU)
  int opt_act_fused;    // RB_OPTS act_fused (default 1): Agent.act as ONE launch
  int rows_cap;         // image rows the forward buffers (act, hpart, h, feat_b, h_b, logits) hold: 3B, grown by act_batch
  int hs, xs, ws[3];    // split counts
  int dw_slices[3];     // slices actually written by the last conv weight-grad launch of each layer
  ImgSrc cur_src;       // input frames of the learn step in flight
  int sink_done;        // the last learn() performed the priority write-back itself
  const int32_t* batch_status;   // device word (the sink replay's header.last_status): non-zero = the sampler gave up on the
                                 // batch in flight; the optimiser update and its step number are then skipped (k_head, k_clip_adam)
  rb_replay_t* sink;    // priority sink: when set, learn() writes loss^w back into this replay's sum-tree itself
  const int64_t* sink_idx;
  int fast_fc;          // streamed 16x16x4 noisy-linear kernels usable (alignment preconditions hold)
  int fast_conv;        // LDS-resident conv kernels usable (history <= 4, standard channel counts)
  // replica exchange: world > 1 defers the noisy-linear WEIGHT gradients — instead of all-reducing 27 MB of
  // gradient, the replicas all-gather the two factors of every FC gradient (dY and X rows, 0.7 MB per rank) and each
  // computes the replica-mean gradient from the gathered rows itself (rb_learner_finish_grads)
  int world;
  float* fact_local;        // [fact_stride] this rank's factor block, written by the learn call
  const float* fact_all;    // [world][fact_stride] every rank's block (the all-gather's output)
  int64_t fact_off[6];      // dlogits [B][NZ] | h [B][2H] | dh [B][2H] | feat [B][F] | this rank's online noise [n_noise] |
                            // this rank's conv gradients (the leading h_mu floats of the flat gradient)
  int64_t fact_stride;
  int exch_pending;         // a learn call left its FC weight gradients to rb_learner_finish_grads
  long long* step_ctr;      // optional device-resident optimiser step counter (rb_learner_set_step_counter)
  int flags;                // RB_LEARNER_FUSE_FC_H_DW | RB_LEARNER_WRITE_FUSED_GRADS (rb_learner_set_flags)
  int dw_deferred;          // the last learn call computed the hidden layer's weight gradient for its norm only: the
                            // optimiser pass (rb_learner_clip_adam) recomputes the tiles while it streams the parameters
  // RB_LEARNER_DEFER_UPDATE: rb_learner_train_step leaves its optimiser pass PENDING; the next train_step's sampler launch
  // hosts it as extra workgroups (adam_body.h), every other entry point that touches parameters, moments, gradients or
  // the norm runs it first as a launch of its own (flush_update)
  ClipAdamArgs* adam_args_dev;   // the pending pass's arguments in device memory (rewritten only when they change)
  ClipAdamArgs adam_args_host;   // ... and what that memory holds
  int adam_args_valid, adam_pending, adam_blocks;
  // The early draw (RB_OPTS spec_draw=1, OFF by default; replay_internal.h rb_replay_spec_launch): from the second back-to-back
  // rb_learner_train_step on the same replay with nothing in between, the priority write-back leaves the hidden layer's backward
  // launch and runs — together with the NEXT call's draw — on the replay's own stream as soon as the head kernel is done; the next
  // call's sampler launch accepts the draw and carries only the noise and the pending optimiser pass.  Only an append-free,
  // constant-beta loop ever arms it (a PER benchmark; never main.py's loop), every wait is bounded at ~2 ms and fails safe, and the
  // first expiry disables it on the handle.  opt_spec_stall (RB_OPTS spec_stall=1, test hook): the launch behind the head kernel does
  // not store the go flag — the gate in front of the pair expires.
  // (A SPLIT optimiser pass — the (mu, sigma) pair workgroups on a second stream beside the sampler and the conv forward, the hidden
  // layer's forward waiting in-kernel for their arrival — was built in round 5, bit-identical, and measured 177 us per step against
  // 161.5: profiles/round5_split_experiments.txt; removed, the code is commit 645f60a.)
  int opt_spec_draw, opt_spec_stall;
  unsigned* go_flag;          // device word: epoch of the last head kernel known complete (stored by the launch behind it)
  unsigned go_epoch;
  int spec_now;               // this train_step: the write-back and the next draw go to the replay's stream
  rb_spec_request spec_req;
  struct { rb_replay_t* replay; int32_t batch, max_attempts; double beta; int64_t* tree_idx; int64_t* actions; float* returns; float* nonterm;
           float* weights; unsigned long long mut_after; int valid, streak; } ts_last;
  // RB_LEARNER_IMPLICIT_SIGMA: the hidden layer's sigma-weight gradient is not stored by the backward; the hosted optimiser
  // pass forms it from g_mu and the noise the backward used (adam_body.h rb_adam_hosted_pairs).  sigma_implicit = the flat
  // gradient lacks that range right now; every other consumer of the gradient materialises it first (materialize_sigma)
  int sigma_implicit;
  float* noise_snap;        // [n_noise] the online noise of the learn call in flight, copied by its last backward launch
  int32_t* status_copy;     // this learn call's batch_status, copied by its head kernel: the hosted pass shares a launch with
                            // the NEXT call's sampler, which overwrites the replay header's word
  float gamma_n;        // float32(discount ** n)        agent.py:79
  float delta_z;        // float32((Vmax - Vmin)/(Z-1))   agent.py:19,82
};

#ifndef RB_SPEC_DRAW_DEFAULT
#define RB_SPEC_DRAW_DEFAULT 0    // RB_OPTS spec_draw: see rb_learner::opt_spec_draw (opt-in: only append-free loops ever arm it)
#endif
static int flush_update(rb_learner* l, hipStream_t stream);
#define RB_FLUSH_UPDATE(l, stream)                                  \
  do {                                                              \
    const int rcf_ = flush_update((l), (hipStream_t)(stream));      \
    if (rcf_ != RB_OK) return rcf_;                                 \
  } while (0)

// ------------------------------------------------------------------------ noise --
// f(x) = sign(x) * sqrt(|x|)  (model.py:32-34).  raw == NULL: N(0,1) from Philox + Box-Muller.
// Draw order = the reference's: per layer randn(in) then randn(out); fc_h_v, fc_h_a, fc_z_v,
// fc_z_a (model.py:36-38, 82-85).
__global__ __launch_bounds__(256) void k_noise(float* noise, float* noise2, const float* raw, NoiseMap map, uint64_t seed,
                                                unsigned long long* ctr) {
  rb_noise_body(noise, noise2, raw, map, seed, ctr, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y, (int)gridDim.y);
}

// ----------------------------------------------------------- small fused passes --
// h[img][n] = relu(sum_s part[s][img][n] + (bias_mu + bias_sigma*eps_out)[n])        model.py:44,72-73
__global__ __launch_bounds__(256) void k_fc_h_finish(const float* part, int splits, int NI, int H2, int n_online,
                                                      NetPtrs on, NetPtrs tg, float* h, float* h_blocked) {
  const int64_t total = (int64_t)NI * H2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int img = (int)(i / H2), n = (int)(i % H2);
    const NetPtrs& p = img < n_online ? on : tg;
    float acc = 0.0f;
    for (int s = 0; s < splits; ++s) acc += part[(int64_t)s * total + i];
    const float bias = p.h_bmu[n] + p.h_bsigma[n] * p.h_eout[n];
    const float o = fmaxf(acc + bias, 0.0f);
    h[i] = o;
    if (h_blocked) h_blocked[((int64_t)(n >> 4) * NI + img) * 16 + (n & 15)] = o;
  }
}

// row-major [rows][K] -> k-blocked copy (only when the generic conv path feeds the streamed FC kernels)
__global__ __launch_bounds__(256) void k_block_copy(const float* x, int rows, int K, float* xb) {
  const int64_t total = (int64_t)rows * K;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / K), k = (int)(i % K);
    xb[((int64_t)(k >> 4) * rows + row) * 16 + (k & 15)] = x[i];
  }
}

// dfeat[b][k] = (feat[b][k] > 0) * sum_s part[s][b][k]
__global__ __launch_bounds__(256) void k_dfeat_finish(const float* part, int splits, int64_t total, const float* feat,
                                                       float* dfeat) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const float fv = feat[i];
    float acc = 0.0f;
    for (int s0 = 0; s0 < splits; s0 += 8) {             // 8 partial loads in flight (a runtime-length loop of load-then-add
      float v[8];                                        // made every split its own dependent round trip)
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(s0 + u < splits ? s0 + u : splits - 1) * total + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += (s0 + u < splits) ? v[u] : 0.0f;
    }
    dfeat[i] = fv > 0.0f ? acc : 0.0f;
  }
}

// conv weight/bias grads: sum the split-K slices in a fixed order (deterministic)
__global__ __launch_bounds__(256) void k_reduce_conv_dw(const float* part, int splits, int cout, int K, float* gw,
                                                         float* gb) {
  const int64_t total = (int64_t)cout * (K + 1);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    float acc = 0.0f;
#pragma unroll 8
    for (int s = 0; s < splits; ++s) acc += part[(int64_t)s * total + i];   // independent loads, fixed add order
    const int co = (int)(i / (K + 1)), col = (int)(i % (K + 1));
    if (col < K) gw[(int64_t)co * K + col] = acc;
    else gb[co] = acc;
  }
}

// all conv layers' split slices in ONE launch (saves two dependent ~5 us launches per step)
struct ReduceLayer {
  const float* part;
  float *gw, *gb;
  int slices, cout, K;
  int64_t begin;          // first flat output index of this layer in the fused index space
};
struct ReduceAllArgs {
  ReduceLayer layer[3];
  int n_layers;
  int64_t total;
  float* sq_part;         // optional: one slot per block = sum of squares of the gradients this block produced
  // replica exchange: every reduced element is ALSO stored at copy_base + (its offset inside the flat gradient), i.e. into
  // the conv segment of this rank's exchange block
  const float* grads_base;
  float* copy_base;
  // tenant blocks behind the reduction's own: copy snap_n floats (the learn call's online noise, for the optimiser pass that
  // forms the hidden layer's sigma gradient itself: the launch hosting that pass resamples the noise)
  const float* snap_src;
  float* snap_dst;
  int snap_n;
  int32_t* snap_clear;      // ... and clear this word (ClipAdamArgs::pair_clipped: no scaled gradient has been stored for this step yet)
};
template <int N>
__device__ __forceinline__ float rb_sum_slices(const float* part, int64_t per, int64_t j, int slices) {
  float v[N];
#pragma unroll
  for (int u = 0; u < N; ++u) v[u] = part[(int64_t)(u < slices ? u : slices - 1) * per + j];   // clamped: always legal
  float acc = 0.0f;
#pragma unroll
  for (int u = 0; u < N; ++u) acc += (u < slices) ? v[u] : 0.0f;
  return acc;
}
__global__ __launch_bounds__(64) void k_reduce_conv_dw_all(ReduceAllArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ((a.total + 63) / 64) * 64) {                  // block-uniform: a snapshot tenant
    const int64_t j = i - ((a.total + 63) / 64) * 64;
    if (j < a.snap_n) a.snap_dst[j] = a.snap_src[j];
    if (j == 0 && a.snap_clear) *a.snap_clear = 0;
    return;
  }
  float my = 0.0f;
  if (i < a.total) {
  int li = 0;
  if (a.n_layers > 1 && i >= a.layer[1].begin) li = 1;
  if (a.n_layers > 2 && i >= a.layer[2].begin) li = 2;
  const ReduceLayer L = a.layer[li];
  const int64_t j = i - L.begin;
  const int64_t per = (int64_t)L.cout * (L.K + 1);
  // fixed add order (slice 0, 1, 2, ...), ALL slice loads of an element in flight at once: one memory round trip instead of
  // one per group of 32 (the first layer's 96 slices were three dependent trips: 4.5 of the kernel's 6 us).  Slice counts:
  // 96 / 64 / 64 at batch 32 (B images x row chunks), the same at larger batches (image groups).
  float acc = 0.0f;
  if (L.slices <= 32) acc = rb_sum_slices<32>(L.part, per, j, L.slices);           // (a layer's elements share the branch)
  else if (L.slices <= 64) acc = rb_sum_slices<64>(L.part, per, j, L.slices);
  else if (L.slices <= 96) acc = rb_sum_slices<96>(L.part, per, j, L.slices);
  else if (L.slices <= 128) acc = rb_sum_slices<128>(L.part, per, j, L.slices);    // data-efficient first layer: 4 chunks x 32
  else {                                                                            // (same left-to-right order, a trip per 32)
    for (int s0 = 0; s0 < L.slices; ++s0) acc += L.part[(int64_t)s0 * per + j];
  }
  const int co = (int)(j / (L.K + 1)), col = (int)(j % (L.K + 1));
  float* dst = col < L.K ? L.gw + (int64_t)co * L.K + col : L.gb + co;
  *dst = acc;
  if (a.copy_base) a.copy_base[dst - a.grads_base] = acc;
  my = acc * acc;
  }
  if (a.sq_part) {
    my = rb_wave_sum(my);
    if (threadIdx.x == 0) a.sq_part[blockIdx.x] = my;
  }
}

// the four factor matrices of the FC weight gradients, rows [0, B), packed into one block for the replica all-gather
struct PackArgs {
  const float* src[5];
  int64_t count[5];
  int64_t dst_off[5];
  float* dst;
};
__global__ __launch_bounds__(256) void k_pack_factors(PackArgs a) {
  const int which = (int)blockIdx.y;
  const int64_t n4 = a.count[which] >> 2;      // all segment sizes are multiples of 4 floats (fast_fc preconditions)
  const float* src = a.src[which];
  float* dst = a.dst + a.dst_off[which];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
    rb_st4(dst + 4 * i, rb_ld4(src + 4 * i));
  if (blockIdx.x == 0)
    for (int64_t i = (n4 << 2) + threadIdx.x; i < a.count[which]; i += blockDim.x) dst[i] = src[i];
}

// ------------------------------------------------------------------------- head --
// One workgroup per sample b.  Dueling combine (model.py:74-75), log-softmax of
// the taken action (agent.py:66-67), double-Q argmax on the online net (agent.py:71-73), target
// probabilities of that action (agent.py:75-76), C51 projection with the atom bins staged in LDS
// and accumulated in the reference's order (agent.py:79-92), cross-entropy (agent.py:94) and
// d loss / d logits for mean(w * loss) (agent.py:96).
#define RB_MAX_ATOMS 256
#define RB_MAX_ACTIONS 64


// One workgroup per sample.  The three logit rows (3*(Z + A*Z) floats) are pulled into LDS with one coalesced sweep; after
// that the kernel touches global memory only for its outputs.  ALL softmaxes of the sample are independent tasks spread over
// the waves in ONE phase: the A double-Q softmaxes of online(next_states) (agent.py:71-73), the A candidate softmaxes of
// target(next_states) — computed for every action while a* is still unknown instead of for a* alone afterwards (round 3's
// per-workgroup timeline: 1.4 us double-Q, then 2.0 us for the two remaining softmaxes on two of eight waves) — and the
// log-softmax of online(states)[action].  Every reduction over atoms is a wave64 DPP reduction (each lane owns atoms
// z = lane, lane + 64, ...; ZI = ceil(Z / 64) is a template parameter: 51 atoms are ONE slot per lane, the former fixed four
// slots quadrupled the instruction count of a phase that runs at one lone wave's issue rate).
#define RB_MAX_NZ RB_HEAD_MAX_NZ

template <int ZI>
struct HeadWave {
  int lane;
  // dueling mean over actions for this lane's atoms: a.mean(1)            model.py:75
  __device__ void mean_of(const float* lg, int Z, int A, float* mean) const {
#pragma unroll
    for (int i = 0; i < ZI; ++i) {
      const int z = lane + 64 * i;
      float acc = 0.0f;
      if (z < Z)
        for (int a = 0; a < A; ++a) acc += lg[Z + a * Z + z];
      mean[i] = acc / (float)A;
    }
  }
  // e[i] = exp(q - max), qm[i] = q - max for this lane's atoms; returns the wave-wide sum of e
  __device__ float softmax_of(const float* lg, int Z, const float* mean, int a, float* e, float* qm) const {
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < ZI; ++i) {
      const int z = lane + 64 * i;
      qm[i] = z < Z ? (lg[z] + lg[Z + a * Z + z]) - mean[i] : -INFINITY;   // q = v + a - mean_a(a)
      mx = fmaxf(mx, qm[i]);
    }
    mx = rb_wave_max(mx);
    float se = 0.0f;
#pragma unroll
    for (int i = 0; i < ZI; ++i) {
      const int z = lane + 64 * i;
      qm[i] = z < Z ? qm[i] - mx : 0.0f;
      e[i] = z < Z ? expf(qm[i]) : 0.0f;
      se += e[i];
    }
    return rb_wave_sum(se);
  }
};

struct HeadTenants {
  ConvWtJob job[2];
  int per_job;           // workgroups per job (0: no tenants)
};
#define RB_HEAD_THREADS 1024      // launch bound; the launch uses 64 x min(16, max(8, 2A + 1)) threads
template <int ZI>
__global__ __launch_bounds__(RB_HEAD_THREADS) void k_head(int B, int Z, int A, const float* logits, const int64_t* actions,
                                               const float* returns, const float* nonterminals, const float* weights,
                                               const float* support, float v_min, float v_max, float gamma_n,
                                               float delta_z, float* log_ps_a_out, float* pns_a_out, float* m_out,
                                               int32_t* a_star_out, float* loss_out, float* dlogits, long long* step_ctr,
                                               const int32_t* batch_status, int32_t* status_copy, float* dlogitsT, HeadTenants tn) {
  // tenant workgroups behind the B samples: the conv input-gradient kernels' weight operand of THIS step (conv_lds.h
  // rb_conv_wt_block) — independent of the head, on CUs this launch leaves idle (32 of 256 busy), two launches ahead of its
  // first reader
  if ((int)blockIdx.x >= B) {
    const int tb = (int)blockIdx.x - B;
    if (tb < tn.per_job) rb_conv_wt_block(tn.job[0], tb, tn.per_job);
    else rb_conv_wt_block(tn.job[1], tb - tn.per_job, tn.per_job);
    return;
  }
  __shared__ float s_lg[3][RB_MAX_NZ];               // rows: online(states), online(next), target(next)
  __shared__ float s_pt[RB_MAX_NZ];                  // target(next) probabilities of EVERY action: [a][z] at a * Z + z
  __shared__ float s_lo[RB_MAX_ATOMS], s_hi[RB_MAX_ATOMS], s_m[RB_MAX_ATOMS], s_logp[RB_MAX_ATOMS], s_sup[RB_MAX_ATOMS];
  __shared__ int s_l[RB_MAX_ATOMS], s_u[RB_MAX_ATOMS];
  __shared__ float s_ev[RB_MAX_ACTIONS];
  __shared__ float s_scal[2];                        // sum(m), -loss
  const int t = (int)threadIdx.x, T = (int)blockDim.x, lane = rb_lane(), wave = rb_wave(), nw = T >> 6;
  const int b = (int)blockIdx.x;
  const int NZ = Z + A * Z;
  RB_WGT(7, b, 0);
  RB_WGT_HW(7, b);
  for (int i = t; i < NZ; i += T) {
    s_lg[0][i] = logits[(int64_t)b * NZ + i];
    s_lg[1][i] = logits[(int64_t)(B + b) * NZ + i];
    s_lg[2][i] = logits[(int64_t)(2 * B + b) * NZ + i];
  }
  for (int z = t; z < Z; z += T) s_sup[z] = support[z];  // requested with the logits: one memory round trip, not two
  const float R = returns[b], nt = nonterminals[b], wgt = weights[b];
  const int act = (int)actions[b];
  __syncthreads();
  RB_WGT(7, b, 1);
  HeadWave<ZI> hw;
  hw.lane = lane;
  float mean[ZI], e[ZI], qm[ZI];

  // ---------------- every softmax of the sample, one task per wave (round-robin when 2A + 1 exceeds the wave count)
  for (int task = wave; task < 2 * A + 1; task += nw) {                 // wave-uniform
    if (task < A) {
      // double-Q selection on online(next_states)   agent.py:71-73
      hw.mean_of(s_lg[1], Z, A, mean);
      const float se = hw.softmax_of(s_lg[1], Z, mean, task, e, qm);
      float sv = 0.0f;
#pragma unroll
      for (int i = 0; i < ZI; ++i) sv += (lane + 64 * i < Z ? s_sup[lane + 64 * i] : 0.0f) * e[i];
      sv = rb_wave_sum(sv);
      if (lane == 0) s_ev[task] = sv / se;                            // sum_z z * p(z)
    } else if (task < 2 * A) {
      // target(next_states)[a] probabilities for candidate a   agent.py:75-76
      const int a = task - A;
      hw.mean_of(s_lg[2], Z, A, mean);
      const float se = hw.softmax_of(s_lg[2], Z, mean, a, e, qm);
#pragma unroll
      for (int i = 0; i < ZI; ++i) {
        const int z = lane + 64 * i;
        if (z < Z) s_pt[a * Z + z] = e[i] / se;
      }
    } else {
      // online(states): log p(s_t, a_t)            agent.py:66-67
      hw.mean_of(s_lg[0], Z, A, mean);
      const float se = hw.softmax_of(s_lg[0], Z, mean, act, e, qm);
      const float lse = logf(se);
#pragma unroll
      for (int i = 0; i < ZI; ++i) {
        const int z = lane + 64 * i;
        if (z < Z) {
          const float lp = qm[i] - lse;                             // log_softmax = (q - max) - log(sum exp(q - max))
          s_logp[z] = lp;
          log_ps_a_out[(int64_t)b * Z + z] = lp;
        }
      }
    }
  }
  __syncthreads();
  RB_WGT(7, b, 2);
  int a_star = 0;
  {
    float best = s_ev[0];
    for (int a = 1; a < A; ++a)
      if (s_ev[a] > best) { best = s_ev[a]; a_star = a; }         // argmax, first maximum
  }
  if (t == 0) a_star_out[b] = a_star;
  // this learn call's optimiser step number (1-based); a batch the sampler gave up on does not count (no update follows)
  if (b == 0 && t == 0) {
    const int32_t st = batch_status ? *batch_status : 0;
    if (status_copy) *status_copy = st;
    if (step_ctr && st == 0) *step_ctr = *step_ctr + 1;
  }
  // ---------------- projection inputs from the selected action's probabilities      agent.py:79-86
  for (int z = t; z < Z; z += T) {
    const float p = s_pt[a_star * Z + z];
    pns_a_out[(int64_t)b * Z + z] = p;
    float Tz = R + (nt * gamma_n) * s_sup[z];                 // agent.py:79
    Tz = fminf(fmaxf(Tz, v_min), v_max);                      // agent.py:80
    const float bq = (Tz - v_min) / delta_z;                  // agent.py:82
    int l = (int)floorf(bq), u = (int)ceilf(bq);              // agent.py:83
    if (u > 0 && l == u) l -= 1;                              // agent.py:85
    if (l < Z - 1 && l == u) u += 1;                          // agent.py:86
    s_l[z] = l; s_u[z] = u;
    s_lo[z] = p * ((float)u - bq);                            // agent.py:91
    s_hi[z] = p * (bq - (float)l);                            // agent.py:92
    s_m[z] = 0.0f;
  }
  __syncthreads();
  RB_WGT(7, b, 3);
  // ---------------- scatter into atom bins in the reference's accumulation order   agent.py:89-92
  // b is monotone in the atom index (support increasing, nt*gamma^n >= 0), so equal l (and equal u) form
  // contiguous runs: the first atom of a run owns its bin and adds the run left to right — exactly the order of
  // the reference's first index_add_ (all l bins, j ascending) followed by the second (u bins) on the same m.
  // The run's adds are inherently serial (float adds in the reference's order), but their OPERANDS need not be: walking the run with
  // `jj < Z && s_l[jj] == key` made every atom two dependent LDS round trips (~130 cycles), fine for the usual one to three atoms per
  // bin, 2 x 51 steps = 7 us for a TERMINAL transition, whose atoms all land in one bin — and with 256 samples per batch there is
  // almost always one: the launch was 12 us for 5.4 us workgroups (profiles/round6_wg_timeline_b256.txt).  For Z <= 64 the run lengths
  // come from one ballot of the run starts, and an owner fetches its run eight atoms per round trip, then adds them in order.
#if defined(RB_HEAD_NO_SCAN)      // (variant build for A/B runs)
  const bool by_ballot = false;
#else
  const bool by_ballot = Z <= 64;
#endif
  auto scatter_runs = [&](const int* s_key, const float* s_x, bool second) {      // wave 0, lane = atom
    const bool valid = lane < Z;
    const int key = s_key[valid ? lane : Z - 1];
    const int prev = __shfl_up(key, 1);
    const bool start = valid && (lane == 0 || prev != key);
    const unsigned long long starts = __ballot(start ? 1 : 0);
    const unsigned long long rest = lane < 63 ? starts >> (lane + 1) : 0ull;       // run starts behind this atom
    const int len = rest ? __builtin_ctzll(rest) + 1 : Z - lane;                   // atoms of the run that starts here
    if (start) {
      float acc = second ? s_m[key] : 0.0f;
      for (int i0 = 0; i0 < len; i0 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int jj = lane + i0 + u; v[u] = s_x[jj < Z ? jj : Z - 1]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = i0 + u < len ? acc + v[u] : acc;
      }
      s_m[key] = acc;
    }
  };
  {
    if (by_ballot) {
      if (wave == 0) scatter_runs(s_l, s_lo, false);
    } else {
      for (int j = t; j < Z; j += T) {
        const int key = s_l[j];
        if (j == 0 || s_l[j - 1] != key) {
          float acc = 0.0f;
          for (int jj = j; jj < Z && s_l[jj] == key; ++jj) acc += s_lo[jj];
          s_m[key] = acc;
        }
      }
    }
    __syncthreads();
    if (by_ballot) {
      if (wave == 0) scatter_runs(s_u, s_hi, true);
    } else {
      for (int j = t; j < Z; j += T) {
        const int key = s_u[j];
        if (j == 0 || s_u[j - 1] != key) {
          float acc = s_m[key];
          for (int jj = j; jj < Z && s_u[jj] == key; ++jj) acc += s_hi[jj];
          s_m[key] = acc;
        }
      }
    }
  }
  __syncthreads();
  RB_WGT(7, b, 4);
  for (int k = t; k < Z; k += T) m_out[(int64_t)b * Z + k] = s_m[k];
  if (wave == 0) {                                                // loss = -sum m * log p   agent.py:94
    float pl = 0.0f, pm = 0.0f;
    for (int z = lane; z < Z; z += 64) { pl += s_m[z] * s_logp[z]; pm += s_m[z]; }
    pl = rb_wave_sum(pl);
    pm = rb_wave_sum(pm);
    if (lane == 0) { s_scal[0] = pm; s_scal[1] = pl; loss_out[b] = -pl; }
  }
  __syncthreads();
  // ---------------- backward of mean(w * loss) to the logits     agent.py:96
  // d/dq[z] = (w/B) * (p[z] * sum(m) - m[z]) on the taken action; dueling adjoint:
  // dv[z] = g[z] ; da[a'][z] = (delta(a',act) - 1/A) * g[z]
  RB_WGT(7, b, 5);
  const float coef = wgt / (float)B;
  const float msum = s_scal[0];
  float* dl = dlogits + (int64_t)b * NZ;
  for (int i = t; i < NZ; i += T) {
    const int z = i < Z ? i : (i - Z) % Z;
    const float g = coef * (expf(s_logp[z]) * msum - s_m[z]);
    float o;
    if (i < Z) o = g;
    else o = ((i - Z) / Z == act ? g : 0.0f) - g / (float)A;
    dl[i] = o;
    if (dlogitsT) dlogitsT[(int64_t)i * B + b] = o;      // [NZ][B]: the output layer's input gradient reads 16 consecutive samples of a row
  }
  RB_WGT(7, b, 6);
}

// Agent.act / evaluate_q head (agent.py:53-55, 110-112) for ONE image at logits row `row` (act_path.h rb_head_act_body).
__global__ __launch_bounds__(256) void k_head_act(int Z, int A, const float* logits, int row, const float* support,
                                                   int32_t* action_out, float* q_out) {
  __shared__ float s_mean[RB_MAX_ATOMS];
  __shared__ float s_ev[RB_MAX_ACTIONS];
  row += (int)blockIdx.x;                       // batched acting: one workgroup per state, outputs indexed alike
  rb_head_act_body(Z, A, logits + (int64_t)row * (Z + A * Z), support, s_mean, s_ev, action_out ? action_out + blockIdx.x : nullptr,
                   q_out ? q_out + blockIdx.x : nullptr, nullptr);
}

// -------------------------------------------------------------- global-norm clip --
// clip_grad_norm_ (agent.py:97).  Stage 1: per-block sum of squares (fixed tree order).
__global__ __launch_bounds__(256) void k_sumsq(const float* g, int64_t n, float* part) {
  __shared__ float s_red[16];
  float acc = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    acc = fmaf(g[i], g[i], acc);
  acc = rb_block_sum(acc, s_red);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
// rb_learner_finish_grads as ONE launch (three dependent-free jobs would otherwise queue as three ~10 us kernels on the
// replica step's critical path): block ranges [fc_z dW tiles | fc_h dW tiles | sum of squares of the all-reduced conv range]
struct FinishArgs {
  NlDwArgs z, h;
  int z_x, z_n, h_x, h_n;      // grid.x and block count of each weight-gradient problem
  // conv range: g[i] = (sum over ranks, in rank order, of blocks[r * bstride + i]) * scale; part[b] = this block's sum of squares
  float* g; int64_t n; float* part; int nparts;
  const float* blocks; int64_t bstride; int world; float scale;
};
__global__ __launch_bounds__(256) void k_finish_grads(FinishArgs a) {
  __shared__ float s_red[16];
  int b = (int)blockIdx.x;
  if (b < a.z_n) { rb_nl_dw_body_ranks(a.z, b % a.z_x, b / a.z_x, 4 * b); return; }
  b -= a.z_n;
  if (b < a.h_n) { rb_nl_dw_body_ranks(a.h, b % a.h_x, b / a.h_x, 4 * b); return; }
  b -= a.h_n;
  float acc = 0.0f;
  for (int64_t i = (int64_t)b * 256 + threadIdx.x; i < a.n; i += (int64_t)a.nparts * 256) {
    float v = 0.0f;
    for (int r = 0; r < a.world; ++r) v += a.blocks[(int64_t)r * a.bstride + i];
    v *= a.scale;
    a.g[i] = v;
    acc = fmaf(v, v, acc);
  }
  acc = rb_block_sum(acc, s_red);
  if (threadIdx.x == 0) a.part[b] = acc;
}
// The same launch with the hidden layer's weight gradient on 128 x 128 LDS tiles (fc_gemm.h rb_fc_gemm_dw_ranks; 512-thread
// workgroups): block ranges [fc_h tiles | fc_z 16-row tiles (first four waves) | conv range]
__global__ __launch_bounds__(RB_TG_THREADS) void k_finish_grads_tiled(FinishArgs a, int h_nt, int h_kt) {
  __shared__ __attribute__((aligned(16))) float lds[RB_TG_LDS];
  __shared__ float s_red[16];
  int b = (int)blockIdx.x;
  const int h_n = h_nt * h_kt;
  if (b < h_n) {
    // the conv range rides in the tile workgroups (a.nparts == h_n: one slice and one partial per workgroup): its per-element chain —
    // `world` loads, one add each — as 20 workgroups of their own was the launch's pole (22 of 40 us with the tile loop ablated:
    // every thread walked 8 elements x 8 ranks one dependent load at a time).  Here: one element per thread and trip, all ranks'
    // loads in flight, in front of the tile loop.
    float acc = 0.0f;
    for (int64_t i = (int64_t)b * blockDim.x + threadIdx.x; i < a.n; i += (int64_t)a.nparts * blockDim.x) {
      float v = 0.0f;
      for (int r0 = 0; r0 < a.world; r0 += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = a.blocks[(int64_t)(r0 + u < a.world ? r0 + u : a.world - 1) * a.bstride + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += (r0 + u < a.world) ? t[u] : 0.0f;      // rank order
      }
      v *= a.scale;
      a.g[i] = v;
      acc = fmaf(v, v, acc);
    }
    acc = rb_block_sum(acc, s_red);
    if (threadIdx.x == 0) a.part[b] = acc;
    rb_fc_gemm_dw_ranks(a.h, b / h_kt, b % h_kt, 8 * b, lds);
    return;
  }
  b -= h_n;
  if (b < a.z_n) rb_nl_dw_body_ranks(a.z, b % a.z_x, b / a.z_x, 8 * b);      // (all eight waves: 512 columns per workgroup)
}
// Stage 2: every block re-reduces the partials (same order everywhere), then scales its slice.
__global__ __launch_bounds__(256) void k_clip_scale(float* g, int64_t n, const float* part, int nparts, float max_norm,
                                                     float* norm_out) {
  __shared__ float s_red[16];
  float acc = 0.0f;
  for (int i = (int)threadIdx.x; i < nparts; i += (int)blockDim.x) acc += part[i];
  acc = rb_block_sum(acc, s_red);
  const float total = sqrtf(acc);
  float coef = max_norm / (total + 1e-6f);
  if (coef > 1.0f) coef = 1.0f;                                    // clamp(max=1.0)
  if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) *norm_out = total;
  if (coef < 1.0f)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
      g[i] *= coef;
}

// (ClipAdamArgs, rb_adam_elem / rb_adam_quad and the hosted form of the pass: adam_body.h)
// Tile part of the fused optimiser pass (batch <= 32).  The hidden layer's weight gradient is a rank-B product,
// g_mu = dY^T X  (dY [B][2H], X [B][F], both L2-resident: 0.5 MB), g_sigma = g_mu * (eps_out x eps_in).  Writing it in the
// backward and reading it back here costs 2 x 25.7 MB of HBM traffic per step; instead a wave recomputes its 16 x 64
// tile with 32 MFMAs (same operand order as rb_nl_dw_body_pipe, so the bits equal those of the backward's norm-only
// pass) while its p / m / v loads are in flight, and applies clip + Adam to mu and sigma right there: 6 array passes
// over the 6.4 M weights instead of 9.
struct FusedDwAdamArgs {
  NlDwArgs dw;                 // operands of the weight gradient (g_* unused)
  int64_t mu_off, sigma_off;   // offsets of the [2H][F] mu / sigma arrays inside p, m, v (and g)
  int dw_x, n_tile_blocks;     // 256-column block columns; 256-thread tile blocks = dw_x * (2H / 16)
  int write_grads;             // tests: also store the (unclipped... as clip_grad_norm_ leaves it: clipped) gradient tile
};
// Each tile is taken by TWO workgroup slots: slot 0 updates mu, slot 1 sigma (both recompute the same 32 MFMAs — 0.4 GFLOP
// extra per step against 24 fewer live registers per lane: 4 waves per SIMD instead of 2, no spills; the single-slot
// version measured 36.5 us per launch against 35.0 for the plain streaming pass, i.e. slower despite 13 % fewer bytes).
template <bool WT>
__device__ __forceinline__ void rb_fused_dw_adam_tile(const ClipAdamArgs& a, const FusedDwAdamArgs& f, int b2, float coef) {
  const NlDwArgs& d = f.dw;
  const int lane = rb_lane(), wave = rb_wave();
  const int which = b2 & 1, b = b2 >> 1;                 // 0: mu, 1: sigma
  const int bx = b % f.dw_x, by = b / f.dw_x;
  const int kt = bx * 256 + wave * 64;
  if (kt >= d.K) return;                                  // wave-uniform
  const int g = (d.n_prob > 1 && by >= d.prob[1].tile_begin) ? 1 : 0;
  const NlDwProblem pr = d.prob[g];
  const int row0 = pr.row_begin + (by - pr.tile_begin) * 16;
  const int row_end = pr.row_begin + pr.row_cnt;
  const int c = lane & 15, q = lane >> 4;
  int col4 = kt + 4 * c;
  const bool cv = col4 < d.K;
  if (!cv) col4 = d.K - 4;
  int arow = row0 + c;
  const bool av_ok = arow < row_end;
  if (!av_ok) arow = row_end - 1;
  const int64_t arr = which ? f.sigma_off : f.mu_off;
  // operands of the gradient tile first (they gate the MFMAs), then the 12 parameter / moment quads (they gate the update)
  float avs[8];
  float4 xs[8];
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    const int m = 4 * st + q;
    const bool mv = m < d.M;
    const int mcl = mv ? m : d.M - 1;
    avs[st] = (mv && av_ok) ? d.dy[(int64_t)mcl * d.ldy + arow] : 0.0f;
    xs[st] = rb_ld4(d.x + (int64_t)mcl * d.ldx + pr.x_off + col4);
    if (!mv) { xs[st].x = 0.0f; xs[st].y = 0.0f; xs[st].z = 0.0f; xs[st].w = 0.0f; }
  }
  const float4 e4 = rb_ld4(d.ein + pr.ein_off + col4);
  float eo4[4];
  int64_t off[4];
  float4 P[4], M[4], V[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    int n = row0 + 4 * q + e;
    if (n > row_end - 1) n = row_end - 1;                 // clamped rows are loaded (legal) and never stored
    eo4[e] = d.eout[n];
    off[e] = arr + (int64_t)n * d.K + col4;
    P[e] = rb_ld4(a.p + off[e]); M[e] = rb_ld4(a.m + off[e]); V[e] = rb_ld4(a.v + off[e]);
  }
  rb_f32x4 acc[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { acc[0][e] = 0.0f; acc[1][e] = 0.0f; acc[2][e] = 0.0f; acc[3][e] = 0.0f; }
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    if (4 * st < d.M) {                                   // uniform
      acc[0] = rb_mfma16(avs[st], xs[st].x, acc[0]);
      acc[1] = rb_mfma16(avs[st], xs[st].y, acc[1]);
      acc[2] = rb_mfma16(avs[st], xs[st].z, acc[2]);
      acc[3] = rb_mfma16(avs[st], xs[st].w, acc[3]);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int n = row0 + 4 * q + e;
    if (n < row_end && cv) {
      float4 gr;
      gr.x = acc[0][e]; gr.y = acc[1][e]; gr.z = acc[2][e]; gr.w = acc[3][e];
      if (which) {                                        // block-uniform: g_sigma = g_mu * (eps_out * eps_in), model.py:39,44
        const float eo = eo4[e];
        gr.x = gr.x * (eo * e4.x); gr.y = gr.y * (eo * e4.y); gr.z = gr.z * (eo * e4.z); gr.w = gr.w * (eo * e4.w);
      }
      rb_adam_quad(P[e], gr, M[e], V[e], coef, a);
      if (WT) {
        const unsigned o = (unsigned)(4 * off[e]);
        rb_st4_wt(a.p, o, P[e]); rb_st4_wt(a.m, o, M[e]); rb_st4_wt(a.v, o, V[e]);
      } else {
        rb_st4(a.p + off[e], P[e]); rb_st4(a.m + off[e], M[e]); rb_st4(a.v + off[e], V[e]);
      }
      if (f.write_grads) rb_st4(a.g + off[e], gr);         // as clip_grad_norm_ leaves .grad: scaled when the clip bites
    }
  }
}
#ifndef RB_ADAM_MINWAVES
#define RB_ADAM_MINWAVES 1
#endif
template <int RB_ADAM_UNROLL, bool WT, bool FUSED>   // float4 quadruples (p, g, m, v) in flight per thread; WT: write-through stores
__global__ __launch_bounds__(256, RB_ADAM_MINWAVES) void k_clip_adam(ClipAdamArgs a, FusedDwAdamArgs f) {
  __shared__ float s_red[18];      // [0, 16) rb_block_sum's wave slots; [16], [17] the bias-correction scalars (slots of their own:
                                   // thread 0 writes them while other waves may still be reading the wave slots of the sum —
                                   // the host interpreter's schedule turned that into a wrong clip coefficient for every thread
                                   // but thread 0 whenever the clip bit and the step number came from the device counter)
  const bool tile_block = FUSED && (int)blockIdx.x < f.n_tile_blocks;
  const int64_t n4 = (a.n >> 2) - (FUSED ? a.skip_len4 : 0);
  const int eb = FUSED ? (int)blockIdx.x - f.n_tile_blocks : (int)blockIdx.x;
  const int64_t base = (int64_t)eb * (256 * RB_ADAM_UNROLL) + threadIdx.x;
  float4 P[RB_ADAM_UNROLL], G[RB_ADAM_UNROLL], M[RB_ADAM_UNROLL], V[RB_ADAM_UNROLL];
  int64_t idx[RB_ADAM_UNROLL];
  if (!tile_block) {
#pragma unroll
    for (int u = 0; u < RB_ADAM_UNROLL; ++u) {
      int64_t i = base + u * 256;
      if (i >= n4) i = n4 > 0 ? n4 - 1 : 0;          // clamped load (always legal), masked store
      if (FUSED && i >= a.skip_lo4) i += a.skip_len4;
      idx[u] = i;
      P[u] = rb_ld4(a.p + 4 * i); G[u] = rb_ld4(a.g + 4 * i); M[u] = rb_ld4(a.m + 4 * i); V[u] = rb_ld4(a.v + 4 * i);
    }
  }
  if (a.batch_status && *a.batch_status != 0) {                     // block-uniform (every block reads the same word)
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.norm_out) *a.norm_out = 0.0f;
    return;
  }
  float acc = 0.0f;
  for (int i = (int)threadIdx.x; i < a.nparts; i += 256) acc += a.part[i];
  acc = rb_block_sum(acc, s_red);
  const float total = sqrtf(acc);
  float coef = a.max_norm / (total + 1e-6f);
  if (coef > 1.0f) coef = 1.0f;                                    // clamp(max=1.0)
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.norm_out) *a.norm_out = total;
  if (a.step_dev) {                                                // block-uniform
    if (threadIdx.x == 0) {
      const double t = (double)*a.step_dev;
      const double bc1 = 1.0 - pow(a.beta1, t), bc2 = 1.0 - pow(a.beta2, t);
      s_red[16] = (float)(-(a.lr / bc1));
      s_red[17] = (float)sqrt(bc2);
    }
    __syncthreads();
    a.neg_step_size = s_red[16];
    a.bc2_sqrt = s_red[17];
  }
  if (tile_block) {
    rb_fused_dw_adam_tile<WT>(a, f, (int)blockIdx.x, coef);
    return;
  }
#pragma unroll
  for (int u = 0; u < RB_ADAM_UNROLL; ++u) {
    if (base + u * 256 >= n4) continue;
    const int64_t i = idx[u];
    rb_adam_quad(P[u], G[u], M[u], V[u], coef, a);
    if (WT) {
      const unsigned off = (unsigned)(16 * i);
      rb_st4_wt(a.p, off, P[u]); rb_st4_wt(a.m, off, M[u]); rb_st4_wt(a.v, off, V[u]);
    } else {
      rb_st4(a.p + 4 * i, P[u]); rb_st4(a.m + 4 * i, M[u]); rb_st4(a.v + 4 * i, V[u]);
    }
    if (coef < 1.0f) rb_st4(a.g + 4 * i, G[u]);
  }
  // tail (n % 4 elements): last block's first threads
  if (blockIdx.x == gridDim.x - 1) {
    const int64_t t = ((a.n >> 2) << 2) + threadIdx.x;
    if (t < a.n) {
      float p = a.p[t], g = a.g[t], m = a.m[t], v = a.v[t];
      rb_adam_elem(p, g, m, v, coef, a);
      a.p[t] = p; a.m[t] = m; a.v[t] = v;
      if (coef < 1.0f) a.g[t] = g;
    }
  }
}

// ========================================================================= host ==
static int pick_splits(int64_t tiles, int ksteps, int64_t target_blocks) {
  int64_t s = target_blocks / (tiles > 0 ? tiles : 1);
  if (s < 1) s = 1;
  if (s > ksteps) s = ksteps;
  if (s > 64) s = 64;
  const int64_t per = (ksteps + s - 1) / s;   // no empty trailing split (every partial slice gets written)
  s = (ksteps + per - 1) / per;
  return (int)s;
}

template <class G>
static int launch_conv_fwd(rb_learner* l, int layer, int n_on, int n_tg, const ImgSrc& src, const NetPtrs& on,
                           const NetPtrs& tg, hipStream_t stream) {
  const ConvLayer& c = l->L.conv[layer];
  const int n_max = (n_on > n_tg ? n_on : n_tg) * G::P;
  if (layer == 0) {
    ConvFwdProb<G, true> p;
    p.cin = c.cin; p.cout = c.cout;
    p.n_img[0] = n_on; p.n_img[1] = n_tg; p.img_base[0] = 0; p.img_base[1] = n_on;
    p.w[0] = on.conv_w[0]; p.w[1] = tg.conv_w[0]; p.bias[0] = on.conv_b[0]; p.bias[1] = tg.conv_b[0];
    p.src = src; p.in_f = nullptr; p.out = l->act[0];
    RB_LAUNCH((k_gemm<1, 2, ConvFwdProb<G, true>>), dim3(1, (unsigned)rb_div_up(n_max, 64), 2), dim3(128), stream, p);
  } else {
    ConvFwdProb<G, false> p;
    p.cin = c.cin; p.cout = c.cout;
    p.n_img[0] = n_on; p.n_img[1] = n_tg; p.img_base[0] = 0; p.img_base[1] = n_on;
    p.w[0] = on.conv_w[layer]; p.w[1] = tg.conv_w[layer]; p.bias[0] = on.conv_b[layer]; p.bias[1] = tg.conv_b[layer];
    p.src = src; p.in_f = l->act[layer - 1]; p.out = l->act[layer];
    RB_LAUNCH((k_gemm<2, 1, ConvFwdProb<G, false>>), dim3(1, (unsigned)rb_div_up(n_max, 32), 2), dim3(128), stream, p);
  }
  RB_LAUNCH_CHECK();
  return RB_OK;
}

template <class G, int NT, int PR, int KMAX, bool FIRST, int PCH = 32 * NT>
static int launch_conv_fwd_lds(rb_learner* l, int layer, int n_on, int n_tg, const ImgSrc& src, const NetPtrs& on,
                               const NetPtrs& tg, hipStream_t stream) {
  const ConvLayer& c = l->L.conv[layer];
  ConvLdsFwdArgs a;
  a.cin = c.cin; a.cout = c.cout; a.n_on = n_on;
  a.w[0] = on.conv_w[layer]; a.w[1] = tg.conv_w[layer]; a.bias[0] = on.conv_b[layer]; a.bias[1] = tg.conv_b[layer];
  a.src = src; a.in_f = layer > 0 ? l->act[layer - 1] : nullptr; a.out = l->act[layer];
  a.out_blocked = (layer == l->L.nconv - 1 && l->fast_fc) ? l->feat_b : nullptr;
  a.rows_total = n_on + n_tg;
  a.ipb = 1;
  a.img_fast = 0;
  static const char* const tags[3] = {"conv1_fwd:k_conv_fwd_lds", "conv2_fwd:k_conv_fwd_lds", "conv3_fwd:k_conv_fwd_lds"};
  // large batches: one round of workgroups, each keeping its weight slab for ipb images of one net (conv_lds.h)
  const bool multi_forced = l->opt_conv_multi >= 0;                      // RB_CONV_MULTI: images per workgroup (0 = off)
  int ipb = 0;
  if (multi_forced) ipb = l->opt_conv_multi;
  else if (n_on + n_tg >= 256) {
    const int per_img = (int)(rb_div_up(G::P, PCH) * rb_div_up(c.cout, 32));
    ipb = (int)rb_div_up((int64_t)(n_on + n_tg) * per_img, 256);
  }
  if constexpr (FIRST && ConvFwdFullLds<G, KMAX>::FITS) {
    // first layer: whole image per workgroup, whole reduction per wave (RB_CONV_FULL=0: the chunked kernel below)
    const bool full_off = !l->opt_conv_full;
    if (ipb > 0 && !src.f32 && c.cout <= 32 && !a.out_blocked && !full_off && c.cin * G::KK == KMAX && (KMAX & 1) == 0) {
      int fi = ipb;
      if (!multi_forced) fi = (int)rb_div_up(n_on + n_tg, 256);         // one round of workgroups
      a.ipb = fi;
      RB_LAUNCH_T(tags[layer], (k_conv_fwd_full<G, KMAX>), dim3(1, 1, (unsigned)rb_div_up(n_on + n_tg, fi)),
                  dim3(RB_CONV_THREADS), stream, a);
      RB_LAUNCH_CHECK();
      return RB_OK;
    }
  }
  if constexpr (!FIRST && ConvFwdMultiLds<G, PR, KMAX>::FITS) {      // (first layers: k_conv_fwd_full above)
    if (ipb > 0) {
      a.ipb = ipb;
      const unsigned ngroups = (unsigned)rb_div_up(n_on + n_tg, ipb);
      dim3 gridm((unsigned)rb_div_up(G::P, PCH), (unsigned)rb_div_up(c.cout, 32), ngroups);
      if (l->opt_img_fast && ngroups % 8 == 0) {     // image-group-fastest block order (layers 2 and 3 use the same ipb)
        a.img_fast = 1;
        gridm = dim3(ngroups, (unsigned)rb_div_up(c.cout, 32), (unsigned)rb_div_up(G::P, PCH));
      }
      if constexpr (KMAX % 16 == 0 && (KMAX / G::KK) % 4 == 0 && 2 * ((PCH + 15) / 16) <= 16 && (PCH % 16 == 0 || PCH >= G::P) && G::P > 16) {
        // whole-K 16x16x4 tiles in the image loop as well (conv_lds.h k_conv_fwd_multi_t16; RB_OPTS conv_multi_t16=0: the split-K body)
        if (l->opt_conv_multi_t16 && ((l->opt_t16 >> layer) & 1) && c.cin * G::KK == KMAX && c.cout % 32 == 0) {
          constexpr int NWV = ConvFwdWaves<G, NT, PR, KMAX, false, PCH, false, 1>::NWV;
          RB_LAUNCH_T(tags[layer], (k_conv_fwd_multi_t16<G, NT, PR, KMAX, PCH>), gridm, dim3(64 * NWV), stream, a);
          RB_LAUNCH_CHECK();
          return RB_OK;
        }
      }
      RB_LAUNCH_T(tags[layer], (k_conv_fwd_multi<G, NT, PR, KMAX, FIRST, PCH>), gridm, dim3(RB_CONV_THREADS), stream, a);
      RB_LAUNCH_CHECK();
      return RB_OK;
    }
  }
  dim3 grid1((unsigned)rb_div_up(G::P, PCH), (unsigned)rb_div_up(c.cout, 32), (unsigned)(n_on + n_tg));
  if (l->opt_img_fast && (n_on + n_tg) % 8 == 0) {     // RB_CONV_IMGFAST: image-fastest block order (XCD = image mod 8 in every layer)
    a.img_fast = 1;
    grid1 = dim3((unsigned)(n_on + n_tg), (unsigned)rb_div_up(c.cout, 32), (unsigned)rb_div_up(G::P, PCH));
  }
  if constexpr (KMAX % 16 == 0 && (KMAX / G::KK) % 4 == 0 && 2 * ((PCH + 15) / 16) <= 16 && (PCH % 16 == 0 || PCH >= G::P) && G::P > 16) {
    // whole-K 16x16x4 tiles, one wave per tile, no cross-wave reduction (conv_lds.h T16): the learn step's u8 / f32 inputs
    if (((l->opt_t16 >> layer) & 1) && !(FIRST && src.f32) && c.cin * G::KK == KMAX && c.cout % 32 == 0) {
      constexpr int CTW = 1;                    // channel tiles per wave (2 measured slower for the first layer: 5-wave staging)
      constexpr int NWV = ConvFwdWaves<G, NT, PR, KMAX, FIRST, PCH, false, CTW>::NWV;
      RB_LAUNCH_T(tags[layer], (k_conv_fwd_t16<G, NT, PR, KMAX, FIRST, PCH, CTW>), grid1, dim3(64 * NWV), stream, a);
      RB_LAUNCH_CHECK();
      return RB_OK;
    }
  }
  if (FIRST && src.f32) {       // float states (act / evaluate): an instantiation of its own (conv_lds.h F32SRC)
    RB_LAUNCH_T(tags[layer], (k_conv_fwd_lds<G, NT, PR, KMAX, FIRST, PCH, FIRST>), grid1, dim3(RB_CONV_THREADS), stream, a);
  } else {
    RB_LAUNCH_T(tags[layer], (k_conv_fwd_lds<G, NT, PR, KMAX, FIRST, PCH>), grid1, dim3(RB_CONV_THREADS), stream, a);
  }
  RB_LAUNCH_CHECK();
  return RB_OK;
}

static int conv_fwd(rb_learner* l, int layer, int n_on, int n_tg, const ImgSrc& src, const NetPtrs& on,
                    const NetPtrs& tg, hipStream_t stream) {
  const ConvLayer& c = l->L.conv[layer];
  if (l->fast_conv) {
    if (c.ks == 8) {
      // 80 positions (4 output rows) per workgroup: 5 x 96 = 480 workgroups at batch 32, one round at two per CU
      // (64 positions gave 672, the seventh chunk of each image nearly empty: 224.4 vs 222.6 us per step; 100 positions
      // = 384 workgroups measured 225)
      return launch_conv_fwd_lds<GeomC1, 3, 20, 256, true, 80>(l, layer, n_on, n_tg, src, on, tg, stream);
    }
    if (c.ks == 4) return launch_conv_fwd_lds<GeomC2, 3, 20, 512, false>(l, layer, n_on, n_tg, src, on, tg, stream);
    if (c.ks == 3) return launch_conv_fwd_lds<GeomC3, 2, 9, 576, false>(l, layer, n_on, n_tg, src, on, tg, stream);
    if (c.ih == 84) return launch_conv_fwd_lds<GeomD1, 2, 20, 100, true>(l, layer, n_on, n_tg, src, on, tg, stream);
    return launch_conv_fwd_lds<GeomD2, 1, 16, 800, false>(l, layer, n_on, n_tg, src, on, tg, stream);
  }
  if (c.ks == 8) return launch_conv_fwd<GeomC1>(l, layer, n_on, n_tg, src, on, tg, stream);
  if (c.ks == 4) return launch_conv_fwd<GeomC2>(l, layer, n_on, n_tg, src, on, tg, stream);
  if (c.ks == 3) return launch_conv_fwd<GeomC3>(l, layer, n_on, n_tg, src, on, tg, stream);
  if (c.ih == 84) return launch_conv_fwd<GeomD1>(l, layer, n_on, n_tg, src, on, tg, stream);
  return launch_conv_fwd<GeomD2>(l, layer, n_on, n_tg, src, on, tg, stream);
}

static NlWeights nl_h(const NetPtrs& p) {
  NlWeights w;
  w.mu = p.h_mu; w.sigma = p.h_sigma; w.eout = p.h_eout; w.ein = p.h_ein; w.bmu = p.h_bmu; w.bsigma = p.h_bsigma;
  return w;
}
static NlWeights nl_z(const NetPtrs& p) {
  NlWeights w;
  w.mu = p.z_mu; w.sigma = p.z_sigma; w.eout = p.z_eout; w.ein = p.z_ein; w.bmu = p.z_bmu; w.bsigma = p.z_bsigma;
  return w;
}

// Forward of n_on online images + n_tg target images up to the logits.
static int forward(rb_learner* l, int n_on, int n_tg, const ImgSrc& src, const NetPtrs& on, const NetPtrs& tg,
                   hipStream_t stream) {
  const Layout& L = l->L;
  const int NI = n_on + n_tg;
  for (int layer = 0; layer < L.nconv; ++layer) {
    int rc = conv_fwd(l, layer, n_on, n_tg, src, on, tg, stream);
    if (rc != RB_OK) return rc;
  }
  const float* feat = l->act[L.nconv - 1];
  const int m_max = n_on > n_tg ? n_on : n_tg;
  const unsigned mchunks = (unsigned)rb_div_up(m_max, 64);
  if (l->fast_fc) {
    if (!l->fast_conv) {
      RB_LAUNCH(k_block_copy, dim3((unsigned)rb_div_up((int64_t)NI * L.F, 256)), dim3(256), stream, feat, NI, L.F, l->feat_b);
      RB_LAUNCH_CHECK();
    }
    // hidden layer: both streams, both nets, weights streamed once, bias + ReLU fused, no partials (noisy_linear.h)
    NlFwd2Args a;
    a.x = l->feat_b;
    a.m_base[0] = 0; a.m_cnt[0] = n_on; a.m_base[1] = n_on; a.m_cnt[1] = n_tg;
    a.w[0] = nl_h(on); a.w[1] = nl_h(tg);
    a.K = L.F; a.n_groups = 2;
    const int ht16 = (int)rb_div_up(L.H, 16);
    a.grp[0] = NlRowGroup{0, L.H, 0, 0, 0};
    a.grp[1] = NlRowGroup{L.H, L.H, 0, L.F, ht16};
    a.out = l->h; a.out_blocked = l->h_b; a.ld_out = 2 * L.H; a.rows_total = NI; a.relu = 1;
    // batch 256: 64-row m-chunks halve the passes over the weights (at batch 32 one 64-row chunk for the online net's rows
    // reads every tile once instead of twice and is 7 us per step SLOWER: a workgroup's MFMAs are serial on its CU)
    const bool wide = m_max >= 128;
    const unsigned mch32 = (unsigned)rb_div_up(m_max, wide ? 64 : RB_FWD2_MROWS);
    // (16-row m-chunks — 384 workgroups, every CU busy — measured 20.8 us against 16.2: the tiles are re-read four times)
    const dim3 hg((unsigned)(2 * ht16), 1, 2 * mch32), hb(64 * RB_NL_FWD_WAVES);
    // from 128 rows per net on the layer is a GEMM, not a weight stream: 128 x 128 LDS tiles, split-K over the idle CUs (fc_gemm.h)
    const bool gemm = l->gemm_part && (l->opt_fc_gemm == 1 || (l->opt_fc_gemm < 0 && m_max >= 128));
    if (gemm) {
      FcGemmFwdArgs ga;
      ga.f = a;
      ga.mt[0] = (int)rb_div_up(n_on, RB_TG_T); ga.mt[1] = (int)rb_div_up(n_tg, RB_TG_T);
      ga.nt = (int)rb_div_up(2 * L.H, RB_TG_T);
      const int tiles = (ga.mt[0] + ga.mt[1]) * ga.nt;
      int S = l->n_cu / tiles;
      if (S > 8) S = 8;
      if (S > L.F / RB_TG_KS) S = L.F / RB_TG_KS;
      if (S < 1 || tiles > 1024) S = 1;
      ga.S = S; ga.part = l->gemm_part; ga.ctr = l->gemm_ctr;
      RB_LAUNCH_T("fc_h_fwd:k_fc_gemm_fwd", k_fc_gemm_fwd, dim3((unsigned)(tiles * S)), dim3(RB_TG_THREADS), stream, ga);
    } else if (wide) { RB_LAUNCH_T("fc_h_fwd:k_nl_fwd3", k_nl_fwd3<4>, hg, hb, stream, a); }
    else { RB_LAUNCH_T("fc_h_fwd:k_nl_fwd3", k_nl_fwd3<2>, hg, hb, stream, a); }
    RB_LAUNCH_CHECK();
    // output layer: value rows read h[:, :H], advantage rows read h[:, H:]; bias fused
    NlFwd2Args z;
    z.x = l->h_b;
    z.m_base[0] = 0; z.m_cnt[0] = n_on; z.m_base[1] = n_on; z.m_cnt[1] = n_tg;
    z.w[0] = nl_z(on); z.w[1] = nl_z(tg);
    z.K = L.H; z.n_groups = 2;
    const int vt16 = (int)rb_div_up(L.Z, 16), at16 = (int)rb_div_up(L.NZ - L.Z, 16);
    z.grp[0] = NlRowGroup{0, L.Z, 0, 0, 0};
    z.grp[1] = NlRowGroup{L.Z, L.NZ - L.Z, L.H, L.H, vt16};
    z.out = l->logits; z.out_blocked = nullptr; z.ld_out = L.NZ; z.rows_total = NI; z.relu = 0;
    const dim3 zgrid((unsigned)(vt16 + at16), 1, 2 * mch32), zblock(64 * RB_NL_FWD_WAVES);
    if (wide) { RB_LAUNCH_T("fc_z_fwd:k_nl_fwd3", k_nl_fwd3<4>, zgrid, zblock, stream, z); }
    else { RB_LAUNCH_T("fc_z_fwd:k_nl_fwd3", k_nl_fwd3<2>, zgrid, zblock, stream, z); }
    RB_LAUNCH_CHECK();
    return RB_OK;
  }
  {
    FcHFwdProb p;
    p.F = L.F; p.H = L.H; p.NI = NI; p.splits = l->hs;
    p.n_img[0] = n_on; p.n_img[1] = n_tg; p.img_base[0] = 0; p.img_base[1] = n_on;
    p.feat = feat; p.net[0] = on; p.net[1] = tg; p.part = l->hpart;
    RB_LAUNCH((k_gemm<2, 2, FcHFwdProb>),
              dim3((unsigned)rb_div_up(m_max, 64), (unsigned)rb_div_up(2 * L.H, 64), (unsigned)(2 * l->hs)), dim3(256),
              stream, p);
    RB_LAUNCH_CHECK();
    const int64_t total = (int64_t)NI * 2 * L.H;
    RB_LAUNCH(k_fc_h_finish, dim3((unsigned)rb_div_up(total, 256)), dim3(256), stream, (const float*)l->hpart, l->hs, NI,
              2 * L.H, n_on, on, tg, l->h, (float*)nullptr);
    RB_LAUNCH_CHECK();
  }
  {
    FcZFwdProb p;
    p.H = L.H; p.Z = L.Z; p.NZ = L.NZ;
    p.n_img[0] = n_on; p.n_img[1] = n_tg; p.img_base[0] = 0; p.img_base[1] = n_on;
    p.h = l->h; p.net[0] = on; p.net[1] = tg; p.logits = l->logits;
    RB_LAUNCH((k_gemm<1, 1, FcZFwdProb>),
              dim3((unsigned)rb_div_up(m_max, 32), (unsigned)rb_div_up(L.NZ - L.Z, 32), 4), dim3(64), stream, p);
    RB_LAUNCH_CHECK();
  }
  return RB_OK;
}

// The data gradient of conv layer `layer` (>= 1) on the whole-K 16x16x4 tile kernel (conv_lds.h k_conv_dx_t16_multi): the image-loop
// form, i.e. batches of 64 and more (or RB_OPTS dx_ipb > 1, the test hook), the canonical later layers' geometries (64 output
// channels, kernel size a multiple of the stride).  RB_OPTS dx_t16=0: k_conv_dx_lds<..., MULTI>.  Decides the layout of conv_wT too.
static bool dx_uses_t16(const rb_learner* l, int layer) {
  const Layout& L = l->L;
  if (layer < 1 || layer >= L.nconv || !l->fast_conv || !l->conv_wT[layer] || !l->opt_dx_t16) return false;
  const ConvLayer& c = L.conv[layer];
  if (c.cout != 64 || c.cin % 32 != 0 || c.ks % c.s != 0) return false;
  if (!((c.ks == 4 && c.s == 2 && c.ih == 20) || (c.ks == 3 && c.s == 1 && c.ih == 9))) return false;      // GeomC2 / GeomC3 (ConvDxT16<G, 64>::OK)
  return L.B >= 64 || l->opt_dx_ipb > 1;
}

// mode bit 0: weight/bias grads (+ split reduction); bit 1: data grads into dact[layer-1]
template <class G>
static int launch_conv_bwd(rb_learner* l, int layer, const uint8_t* states, hipStream_t stream, int mode) {
  const Layout& L = l->L;
  const ConvLayer& c = L.conv[layer];
  const int K = c.K();
  float* gw = l->grads + L.conv_w[layer];
  float* gb = l->grads + L.conv_b[layer];
  const int splits = l->ws[layer];
  if (!(mode & 1)) {
  } else if (layer == 0) {
    ConvDwProb<G, true> p;
    p.B = L.B; p.cin = c.cin; p.cout = c.cout; p.splits = splits;
    p.dy = l->dact[0]; p.x_u8 = states; p.x_f = nullptr; p.part = l->dw_part[0];
    RB_LAUNCH((k_gemm<1, 2, ConvDwProb<G, true>>),
              dim3((unsigned)rb_div_up(c.cout, 32), (unsigned)rb_div_up(K + 1, 64), (unsigned)splits), dim3(128), stream, p);
  } else {
    ConvDwProb<G, false> p;
    p.B = L.B; p.cin = c.cin; p.cout = c.cout; p.splits = splits;
    p.dy = l->dact[layer]; p.x_u8 = nullptr; p.x_f = l->act[layer - 1]; p.part = l->dw_part[layer];
    RB_LAUNCH((k_gemm<2, 2, ConvDwProb<G, false>>),
              dim3((unsigned)rb_div_up(c.cout, 64), (unsigned)rb_div_up(K + 1, 64), (unsigned)splits), dim3(256), stream, p);
  }
  if (mode & 1) {
    RB_LAUNCH_CHECK();
    l->dw_slices[layer] = splits;   // summed by k_reduce_conv_dw_all after the last layer
    (void)gw; (void)gb;
  }
  if (!(mode & 2)) return RB_OK;
  if constexpr (G::IH == 84) {
    // first-layer geometries never need a data gradient (frames are not differentiated)
  } else if (layer > 0 && l->fast_conv && l->conv_wT[layer] && ((l->lazy_dfeat && layer == L.nconv - 1) == RB_LAST_CONV_GEOM(G))) {
    // (the last layer's LDS kernel exists in its LAZY form only — dY summed from the hidden layer's row-split partials while
    // it is staged; when those are not what the step produced, i.e. the generic FC path ran, the generic kernel below runs)
    ConvLdsDxArgs a;
    a.cin = c.cin; a.cout = c.cout;
    a.w = l->p_online + L.conv_w[layer]; a.dy = l->dact[layer]; a.x_act = l->act[layer - 1]; a.dx = l->dact[layer - 1];
    a.wT = l->conv_wT[layer];
    constexpr bool lazy = RB_LAST_CONV_GEOM(G);
    a.dy_part = l->dfeat_part; a.dy_mask = l->act[layer]; a.dy_stride = (int64_t)L.B * L.F; a.dy_splits = lazy ? l->lazy_splits : 0;
    constexpr int NPOS = ((G::IH + G::S - 1) / G::S) * ((G::IH + G::S - 1) / G::S);
    constexpr int NT_ALL = (NPOS + 31) / 32;
    // few images at batch 32: spread each phase's positions over several workgroups (weights are re-staged from L2)
    constexpr int NT = NT_ALL >= 4 ? 2 : 1;
    const unsigned groups = (unsigned)rb_div_up(NT_ALL, NT);
    static const char* const tags[3] = {"conv1_dx:k_conv_dx_lds", "conv2_dx:k_conv_dx_lds", "conv3_dx:k_conv_dx_lds"};
    // batches of 64 and more: about one round of workgroups over the chip, each keeping its weight slab for ipb images
    const int ipb_env = l->opt_dx_ipb;                                            // RB_DX_IPB (1 = one image each)
    const int per_img = (G::S * G::S) * (int)groups * (int)rb_div_up(c.cin, 32);
    int ipb = 1;
    if (ipb_env > 0) ipb = ipb_env;
    else if (L.B >= 64) {                             // ONE round of workgroups (their LDS footprint allows one per CU)
      while (per_img * (int)rb_div_up(L.B, ipb) > 256) ++ipb;
      // image-group-fastest order wants a group count that is a multiple of 8 — and the same groups as the next layer's launch and
      // the weight-gradient launch (8 images each at batch 256), so that a group's dY stays in one XCD's L2 down the chain
      if (l->opt_img_fast)
        while (ipb < L.B && (rb_div_up(L.B, ipb) % 8 != 0 || L.B % ipb != 0)) ++ipb;
    }
    a.ipb = ipb; a.batch = L.B;
    dim3 grid((unsigned)(G::S * G::S) * groups, (unsigned)rb_div_up(c.cin, 32), (unsigned)rb_div_up(L.B, ipb));
    a.img_fast = 0;
    if (l->opt_img_fast && L.B % ipb == 0 && (L.B / ipb) % 8 == 0) {      // image(-group)-fastest block order: image i on XCD i mod 8 in every conv launch
      a.img_fast = 1;
      grid = dim3((unsigned)(L.B / ipb), (unsigned)rb_div_up(c.cin, 32), (unsigned)(G::S * G::S) * groups);
    }
    if constexpr (ConvDxT16<G, 64>::OK) {
      if (dx_uses_t16(l, layer)) {
        // whole-K tiles: a workgroup per (phase, 32 input channels, image group), about one round of 256
        const int units = G::S * G::S * (int)rb_div_up(c.cin, 32);
        int tp = ipb_env > 0 ? ipb_env : (int)rb_div_up((int64_t)units * L.B, 256);
        if (tp < 1) tp = 1;
        if (ipb_env <= 0 && l->opt_img_fast)
          while (tp < L.B && (rb_div_up(L.B, tp) % 8 != 0 || L.B % tp != 0)) ++tp;
        a.ipb = tp;
        const unsigned ng = (unsigned)rb_div_up(L.B, tp);
        a.img_fast = (l->opt_img_fast && L.B % tp == 0 && ng % 8 == 0) ? 1 : 0;
        const dim3 gt = a.img_fast ? dim3(ng, (unsigned)rb_div_up(c.cin, 32), (unsigned)(G::S * G::S))
                                   : dim3((unsigned)(G::S * G::S), (unsigned)rb_div_up(c.cin, 32), ng);
        RB_LAUNCH_T(tags[layer], (k_conv_dx_t16_multi<G, 64, lazy>), gt, dim3(64 * ConvDxT16<G, 64>::NWV), stream, a);
        RB_LAUNCH_CHECK();
        return RB_OK;
      }
    }
    if (ipb > 1) { RB_LAUNCH_T(tags[layer], (k_conv_dx_lds<G, NT, 64, lazy, true>), grid, dim3(RB_CONV_THREADS), stream, a); }
    else { RB_LAUNCH_T(tags[layer], (k_conv_dx_lds<G, NT, 64, lazy, false>), grid, dim3(RB_CONV_THREADS), stream, a); }
    RB_LAUNCH_CHECK();
  } else if (layer > 0) {
    ConvDxProb<G> p;
    p.B = L.B; p.cin = c.cin; p.cout = c.cout;
    p.w = l->p_online + L.conv_w[layer]; p.dy = l->dact[layer]; p.x_act = l->act[layer - 1]; p.dx = l->dact[layer - 1];
    const int nyy = (G::IH + G::S - 1) / G::S;
    const int n_max = L.B * nyy * nyy;
    if (c.cin <= 32) {
      RB_LAUNCH((k_gemm<1, 2, ConvDxProb<G>>), dim3(1, (unsigned)rb_div_up(n_max, 64), (unsigned)(G::S * G::S)),
                dim3(128), stream, p);
    } else {
      RB_LAUNCH((k_gemm<2, 1, ConvDxProb<G>>),
                dim3((unsigned)rb_div_up(c.cin, 64), (unsigned)rb_div_up(n_max, 32), (unsigned)(G::S * G::S)), dim3(128),
                stream, p);
    }
    RB_LAUNCH_CHECK();
  }
  return RB_OK;
}

// all conv layers' weight gradients in one launch (LDS kernels); fills dw_slices for the fused slice reduction
static int conv_dw_all(rb_learner* l, hipStream_t stream) {
  const Layout& L = l->L;
  ConvDwAllArgs a;
  a.batch = L.B;
  const int ipb_all = L.B > 32 ? (int)rb_div_up(L.B, 32) : 1;      // keep about 32 image groups: the slice count stays at its batch-32 size
  bool uniform = true;
  unsigned total = 0;
  for (int i = 0; i < 3; ++i) a.ipb[i] = ipb_all;
  // Batches beyond 32, the canonical stack: images per workgroup chosen PER LAYER.  A workgroup walks its images one after the other
  // and the layers' images cost differently (7.5 / 7.4 / 6.2 us per image at batch 256, tools/wg_timeline.py): with 8 images
  // everywhere the launch was 224 workgroups of 60 / 59 / 50 us on 256 CUs; 7 / 7 / 8 images are 249 workgroups of 52 / 52 / 50 us
  // (-7.6 us per step, profiles/round6_dw_layer_ipb_ab.txt).  Smallest longest workgroup that still fits ONE round over the CUs.
  if (L.B > 32 && L.nconv == 3 && l->opt_dw_balance) {
    const int cost[3] = {75, 74, 62};
    const int chunks0 = (L.conv[0].oh + 6) / 7, ct[3] = {(int)rb_div_up(L.conv[0].cout, 32), (int)rb_div_up(L.conv[1].cout, 32), (int)rb_div_up(L.conv[2].cout, 32)};
    int best_t = ipb_all * cost[0], best[3] = {ipb_all, ipb_all, ipb_all};
    for (int i0 = 1; i0 <= ipb_all; ++i0)
      for (int i1 = 1; i1 <= ipb_all + 4; ++i1)
        for (int i2 = 1; i2 <= ipb_all + 4; ++i2) {
          const int wgs = chunks0 * ct[0] * (int)rb_div_up(L.B, i0) + ct[1] * (int)rb_div_up(L.B, i1) + ct[2] * (int)rb_div_up(L.B, i2);
          if (wgs > l->n_cu) continue;
          int t = i0 * cost[0];
          if (i1 * cost[1] > t) t = i1 * cost[1];
          if (i2 * cost[2] > t) t = i2 * cost[2];
          if (t < best_t) { best_t = t; best[0] = i0; best[1] = i1; best[2] = i2; }
        }
    for (int i = 0; i < 3; ++i) a.ipb[i] = best[i];
  }
  for (int i = 0; i < L.nconv; ++i) {
    if (l->opt_dw_ipb[i] > 0) a.ipb[i] = l->opt_dw_ipb[i];
    if (a.ipb[i] > L.B) a.ipb[i] = L.B;
    if (a.ipb[i] != ipb_all) uniform = false;
    const int groups = (int)rb_div_up(L.B, a.ipb[i]);
    const ConvLayer& c = L.conv[i];
    ConvLdsDwArgs& d = a.layer[i];
    d.cin = c.cin; d.cout = c.cout; d.dy = l->dact[i]; d.part = l->dw_part[i];
    d.src = l->cur_src; d.x_f = i > 0 ? l->act[i - 1] : nullptr;
    d.dy_part = l->dfeat_part; d.dy_mask = l->act[i]; d.dy_stride = (int64_t)L.B * L.F;
    d.dy_splits = (l->lazy_dfeat && i == L.nconv - 1 && i > 0) ? l->lazy_splits : 0;
    // first layer: 7-row chunks (3 per image) so that all layers together are 96 + 64 + 64 = 224 workgroups at batch 32,
    // ONE round over the 256 CUs (5-row chunks gave 288 workgroups at one per CU: a second round for 32 of them)
    const int rc = i == 0 ? (c.ks == 8 ? 7 : 4) : c.oh;                 // later layers: the whole image is one chunk
    const int chunks = (c.oh + rc - 1) / rc;
    a.cotiles[i] = (int)rb_div_up(c.cout, 32);
    a.nblocks[i] = chunks * a.cotiles[i] * groups;
    l->dw_slices[i] = chunks * groups;
    total += (unsigned)a.nblocks[i];
  }
  for (int i = L.nconv; i < 3; ++i) { a.nblocks[i] = 0; a.cotiles[i] = 1; a.layer[i] = a.layer[0]; }
  // image-fastest decode (an image group's workgroups of every layer on XCD group mod 8, where the input-gradient chain left
  // its dY): block ranges and the group count must be multiples of 8
  a.img_fast = (l->opt_img_fast && uniform && (int)rb_div_up(L.B, ipb_all) % 8 == 0 && a.nblocks[0] % 8 == 0 && a.nblocks[1] % 8 == 0) ? 1 : 0;
  // (a pipelined body — two operand sets in LDS, the next image's loads in flight under this image's MFMAs — was built in round 5,
  // bit-identical, and measured SLOWER at batch 256: 75.9 against 64.5 us for this launch, profiles/round5_experiments.txt; removed)
  if (L.nconv == 3) {
    RB_LAUNCH_T("conv_dw_all", (k_conv_dw_all<GeomC1, 7, GeomC2, 9, 512, GeomC3, 7, 576, 3>), dim3(total), dim3(RB_CONV_THREADS), stream, a);
  } else {
    RB_LAUNCH_T("conv_dw_all", (k_conv_dw_all<GeomD1, 4, GeomD2, 3, 800, GeomD2, 3, 800, 2>), dim3(total), dim3(RB_CONV_THREADS), stream, a);
  }
  RB_LAUNCH_CHECK();
  return RB_OK;
}

static int conv_bwd(rb_learner* l, int layer, const uint8_t* states, hipStream_t stream, int mode) {
  const ConvLayer& c = l->L.conv[layer];
  if (c.ks == 8) return launch_conv_bwd<GeomC1>(l, layer, states, stream, mode);
  if (c.ks == 4) return launch_conv_bwd<GeomC2>(l, layer, states, stream, mode);
  if (c.ks == 3) return launch_conv_bwd<GeomC3>(l, layer, states, stream, mode);
  if (c.ih == 84) return launch_conv_bwd<GeomD1>(l, layer, states, stream, mode);
  return launch_conv_bwd<GeomD2>(l, layer, states, stream, mode);
}

// torch.linspace(start, end, steps) float32 semantics (agent.py:18): step = (end-start)/(steps-1);
// first half counts up from start, second half counts down from end.
static void linspace_f32(float start, float end, int steps, float* out) {
  const float step = (end - start) / (float)(steps - 1);
  const int half = steps / 2;
  for (int i = 0; i < steps; ++i)
    out[i] = i < half ? start + step * (float)i : end - step * (float)(steps - i - 1);
}

extern "C" {

int rb_learner_sizes(const rb_learner_config_t* cfg, int64_t* n_params, int64_t* n_noise) {
  Layout L;
  int rc = make_layout(cfg, &L);
  if (rc != RB_OK) return rc;
  if (n_params) *n_params = L.n_params;
  if (n_noise) *n_noise = L.n_noise;
  return RB_OK;
}

static void set_desc(rb_tensor_desc_t* d, const char* name, int64_t off, int ndim, int s0, int s1, int s2, int s3) {
  memset(d, 0, sizeof(*d));
  snprintf(d->name, sizeof(d->name), "%s", name);
  d->offset = off; d->ndim = ndim;
  d->shape[0] = s0; d->shape[1] = s1; d->shape[2] = s2; d->shape[3] = s3;
}

int rb_learner_param_layout(const rb_learner_config_t* cfg, rb_tensor_desc_t* descs, int32_t* n) {
  Layout L;
  int rc = make_layout(cfg, &L);
  if (rc != RB_OK) return rc;
  RB_REQUIRE(n != nullptr, "rb_learner_param_layout: n is NULL");
  const int count = 2 * L.nconv + 16;
  if (!descs) { *n = count; return RB_OK; }
  RB_REQUIRE(*n >= count, "rb_learner_param_layout: need room for %d descriptors", count);
  int i = 0;
  char name[48];
  for (int l = 0; l < L.nconv; ++l) {
    const ConvLayer& c = L.conv[l];
    snprintf(name, sizeof(name), "convs.%d.weight", 2 * l);
    set_desc(&descs[i++], name, L.conv_w[l], 4, c.cout, c.cin, c.ks, c.ks);
    snprintf(name, sizeof(name), "convs.%d.bias", 2 * l);
    set_desc(&descs[i++], name, L.conv_b[l], 1, c.cout, 0, 0, 0);
  }
  const int64_t HF = (int64_t)L.H * L.F, ZH = (int64_t)L.Z * L.H;
  const int AZ = L.A * L.Z;
  set_desc(&descs[i++], "fc_h_v.weight_mu", L.h_mu, 2, L.H, L.F, 0, 0);
  set_desc(&descs[i++], "fc_h_v.weight_sigma", L.h_sigma, 2, L.H, L.F, 0, 0);
  set_desc(&descs[i++], "fc_h_v.bias_mu", L.h_bmu, 1, L.H, 0, 0, 0);
  set_desc(&descs[i++], "fc_h_v.bias_sigma", L.h_bsigma, 1, L.H, 0, 0, 0);
  set_desc(&descs[i++], "fc_h_a.weight_mu", L.h_mu + HF, 2, L.H, L.F, 0, 0);
  set_desc(&descs[i++], "fc_h_a.weight_sigma", L.h_sigma + HF, 2, L.H, L.F, 0, 0);
  set_desc(&descs[i++], "fc_h_a.bias_mu", L.h_bmu + L.H, 1, L.H, 0, 0, 0);
  set_desc(&descs[i++], "fc_h_a.bias_sigma", L.h_bsigma + L.H, 1, L.H, 0, 0, 0);
  set_desc(&descs[i++], "fc_z_v.weight_mu", L.z_mu, 2, L.Z, L.H, 0, 0);
  set_desc(&descs[i++], "fc_z_v.weight_sigma", L.z_sigma, 2, L.Z, L.H, 0, 0);
  set_desc(&descs[i++], "fc_z_v.bias_mu", L.z_bmu, 1, L.Z, 0, 0, 0);
  set_desc(&descs[i++], "fc_z_v.bias_sigma", L.z_bsigma, 1, L.Z, 0, 0, 0);
  set_desc(&descs[i++], "fc_z_a.weight_mu", L.z_mu + ZH, 2, AZ, L.H, 0, 0);
  set_desc(&descs[i++], "fc_z_a.weight_sigma", L.z_sigma + ZH, 2, AZ, L.H, 0, 0);
  set_desc(&descs[i++], "fc_z_a.bias_mu", L.z_bmu + L.Z, 1, AZ, 0, 0, 0);
  set_desc(&descs[i++], "fc_z_a.bias_sigma", L.z_bsigma + L.Z, 1, AZ, 0, 0, 0);
  *n = i;
  return RB_OK;
}

int rb_learner_noise_layout(const rb_learner_config_t* cfg, rb_tensor_desc_t* descs, int32_t* n) {
  Layout L;
  int rc = make_layout(cfg, &L);
  if (rc != RB_OK) return rc;
  RB_REQUIRE(n != nullptr, "rb_learner_noise_layout: n is NULL");
  if (!descs) { *n = 8; return RB_OK; }
  RB_REQUIRE(*n >= 8, "rb_learner_noise_layout: need room for 8 descriptors");
  int i = 0;
  set_desc(&descs[i++], "fc_h_v.eps_in", L.h_ein, 1, L.F, 0, 0, 0);
  set_desc(&descs[i++], "fc_h_v.eps_out", L.h_eout, 1, L.H, 0, 0, 0);
  set_desc(&descs[i++], "fc_h_a.eps_in", L.h_ein + L.F, 1, L.F, 0, 0, 0);
  set_desc(&descs[i++], "fc_h_a.eps_out", L.h_eout + L.H, 1, L.H, 0, 0, 0);
  set_desc(&descs[i++], "fc_z_v.eps_in", L.z_ein, 1, L.H, 0, 0, 0);
  set_desc(&descs[i++], "fc_z_v.eps_out", L.z_eout, 1, L.Z, 0, 0, 0);
  set_desc(&descs[i++], "fc_z_a.eps_in", L.z_ein + L.H, 1, L.H, 0, 0, 0);
  set_desc(&descs[i++], "fc_z_a.eps_out", L.z_eout + L.Z, 1, L.A * L.Z, 0, 0, 0);
  *n = i;
  return RB_OK;
}

int64_t rb_learner_noise_draws(const rb_learner_config_t* cfg) {
  Layout L;
  if (make_layout(cfg, &L) != RB_OK) return -1;
  return 2 * (int64_t)L.F + 2 * (int64_t)L.H + 2 * (int64_t)L.H + L.NZ;
}

int rb_learner_destroy(rb_learner_t* l) {
  if (!l) return RB_OK;
  float** owned[] = {&l->feat_b, &l->h_b, &l->act[0], &l->act[1], &l->act[2], &l->dact[0], &l->dact[1], &l->dact[2], &l->hpart, &l->h,
                     &l->logits, &l->dlogits, &l->dlogitsT, &l->dh, &l->dhT, &l->dfeat_part, &l->dw_part[0], &l->dw_part[1], &l->dw_part[2],
                     &l->conv_wT[0], &l->conv_wT[1], &l->conv_wT[2],
                     &l->log_ps_a, &l->pns_a, &l->m, &l->support, &l->zero_noise, &l->norm_part, &l->gemm_part, &l->noise_snap};
  for (float** p : owned)
    if (*p) rb_dev_free(*p);
  if (l->a_star) rb_dev_free(l->a_star);
  if (l->noise_ctr) rb_dev_free(l->noise_ctr);
  if (l->act_ctr) rb_dev_free(l->act_ctr);
  if (l->gemm_ctr) rb_dev_free(l->gemm_ctr);
  if (l->job_dev) rb_dev_free(l->job_dev);
  if (l->status_copy) rb_dev_free(l->status_copy);
  if (l->adam_args_dev) rb_dev_free(l->adam_args_dev);
  if (l->go_flag) rb_dev_free(l->go_flag);
  delete l;
  return RB_OK;
}

static int upload_noise_jobs(rb_learner* l);
int rb_learner_create(rb_learner_t** out, const rb_learner_config_t* cfg, float* online_params_dev,
                      float* target_params_dev, float* grads_dev, float* online_noise_dev, float* target_noise_dev,
                      uint64_t seed) {
  RB_REQUIRE(out && cfg && online_params_dev && target_params_dev && grads_dev && online_noise_dev && target_noise_dev,
             "rb_learner_create: NULL argument");
  Layout L;
  int rc = make_layout(cfg, &L);
  if (rc != RB_OK) return rc;
  rb_learner* l = new (std::nothrow) rb_learner();
  if (!l) { rb_set_error("rb_learner_create: host OOM"); return RB_ERR_OOM; }
  memset(l, 0, sizeof(*l));
  l->cfg = *cfg; l->L = L;
  l->p_online = online_params_dev; l->p_target = target_params_dev; l->grads = grads_dev;
  l->n_online = online_noise_dev; l->n_target = target_noise_dev;
  l->seed = seed; l->noise_epoch = 0;
  l->world = 1;
  {
    const int64_t seg[6] = {(int64_t)L.B * L.NZ, (int64_t)L.B * 2 * L.H, (int64_t)L.B * 2 * L.H, (int64_t)L.B * L.F, L.n_noise, L.h_mu};
    int64_t off = 0;
    for (int i = 0; i < 6; ++i) { l->fact_off[i] = off; off = align64(off + seg[i]); }
    l->fact_stride = off;
  }
  l->gamma_n = (float)pow(cfg->discount, (double)cfg->multi_step);
  l->delta_z = (float)(((double)cfg->v_max - (double)cfg->v_min) / (double)(cfg->atoms - 1));
  const int B = L.B, NI = 3 * B;
  // split-K factors: aim for >= ~2 workgroups per CU on the 256-CU part
  // RB_OPTS (rb_common.h): generic=1 forces the gemm_core fallback for every contraction, generic=2 for the noisy-linear
  // layers only (both exercised by the CPU tests); the rest are test hooks that force the large-batch paths onto small fixtures
  const int generic = rb_opt("generic", 0);
  l->fast_fc = (L.F % 32 == 0 && L.H % 32 == 0 && L.F <= RB_FWD2_KMAX && L.H <= RB_FWD2_KMAX && generic == 0) ? 1 : 0;
  l->fast_conv = (L.hist <= 4 && generic != 1) ? 1 : 0;
  l->opt_conv_multi = rb_opt("conv_multi", -1);       // images per workgroup of the conv forward (-1: by image count)
  l->opt_conv_multi_t16 = rb_opt("conv_multi_t16", 1);    // the image loop on whole-K 16x16x4 tiles (0: the split-K body)
  l->opt_dx_t16 = rb_opt("dx_t16", 1);                    // the image-loop conv data gradient on whole-K 16x16x4 tiles (0: the split-K body)
  l->opt_finish_tiled = rb_opt("finish_tiled", 1);        // rb_learner_finish_grads: the hidden layer's replica-mean weight gradient on 128 x 128 tiles
  l->opt_conv_full = rb_opt("conv_full", 1);          // first layer's whole-image kernel at large batches
  l->opt_wt_blocks = rb_opt("wt_blocks", 48);          // tenant workgroups of the head launch per weight-operand job
  l->opt_dw_balance = rb_opt("dw_balance", 1);         // 0: the same number of images per workgroup in every layer of the weight-gradient launch
  l->opt_dw_ipb[0] = rb_opt("dw_ipb0", 0); l->opt_dw_ipb[1] = rb_opt("dw_ipb1", 0); l->opt_dw_ipb[2] = rb_opt("dw_ipb2", 0);
  l->opt_dx_ipb = rb_opt("dx_ipb", 0);                // images per workgroup of the conv input gradients (0: by batch)
  l->opt_img_fast = rb_opt("img_fast", 1);            // image-fastest block order of the conv launches (0: the fallback order)
  l->opt_z_tall = rb_opt("z_tall", 1);
  l->opt_z_narrow = rb_opt("z_narrow", 1);
  l->opt_h_deep = rb_opt("h_deep", 1);
  l->opt_wb_auto = rb_opt("wb_auto", 1);
  l->opt_z_deep = rb_opt("z_deep", 1);
  l->opt_h_dw_deep = rb_opt("h_dw_deep", 1);
  l->opt_z_ct = rb_opt("z_ct", 2); if (l->opt_z_ct < 1) l->opt_z_ct = 1;
  l->opt_h_ct = rb_opt("h_ct", 4); if (l->opt_h_ct < 1) l->opt_h_ct = 1;
  l->opt_t16 = rb_opt("t16", 7);                      // conv forward layers on k_conv_fwd_t16 (bit per layer)
  l->opt_implicit_small = rb_opt("implicit_small", 0);   // test hook: RB_LEARNER_IMPLICIT_SIGMA on hidden layers of any size
  l->opt_fc_gemm = rb_opt("fc_gemm", -1);             // hidden layer as LDS-tiled GEMMs (fc_gemm.h): -1 = from 128 rows on
  if (l->fast_fc) {
    l->hs = pick_splits(2 * rb_div_up(L.H, 32) * 2 * rb_div_up(2 * B, 64), L.F / 16 / RB_NL_FWD_WAVES, 512);
    // input-gradient row splits: 256 weight rows per workgroup (64 per wave = 4 sixteen-row iterations); measured
    // 225.1 us per step against 228.3 with the former ~100-row splits (xs 10) and 239 without splitting
    // (at most 4: the consumers of the partials — the last conv layer's dX and dW kernels — sum up to 4 of them while staging)
    l->xs = (int)rb_div_up(2 * L.H, 256);
    if (l->xs > 4) l->xs = 4;
    l->xs = rb_opt("xs", l->xs);
  } else {
    l->hs = pick_splits(rb_div_up(2 * B, 64) * rb_div_up(2 * L.H, 64) * 2, (L.F + 15) / 16, 512);
    l->xs = pick_splits(rb_div_up(B, 32) * rb_div_up(L.F, 64), (2 * L.H + 15) / 16, 512);
  }
  for (int i = 0; i < L.nconv; ++i) {
    const ConvLayer& c = L.conv[i];
    const int64_t tiles = i == 0 ? rb_div_up(c.cout, 32) * rb_div_up(c.K() + 1, 64)
                                 : rb_div_up(c.cout, 64) * rb_div_up(c.K() + 1, 64);
    l->ws[i] = pick_splits(tiles, (B * c.P() + 15) / 16, 512);
  }
#define RB_ALLOC(ptr, count)                                                                         \
  do {                                                                                               \
    hipError_t e_ = rb_dev_malloc((void**)&(ptr), (size_t)(count) * 4);                                  \
    if (e_ != hipSuccess) {                                                                          \
      rb_set_error("rb_learner_create: hipMalloc(%lld B) failed: %s", (long long)(count) * 4, hipGetErrorString(e_)); \
      rb_learner_destroy(l);                                                                         \
      return RB_ERR_OOM;                                                                             \
    }                                                                                                \
  } while (0)
  for (int i = 0; i < L.nconv; ++i) {
    const ConvLayer& c = L.conv[i];
    RB_ALLOC(l->act[i], (int64_t)NI * c.cout * c.P());
    RB_ALLOC(l->dact[i], (int64_t)B * c.cout * c.P());
    {
      int64_t slices = l->ws[i];
      if (slices < (int64_t)B * 5) slices = (int64_t)B * 5;   // LDS weight-grad kernels: B images x <=5 row chunks
      RB_ALLOC(l->dw_part[i], slices * c.cout * (c.K() + 1));
    }
    if (i > 0 && l->fast_conv && c.cin % 32 == 0) {
      const int tmax = (c.ks + c.s - 1) / c.s;
      const int64_t n = (int64_t)c.s * c.s * (c.cin / 32) * (rb_div_up(c.cout * tmax * tmax, 16) * 16) * 32;
      RB_ALLOC(l->conv_wT[i], n);
      RB_HIP_TRY(hipMemset(l->conv_wT[i], 0, (size_t)n * 4));
    }
  }
  l->rows_cap = NI;
  RB_ALLOC(l->hpart, (int64_t)l->hs * NI * 2 * L.H);
  RB_ALLOC(l->h, (int64_t)NI * 2 * L.H);
  RB_ALLOC(l->feat_b, (int64_t)NI * (L.F + 16));
  RB_ALLOC(l->h_b, (int64_t)NI * (2 * L.H + 16));
  RB_ALLOC(l->logits, (int64_t)NI * L.NZ);
  RB_ALLOC(l->dlogits, (int64_t)B * L.NZ);
  RB_ALLOC(l->dlogitsT, (int64_t)B * L.NZ);
  RB_ALLOC(l->dh, (int64_t)B * 2 * L.H);
  RB_ALLOC(l->dhT, (int64_t)B * 2 * L.H);
  RB_ALLOC(l->dfeat_part, (int64_t)l->xs * B * L.F);
  RB_ALLOC(l->log_ps_a, (int64_t)B * L.Z);
  RB_ALLOC(l->pns_a, (int64_t)B * L.Z);
  RB_ALLOC(l->m, (int64_t)B * L.Z);
  RB_ALLOC(l->a_star, (int64_t)B);
  RB_ALLOC(l->support, (int64_t)L.Z);
  RB_ALLOC(l->zero_noise, L.n_noise);
  RB_ALLOC(l->noise_snap, L.n_noise);
  RB_ALLOC(l->norm_part, 16384);
  RB_ALLOC(l->noise_ctr, 4);
  RB_ALLOC(l->status_copy, 4);
  RB_ALLOC(l->adam_args_dev, (sizeof(ClipAdamArgs) + 3) / 4);
  RB_ALLOC(l->act_ctr, 6 * RB_FAN_SHARDS * RB_FAN_STRIDE + 32);
#undef RB_ALLOC
  RB_HIP_TRY(hipMemset(l->act_ctr, 0, (6 * RB_FAN_SHARDS * RB_FAN_STRIDE + 32) * 4));
  l->opt_act_fused = rb_opt("act_fused", 1);
  l->opt_spec_draw = rb_opt("spec_draw", RB_SPEC_DRAW_DEFAULT);
  l->opt_spec_stall = rb_opt("spec_stall", 0);
  if (l->opt_spec_draw) {
    hipError_t e = rb_dev_malloc((void**)&l->go_flag, 64);
    if (e != hipSuccess) { rb_set_error("rb_learner_create: hipMalloc failed: %s", hipGetErrorString(e)); rb_learner_destroy(l); return RB_ERR_OOM; }
    RB_HIP_TRY(hipMemset(l->go_flag, 0, 64));
  }
#if defined(RB_HOST_INTERP)
  l->n_cu = 8;
#else
  {
    int dev = 0, cus = 0;
    RB_HIP_TRY(hipGetDevice(&dev));
    RB_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    l->n_cu = cus;
  }
#endif
  if (l->fast_fc && l->opt_fc_gemm != 0) {
    // split-K scratch of k_fc_gemm_fwd: tiles * S <= n_cu whenever S > 1, one 128 x 128 partial tile each
    hipError_t e = rb_dev_malloc((void**)&l->gemm_part, (size_t)l->n_cu * RB_TG_T * RB_TG_T * 4);
    if (e == hipSuccess) e = rb_dev_malloc((void**)&l->gemm_ctr, 1024 * 4);
    if (e != hipSuccess) { rb_set_error("rb_learner_create: hipMalloc failed: %s", hipGetErrorString(e)); rb_learner_destroy(l); return RB_ERR_OOM; }
    RB_HIP_TRY(hipMemset(l->gemm_ctr, 0, 1024 * 4));
  }
  RB_HIP_TRY(hipMemset(l->noise_ctr, 0, 16));
  RB_HIP_TRY(hipMemset(l->status_copy, 0, 16));
  float sup[RB_MAX_ATOMS];
  linspace_f32(cfg->v_min, cfg->v_max, L.Z, sup);
  RB_HIP_TRY(hipMemcpy(l->support, sup, L.Z * sizeof(float), hipMemcpyHostToDevice));
  RB_HIP_TRY(hipMemset(l->zero_noise, 0, L.n_noise * sizeof(float)));
  RB_HIP_TRY(hipMemset(l->hpart, 0, (size_t)l->hs * NI * 2 * L.H * 4));
  {
    const int rc = upload_noise_jobs(l);
    if (rc != RB_OK) return rc;
  }
  *out = l;
  return RB_OK;
}

static NoiseMap noise_map(const Layout& L) {
  NoiseMap map;
  const int64_t counts[8] = {L.F, L.H, L.F, L.H, L.H, L.Z, L.H, (int64_t)L.A * L.Z};
  const int64_t dst[8] = {L.h_ein, L.h_eout, L.h_ein + L.F, L.h_eout + L.H, L.z_ein, L.z_eout, L.z_ein + L.H, L.z_eout + L.Z};
  map.seg_begin[0] = 0;
  for (int i = 0; i < 8; ++i) { map.seg_begin[i + 1] = map.seg_begin[i] + (int32_t)counts[i]; map.dst[i] = (int32_t)dst[i]; }
  return map;
}

static NoiseJob make_noise_job(rb_learner* l, int which) {
  NoiseJob j;
  j.noise = which == 1 ? l->n_target : l->n_online;
  j.noise2 = which == 2 ? l->n_target : nullptr;
  j.map = noise_map(l->L);
  j.seed = l->seed; j.ctr = l->noise_ctr;
  j.nblk = (int)rb_div_up(j.map.seg_begin[8], 256); j.nets = which == 2 ? 2 : 1;
  j.dev = l->job_dev + which;
  j.adam_dev = nullptr; j.adam_blocks = 0;
  return j;
}
// device copies of the three job variants (a hosting kernel reads them through NoiseJob::dev): at creation and whenever the
// seed changes — never from rb_learner_noise_job, which may be called while a stream is capturing
static int upload_noise_jobs(rb_learner* l) {
  if (!l->job_dev) RB_HIP_TRY(rb_dev_malloc((void**)&l->job_dev, 3 * sizeof(NoiseJob)));
  NoiseJob j[3];
  for (int which = 0; which < 3; ++which) j[which] = make_noise_job(l, which);
  RB_HIP_TRY(hipMemcpy(l->job_dev, j, sizeof(j), hipMemcpyHostToDevice));
  return RB_OK;
}

int rb_learner_noise_job(rb_learner_t* l, int32_t which, rb_noise_job_t* out) {
  RB_REQUIRE(l && out, "rb_learner_noise_job: NULL argument");
  RB_REQUIRE(which >= 0 && which <= 2, "rb_learner_noise_job: which must be 0 (online), 1 (target) or 2 (both)");
  static_assert(sizeof(NoiseJob) <= sizeof(rb_noise_job_t), "rb_noise_job_t too small");
  RB_REQUIRE(l->job_dev != nullptr, "rb_learner_noise_job: handle has no device job table");
  const NoiseJob j = make_noise_job(l, which);
  memset(out, 0, sizeof(*out));
  memcpy(out, &j, sizeof(j));
  return RB_OK;
}

int rb_learner_reset_noise(rb_learner_t* l, int32_t which, const float* raw_normals_dev, rb_stream_t stream) {
  RB_REQUIRE(l != nullptr, "rb_learner_reset_noise: NULL handle");
  RB_REQUIRE(which >= 0 && which <= 2, "rb_learner_reset_noise: which must be 0 (online), 1 (target) or 2 (both)");
  RB_REQUIRE(!(which == 2 && raw_normals_dev), "rb_learner_reset_noise: injected normals need one call per net");
  const NoiseMap map = noise_map(l->L);
  float* noise = which == 1 ? l->n_target : l->n_online;
  float* noise2 = which == 2 ? l->n_target : nullptr;
  RB_LAUNCH(k_noise, dim3((unsigned)rb_div_up(map.seg_begin[8], 256), which == 2 ? 2u : 1u), dim3(256), stream, noise, noise2,
            raw_normals_dev, map, l->seed, l->noise_ctr);
  RB_LAUNCH_CHECK();
  return RB_OK;
}

// One state through the act path (act_path.h).  RB_ERR_STATE (without touching the error string) = geometry not
// covered, the caller falls back to the training kernels.
// Returns RB_OK (logits ready: the caller launches the head), 1 (the one-launch path ran the head as well and wrote
// head_action_out / head_q_out), or an error.
static int act_forward_single(rb_learner* l, const float* state_dev, const NetPtrs& on, int noisy, hipStream_t stream,
                              int32_t* head_action_out, float* head_q_out) {
  const Layout& L = l->L;
  if (!l->fast_fc || (L.F & 3) || (L.H & 3)) return RB_ERR_STATE;   // RB_GENERIC_GEMM_ONLY=1 also lands here
  int rg[3];
  for (int layer = 0; layer < L.nconv; ++layer) {   // output rows per workgroup: <= 128 positions, patch fits the LDS
    const ConvLayer& c = L.conv[layer];
    int r = RB_ACT_MAXPOS / c.oh;
    if (r > c.oh) r = c.oh;
    while (r >= 1 && (int64_t)c.cin * ((r - 1) * c.s + c.ks) * c.ih > RB_ACT_LDS) --r;
    if (r < 1 || c.K() > RB_ACT_KMAX) return RB_ERR_STATE;
    rg[layer] = r;
  }
  ActFusedArgs f;
  memset(&f, 0, sizeof(f));
  const float* x = state_dev;
  for (int layer = 0; layer < L.nconv; ++layer) {
    const ConvLayer& c = L.conv[layer];
    ActConvArgs& a = f.conv[layer];
    a.x = x; a.w = on.conv_w[layer]; a.bias = on.conv_b[layer]; a.y = l->act[layer];
    a.cin = c.cin; a.cout = c.cout; a.KS = c.ks; a.S = c.s; a.IH = c.ih; a.OH = c.oh; a.RG = rg[layer];
    x = l->act[layer];
  }
  f.nconv = L.nconv;
  ActFcArgs& h = f.h;
  h.x = x; h.w = nl_h(on); h.K = L.F; h.n_rows = 2 * L.H; h.split_row = L.H; h.x_off1 = 0; h.ein_off1 = L.F;
  h.out = l->h; h.relu = 1; h.mu_only = noisy ? 0 : 1;
  ActFcArgs& z = f.z;
  z.x = l->h; z.w = nl_z(on); z.K = L.H; z.n_rows = L.NZ; z.split_row = L.Z; z.x_off1 = L.H; z.ein_off1 = L.H;
  z.out = l->logits; z.relu = 0; z.mu_only = noisy ? 0 : 1;
  bool can_fuse = l->opt_act_fused != 0;
  // (a captured launch would replay a stale launch number: under stream capture the per-layer launches below run instead)
  if (can_fuse && rb_stream_capturing(stream)) can_fuse = false;
  if (can_fuse) {
    // ONE persistent launch (act_path.h k_act_fused): G workgroups, one per CU, all resident — the in-launch waits need that
    f.Z = L.Z; f.A = L.A; f.logits = l->logits; f.support = l->support; f.action_out = head_action_out; f.q_out = head_q_out;
    f.ctr = l->act_ctr; f.err = l->act_ctr + 6 * RB_FAN_SHARDS * RB_FAN_STRIDE;
    int G = (l->n_cu < 256 ? l->n_cu : 256) / RB_FAN_SHARDS * RB_FAN_SHARDS;       // a multiple of the counter shards
    if (G < RB_FAN_SHARDS) G = RB_FAN_SHARDS;
#if defined(RB_HOST_INTERP)
    // the host interpreter runs workgroups one after the other: one launch per phase (no in-launch dependency), same bodies
    for (int ph = 0; ph < 6; ++ph) {
      if (ph < 3 && ph >= L.nconv) continue;
      f.phase_lo = ph; f.phase_hi = ph + 1; f.epoch = 0;
      RB_LAUNCH(k_act_fused<0>, dim3((unsigned)G), dim3(256), stream, f);
    }
#else
    f.phase_lo = 0; f.phase_hi = 6; f.epoch = l->act_epoch + 1;        // (counted below, once the launch is in the stream)
    const int hq = (int)rb_div_up(L.F, 256);
    if (hq <= 3) { RB_LAUNCH_T("act:k_act_fused", k_act_fused<3>, dim3((unsigned)G), dim3(256), stream, f); }
    else if (hq <= 13) { RB_LAUNCH_T("act:k_act_fused", k_act_fused<13>, dim3((unsigned)G), dim3(256), stream, f); }
    else { RB_LAUNCH_T("act:k_act_fused", k_act_fused<0>, dim3((unsigned)G), dim3(256), stream, f); }
#endif
    RB_LAUNCH_CHECK();
#if !defined(RB_HOST_INTERP)
    ++l->act_epoch;       // only a launch that went out advances the monotonic arrival targets (a refused one signalled nothing)
#endif
    return 1;                                              // the head ran inside the launch
  }
  for (int layer = 0; layer < L.nconv; ++layer) {
    const ConvLayer& c = L.conv[layer];
    RB_LAUNCH(k_act_conv, dim3((unsigned)c.cout, (unsigned)rb_div_up(c.oh, rg[layer])), dim3(256), stream, f.conv[layer]);
    RB_LAUNCH_CHECK();
  }
  RB_LAUNCH(k_act_fc, dim3((unsigned)rb_div_up(h.n_rows, 4)), dim3(256), stream, h);
  RB_LAUNCH_CHECK();
  RB_LAUNCH(k_act_fc, dim3((unsigned)rb_div_up(z.n_rows, 4)), dim3(256), stream, z);
  RB_LAUNCH_CHECK();
  return RB_OK;
}

int rb_learner_act(rb_learner_t* l, const float* state_dev, int32_t noisy, int32_t* action_dev, float* q_dev,
                   rb_stream_t stream) {
  RB_REQUIRE(l && state_dev, "rb_learner_act: NULL argument");
  RB_FLUSH_UPDATE(l, stream);
  const Layout& L = l->L;
  ImgSrc src;
  memset(&src, 0, sizeof(src));
  src.f32 = state_dev; src.B = 1;
  const NetPtrs on = net_ptrs(L, l->p_online, noisy ? l->n_online : l->zero_noise);
  int rc = act_forward_single(l, state_dev, on, noisy, (hipStream_t)stream, action_dev, q_dev);
  if (rc == 1) return RB_OK;                                                          // one launch, head included
  if (rc == RB_ERR_STATE) rc = forward(l, 1, 0, src, on, on, (hipStream_t)stream);   // geometry outside the act path
  if (rc != RB_OK) return rc;
  RB_LAUNCH(k_head_act, dim3(1), dim3(256), stream, L.Z, L.A, (const float*)l->logits, 0, (const float*)l->support,
            action_dev, q_dev);
  RB_LAUNCH_CHECK();
  return RB_OK;
}

// rb_learner_act + waiting for its result on the host, in one call (include/rainbow_hip.h): the action word is preset, the launch
// goes out, and the pinned word is polled HERE — a compiled loop sees the head's store within tens of nanoseconds, a Python loop
// over a numpy scalar within a microsecond or two, and the caller saves the interpreter's share of a 43 us act().
int rb_learner_act_wait(rb_learner_t* l, const float* state_dev, int32_t noisy, int32_t* action_pinned, float* q_pinned,
                        int32_t* action_out, float* q_out, rb_stream_t stream) {
  RB_REQUIRE(l && state_dev && action_pinned && q_pinned, "rb_learner_act_wait: NULL argument");
  constexpr int32_t PENDING = -7;
  int attempts = 0;
  for (;;) {
    *(volatile int32_t*)action_pinned = PENDING;
    const int rc = rb_learner_act(l, state_dev, noisy, action_pinned, q_pinned, stream);
    if (rc != RB_OK) return rc;
#if !defined(RB_HOST_INTERP)
    bool seen = false;
    for (long spin = 0; spin < 4000000L; ++spin) {           // ~10 ms of polling, then the stream is synchronised instead
      if (*(volatile int32_t*)action_pinned != PENDING) { seen = true; break; }
    }
    if (!seen) RB_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
#endif
    int32_t a = *(volatile int32_t*)action_pinned;
#if !defined(RB_HOST_INTERP)
    if (a < 0) {                                             // an error code (or a torn view): the final word after a synchronise
      RB_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
      a = *(volatile int32_t*)action_pinned;
    }
#endif
    if (a >= 0) {
      if (action_out) *action_out = a;
      if (q_out) *q_out = *(volatile float*)q_pinned;        // (the head stores q, fences, then the action)
      return RB_OK;
    }
    // the one-launch path reported an expired in-launch wait of THAT launch (its workgroups were not co-resident); the failure is
    // tagged with the launch number, so the next launch starts clean: once more, then give up
    if (++attempts >= 2) {
      rb_set_error("rb_learner_act_wait: the one-launch act path reported an expired in-launch wait twice (action %d); "
                   "RB_OPTS=act_fused=0 selects the per-layer launches", (int)a);
      return RB_ERR_STATE;
    }
  }
}

// The forward buffers are sized for the learn step's 3B images; batched evaluation (test.py:38-39 over a 500-state
// validation memory) may ask for more rows: grow them (synchronising; happens once per size).
static int ensure_rows(rb_learner* l, int rows) {
  if (rows <= l->rows_cap) return RB_OK;
  const Layout& L = l->L;
  RB_HIP_TRY(hipDeviceSynchronize());
  auto regrow = [&](float** p, int64_t count) -> int {
    if (*p) rb_dev_free(*p);
    *p = nullptr;
    hipError_t e = rb_dev_malloc((void**)p, (size_t)count * 4);
    if (e != hipSuccess) { rb_set_error("rb_learner_act_batch: hipMalloc(%lld B) failed: %s", (long long)count * 4, hipGetErrorString(e)); return RB_ERR_OOM; }
    return RB_OK;
  };
  int rc;
  for (int i = 0; i < L.nconv; ++i)
    if ((rc = regrow(&l->act[i], (int64_t)rows * L.conv[i].cout * L.conv[i].P())) != RB_OK) return rc;
  if ((rc = regrow(&l->hpart, (int64_t)l->hs * rows * 2 * L.H)) != RB_OK) return rc;
  if ((rc = regrow(&l->h, (int64_t)rows * 2 * L.H)) != RB_OK) return rc;
  if ((rc = regrow(&l->feat_b, (int64_t)rows * (L.F + 16))) != RB_OK) return rc;
  if ((rc = regrow(&l->h_b, (int64_t)rows * (2 * L.H + 16))) != RB_OK) return rc;
  if ((rc = regrow(&l->logits, (int64_t)rows * L.NZ)) != RB_OK) return rc;
  RB_HIP_TRY(hipMemset(l->hpart, 0, (size_t)l->hs * rows * 2 * L.H * 4));
  l->rows_cap = rows;
  return RB_OK;
}

int rb_learner_act_batch(rb_learner_t* l, const float* states_dev, int32_t n, int32_t noisy, int32_t* actions_dev,
                         float* q_dev, rb_stream_t stream) {
  RB_REQUIRE(l && states_dev, "rb_learner_act_batch: NULL argument");
  RB_FLUSH_UPDATE(l, stream);
  const Layout& L = l->L;
  RB_REQUIRE(n >= 1 && n <= 4096, "rb_learner_act_batch: n must be in [1, 4096]");
  {
    int rc0 = ensure_rows(l, n);
    if (rc0 != RB_OK) return rc0;
  }
  if (n == 1) return rb_learner_act(l, states_dev, noisy, actions_dev, q_dev, stream);
  ImgSrc src;
  memset(&src, 0, sizeof(src));
  src.f32 = states_dev; src.B = n;
  const NetPtrs on = net_ptrs(L, l->p_online, noisy ? l->n_online : l->zero_noise);
  int rc = forward(l, n, 0, src, on, on, (hipStream_t)stream);     // the training kernels: n images share every weight read
  if (rc != RB_OK) return rc;
  RB_LAUNCH(k_head_act, dim3((unsigned)n), dim3(256), stream, L.Z, L.A, (const float*)l->logits, 0, (const float*)l->support,
            actions_dev, q_dev);
  RB_LAUNCH_CHECK();
  return RB_OK;
}

static int learn_impl(rb_learner_t* l, const ImgSrc& src, const uint8_t* states_dev, const int64_t* actions_dev,
                      const float* returns_dev, const float* nonterminals_dev, const float* weights_dev, float* loss_dev,
                      hipStream_t stream);

// Weight-gradient problem of one noisy layer pair (which = 0: fc_z_v | fc_z_a, 1: fc_h_v | fc_h_a) over M reduction rows
// of dy / x.  ct > 0 selects the pipelined body (M <= 32).  slots = sum-of-squares partials the launch writes.
struct FcDwPlan {
  NlDwArgs a;
  int dw_x, dw_y, slots;
};
static FcDwPlan fc_dw_plan(rb_learner* l, const NetPtrs& on, int which, const float* dy, const float* x, int M, int ct) {
  const Layout& L = l->L;
  FcDwPlan p;
  NlDwArgs& w = p.a;
  memset(&w, 0, sizeof(w));
  w.dy = dy; w.x = x; w.M = M; w.n_prob = 2; w.ct = ct; w.rpb = 0; w.bstride = 0; w.scale = 1.0f; w.sq_part = nullptr;
  w.noise_blocks = nullptr; w.eout_noff = 0; w.ein_noff = 0; w.norm_only = 0; w.no_sigma = 0;
  if (which == 0) {
    const int vt = (int)rb_div_up(L.Z, 16), at = (int)rb_div_up(L.NZ - L.Z, 16);
    w.ldy = L.NZ; w.ldx = 2 * L.H; w.K = L.H;
    w.prob[0] = NlDwProblem{0, L.Z, 0, 0, 0};
    w.prob[1] = NlDwProblem{L.Z, L.NZ - L.Z, L.H, L.H, vt};
    w.g_mu = l->grads + L.z_mu; w.g_sigma = l->grads + L.z_sigma; w.g_bmu = l->grads + L.z_bmu; w.g_bsigma = l->grads + L.z_bsigma;
    w.eout = on.z_eout; w.ein = on.z_ein;
    p.dw_y = vt + at;
  } else {
    const int ht = (int)rb_div_up(L.H, 16);
    w.ldy = 2 * L.H; w.ldx = L.F; w.K = L.F;
    w.prob[0] = NlDwProblem{0, L.H, 0, 0, 0};
    w.prob[1] = NlDwProblem{L.H, L.H, 0, L.F, ht};
    w.g_mu = l->grads + L.h_mu; w.g_sigma = l->grads + L.h_sigma; w.g_bmu = l->grads + L.h_bmu; w.g_bsigma = l->grads + L.h_bsigma;
    w.eout = on.h_eout; w.ein = on.h_ein;
    p.dw_y = 2 * ht;
  }
  p.dw_x = (int)rb_div_up(w.K, 256 * (ct > 0 ? ct : 1));
  p.slots = 4 * p.dw_x * p.dw_y;
  return p;
}

int rb_learner_learn(rb_learner_t* l, const uint8_t* states_dev, const uint8_t* next_states_dev,
                     const int64_t* actions_dev, const float* returns_dev, const float* nonterminals_dev,
                     const float* weights_dev, float* loss_dev, rb_stream_t stream_) {
  RB_REQUIRE(l && states_dev && next_states_dev && actions_dev && returns_dev && nonterminals_dev && weights_dev && loss_dev,
             "rb_learner_learn: NULL argument");
  RB_FLUSH_UPDATE(l, stream_);
  ImgSrc src;
  memset(&src, 0, sizeof(src));
  src.u8_states = states_dev; src.u8_next = next_states_dev; src.B = l->L.B;
  return learn_impl(l, src, states_dev, actions_dev, returns_dev, nonterminals_dev, weights_dev, loss_dev, (hipStream_t)stream_);
}

int rb_learner_learn_windows(rb_learner_t* l, const uint8_t* frames_dev, const int32_t* windows_dev, int32_t window_len,
                             const int64_t* actions_dev, const float* returns_dev, const float* nonterminals_dev,
                             const float* weights_dev, float* loss_dev, rb_stream_t stream_) {
  RB_REQUIRE(l && frames_dev && windows_dev && actions_dev && returns_dev && nonterminals_dev && weights_dev && loss_dev,
             "rb_learner_learn_windows: NULL argument");
  RB_FLUSH_UPDATE(l, stream_);
  RB_REQUIRE(window_len == l->L.hist + l->cfg.multi_step, "rb_learner_learn_windows: window_len must be history + multi_step");
  if (!l->fast_conv) {
    rb_set_error("rb_learner_learn_windows: zero-copy frames need the LDS conv kernels (history <= 4); gather the stacks and "
                 "call rb_learner_learn instead");
    return RB_ERR_STATE;
  }
  ImgSrc src;
  memset(&src, 0, sizeof(src));
  src.B = l->L.B; src.ring = frames_dev; src.win = windows_dev; src.win_len = window_len; src.n_step = l->cfg.multi_step;
  return learn_impl(l, src, nullptr, actions_dev, returns_dev, nonterminals_dev, weights_dev, loss_dev, (hipStream_t)stream_);
}

static int learn_impl(rb_learner_t* l, const ImgSrc& src, const uint8_t* states_dev, const int64_t* actions_dev,
                      const float* returns_dev, const float* nonterminals_dev, const float* weights_dev, float* loss_dev,
                      hipStream_t stream) {
  const Layout& L = l->L;
  const int B = L.B;
  l->cur_src = src;
  const NetPtrs on = net_ptrs(L, l->p_online, l->n_online);
  const NetPtrs tg = net_ptrs(L, l->p_target, l->n_target);
  int rc = forward(l, 2 * B, B, src, on, tg, stream);
  if (rc != RB_OK) return rc;
  {
    // threads: one wave per softmax task (2A + 1) up to 16 waves, at least 8 (the logits sweep and the dlogits store want lanes)
    int hwaves = 2 * L.A + 1;
    if (hwaves < 8) hwaves = 8;
    if (hwaves > 16) hwaves = 16;
    HeadTenants tn;
    memset(&tn, 0, sizeof(tn));
    int n_jobs = 0;
    if (l->fast_conv) {
      for (int layer = 1; layer < L.nconv && n_jobs < 2; ++layer) {
        if (!l->conv_wT[layer]) continue;
        const ConvLayer& c = L.conv[layer];
        const int tmax = (c.ks + c.s - 1) / c.s;
        tn.job[n_jobs++] = ConvWtJob{on.conv_w[layer], l->conv_wT[layer], c.cin, c.cout, c.ks, c.s, (int)rb_div_up(c.cout * tmax * tmax, 16) * 16,
                                     dx_uses_t16(l, layer) ? 1 : 0};
      }
      if (n_jobs == 1) tn.job[1] = tn.job[0];
      tn.per_job = n_jobs > 0 ? l->opt_wt_blocks : 0;                  // (one element or two per thread: the tenants must stay shorter than the head)
    }
    const dim3 hgrid((unsigned)(B + (n_jobs > 0 ? 2 * tn.per_job : 0))), hblock((unsigned)(64 * hwaves));
#define RB_HEAD_ARGS B, L.Z, L.A, (const float*)l->logits, actions_dev, returns_dev, nonterminals_dev, weights_dev, (const float*)l->support, \
    l->cfg.v_min, l->cfg.v_max, l->gamma_n, l->delta_z, l->log_ps_a, l->pns_a, l->m, l->a_star, loss_dev, l->dlogits, l->step_ctr,          \
    l->batch_status, l->status_copy, l->dlogitsT, tn
    if (L.Z <= 64) { RB_LAUNCH_T("head:k_head", k_head<1>, hgrid, hblock, stream, RB_HEAD_ARGS); }
    else if (L.Z <= 128) { RB_LAUNCH_T("head:k_head", k_head<2>, hgrid, hblock, stream, RB_HEAD_ARGS); }
    else { RB_LAUNCH_T("head:k_head", k_head<4>, hgrid, hblock, stream, RB_HEAD_ARGS); }
#undef RB_HEAD_ARGS
  }
  RB_LAUNCH_CHECK();

  // ---- backward (online net, images [0,B)).  The input-gradient chain (fc_z dX -> fc_h dX -> conv dX ...) is the critical
  // path; the weight-gradient work rides in the same launches as block ranges (side streams measured slower, round 1).
  const float* feat = l->act[L.nconv - 1];
  const bool exch = l->world > 1 && l->fact_local != nullptr && l->fast_fc;   // replica exchange: FC weight grads deferred
  l->exch_pending = 0;
  l->dw_deferred = 0;
  if (l->fast_fc) {
    // ---- output layer: weight/bias grads and (ReLU-masked) input grads in one launch
    // sum-of-squares slots (clip_grad_norm_ without re-reading the gradient): [fc_z dW waves | fc_h dW waves | conv reduce blocks]
    // pipelined weight-gradient body (one reduction pass per tile, i.e. batch <= 32): column tiles per wave
    const bool pipe = B <= 32 && !exch;
    const int z_ct = pipe ? l->opt_z_ct : 0, h_ct = pipe ? l->opt_h_ct : 0;     // (RB_OPTS z_ct / h_ct: 256-column tiles per wave and workgroup)
    FcDwPlan zp = fc_dw_plan(l, on, 0, l->dlogits, l->h, B, z_ct);
    FcDwPlan hp = fc_dw_plan(l, on, 1, l->dh, feat, B, h_ct);
    hp.a.deep = l->opt_h_dw_deep;
    // batch >= 128: the hidden layer's two gradients as LDS-tiled GEMMs in one launch (fc_gemm.h k_fc_gemm_bwd)
    const bool gemm_bwd = !exch && l->gemm_part && (l->opt_fc_gemm == 1 || (l->opt_fc_gemm < 0 && B >= 128));
    const int g_nt = (int)rb_div_up(2 * L.H, RB_TG_T), g_kt = (int)rb_div_up(L.F, RB_TG_T);
    if (gemm_bwd) hp.slots = 8 * g_nt * g_kt;             // one sum-of-squares slot per wave of a weight-gradient tile
    int64_t conv_out = 0;
    for (int layer = 0; layer < L.nconv; ++layer) conv_out += (int64_t)L.conv[layer].cout * (L.conv[layer].K() + 1);
    const int c_slots = (int)rb_div_up(conv_out, 64);
    const bool fuse_norm = !exch && zp.slots + hp.slots + c_slots <= 16384;
    NlDwArgs& zw = zp.a;
    NlDwArgs& hw_ = hp.a;
    const bool defer_dw = (l->flags & RB_LEARNER_FUSE_FC_H_DW) && pipe && h_ct > 0 && fuse_norm && l->fast_conv;
    hw_.norm_only = defer_dw ? 1 : 0;
    l->dw_deferred = defer_dw ? 1 : 0;
    // RB_LEARNER_IMPLICIT_SIGMA: g_sigma = g_mu * (eps_out x eps_in) is left to the optimiser pass (its square still enters
    // the norm here).  Needs the pipelined weight-gradient body (batch <= 32) or the tiled GEMM (batch >= 128), the fused norm and
    // adjacent mu | sigma arrays
    const bool implicit_sigma = (l->flags & RB_LEARNER_IMPLICIT_SIGMA) && ((pipe && h_ct > 0) || gemm_bwd) && fuse_norm && !defer_dw &&
                                L.h_sigma == L.h_mu + (int64_t)2 * L.H * L.F && (L.F % 4) == 0 && (L.h_mu % 4) == 0 &&
                                ((int64_t)2 * L.H * L.F >= ((int64_t)1 << 20) || l->opt_implicit_small);   // (the data-efficient
                                // net's 0.3 M-element layer: +0.8 us per step with the pairing — it pays from megabytes on)
    hw_.no_sigma = implicit_sigma ? 1 : 0;
    l->sigma_implicit = implicit_sigma ? 1 : 0;
    zw.sq_part = fuse_norm ? l->norm_part : nullptr;
    hw_.sq_part = fuse_norm ? l->norm_part + zp.slots : nullptr;
    l->norm_slots = fuse_norm ? zp.slots + hp.slots + c_slots : 0;
    l->norm_conv_base = zp.slots + hp.slots;
    const int vt = (int)rb_div_up(L.Z, 16), at = (int)rb_div_up(L.NZ - L.Z, 16), ht = (int)rb_div_up(L.H, 16);
    NlDxArgs zx;
    zx.dy = l->dlogits; zx.ldy = L.NZ; zx.M = B; zx.w = nl_z(on); zx.K = L.H; zx.n_prob = 2;
    zx.prob[0] = NlDxProblem{0, L.Z, 1 << 30, 0, 0, 0};
    zx.prob[1] = NlDxProblem{L.Z, L.NZ - L.Z, 1 << 30, L.H, L.H, L.H};
    zx.rows_per_split = (int)rb_div_up(L.NZ, 16) * 16;
    zx.out = l->dh; zx.ld_out = 2 * L.H; zx.mask_src = l->h;
    zx.dyT = l->dlogitsT; zx.ldyT = B; zx.outT = l->dhT;
    // the output layer's input gradient with eight waves per workgroup (noisy_linear.h rb_nl_dx_body_tall) at batch <= 32
    // (RB_OPTS z_tall=0: the four-wave body)
    const int z_tall = (l->opt_z_tall && B <= 32) ? 1 : 0;
    // ... on 32-column tiles (RB_OPTS z_narrow=0: 64-column tiles): twice the workgroups, half the weight bytes through each CU
    const int z_narrow = (z_tall && l->opt_z_narrow && (L.H % 32) == 0) ? 1 : 0;
    NlBwdGrid zg{exch ? 0 : zp.dw_x, exch ? 0 : vt + at, (int)rb_div_up(L.H, z_narrow ? 32 : 64), 1, 2 * (int)rb_div_up(B, 64), z_tall ? z_narrow : (l->opt_z_deep ? 2 : 0)};
    // ---- hidden layer
    NlDxArgs hx;
    hx.dy = l->dh; hx.ldy = 2 * L.H; hx.M = B; hx.w = nl_h(on); hx.K = L.F; hx.n_prob = 1;
    hx.prob[0] = NlDxProblem{0, 2 * L.H, L.H, 0, L.F, 0};
    hx.prob[1] = hx.prob[0];
    hx.rows_per_split = (int)rb_div_up(rb_div_up(2 * L.H, l->xs), 16) * 16;
    const int hsplits = (int)rb_div_up(2 * L.H, hx.rows_per_split);
    hx.out = l->dfeat_part; hx.ld_out = L.F; hx.mask_src = nullptr;
    hx.dyT = l->dhT; hx.ldyT = B; hx.outT = nullptr;
    NlBwdGrid hg{exch ? 0 : hp.dw_x, exch ? 0 : hp.dw_y, (int)rb_div_up(L.F, 64), hsplits, (int)rb_div_up(B, 64), (B <= 32 && l->opt_h_deep) ? 1 : 0};
    NlPriorityUpdate up;
    memset(&up, 0, sizeof(up));
    // the write-back leaves this launch for the replay's stream (decided HERE, once: an expiry seen later only affects the next call)
    const bool spec = l->spec_now && l->sink && B <= 256 && !exch && rb_replay_spec_allowed(l->sink);
    if (l->sink && B <= 256 && !spec) {
      up.enabled = l->opt_wb_auto ? 2 : 1; up.tree_idx = l->sink_idx; up.loss = loss_dev; up.n = B;
      if (rb_replay_internal_view(l->sink, &up.view, &up.omega) != RB_OK) {
        rb_set_error("rb_learner_learn: bad priority sink");
        return RB_ERR_STATE;
      }
    }
    {
      NlPriorityUpdate none;
      memset(&none, 0, sizeof(none));
      if (spec) { none.go_flag = l->opt_spec_stall ? nullptr : l->go_flag; none.go_epoch = ++l->go_epoch; }
      const dim3 zgrid_((unsigned)(zg.dw_x * zg.dw_y + zg.dx_x * zg.dx_y * zg.dx_z));
      if (z_tall) { RB_LAUNCH_T("fc_z_bwd:k_nl_bwd", k_nl_bwd<true>, zgrid_, dim3(64 * RB_NL_DXT_WAVES), stream, zw, zx, zg, none); }
      else { RB_LAUNCH_T("fc_z_bwd:k_nl_bwd", k_nl_bwd<false>, zgrid_, dim3(256), stream, zw, zx, zg, none); }
      if (spec) {
        // the head is complete once the launch above has started: the write-back of THIS call and the draw of the NEXT one, on the
        // replay's stream, behind that launch's flag (submitted after it: a serialising profiler still terminates)
        RB_LAUNCH_CHECK();
        rb_spec_request q = l->spec_req;
        q.upd_idx = l->sink_idx; q.upd_loss = loss_dev; q.upd_n = B;
        q.go_flag = l->go_flag; q.go_epoch = l->go_epoch;
        const int rcs = rb_replay_spec_launch(l->sink, q);
        if (rcs != RB_OK) return rcs;
      }
      if (exch) {
        // every factor of the FC weight gradients exists now (dlogits, h, dh, feat rows [0, B)): pack them into this rank's
        // exchange block; the conv gradients join it at the end of the backward (k_reduce_conv_dw_all stores them twice)
        PackArgs pk;
        pk.src[0] = l->dlogits; pk.src[1] = l->h; pk.src[2] = l->dh; pk.src[3] = feat; pk.src[4] = l->n_online;
        pk.count[0] = (int64_t)B * L.NZ; pk.count[1] = (int64_t)B * 2 * L.H; pk.count[2] = (int64_t)B * 2 * L.H; pk.count[3] = (int64_t)B * L.F;
        pk.count[4] = L.n_noise;
        for (int i = 0; i < 5; ++i) pk.dst_off[i] = l->fact_off[i];
        pk.dst = l->fact_local;
        RB_LAUNCH(k_pack_factors, dim3(16, 5), dim3(256), stream, pk);
        RB_LAUNCH_CHECK();
        l->exch_pending = 1;
      }
      // the priority write-back (a single-workgroup latency chain of ~11 us) rides in the LONGER of the two backward
      // launches: as a tenant of the output layer's launch (~8 us of real work) it was that launch's long pole
      const unsigned h_blocks = (unsigned)(hg.dw_x * hg.dw_y + hg.dx_x * hg.dx_y * hg.dx_z + (up.enabled ? 1 : 0));
      if (gemm_bwd) {
        FcGemmBwdGrid gg;
        gg.first = up.enabled ? 8 : 0;
        gg.dx_mt = (int)rb_div_up(B, RB_TG_T); gg.dx_kt = g_kt; gg.dx_splits = hsplits;
        gg.dx_combos = (int)rb_div_up(g_kt * hsplits, 8) * 8;
        gg.dw_nt = g_nt; gg.dw_kt = g_kt;
        const unsigned gb = (unsigned)(gg.first + gg.dx_mt * gg.dx_combos + g_nt * g_kt);
        RB_LAUNCH_T("fc_h_bwd:k_fc_gemm_bwd", k_fc_gemm_bwd, dim3(gb), dim3(RB_TG_THREADS), stream, hw_, hx, gg, up);
      } else if (h_blocks > 0) { RB_LAUNCH_T("fc_h_bwd:k_nl_bwd", k_nl_bwd<false>, dim3(h_blocks), dim3(256), stream, hw_, hx, hg, up); }
    }
    l->sink_done = (up.enabled || spec) ? 1 : 0;
    RB_LAUNCH_CHECK();
    // d(conv output) = relu' * sum of the row-split partials: formed by its two consumers (the last conv layer's dX and
    // dW kernels) while they stage it, instead of a ~5 us launch of its own between two dependent kernels
    l->lazy_dfeat = (l->fast_conv && L.nconv >= 2 && hsplits <= 4) ? 1 : 0;
    l->lazy_splits = hsplits;
    if (!l->lazy_dfeat) {
      const int64_t total = (int64_t)B * L.F;
      RB_LAUNCH(k_dfeat_finish, dim3((unsigned)rb_div_up(total, 256)), dim3(256), stream, (const float*)l->dfeat_part,
                hsplits, total, feat, l->dact[L.nconv - 1]);
      RB_LAUNCH_CHECK();
    }
  } else {
  l->lazy_dfeat = 0;
  l->norm_slots = 0;
  l->sink_done = 0;
  FcGradOut gz;
  gz.g_mu = l->grads + L.z_mu; gz.g_sigma = l->grads + L.z_sigma; gz.g_bmu = l->grads + L.z_bmu;
  gz.g_bsigma = l->grads + L.z_bsigma; gz.eout = on.z_eout; gz.ein = on.z_ein;
  {
    FcZDwProb p;
    p.B = B; p.H = L.H; p.Z = L.Z; p.NZ = L.NZ; p.dlogits = l->dlogits; p.h = l->h; p.o = gz;
    RB_LAUNCH((k_gemm<1, 2, FcZDwProb>),
              dim3((unsigned)rb_div_up(L.NZ - L.Z, 32), (unsigned)rb_div_up(L.H + 1, 64), 2), dim3(128), stream, p);
    RB_LAUNCH_CHECK();
  }
  {
    FcZDxProb p;
    p.B = B; p.H = L.H; p.Z = L.Z; p.NZ = L.NZ; p.dlogits = l->dlogits; p.h = l->h; p.net = on; p.dh = l->dh;
    RB_LAUNCH((k_gemm<1, 1, FcZDxProb>), dim3((unsigned)rb_div_up(B, 32), (unsigned)rb_div_up(L.H, 32), 2), dim3(64),
              stream, p);
    RB_LAUNCH_CHECK();
  }
  {
    FcHDwProb p;
    p.B = B; p.H = L.H; p.F = L.F; p.dh = l->dh; p.feat = feat;
    p.o.g_mu = l->grads + L.h_mu; p.o.g_sigma = l->grads + L.h_sigma; p.o.g_bmu = l->grads + L.h_bmu;
    p.o.g_bsigma = l->grads + L.h_bsigma; p.o.eout = on.h_eout; p.o.ein = on.h_ein;
    RB_LAUNCH((k_gemm<2, 2, FcHDwProb>), dim3((unsigned)rb_div_up(2 * L.H, 64), (unsigned)rb_div_up(L.F + 1, 64), 1),
              dim3(256), stream, p);
    RB_LAUNCH_CHECK();
  }
  {
    FcHDxProb p;
    p.B = B; p.H = L.H; p.F = L.F; p.splits = l->xs; p.dh = l->dh; p.net = on; p.part = l->dfeat_part;
    RB_LAUNCH((k_gemm<1, 2, FcHDxProb>), dim3((unsigned)rb_div_up(B, 32), (unsigned)rb_div_up(L.F, 64), (unsigned)l->xs),
              dim3(128), stream, p);
    RB_LAUNCH_CHECK();
    const int64_t total = (int64_t)B * L.F;
    RB_LAUNCH(k_dfeat_finish, dim3((unsigned)rb_div_up(total, 256)), dim3(256), stream, (const float*)l->dfeat_part,
              l->xs, total, feat, l->dact[L.nconv - 1]);
    RB_LAUNCH_CHECK();
  }
  }
  if (l->fast_conv) {
    for (int layer = L.nconv - 1; layer > 0; --layer)                 // the input-gradient chain first ...
      if ((rc = conv_bwd(l, layer, states_dev, stream, 2)) != RB_OK) return rc;
    if ((rc = conv_dw_all(l, stream)) != RB_OK) return rc;            // ... then every weight gradient in one launch
  } else {
    for (int layer = L.nconv - 1; layer >= 0; --layer)
      if ((rc = conv_bwd(l, layer, states_dev, stream, 3)) != RB_OK) return rc;
  }
  {   // one fixed-order reduction of every conv layer's split slices into the gradient buffer
    ReduceAllArgs ra;
    int64_t off = 0;
    for (int layer = 0; layer < L.nconv; ++layer) {
      const ConvLayer& c = L.conv[layer];
      ra.layer[layer] = ReduceLayer{l->dw_part[layer], l->grads + L.conv_w[layer], l->grads + L.conv_b[layer],
                                    l->dw_slices[layer], c.cout, c.K(), off};
      off += (int64_t)c.cout * (c.K() + 1);
    }
    ra.n_layers = L.nconv; ra.total = off;
    ra.sq_part = l->norm_slots > 0 ? l->norm_part + l->norm_conv_base : nullptr;
    ra.grads_base = l->grads;
    ra.copy_base = exch ? l->fact_local + l->fact_off[5] : nullptr;
    ra.snap_src = nullptr; ra.snap_dst = nullptr; ra.snap_n = 0; ra.snap_clear = nullptr;
    if (l->sigma_implicit) { ra.snap_src = l->n_online; ra.snap_dst = l->noise_snap; ra.snap_n = (int)L.n_noise; ra.snap_clear = l->status_copy + 2; }
    RB_LAUNCH(k_reduce_conv_dw_all, dim3((unsigned)(rb_div_up(off, 64) + rb_div_up(ra.snap_n, 64))), dim3(64), stream, ra);
    RB_LAUNCH_CHECK();
  }
  return RB_OK;
}

static int clip_adam_impl(rb_learner* l, float max_norm, float* exp_avg, float* exp_avg_sq, double lr, double beta1, double beta2,
                          double eps, int64_t step, float* norm_dev, hipStream_t stream, bool defer);

// The pending optimiser pass (RB_LEARNER_DEFER_UPDATE) as a launch of its own: every entry point that reads or writes
// parameters, moments, gradients or the norm calls this first — only the next rb_learner_train_step hosts it instead.
// g_sigma = g_mu * (eps_out[n] * eps_in[k]) for the hidden layer, from the noise snapshot of the learn call that produced g_mu:
// what RB_LEARNER_IMPLICIT_SIGMA's backward left out, for every consumer of the flat gradient other than the hosted pass
__global__ __launch_bounds__(256) void k_materialize_sigma(float* g, int64_t mu4, int64_t len4, int f4, int split_row,
                                                            const float* eout, const float* ein, const int32_t* clipped) {
  if (*clipped != 0) return;          // the optimiser pass has stored the scaled gradients already (block-uniform)
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < len4; j += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(j / f4), cq = (int)(j - (int64_t)row * f4);
    const float eo = eout[row];
    const float4 e = rb_ld4(ein + 4 * (int64_t)(cq + (row >= split_row ? f4 : 0)));
    const float4 gm = rb_ld4(g + 4 * (mu4 + j));
    float4 gs;
    gs.x = gm.x * (eo * e.x); gs.y = gm.y * (eo * e.y); gs.z = gm.z * (eo * e.z); gs.w = gm.w * (eo * e.w);
    rb_st4(g + 4 * (mu4 + len4 + j), gs);
  }
}
static int materialize_sigma(rb_learner* l, hipStream_t stream) {
  if (!l->sigma_implicit) return RB_OK;
  const Layout& L = l->L;
  const NetPtrs sn = net_ptrs(L, l->p_online, l->noise_snap);
  const int64_t len4 = (int64_t)2 * L.H * L.F / 4;
  RB_LAUNCH(k_materialize_sigma, dim3((unsigned)rb_div_up(len4, 256 * 4)), dim3(256), stream, l->grads, L.h_mu / 4, len4, L.F / 4, L.H,
            sn.h_eout, sn.h_ein, (const int32_t*)(l->status_copy + 2));
  RB_LAUNCH_CHECK();
  l->sigma_implicit = 0;
  return RB_OK;
}
#define RB_MATERIALIZE_SIGMA(l, stream)                             \
  do {                                                              \
    const int rcm_ = materialize_sigma((l), (hipStream_t)(stream)); \
    if (rcm_ != RB_OK) return rcm_;                                 \
  } while (0)

static int flush_update(rb_learner* l, hipStream_t stream) {
  if (!l->adam_pending) return RB_OK;
  FusedDwAdamArgs f;
  memset(&f, 0, sizeof(f));
  ClipAdamArgs a = l->adam_args_host;
  int blocks = l->adam_blocks;
  if (a.pair_len4 > 0) {        // the pending pass forms the sigma gradient itself: the hosted body as a launch of its own
    const int rc = rb_launch_adam_pending(l->adam_args_dev, blocks, stream);      // (its arguments are in device memory already)
    if (rc != RB_OK) return rc;
    l->adam_pending = 0;
    return RB_OK;
  }
  RB_LAUNCH_T("clip_adam:k_clip_adam", (k_clip_adam<4, true, false>), dim3((unsigned)blocks), dim3(256), stream, a, f);
  RB_LAUNCH_CHECK();
  l->adam_pending = 0;
  return RB_OK;
}

int rb_learner_flush(rb_learner_t* l, rb_stream_t stream) {
  RB_REQUIRE(l != nullptr, "rb_learner_flush: NULL handle");
  const int rc = flush_update(l, (hipStream_t)stream);
  if (rc != RB_OK) return rc;
  return materialize_sigma(l, (hipStream_t)stream);      // (a caller about to read grads_dev: RB_LEARNER_IMPLICIT_SIGMA)
}

// The same hosting for a caller that issues the step's entry points one by one (Agent's eager path, the replica exchange):
// attach fills `job_out` = `job_in` + the pending pass (returns 1) or leaves it a plain copy (0); the caller passes job_out
// to rb_replay_sample_fused_noise and, once that launch is in the stream, calls rb_learner_pending_launched.
int rb_learner_attach_pending(rb_learner_t* l, const rb_noise_job_t* job_in, int32_t batch, rb_noise_job_t* job_out) {
  if (!l || !job_in || !job_out) { rb_set_error("rb_learner_attach_pending: NULL argument"); return RB_ERR_INVALID; }
  memcpy(job_out, job_in, sizeof(*job_out));
  if (!l->adam_pending || batch > 256) return 0;
  NoiseJob* nj = reinterpret_cast<NoiseJob*>(job_out);
  nj->adam_dev = l->adam_args_dev; nj->adam_blocks = l->adam_blocks;
  return 1;
}
int rb_learner_pending_launched(rb_learner_t* l) {
  RB_REQUIRE(l != nullptr, "rb_learner_pending_launched: NULL handle");
  l->adam_pending = 0;
  return RB_OK;
}
// rb_learner_clip_adam that leaves the pass pending when the handle's flags say so (RB_LEARNER_DEFER_UPDATE) and it can
// (step = 0 with a device step counter, norm partials from the learn call); otherwise exactly rb_learner_clip_adam.
int rb_learner_clip_adam_deferred(rb_learner_t* l, float max_norm, float* exp_avg, float* exp_avg_sq, double lr, double beta1,
                                  double beta2, double eps, int64_t step, float* norm_dev, rb_stream_t stream) {
  RB_REQUIRE(l != nullptr, "rb_learner_clip_adam_deferred: NULL handle");
  const int rc = flush_update(l, (hipStream_t)stream);
  if (rc != RB_OK) return rc;
  return clip_adam_impl(l, max_norm, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, norm_dev, (hipStream_t)stream,
                        (l->flags & RB_LEARNER_DEFER_UPDATE) != 0);
}

static int train_step_impl(rb_learner_t* l, const rb_train_step_t* a, rb_comm_t* comm, rb_stream_t stream);
int rb_learner_train_step(rb_learner_t* l, const rb_train_step_t* a, rb_stream_t stream) {
  return train_step_impl(l, a, nullptr, stream);
}
int rb_learner_train_step_dist(rb_learner_t* l, const rb_train_step_t* a, rb_comm_t* comm, rb_stream_t stream) {
  RB_REQUIRE(comm != nullptr, "rb_learner_train_step_dist: NULL communicator");
  return train_step_impl(l, a, comm, stream);
}
static int train_step_impl(rb_learner_t* l, const rb_train_step_t* a, rb_comm_t* comm, rb_stream_t stream) {
  RB_REQUIRE(l != nullptr && a != nullptr && a->replay != nullptr, "rb_learner_train_step: NULL argument");
  // the previous call's optimiser pass, if it was left pending, rides in this call's sampler launch (adam_body.h)
  rb_noise_job_t hosted_job;
  const rb_noise_job_t* job = a->noise_job;
  bool hosted = false;
  if (l->adam_pending) {
    if (job != nullptr && a->batch <= 256) {
      memcpy(&hosted_job, job, sizeof(hosted_job));
      NoiseJob* nj = reinterpret_cast<NoiseJob*>(&hosted_job);
      nj->adam_dev = l->adam_args_dev; nj->adam_blocks = l->adam_blocks;
      job = &hosted_job;
      hosted = true;
    } else {
      const int rc = flush_update(l, (hipStream_t)stream);
      if (rc != RB_OK) return rc;
    }
  }
  // the early draw (RB_OPTS spec_draw=1; off by default): from the SECOND consecutive call with the same replay, batch, beta and
  // buffers — and nothing else having touched the replay in between — this call's write-back and the next call's draw leave for the
  // replay's stream behind the head kernel.  main.py's loop never meets the condition (it appends and anneals beta between learn
  // calls, main.py:157,161): this is for append-free loops only (a PER benchmark, bench.py).  Never under stream capture: the pair
  // would be launched now, outside the graph, waiting for a flag the captured kernels only store on replay.
  {
    auto& t = l->ts_last;
    const bool capturing = l->opt_spec_draw && rb_stream_capturing(stream);
    const bool same = !capturing && t.valid && t.replay == a->replay && t.batch == a->batch && t.max_attempts == a->max_attempts &&
                      t.beta == a->priority_weight &&
                      t.tree_idx == a->tree_idx_dev && t.actions == a->actions_dev && t.returns == a->returns_dev &&
                      t.nonterm == a->nonterminals_dev && t.weights == a->weights_dev && t.mut_after == rb_replay_mutations(a->replay);
    t.streak = same ? t.streak + 1 : 0;
    l->spec_now = (l->opt_spec_draw && t.streak >= 1 && comm == nullptr && a->noise_job != nullptr && a->batch <= 256 && l->fast_fc &&
                   l->sink == a->replay && l->sink_idx == a->tree_idx_dev && rb_replay_spec_allowed(a->replay)) ? 1 : 0;
    if (l->spec_now) {
      rb_spec_request& q = l->spec_req;
      memset(&q, 0, sizeof(q));
      q.batch = a->batch; q.priority_weight = a->priority_weight; q.max_attempts = a->max_attempts;
      q.tree_idx = a->tree_idx_dev; q.actions = a->actions_dev; q.returns = a->returns_dev; q.nonterminals = a->nonterminals_dev;
      q.weights = a->weights_dev;
    }
    if (l->opt_spec_draw && !capturing) rb_replay_spec_arm_accept(a->replay);     // only THIS caller reads the table of the accepted draw
  }
  int rc = rb_replay_sample_fused_noise(a->replay, a->batch, a->priority_weight, nullptr, a->max_attempts, a->tree_idx_dev, nullptr,
                                        nullptr, a->actions_dev, a->returns_dev, a->nonterminals_dev, a->weights_dev,
                                        job, stream);
  if (rc != RB_OK) { l->spec_now = 0; return rc; }
  if (hosted) l->adam_pending = 0;
  // (the window table of THIS draw: an accepted early draw filled the replay's other table)
  rc = rb_learner_learn_windows(l, a->frames_dev, rb_replay_current_windows(a->replay), a->window_len, a->actions_dev, a->returns_dev,
                                a->nonterminals_dev, a->weights_dev, a->loss_dev, stream);
  l->spec_now = 0;
  {
    auto& t = l->ts_last;
    t.replay = a->replay; t.batch = a->batch; t.max_attempts = a->max_attempts; t.beta = a->priority_weight; t.tree_idx = a->tree_idx_dev;
    t.actions = a->actions_dev; t.returns = a->returns_dev; t.nonterm = a->nonterminals_dev; t.weights = a->weights_dev;
    t.mut_after = rb_replay_mutations(a->replay); t.valid = rc == RB_OK ? 1 : 0;
  }
  if (rc != RB_OK) return rc;
  if (comm) {      // replicas: the factor all-gather + the finishing launch between backward and clip (agent.py:96-97)
    rc = rb_learner_exchange_rccl(l, comm, stream);
    if (rc != RB_OK) return rc;
  }
  return clip_adam_impl(l, a->max_norm, a->exp_avg_dev, a->exp_avg_sq_dev, a->lr, a->beta1, a->beta2, a->eps, a->step,
                        a->norm_dev, (hipStream_t)stream, (l->flags & RB_LEARNER_DEFER_UPDATE) != 0);
}

int rb_learner_clip_grad(rb_learner_t* l, float max_norm, float* norm_dev, rb_stream_t stream) {
  RB_REQUIRE(l != nullptr, "rb_learner_clip_grad: NULL handle");
  RB_FLUSH_UPDATE(l, stream);
  RB_MATERIALIZE_SIGMA(l, stream);
  if (l->dw_deferred) {
    rb_set_error("rb_learner_clip_grad: the last learn call left the hidden layer's weight gradient to the fused optimiser "
                 "pass (RB_LEARNER_FUSE_FC_H_DW); call rb_learner_clip_adam, or clear the flag before learning");
    return RB_ERR_STATE;
  }
  const int64_t n = l->L.n_params;
  int nblocks = (int)rb_div_up(n, 256 * 16);
  if (nblocks > 1024) nblocks = 1024;
  int nparts = l->norm_slots;
  if (nparts > 0) {
    // every block of the scale kernel re-sums the partial list (same order everywhere): keep that redundant work small.
    // The scale loop itself only runs when the norm exceeds max_norm.
    if (nblocks > 256) nblocks = 256;
  } else {   // gradient was produced by the fallback path or modified since (all-reduce): one pass over it
    nparts = nblocks;
    RB_LAUNCH(k_sumsq, dim3((unsigned)nparts), dim3(256), stream, (const float*)l->grads, n, l->norm_part);
    RB_LAUNCH_CHECK();
  }
  l->norm_slots = 0;   // consumed
  RB_LAUNCH(k_clip_scale, dim3((unsigned)nblocks), dim3(256), stream, l->grads, n, (const float*)l->norm_part, nparts,
            max_norm, norm_dev);
  RB_LAUNCH_CHECK();
  return RB_OK;
}

__global__ void k_store_adam_args(ClipAdamArgs a, ClipAdamArgs* dst) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *dst = a;
}

int rb_learner_clip_adam(rb_learner_t* l, float max_norm, float* exp_avg, float* exp_avg_sq, double lr, double beta1,
                         double beta2, double eps, int64_t step, float* norm_dev, rb_stream_t stream) {
  RB_REQUIRE(l != nullptr, "rb_learner_clip_adam: NULL handle");
  const int rc = flush_update(l, (hipStream_t)stream);
  if (rc != RB_OK) return rc;
  return clip_adam_impl(l, max_norm, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, norm_dev, (hipStream_t)stream, false);
}

static int clip_adam_impl(rb_learner* l, float max_norm, float* exp_avg, float* exp_avg_sq, double lr, double beta1, double beta2,
                          double eps, int64_t step, float* norm_dev, hipStream_t stream, bool defer) {
  RB_REQUIRE(exp_avg != nullptr && exp_avg_sq != nullptr, "rb_learner_clip_adam: NULL moment buffer");
  RB_REQUIRE(step >= 1 || (step == 0 && l->step_ctr), "rb_learner_clip_adam: step is 1-based (0 = take it from the device counter set "
             "with rb_learner_set_step_counter)");
  const int64_t n = l->L.n_params;
  int nparts = l->norm_slots;
  if (!(max_norm < INFINITY) && norm_dev == nullptr) {
    nparts = 0;        // plain optimiser.step(): no clip, nobody wants the norm
  } else if (nparts <= 0) {   // gradient came from the fallback path or was modified since (all-reduce): one pass over it
    nparts = (int)rb_div_up(n, 256 * 16);
    if (nparts > 1024) nparts = 1024;
    RB_LAUNCH(k_sumsq, dim3((unsigned)nparts), dim3(256), stream, (const float*)l->grads, n, l->norm_part);
    RB_LAUNCH_CHECK();
  }
  l->norm_slots = 0;   // consumed
  ClipAdamArgs a;
  memset(&a, 0, sizeof(a));
  a.p = l->p_online; a.g = l->grads; a.m = exp_avg; a.v = exp_avg_sq; a.n = n;
  a.part = l->norm_part; a.nparts = nparts; a.max_norm = max_norm; a.norm_out = norm_dev;
  // scalars exactly as torch.optim.adam._single_tensor_adam forms them (python doubles, rounded once to f32 by the op)
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  a.w1 = (float)(1.0 - beta1); a.b2 = (float)beta2; a.w2 = (float)(1.0 - beta2);
  a.neg_step_size = (float)(-(lr / bc1)); a.bc2_sqrt = (float)sqrt(bc2); a.eps = (float)eps;
  a.step_dev = step == 0 ? l->step_ctr : nullptr; a.lr = lr; a.beta1 = beta1; a.beta2 = beta2;
  a.batch_status = l->status_copy;        // (k_head's copy of l->batch_status: see status_copy)
  const int64_t n4 = n >> 2;
  // 4 quadruples per thread: measured best of {2, 4, 8} on MI355X (254.3 / 255.6 / 256.6 us per step)
  // write-through stores (same-box A/B 253.7 -> 250.8 us per step) through buffer instructions: offsets are 31-bit
  RB_REQUIRE(n * 4 < (int64_t)0x7fffffff, "rb_learner_clip_adam: the flat parameter buffer must be smaller than 2 GiB");
  FusedDwAdamArgs f;
  memset(&f, 0, sizeof(f));
  a.skip_lo4 = 0; a.skip_len4 = 0;
  if (l->dw_deferred) {
    const Layout& L = l->L;
    const NetPtrs on = net_ptrs(L, l->p_online, l->n_online);
    FcDwPlan hp = fc_dw_plan(l, on, 1, l->dh, l->act[L.nconv - 1], L.B, 0);
    f.dw = hp.a;
    f.mu_off = L.h_mu; f.sigma_off = L.h_sigma;
    f.dw_x = hp.dw_x; f.n_tile_blocks = 2 * hp.dw_x * hp.dw_y;     // two slots per tile: mu, sigma
    f.write_grads = (l->flags & RB_LEARNER_WRITE_FUSED_GRADS) ? 1 : 0;
    a.skip_lo4 = L.h_mu >> 2; a.skip_len4 = (L.h_bmu - L.h_mu) >> 2;
    const unsigned grid = (unsigned)(f.n_tile_blocks + rb_div_up(n4 - a.skip_len4 > 0 ? n4 - a.skip_len4 : 1, 256 * 4));
    RB_LAUNCH_T("clip_adam:k_clip_adam", (k_clip_adam<4, true, true>), dim3(grid), dim3(256), stream, a, f);
    l->dw_deferred = 0;
  } else {
    unsigned grid = (unsigned)rb_div_up(n4 > 0 ? n4 : 1, 256 * 4);
    const bool will_defer = defer && a.step_dev != nullptr && a.nparts > 0 && l->adam_args_dev != nullptr;
    if (l->sigma_implicit && will_defer) {
      // the hosted pass updates (mu, sigma) quads of the hidden layer together and forms g_sigma itself (adam_body.h)
      const Layout& L = l->L;
      const NetPtrs sn = net_ptrs(L, l->p_online, l->noise_snap);
      a.pair_mu4 = L.h_mu / 4; a.pair_len4 = (int64_t)2 * L.H * L.F / 4;
      a.pair_f4 = L.F / 4; a.pair_split_row = L.H; a.pair_eout = sn.h_eout; a.pair_ein = sn.h_ein;
      a.pair_clipped = l->status_copy + 2;
      a.hole_lo4 = (unsigned)a.pair_mu4; a.hole4 = (unsigned)(2 * a.pair_len4);
      a.pair_blk0 = (int)rb_div_up(n4 - 2 * a.pair_len4 > 0 ? n4 - 2 * a.pair_len4 : 1, 256 * 4);
      grid = (unsigned)(a.pair_blk0 + rb_div_up(a.pair_len4, 256 * 2));   /* adam_body.h rb_adam_hosted_pairs: 2 pairs per thread */
    } else if (l->sigma_implicit) {
      const int rcm = materialize_sigma(l, stream);
      if (rcm != RB_OK) return rcm;
    }
    if (will_defer) {
      // left pending: the next train_step's sampler launch hosts these workgroups (or flush_update launches them).  The
      // arguments are all step-invariant (the step number and the norm partials live on the device): uploaded on change only
      if (!l->adam_args_valid || memcmp(&a, &l->adam_args_host, sizeof(a)) != 0) {
        RB_LAUNCH(k_store_adam_args, dim3(1), dim3(64), stream, a, l->adam_args_dev);
        RB_LAUNCH_CHECK();
        memcpy(&l->adam_args_host, &a, sizeof(a));
        l->adam_args_valid = 1;
      }
      l->adam_pending = 1;
      l->adam_blocks = (int)grid;
      return RB_OK;
    }
    RB_LAUNCH_T("clip_adam:k_clip_adam", (k_clip_adam<4, true, false>), dim3(grid), dim3(256), stream, a, f);
  }
  RB_LAUNCH_CHECK();
  return RB_OK;
}

int rb_learner_set_step_counter(rb_learner_t* l, int64_t* step_dev) {
  RB_REQUIRE(l != nullptr, "rb_learner_set_step_counter: NULL handle");
  l->step_ctr = reinterpret_cast<long long*>(step_dev);
  return RB_OK;
}

int rb_learner_set_flags(rb_learner_t* l, int32_t flags) {
  RB_REQUIRE(l != nullptr, "rb_learner_set_flags: NULL handle");
  RB_REQUIRE((flags & ~(RB_LEARNER_FUSE_FC_H_DW | RB_LEARNER_WRITE_FUSED_GRADS | RB_LEARNER_DEFER_UPDATE | RB_LEARNER_IMPLICIT_SIGMA)) == 0,
             "rb_learner_set_flags: unknown flag bits");
  l->flags = flags;
  return RB_OK;
}

int rb_learner_exchange_layout(rb_learner_t* l, int64_t* factor_floats, int64_t* small_offset, int64_t* small_floats) {
  RB_REQUIRE(l != nullptr, "rb_learner_exchange_layout: NULL handle");
  if (factor_floats) *factor_floats = l->fact_stride;
  if (small_offset) *small_offset = 0;            // the conv parameters lead the flat buffers (make_layout)
  if (small_floats) *small_floats = l->L.h_mu;
  return RB_OK;
}

int rb_learner_set_exchange(rb_learner_t* l, int32_t world, float* factors_local_dev, const float* factors_all_dev) {
  RB_REQUIRE(l != nullptr, "rb_learner_set_exchange: NULL handle");
  RB_REQUIRE(world >= 1 && world <= 64, "rb_learner_set_exchange: world must be in [1,64]");
  if (world == 1) { l->world = 1; l->fact_local = nullptr; l->fact_all = nullptr; return RB_OK; }
  RB_REQUIRE(factors_local_dev && factors_all_dev, "rb_learner_set_exchange: NULL factor buffer");
  if (!l->fast_fc) {
    rb_set_error("rb_learner_set_exchange: the factored exchange needs the streamed noisy-linear kernels (F, H multiples of 32); "
                 "all-reduce the flat gradient and call rb_learner_grads_modified instead");
    return RB_ERR_STATE;
  }
  l->world = world; l->fact_local = factors_local_dev; l->fact_all = factors_all_dev;
  return RB_OK;
}

int rb_learner_wait_factors(rb_learner_t* l, rb_stream_t side_stream) {
  RB_REQUIRE(l != nullptr, "rb_learner_wait_factors: NULL handle");
  RB_REQUIRE(l->exch_pending, "rb_learner_wait_factors: no learn call with a pending exchange");
  (void)side_stream;     // the block is complete in the stream order of the learn call: nothing to wait for (see the header)
  return RB_OK;
}

// ---- RCCL, resolved at run time (include/rainbow_hip.h: rb_comm_*).  Prefers the librccl the process already has (PyTorch
// ships one: the communicator then lives in the same library instance as torch.distributed's), else the ROCm installation's.
#if !defined(RB_HOST_INTERP)
#include <dlfcn.h>
struct RbNcclUniqueId { char internal[128]; };
struct RbNccl {
  void* h;
  int (*GetUniqueId)(RbNcclUniqueId*);
  int (*CommInitRank)(void**, int, RbNcclUniqueId, int);
  int (*CommDestroy)(void*);
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t);
  const char* (*GetErrorString)(int);
};
static RbNccl* rb_nccl() {
  static RbNccl n;
  static int state = 0;        // 0 untried, 1 ok, -1 unavailable
  if (state == 0) {
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so", "/opt/rocm/lib/librccl.so.1"};
    n.h = nullptr;
    for (const char* nm : names) if ((n.h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;     // already loaded?
    if (!n.h) for (const char* nm : names) if ((n.h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
    state = -1;
    if (n.h) {
      n.GetUniqueId = (int (*)(RbNcclUniqueId*))dlsym(n.h, "ncclGetUniqueId");
      n.CommInitRank = (int (*)(void**, int, RbNcclUniqueId, int))dlsym(n.h, "ncclCommInitRank");
      n.CommDestroy = (int (*)(void*))dlsym(n.h, "ncclCommDestroy");
      n.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(n.h, "ncclAllGather");
      n.GetErrorString = (const char* (*)(int))dlsym(n.h, "ncclGetErrorString");
      if (n.GetUniqueId && n.CommInitRank && n.CommDestroy && n.AllGather && n.GetErrorString) state = 1;
    }
  }
  return state == 1 ? &n : nullptr;
}
#define RB_NCCL_TRY(n, expr)                                                                          \
  do {                                                                                                \
    const int r_ = (expr);                                                                            \
    if (r_ != 0) { rb_set_error("%s failed: %s", #expr, (n)->GetErrorString(r_)); return RB_ERR_HIP; } \
  } while (0)
#endif
struct rb_comm {
  void* comm;
  int world, rank;
  hipStream_t last_stream = nullptr;   // stream of the last all-gather (rb_comm_destroy waits for it)
  int used = 0;
};

int rb_comm_available(void) {
#if defined(RB_HOST_INTERP)
  return 0;
#else
  return rb_nccl() ? 1 : 0;        // dlopen + dlsym only: no bootstrap id, no listener thread
#endif
}

int rb_comm_unique_id(void* id128) {
  RB_REQUIRE(id128 != nullptr, "rb_comm_unique_id: NULL argument");
#if defined(RB_HOST_INTERP)
  rb_set_error("rb_comm_unique_id: RCCL is not part of the host-interpreted test build");
  return RB_ERR_STATE;
#else
  RbNccl* n = rb_nccl();
  if (!n) { rb_set_error("rb_comm_unique_id: librccl.so could not be loaded (dlopen)"); return RB_ERR_STATE; }
  static_assert(sizeof(RbNcclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  RB_NCCL_TRY(n, n->GetUniqueId(reinterpret_cast<RbNcclUniqueId*>(id128)));
  return RB_OK;
#endif
}

int rb_comm_create(rb_comm_t** out, const void* id128, int32_t world, int32_t rank) {
  RB_REQUIRE(out && id128, "rb_comm_create: NULL argument");
  RB_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rb_comm_create: rank must be in [0, world)");
#if defined(RB_HOST_INTERP)
  rb_set_error("rb_comm_create: RCCL is not part of the host-interpreted test build");
  return RB_ERR_STATE;
#else
  RbNccl* n = rb_nccl();
  if (!n) { rb_set_error("rb_comm_create: librccl.so could not be loaded (dlopen)"); return RB_ERR_STATE; }
  RbNcclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  void* c = nullptr;
  RB_NCCL_TRY(n, n->CommInitRank(&c, world, id, rank));
  rb_comm* rc = new (std::nothrow) rb_comm();
  if (!rc) { n->CommDestroy(c); rb_set_error("rb_comm_create: host OOM"); return RB_ERR_OOM; }
  rc->comm = c; rc->world = world; rc->rank = rank;
  *out = rc;
  return RB_OK;
#endif
}

int rb_comm_destroy(rb_comm_t* comm) {
  if (!comm) return RB_OK;
#if !defined(RB_HOST_INTERP)
  RbNccl* n = rb_nccl();
  // the all-gather of the last exchange may still be in flight on the stream it was issued on
  if (comm->used) (void)hipStreamSynchronize(comm->last_stream);
  if (n && comm->comm) n->CommDestroy(comm->comm);
#endif
  delete comm;
  return RB_OK;
}

int rb_learner_exchange_rccl(rb_learner_t* l, rb_comm_t* comm, rb_stream_t stream_) {
  RB_REQUIRE(l && comm, "rb_learner_exchange_rccl: NULL argument");
  RB_REQUIRE(l->exch_pending && l->fact_local && l->fact_all, "rb_learner_exchange_rccl: no learn call with a pending exchange");
  RB_REQUIRE(comm->world == l->world || (comm->world == 1 && l->world == 2),
             "rb_learner_exchange_rccl: the communicator has %d ranks, the exchange buffer %d blocks", comm->world, l->world);
#if defined(RB_HOST_INTERP)
  rb_set_error("rb_learner_exchange_rccl: RCCL is not part of the host-interpreted test build");
  return RB_ERR_STATE;
#else
  RbNccl* n = rb_nccl();
  if (!n) { rb_set_error("rb_learner_exchange_rccl: librccl.so could not be loaded (dlopen)"); return RB_ERR_STATE; }
  hipStream_t stream = (hipStream_t)stream_;
  float* all = const_cast<float*>(l->fact_all);
  RB_NCCL_TRY(n, n->AllGather(l->fact_local, all, (size_t)l->fact_stride, /* ncclFloat32 */ 7, comm->comm, stream));
  comm->last_stream = stream; comm->used = 1;
  if (comm->world == 1 && l->world == 2)      // single-GPU plumbing run: the lone block stands for both replicas
    RB_HIP_TRY(hipMemcpyAsync(all + l->fact_stride, all, (size_t)l->fact_stride * 4, hipMemcpyDeviceToDevice, stream));
  return rb_learner_finish_grads(l, stream_);
#endif
}

int rb_learner_finish_grads(rb_learner_t* l, rb_stream_t stream_) {
  RB_REQUIRE(l != nullptr, "rb_learner_finish_grads: NULL handle");
  RB_REQUIRE(l->exch_pending, "rb_learner_finish_grads: no learn call with a pending exchange");
  RB_FLUSH_UPDATE(l, stream_);
  hipStream_t stream = (hipStream_t)stream_;
  const Layout& L = l->L;
  const NetPtrs on = net_ptrs(L, l->p_online, l->n_online);
  const int M = l->world * L.B;
  const float* f = l->fact_all;
  FcDwPlan zp = fc_dw_plan(l, on, 0, f + l->fact_off[0], f + l->fact_off[1], M, 0);
  FcDwPlan hp = fc_dw_plan(l, on, 1, f + l->fact_off[2], f + l->fact_off[3], M, 0);
  const int64_t conv_n = L.h_mu;
  int c_slots = (int)rb_div_up(conv_n, 256 * 16);
  if (c_slots > 1024) c_slots = 1024;
  for (FcDwPlan* p : {&zp, &hp}) {
    p->a.rpb = L.B; p->a.bstride = l->fact_stride; p->a.scale = 1.0f / (float)l->world;
    p->a.noise_blocks = f + l->fact_off[4];
  }
  zp.a.eout_noff = L.z_eout; zp.a.ein_noff = L.z_ein;
  hp.a.eout_noff = L.h_eout; hp.a.ein_noff = L.h_ein;
  // the hidden layer on 128 x 128 LDS tiles (fc_gemm.h rb_fc_gemm_dw_ranks; RB_OPTS finish_tiled=0: the 16-row-tile body): a
  // rank's slab of the gathered factors is read once per 128 weight rows instead of once per 16
  const bool tiled = l->opt_finish_tiled && 2 * L.H >= 64 && L.F >= 64;
  const int h_nt = (int)rb_div_up(2 * L.H, RB_TG_T), h_kt = (int)rb_div_up(L.F, RB_TG_T);
  if (tiled) {
    // (every workgroup of this launch is 512 threads at ~250 registers: ONE per CU.  The output layer's tiles therefore take all
    // eight waves — 512 columns per workgroup, 23 instead of 46 workgroups at the canonical shape — so that the launch stays within
    // one round of 256: with 266 workgroups the last ten waited for a CU and the launch took 40 us instead of 27)
    hp.slots = 8 * h_nt * h_kt;
    zp.dw_x = (int)rb_div_up(zp.a.K, 512);
    zp.slots = 8 * zp.dw_x * zp.dw_y;
    c_slots = h_nt * h_kt;         // the conv range: one slice (and one partial) per tile workgroup
  }
  RB_REQUIRE(zp.slots + hp.slots + c_slots <= 16384, "rb_learner_finish_grads: too many norm partials");
  FinishArgs fa;
  if (tiled) { hp.a.sq_part = l->norm_part; zp.a.sq_part = l->norm_part + hp.slots; }
  else { zp.a.sq_part = l->norm_part; hp.a.sq_part = l->norm_part + zp.slots; }
  fa.z = zp.a; fa.h = hp.a;
  fa.z_x = zp.dw_x; fa.z_n = zp.dw_x * zp.dw_y; fa.h_x = hp.dw_x; fa.h_n = hp.dw_x * hp.dw_y;
  // the conv gradients travel in the same blocks: their replica mean (rank order) and its sum of squares (0.3 MB per rank);
  // tiled: sliced over the hidden layer's tile workgroups (c_slots above)
  fa.g = l->grads; fa.n = conv_n; fa.part = l->norm_part + zp.slots + hp.slots; fa.nparts = c_slots;
  fa.blocks = f + l->fact_off[5]; fa.bstride = l->fact_stride; fa.world = l->world; fa.scale = 1.0f / (float)l->world;
  if (tiled) {
    RB_LAUNCH_T("finish_grads:k_finish_grads", k_finish_grads_tiled, dim3((unsigned)(h_nt * h_kt + fa.z_n)), dim3(RB_TG_THREADS), stream, fa, h_nt, h_kt);
  } else {
    RB_LAUNCH_T("finish_grads:k_finish_grads", k_finish_grads, dim3((unsigned)(fa.z_n + fa.h_n + c_slots)), dim3(256), stream, fa);
  }
  RB_LAUNCH_CHECK();
  l->norm_slots = zp.slots + hp.slots + c_slots;
  l->exch_pending = 0;
  return RB_OK;
}

int rb_learner_get_rng(rb_learner_t* l, uint64_t* seed, uint64_t* epoch, rb_stream_t stream) {
  RB_REQUIRE(l && seed && epoch, "rb_learner_get_rng: NULL argument");
  RB_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  unsigned long long e = 0;
  RB_HIP_TRY(hipMemcpy(&e, l->noise_ctr, sizeof(e), hipMemcpyDeviceToHost));
  *seed = l->seed; *epoch = (uint64_t)e;
  return RB_OK;
}

int rb_learner_set_rng(rb_learner_t* l, uint64_t seed, uint64_t epoch, rb_stream_t stream) {
  RB_REQUIRE(l != nullptr, "rb_learner_set_rng: NULL handle");
  RB_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  const unsigned long long e[2] = {(unsigned long long)epoch, 0ull};
  RB_HIP_TRY(hipMemcpy(l->noise_ctr, e, sizeof(e), hipMemcpyHostToDevice));
  l->seed = seed;
  return upload_noise_jobs(l);   // the device copies of the noise jobs carry the seed (callers re-query the host copy)
}

int rb_learner_set_priority_sink(rb_learner_t* l, rb_replay_t* replay, const int64_t* tree_idx_dev) {
  RB_REQUIRE(l != nullptr, "rb_learner_set_priority_sink: NULL handle");
  RB_REQUIRE((replay == nullptr) == (tree_idx_dev == nullptr), "rb_learner_set_priority_sink: pass both or neither");
  l->sink = replay;
  l->sink_idx = tree_idx_dev;
  l->batch_status = nullptr;
  if (replay) {
    ReplayView v;
    double omega;
    if (rb_replay_internal_view(replay, &v, &omega) != RB_OK) { rb_set_error("rb_learner_set_priority_sink: bad replay handle"); return RB_ERR_INVALID; }
    l->batch_status = &v.hdr->last_status;
  }
  return RB_OK;
}

int rb_learner_zero_copy_ok(rb_learner_t* l) {
  return (l && l->fast_conv) ? 1 : 0;
}

int rb_learner_priority_written(rb_learner_t* l) {
  return (l && l->sink_done) ? 1 : 0;
}

int rb_learner_grads_modified(rb_learner_t* l) {
  RB_REQUIRE(l != nullptr, "rb_learner_grads_modified: NULL handle");
  if (l->dw_deferred) {
    rb_set_error("rb_learner_grads_modified: the hidden layer's weight gradient was not materialised (RB_LEARNER_FUSE_FC_H_DW)");
    return RB_ERR_STATE;
  }
  if (l->sigma_implicit) {
    rb_set_error("rb_learner_grads_modified: the hidden layer's sigma gradient was not materialised (RB_LEARNER_IMPLICIT_SIGMA): "
                 "call rb_learner_flush before reading or modifying grads_dev");
    return RB_ERR_STATE;
  }
  l->norm_slots = 0;
  return RB_OK;
}

int rb_learner_sync_target(rb_learner_t* l, rb_stream_t stream) {
  RB_REQUIRE(l != nullptr, "rb_learner_sync_target: NULL handle");
  RB_FLUSH_UPDATE(l, stream);
  // load_state_dict copies parameters AND the epsilon buffers (agent.py:102-103)
  RB_HIP_TRY(hipMemcpyAsync(l->p_target, l->p_online, (size_t)l->L.n_params * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  RB_HIP_TRY(hipMemcpyAsync(l->n_target, l->n_online, (size_t)l->L.n_noise * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return RB_OK;
}

#if defined(RB_STAMP)
// RB_STAMP builds: what the runtime says about the residency of the first layer's forward kernels
int rb_debug_occupancy(void) {
  int a = 0, b = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, (const void*)(k_conv_fwd_t16<GeomC1, 3, 20, 256, true, 80, 1>), 640, 0);
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, (const void*)(k_conv_fwd_lds<GeomC1, 3, 20, 256, true, 80>), 512, 0);
  hipFuncAttributes fa;
  hipFuncGetAttributes(&fa, (const void*)(k_conv_fwd_t16<GeomC1, 3, 20, 256, true, 80, 1>));
  printf("occupancy API: conv1 t16 (640 threads) %d blocks/CU, conv1 lds (512 threads) %d; t16: regs %d lds %zu maxThreads %d\n", a, b, fa.numRegs, fa.sharedSizeBytes, fa.maxThreadsPerBlock);
  return a;
}
#endif
int rb_learner_debug_read(rb_learner_t* l, int32_t what, void* out_dev, rb_stream_t stream) {
  RB_REQUIRE(l && out_dev, "rb_learner_debug_read: NULL argument");
  if (what != 5) RB_FLUSH_UPDATE(l, stream);     // (5 is an activation of the last learn call: a pending optimiser pass stays pending)
  const Layout& L = l->L;
  const void* src = nullptr;
  size_t bytes = 0;
  switch (what) {
    case 0: src = l->log_ps_a; bytes = (size_t)L.B * L.Z * 4; break;
    case 1: src = l->m; bytes = (size_t)L.B * L.Z * 4; break;
    case 2: src = l->a_star; bytes = (size_t)L.B * 4; break;
    case 3: src = l->pns_a; bytes = (size_t)L.B * L.Z * 4; break;
    case 4: src = l->logits; bytes = (size_t)3 * L.B * L.NZ * 4; break;
    case 5: src = l->h; bytes = (size_t)L.B * 2 * L.H * 4; break;     // hidden activations of the differentiated forward (rows [0, B))
    default: rb_set_error("rb_learner_debug_read: unknown selector %d", what); return RB_ERR_INVALID;
  }
  RB_HIP_TRY(hipMemcpyAsync(out_dev, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return RB_OK;
}

}  // extern "C"
