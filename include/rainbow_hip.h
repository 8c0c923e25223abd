/* rainbow_hip.h — C ABI of librainbow_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary for the Rainbow *learn step* hot path of
 * Kaixhin/Rainbow (reference = /root/reference, Python only, no FFI of its own).
 * The reference's boundary is the Python class surface consumed by main.py/test.py
 * (SURVEY.md §8b); rainbow_amd/{memory,agent}.py keep that surface and forward to the
 * entry points below through ctypes.  Every entry point cites the reference code it
 * replaces as  file:line  relative to /root/reference.
 *
 * Conventions
 *   - plain C types only; no torch types cross this boundary;
 *   - every pointer named *_dev is a DEVICE pointer (HBM); *_host is host memory;
 *   - every call returns 0 on success, <0 on error (rb_last_error() has the text);
 *     nothing throws across the ABI;
 *   - every launch takes an explicit stream (a hipStream_t passed as void*);
 *     calls are asynchronous unless stated otherwise;
 *   - replay/tree/activation memory is OWNED by the library; parameter, gradient and
 *     noise memory is BORROWED from the caller (torch tensors' data_ptr());
 *   - handles are thread-compatible, not thread-safe (one handle per host thread).
 */
#ifndef RAINBOW_HIP_H_
#define RAINBOW_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RB_OK 0
#define RB_ERR_INVALID (-1)
#define RB_ERR_HIP (-2)
#define RB_ERR_OOM (-3)
#define RB_ERR_STATE (-4)

#define RB_FRAME_H 84
#define RB_FRAME_W 84
#define RB_FRAME_BYTES (84 * 84) /* memory.py:7 'state' u8[84,84] */

typedef void* rb_stream_t; /* hipStream_t */
/* Opaque description of a pending noise resample (rb_learner_noise_job) that another launch of the
 * library can host as extra workgroups (rb_replay_sample_fused_noise).                       */
typedef struct { uint64_t opaque[32]; } rb_noise_job_t;
typedef struct rb_replay rb_replay_t;
typedef struct rb_learner rb_learner_t;

const char* rb_last_error(void);
int rb_abi_version(void);
/* Hash of the sources this binary was built from (__graft_entry__.source_hash(), baked in at build time; "" for a build
 * made by hand).  bench.py and smoke() print it, build() rebuilds when it differs from the sources in the tree.        */
const char* rb_source_hash(void);
/* Synchronous copies between host memory and library/torch-owned device memory (state
 * dump/restore of the replay buffer — main.py:94-100,118 pickles the whole memory — and
 * white-box tests).  `stream` is synchronised first.                                     */
int rb_copy_to_host(void* dst_host, const void* src_dev, size_t nbytes, rb_stream_t stream);
int rb_copy_to_device(void* dst_dev, const void* src_host, size_t nbytes, rb_stream_t stream);

/* Live kernel timing for bench.py's roofline line: every launch whose kernel expression
 * contains `kernel_substr` (e.g. "FcHFwdProb") is bracketed by two hipEvents recorded on the
 * stream it is launched on.  NULL/"" disables.  Not for use under stream capture.
 * rb_profile_read synchronises the recorded events, returns their summed elapsed time and
 * the number of launches, and resets the counters.                                        */
int rb_profile_select(const char* kernel_substr);
/* Bracket only every `every`-th matching launch (default 1 = all): an event pair costs the stream ~5 us, so a timed
 * region that samples 1 launch in 8 is perturbed by 0.6 us per step instead of 5.                                    */
int rb_profile_stride(int32_t every);
int rb_profile_read(double* total_ms, int64_t* launches);
/* Cost of an event pair with NOTHING between them on `stream` (mean of n pairs, ms): what the
 * bracketing itself adds to every rb_profile_read sample.  Blocks until the stream drains.     */
int rb_profile_overhead(rb_stream_t stream, int32_t n, double* mean_ms);
/* Test hook (no reference counterpart).  With RB_GUARD=1 in the environment when the library is loaded, every device
 * allocation the library owns (replay ring, sum-tree, learner workspaces) sits between two 4 KiB guard bands; this call
 * synchronises the device and reports how many blocks are live and how many have a band that a kernel wrote into
 * (tests/test_guard_gpu.py).  Without RB_GUARD it reports 0 / 0.                                                     */
int rb_debug_check_guards(int64_t* n_blocks, int64_t* n_bad);

/* ===================================================================== replay ==
 * HBM-resident prioritised replay: SoA ring (frames u8[C][7056], timestep i32[C],
 * action i32[C], reward f32[C], nonterminal u8[C]) + level-order float32 sum-tree
 * with the reference's truncated leaf level (memory.py:17-18).                  */

/* Device-resident header; host reads it with rb_replay_header (synchronising).   */
typedef struct {
  int64_t index;   /* SegmentTree.index   memory.py:14,59 */
  int32_t full;    /* SegmentTree.full    memory.py:16,60 */
  float max;       /* SegmentTree.max     memory.py:20,48,54,61 (p^w domain) */
  float total;     /* sum_tree[0]         memory.py:88-89 */
  int32_t last_attempts; /* sampler attempts used by the last rb_replay_sample */
  int32_t last_status;   /* 0 ok, 1 = gave up after max attempts */
  uint64_t rng_counter;  /* Philox counter of the device sampler */
} rb_replay_header_t;

/* Raw device pointers, for state dump/restore and white-box tests.               */
typedef struct {
  float* sum_tree_dev;    int64_t tree_len;   /* tree_start + capacity */
  int64_t tree_start;
  uint8_t* frames_dev;    /* [capacity][7056] */
  int32_t* timestep_dev;  int32_t* action_dev;
  float* reward_dev;      uint8_t* nonterminal_dev;
  rb_replay_header_t* header_dev;
  int32_t* window_dev;    /* [max_batch][history+multi_step] ring index of every window slot of the last sample */
  int32_t window_len;
} rb_replay_buffers_t;

/* ReplayMemory.__init__ + SegmentTree.__init__  (memory.py:92-102, 13-20).
 * capacity must be even and >= 2 (odd capacities crash the reference). */
int rb_replay_create(rb_replay_t** out, int64_t capacity, int32_t history, int32_t multi_step,
                     double discount, double priority_exponent, uint64_t seed);
int rb_replay_destroy(rb_replay_t* r);
int rb_replay_buffers(rb_replay_t* r, rb_replay_buffers_t* out_host);
/* synchronises `stream`, then copies the header to host */
int rb_replay_header(rb_replay_t* r, rb_replay_header_t* out_host, rb_stream_t stream);

/* Frame preprocessing of the environment wrapper on the device (env.py:27-29 `_get_state` =
 * cv2.resize(ale.getScreenGrayscale() u8 [height][width], (84, 84), INTER_LINEAR) as float32 / 255, and env.py:57-69: the
 * observation of a step is the element-wise max of the states after frames 3 and 4 of the action repeat).
 * frame_a_dev / frame_b_dev: n raw grayscale screens each ([n][height][width] u8; frame_b_dev NULL = no max, env.py:50 reset);
 * out_dev: [n][84][84] float32 in [0, 1] — one slice of the `state` tensor that Agent.act and ReplayMemory.append take.
 * PARITY UNPINNED: cv2 is absent from the build container; the kernel is bit-exact against oracle/frame_oracle.py, a
 * restatement of OpenCV's published 8-bit fixed-point INTER_LINEAR (11-bit coefficients).                                  */
int rb_frame_preprocess(const uint8_t* frame_a_dev, const uint8_t* frame_b_dev, int32_t height, int32_t width, int32_t n,
                        float* out_dev, rb_stream_t stream);

/* ReplayMemory.append (memory.py:105-108) + SegmentTree.append (memory.py:56-61):
 * quantises state_dev[history-1] (f32 in [0,1]) to u8 by x*255 truncation ON DEVICE,
 * stores (timestep, frame, action, reward, nonterminal) at `index`, sets the leaf to
 * the running max priority and walks the sums to the root.                        */
int rb_replay_append(rb_replay_t* r, const float* state_dev, int32_t timestep, int32_t action,
                     float reward, int32_t nonterminal, rb_stream_t stream);
/* n sequential appends of already-quantised frames (same result as n calls of
 * rb_replay_append: all leaves get the current max, memory.py:107).               */
int rb_replay_append_batch(rb_replay_t* r, const uint8_t* frames_dev, const int32_t* timesteps_dev,
                           const int32_t* actions_dev, const float* rewards_dev,
                           const uint8_t* nonterminals_dev, int64_t n, rb_stream_t stream);

/* SegmentTree.find (memory.py:64-82): float64 values against float32 nodes.       */
int rb_replay_find(rb_replay_t* r, const double* values_dev, int32_t n, float* probs_dev,
                   int64_t* data_idx_dev, int64_t* tree_idx_dev, rb_stream_t stream);

/* ReplayMemory.sample (memory.py:124-155), entirely on device:
 * stratified draw (+ whole-batch rejection, memory.py:128-132), tree search, window
 * gather with episode-boundary blanking (memory.py:111-121), n-step return, IS weights.
 * unit_uniforms_dev: NULL = device Philox; else [max_attempts][batch] float64 in [0,1)
 *   (parity hook: the reference's np.random.uniform(0,seg) == seg*u, memory.py:129).
 * Outputs (all device, caller-allocated):
 *   tree_idx i64[B]; states/next_states u8[B][history][7056] (NOT yet /255);
 *   actions i64[B]; returns f32[B]; nonterminals f32[B]; weights f32[B].
 * Status lands in the device header (last_attempts/last_status).                   */
int rb_replay_sample(rb_replay_t* r, int32_t batch, double priority_weight,
                     const double* unit_uniforms_dev, int32_t max_attempts,
                     int64_t* tree_idx_dev, uint8_t* states_dev, uint8_t* next_states_dev,
                     int64_t* actions_dev, float* returns_dev, float* nonterminals_dev,
                     float* weights_dev, rb_stream_t stream);

/* rb_replay_sample + the learner's per-step noise resample (model.py:36-40) in ONE launch: the
 * sampler is a single-workgroup latency chain and the noise draw is independent of it, so the
 * noise workgroups ride along and cost no kernel boundary of their own.  noise_job comes from
 * rb_learner_noise_job (device RNG only).                                                    */
int rb_replay_sample_fused_noise(rb_replay_t* r, int32_t batch, double priority_weight,
                                 const double* unit_uniforms_dev, int32_t max_attempts,
                                 int64_t* tree_idx_dev, uint8_t* states_dev, uint8_t* next_states_dev,
                                 int64_t* actions_dev, float* returns_dev, float* nonterminals_dev,
                                 float* weights_dev, const rb_noise_job_t* noise_job, rb_stream_t stream);

/* The reference's sampler retries until a batch is valid (memory.py:128-132); the device sampler is bounded by
 * max_attempts.  When the bound is hit it writes ZERO importance weights (the learn step that consumes the batch then
 * has an exactly zero gradient), sets last_status = 1 in the device header, marks the draw's index buffer (tree_idx = -1
 * everywhere) and increments a pinned host counter.
 * This call reads that counter WITHOUT synchronising: the number of failed sampler launches that have completed so far. */
int rb_replay_failed_samples(rb_replay_t* r, int64_t* count_host);
/* Zero that counter (after the caller has reported the failure and, e.g., appended more transitions).  A learn step that
 * consumed a failed batch left no trace: its priority write-back (rb_replay_update_priorities and the learner's fused
 * sink: by the mark in the index buffer, rb_replay_dropped_updates), the optimiser update and the optimiser's step number
 * (by last_status of the draw the learn call consumes) are all skipped on the device.  Also zeroes the dropped-update count
 * and the expired-wait count (rb_replay_expired_waits: the early draw is allowed again). */
int rb_replay_reset_failed_samples(rb_replay_t* r);
/* A draw that gave up marks its own index buffer (every tree index = -1).  ReplayMemory.update_priorities (memory.py:157-159)
 * — rb_replay_update_priorities, rb_replay_update_sample and the learner's fused sink — drops exactly the write-back whose
 * indices carry that mark and counts it here (pinned host word, read WITHOUT synchronising: completed launches so far).  The
 * write-back of an earlier, valid batch is applied whatever happened to later draws.                                        */
int rb_replay_dropped_updates(rb_replay_t* r, int64_t* count_host);
/* The early draw (RB_OPTS spec_draw=1, off by default: rb_learner_train_step may issue ReplayMemory.update_priorities of call k and
 * ReplayMemory.sample of call k + 1 — agent.py:100 / :63, memory.py:124-159 — on a stream the replay owns) waits across streams with
 * a bound of ~2 ms and fails safe: an expired wait makes the waiting launch do the work itself (the draw) or drop it (the
 * write-back: also counted by rb_replay_dropped_updates), is counted here (pinned host word, read WITHOUT synchronising) and
 * switches the early draw off on this handle until rb_replay_reset_failed_samples.  0 on a healthy device.                      */
int rb_replay_expired_waits(rb_replay_t* r, int64_t* count_host);
/* SegmentTree.index / .full (memory.py:14,16) from the library's host mirror, without touching the device: exact as long
 * as every append went through this handle (a header restored with rb_copy_to_device is picked up as well).        */
int rb_replay_position(rb_replay_t* r, int64_t* index_host, int32_t* full_host);

/* Graph replay support: when set (non-NULL), rb_replay_sample reads -beta (float32, i.e.
 * float32(-priority_weight), memory.py:153) from this DEVICE location instead of its by-value
 * argument, so main.py:161's per-step annealing works under a captured hipGraph.           */
int rb_replay_set_beta_source(rb_replay_t* r, const float* neg_beta_dev);

/* SegmentTree.update (memory.py:44-48): raw leaf values, duplicates last-write-wins. */
int rb_replay_update_leaves(rb_replay_t* r, const int64_t* tree_idx_dev, const float* values_dev,
                            int32_t n, rb_stream_t stream);
/* ReplayMemory.update_priorities (memory.py:157-159): p = loss^w, then the above.  */
int rb_replay_update_priorities(rb_replay_t* r, const int64_t* tree_idx_dev,
                                const float* losses_dev, int32_t n, rb_stream_t stream);

/* ReplayMemory.update_priorities(idx, loss) of learn step k (agent.py:100, memory.py:157-159) followed by
 * ReplayMemory.sample(batch) of step k + 1 (agent.py:62, memory.py:148-155) as ONE launch of one workgroup: the update's
 * writes to the top of the tree go straight into the LDS copy the search starts from.  Results (tree, header, batch) are
 * bit-identical to rb_replay_update_priorities followed by rb_replay_sample with the same arguments; upd_n above 64 or batch
 * above 256 take exactly those two calls.  states_dev / next_states_dev may be NULL as in rb_replay_sample.                 */
int rb_replay_update_sample(rb_replay_t* r, const int64_t* upd_tree_idx_dev, const float* upd_losses_dev, int32_t upd_n,
                            int32_t batch, double priority_weight, const double* unit_uniforms_dev, int32_t max_attempts,
                            int64_t* tree_idx_dev, uint8_t* states_dev, uint8_t* next_states_dev,
                            int64_t* actions_dev, float* returns_dev, float* nonterminals_dev,
                            float* weights_dev, rb_stream_t stream);

/* ReplayMemory.__next__ (memory.py:167-178): blanked history stack for data index i,
 * as f32 /255, out_dev f32[history][7056].                                          */
int rb_replay_state_at(rb_replay_t* r, int64_t data_index, float* out_dev, rb_stream_t stream);

/* The same for n data indices in ONE launch (the validation pass of test.py:38-39 walks the whole validation memory):
 * data_index_dev i64[n], each in [0, capacity); out_dev f32[n][history][7056].                                        */
int rb_replay_states_at(rb_replay_t* r, const int64_t* data_index_dev, int32_t n, float* out_dev,
                        rb_stream_t stream);

/* u8 -> f32 x/255 (memory.py:137-138 `.div_(255)`), correctly-rounded division.    */
int rb_u8_to_unit_f32(const uint8_t* src_dev, float* dst_dev, int64_t n, rb_stream_t stream);

/* ==================================================================== learner ==
 * Rainbow network (model.py) + learn step (agent.py:61-100) on flat f32 buffers.  */

typedef struct {
  int32_t batch;        /* args.batch_size      agent.py:20 */
  int32_t atoms;        /* args.atoms           agent.py:15 */
  int32_t actions;      /* env.action_space()   agent.py:14 */
  int32_t history;      /* args.history_length  model.py:56 */
  int32_t hidden;       /* args.hidden_size     model.py:64 */
  int32_t architecture; /* 0 canonical, 1 data-efficient  model.py:55-63 */
  int32_t multi_step;   /* args.multi_step      agent.py:21 */
  float v_min, v_max;   /* agent.py:16-17 */
  double discount;      /* agent.py:22 */
} rb_learner_config_t;

/* One named tensor inside the flat parameter/gradient buffer (state-dict names of
 * model.py: convs.{0,2,4}.{weight,bias}, fc_*.{weight,bias}_{mu,sigma}).          */
typedef struct {
  char name[48];
  int64_t offset; /* in floats */
  int32_t ndim;
  int32_t shape[4];
} rb_tensor_desc_t;

/* Sizes of the flat buffers (in floats) for a config.                              */
int rb_learner_sizes(const rb_learner_config_t* cfg, int64_t* n_params, int64_t* n_noise);
/* Fills descs[0..*n) ; pass descs=NULL to query the count.                         */
int rb_learner_param_layout(const rb_learner_config_t* cfg, rb_tensor_desc_t* descs, int32_t* n);
/* Noise buffer layout: factorised vectors f(eps) (model.py:32-40), per layer
 * {name = fc_h_v.eps_in, fc_h_v.eps_out, ...}.                                      */
int rb_learner_noise_layout(const rb_learner_config_t* cfg, rb_tensor_desc_t* descs, int32_t* n);

int rb_learner_create(rb_learner_t** out, const rb_learner_config_t* cfg, float* online_params_dev,
                      float* target_params_dev, float* grads_dev, float* online_noise_dev,
                      float* target_noise_dev, uint64_t seed);
int rb_learner_destroy(rb_learner_t* l);

/* DQN.reset_noise (model.py:82-85, 36-40).  which: 0 online (agent.py:49-50),
 * 1 target (agent.py:74), 2 both in one launch (online first; device RNG only).
 * raw_normals_dev: NULL = device Philox + Box-Muller;
 * else N(0,1) draws in the reference's order (per layer randn(in) then randn(out);
 * layers fc_h_v, fc_h_a, fc_z_v, fc_z_a) — parity hook.                             */
int rb_learner_reset_noise(rb_learner_t* l, int32_t which, const float* raw_normals_dev,
                           rb_stream_t stream);
int64_t rb_learner_noise_draws(const rb_learner_config_t* cfg);
/* The same resample as rb_learner_reset_noise(l, which, NULL, ...) described as a job another
 * launch can host (rb_replay_sample_fused_noise); nothing is launched here.                  */
int rb_learner_noise_job(rb_learner_t* l, int32_t which, rb_noise_job_t* out);

/* Agent.act / evaluate_q (agent.py:53-55, 110-112): single state f32[history][7056]
 * in [0,1]; writes argmax action (i32) and its expected value (f32).
 * noisy=0 is online_net.eval() (mu only, model.py:46).                              */
int rb_learner_act(rb_learner_t* l, const float* state_dev, int32_t noisy, int32_t* action_dev,
                   float* q_dev, rb_stream_t stream);
/* rb_learner_act for a caller that needs the result on the HOST at once (agent.py:53-55 returns a Python int; main.py:153 feeds it to
 * the emulator): action_pinned / q_pinned are PINNED host words mapped on the device (hipHostMalloc / torch pin_memory); the call
 * presets the action word, launches, polls the word in a compiled loop (falling back to a stream synchronise after ~10 ms) and
 * returns the action and its value through action_out / q_out (either may be NULL).  One call instead of launch + a Python poll loop.
 * Where q_pinned == (float*)(action_pinned + 1) and action_pinned is 8-byte aligned, the device writes both with ONE 8-byte store
 * (otherwise: q, a system-scope fence, then the action): rb_learner_act behaves the same way.                                    */
int rb_learner_act_wait(rb_learner_t* l, const float* state_dev, int32_t noisy, int32_t* action_pinned, float* q_pinned,
                        int32_t* action_out, float* q_out, rb_stream_t stream);
/* The same for n states at once (vectorised actors; the reference acts on one state per call,
 * main.py:153 — this is that call batched): states f32[n][history][7056],
 * 1 <= n <= 4096 (beyond the learn step's 3*batch images the forward buffers are regrown once, synchronising — batched
 * evaluation of a validation memory, test.py:38-39); actions_dev i32[n], q_dev f32[n] (either may be NULL).  action_dev /
 * q_dev of both calls may point to pinned host memory mapped on the device, which saves the
 * caller a device-to-host copy: the values are final once the stream has drained.         */
int rb_learner_act_batch(rb_learner_t* l, const float* states_dev, int32_t n, int32_t noisy,
                         int32_t* actions_dev, float* q_dev, rb_stream_t stream);

/* Agent.learn minus sampling/optimiser (agent.py:66-96): three forwards, double-Q
 * select, C51 projection, weighted cross-entropy, full backward into grads_dev.
 * Inputs are rb_replay_sample's outputs.  loss_dev f32[B] = per-sample CE (agent.py:94),
 * i.e. the new raw priorities (agent.py:100).                                        */
int rb_learner_learn(rb_learner_t* l, const uint8_t* states_dev, const uint8_t* next_states_dev,
                     const int64_t* actions_dev, const float* returns_dev,
                     const float* nonterminals_dev, const float* weights_dev, float* loss_dev,
                     rb_stream_t stream);

/* Same step, zero-copy input: instead of gathered stacks the first conv layer reads its frames
 * straight from the replay ring (rb_replay_buffers_t.frames_dev) through the window table the
 * last rb_replay_sample wrote (rb_replay_buffers_t.window_dev: [B][history+multi_step] ring
 * indices, -1 = blanked frame, memory.py:112-121).  Call rb_replay_sample with states_dev =
 * next_states_dev = NULL to skip the gather.                                                 */
int rb_learner_learn_windows(rb_learner_t* l, const uint8_t* frames_dev, const int32_t* windows_dev,
                             int32_t window_len, const int64_t* actions_dev, const float* returns_dev,
                             const float* nonterminals_dev, const float* weights_dev, float* loss_dev,
                             rb_stream_t stream);
/* 1 when rb_learner_learn_windows is available for this handle (the ring-reading first conv
 * layer needs history <= 4 and the LDS conv kernels), else 0: gather and call rb_learner_learn. */
int rb_learner_zero_copy_ok(rb_learner_t* l);

/* clip_grad_norm_ (agent.py:97): global L2 norm of grads_dev, scale in place by
 * max_norm/(norm+1e-6) when that is < 1.  norm_dev (f32[1], may be NULL) gets ||g||. */
int rb_learner_clip_grad(rb_learner_t* l, float max_norm, float* norm_dev, rb_stream_t stream);

/* clip_grad_norm_ + optimiser.step() (agent.py:97-98, optimiser built at agent.py:46) in one pass
 * over the flat buffers: the clip coefficient is applied to the gradient in registers on its way
 * into torch.optim.Adam's update (amsgrad off, weight_decay 0).  exp_avg_dev / exp_avg_sq_dev:
 * caller-owned f32[n_params] moment buffers (zero before step 1); step: 1-based step number for
 * the bias corrections.  grads_dev is rewritten (scaled) only when the norm exceeds max_norm,
 * as clip_grad_norm_ would leave it.  norm_dev (f32[1], may be NULL) gets ||g||.            */
int rb_learner_clip_adam(rb_learner_t* l, float max_norm, float* exp_avg_dev, float* exp_avg_sq_dev,
                         double lr, double beta1, double beta2, double eps, int64_t step,
                         float* norm_dev, rb_stream_t stream);

/* One whole training step of Agent.learn (agent.py:63-100) in ONE call: rb_replay_sample_fused_noise (zero-copy: no stack
 * gather), rb_learner_learn_windows on the frames/windows of `replay`, rb_learner_clip_adam — the same three entry points
 * with the same arguments, back to back.  Why it exists: the HIP runtime lets the launching thread run only about one
 * kernel ahead of the GPU, so host work BETWEEN launches (a Python interpreter between three ctypes calls: ~10 us twice
 * per step) shows up as idle GPU time (measured 2 x 6 us of a 194 us step); inside one call the launches are 3 us apart.
 * The priority write-back needs rb_learner_set_priority_sink(l, replay, tree_idx_dev) set beforehand.               */
typedef struct {
  rb_replay_t* replay;
  int32_t batch, max_attempts, window_len, reserved;
  double priority_weight;
  int64_t* tree_idx_dev;            /* sampler outputs (device, [batch]) */
  int64_t* actions_dev;
  float* returns_dev;
  float* nonterminals_dev;
  float* weights_dev;
  const rb_noise_job_t* noise_job;  /* from rb_learner_noise_job */
  const uint8_t* frames_dev;        /* rb_replay_buffers: frame ring and the sampler's window table */
  const int32_t* windows_dev;
  float* loss_dev;                  /* f32[batch]: per-sample loss (agent.py:94) */
  float* exp_avg_dev;               /* rb_learner_clip_adam arguments */
  float* exp_avg_sq_dev;
  float* norm_dev;
  double lr, beta1, beta2, eps;
  int64_t step;
  float max_norm;
  float reserved2;
} rb_train_step_t;
int rb_learner_train_step(rb_learner_t* l, const rb_train_step_t* a, rb_stream_t stream);

/* Learner options.
 * RB_LEARNER_FUSE_FC_H_DW: the hidden layer's weight gradient (93 % of all gradient bytes) is a rank-B product of two
 *   L2-resident matrices.  With this flag (and batch <= 32) rb_learner_learn* computes it for the global norm only and
 *   does NOT store it; rb_learner_clip_adam then recomputes each 16 x 64 tile on the MFMA units while it streams that
 *   tile's parameters and moments — 2 x 25.7 MB less HBM traffic per step at the canonical shape.  Contract: the next
 *   call on the handle after rb_learner_learn* must be rb_learner_clip_adam (clip_grad / grads_modified refuse), and
 *   grads_dev does not hold the hidden layer's weight gradient afterwards ...
 * RB_LEARNER_WRITE_FUSED_GRADS: ... unless this flag is set as well: the fused pass then also stores the tiles it
 *   computed (as clip_grad_norm_ leaves them), which is how the parity tests read the product path's gradient.
 * RB_LEARNER_DEFER_UPDATE: rb_learner_train_step (called with step = 0: a device step counter is required) leaves its
 *   clip + Adam pass (agent.py:97-98) PENDING instead of launching it.  The next rb_learner_train_step hosts it as extra
 *   workgroups of its sampler launch: the optimiser pass of learn call k does not depend on the sampling of call k + 1 nor
 *   the other way round (the priorities were written back in call k's backward), so 30 us of pure HBM streaming run
 *   beside the sampler's one latency-bound workgroup instead of in front of it.  Same arithmetic in the same order: the
 *   parameters are bit-identical to the undeferred ones.  EVERY other learner entry point that touches parameters,
 *   moments, gradients or the norm (act, act_batch, learn*, clip_*, sync_target, finish_grads, debug_read) first runs the
 *   pending pass as a launch of its own, on the stream it is given — use ONE stream per handle.  A caller that reads the
 *   borrowed buffers itself (params_dev, exp_avg, exp_avg_sq, grads_dev, norm_dev) calls rb_learner_flush first.       */
#define RB_LEARNER_FUSE_FC_H_DW 1
#define RB_LEARNER_WRITE_FUSED_GRADS 2
#define RB_LEARNER_DEFER_UPDATE 4
/* RB_LEARNER_IMPLICIT_SIGMA (with RB_LEARNER_DEFER_UPDATE, batch <= 32): the hidden layer's sigma-weight gradient is
 *   g_mu * (eps_out x eps_in) element for element (the product rule of model.py:44, what the backward forms), so the backward
 *   does not store it and the HOSTED optimiser pass forms it again from g_mu and a snapshot of the noise while it updates the
 *   (mu, sigma) pairs together: 12.85 MB less written and 12.85 MB less read per step at the canonical shape, bit-identical
 *   parameters.  grads_dev lacks that range until rb_learner_flush (which materialises it), rb_learner_clip_grad or a pass
 *   that runs as a launch of its own; rb_learner_grads_modified refuses while it is missing.                          */
#define RB_LEARNER_IMPLICIT_SIGMA 8
int rb_learner_set_flags(rb_learner_t* l, int32_t flags);
/* Run the pending optimiser pass, if any (RB_LEARNER_DEFER_UPDATE), on `stream`.  No-op otherwise.                    */
int rb_learner_flush(rb_learner_t* l, rb_stream_t stream);
/* The same deferral for a caller that issues a step's entry points one by one (rainbow_amd.Agent's eager path; the replica
 * exchange of SURVEY 8(e), where finish_grads sits between the backward and the optimiser pass, agent.py:96-97):
 *   rb_learner_clip_adam_deferred = rb_learner_clip_adam, but the pass is left pending when the flag is set and it can be
 *     (step = 0 with a device step counter; otherwise it runs at once);
 *   rb_learner_attach_pending(l, job_in, batch, job_out): job_out = job_in (from rb_learner_noise_job) plus the pending pass;
 *     returns 1 if one was attached, 0 if job_out is a plain copy, < 0 on error.  Pass job_out to
 *     rb_replay_sample_fused_noise, then call rb_learner_pending_launched(l).  No other call on the handle in between.  */
int rb_learner_clip_adam_deferred(rb_learner_t* l, float max_norm, float* exp_avg_dev, float* exp_avg_sq_dev, double lr,
                                  double beta1, double beta2, double eps, int64_t step, float* norm_dev, rb_stream_t stream);
int rb_learner_attach_pending(rb_learner_t* l, const rb_noise_job_t* job_in, int32_t batch, rb_noise_job_t* job_out);
int rb_learner_pending_launched(rb_learner_t* l);

/* hipGraph replay with the fused optimiser pass: a captured launch cannot take a new `step` by value.  With a counter set
 * (caller-owned i64 on the device), every rb_learner_learn* increments it on the device, and rb_learner_clip_adam called
 * with step = 0 reads the step number from it and forms the bias corrections 1 - beta^t in the kernel (double, one thread
 * per block).  NULL clears.                                                                                           */
int rb_learner_set_step_counter(rb_learner_t* l, int64_t* step_dev);

/* Fused priority write-back (agent.py:100 -> memory.py:157-159).  With a sink set, rb_learner_learn*
 * itself applies  sum_tree[tree_idx] = loss^w  (+ ancestor sums, max) to `replay` as one extra
 * workgroup of its backward launch, i.e. off the step's critical path.  tree_idx_dev must be the
 * index buffer the sampler fills each step (i64[batch]).  rb_learner_priority_written() tells
 * whether the LAST learn call did the write-back (it does not on the generic fallback path or
 * for batch > 256; the caller then calls rb_replay_update_priorities as usual).  Pass NULLs to
 * clear the sink.                                                                           */
int rb_learner_set_priority_sink(rb_learner_t* l, rb_replay_t* replay, const int64_t* tree_idx_dev);
int rb_learner_priority_written(rb_learner_t* l);

/* ---- Replica exchange (the reference is single-device, the insert point is between agent.py:96 and :97).
 * Every replica must apply the MEAN gradient.  27 MB of it are the noisy-linear weight gradients, and those are rank-B
 * products  dW = dY^T X  of two thin matrices: instead of all-reducing dW (2 x 7/8 x 27 MB through each GPU's xGMI links),
 * the replicas all-gather the FACTORS (dlogits, h, dh, feat rows of their batch, their noise vectors) and each computes the
 * mean gradient of the global batch from the gathered rows itself — identical inputs, identical kernel, identical bits on
 * every replica.  The conv gradients (the leading `small_floats` of grads_dev, 0.3 MB) ride in the SAME block, so a step
 * needs exactly ONE collective (an all-gather of `factor_floats` = 1.0 MB per rank at the canonical shape) and no event or
 * side stream (round 2 all-gathered the FC factors on a side stream under the backward and all-reduced the conv range
 * separately: two collectives, two stream joins — 75 us of idle GPU per step in the one-rank plumbing run).
 *   rb_learner_set_exchange(world > 1, local block, gathered blocks [world][factor_floats])   once
 *   per step:  rb_learner_learn*            input-gradient chain, conv grads, the complete local block; NO FC weight grads
 *              [caller: all-gather the local blocks into the gathered buffer, on the learn call's stream]
 *              rb_learner_finish_grads      FC weight/bias gradients of the global batch and the conv gradients' replica
 *                                           mean (rank order, x 1/world) into grads_dev + the norm partials
 *              rb_learner_clip_adam / rb_learner_clip_grad as usual
 * rb_learner_wait_factors is kept for ABI compatibility and does nothing (the block is complete in stream order).
 * world == 1 (default) restores the single-device step.  [small_offset, +small_floats) still names the conv range for
 * hosts that prefer `allreduce` of the flat gradient + rb_learner_grads_modified.                                     */
int rb_learner_exchange_layout(rb_learner_t* l, int64_t* factor_floats, int64_t* small_offset, int64_t* small_floats);
int rb_learner_set_exchange(rb_learner_t* l, int32_t world, float* factors_local_dev, const float* factors_all_dev);
int rb_learner_wait_factors(rb_learner_t* l, rb_stream_t side_stream);
int rb_learner_finish_grads(rb_learner_t* l, rb_stream_t stream);

/* The same exchange WITHOUT a host framework in the loop: the library issues the all-gather itself on an RCCL
 * communicator it owns (librccl is resolved at run time with dlopen — the library has no link-time dependency on it), so a
 * C / C++ host needs no torch.distributed, and the Python host saves the ~16 us per step torch.distributed spends around a
 * 1 MB collective (profiles/round3_defer_dist1.txt).  One communicator per process / GPU:
 *   rank 0:     rb_comm_unique_id(id)                  128 opaque bytes (ncclGetUniqueId); ship them to every rank out of band
 *   every rank: rb_comm_create(&comm, id, world, rank) ncclCommInitRank on the CURRENT device (collective: all ranks call it)
 *   per step:   rb_learner_exchange_rccl(l, comm, stream)  =  ncclAllGather(local block -> gathered blocks) on `stream`
 *                                                             + rb_learner_finish_grads(l, stream)
 * A one-rank communicator with a two-block exchange buffer (the single-GPU plumbing run, RAINBOW_AMD_FORCE_DIST=1) gathers
 * into block 0 and copies it to block 1.  Replaces the [caller: all-gather] step above; insert point agent.py:96-97.        */
typedef struct rb_comm rb_comm_t;
/* 1 when librccl can be loaded by this process (dlopen + symbol lookup only: creates nothing), else 0.  What every rank asks
 * before the group decides whether to build library-owned communicators; only rank 0 then calls rb_comm_unique_id.          */
int rb_comm_available(void);
int rb_comm_unique_id(void* id128);
int rb_comm_create(rb_comm_t** out, const void* id128, int32_t world, int32_t rank);
int rb_comm_destroy(rb_comm_t* comm);   /* waits for the stream of the communicator's last all-gather first */
int rb_learner_exchange_rccl(rb_learner_t* l, rb_comm_t* comm, rb_stream_t stream);
/* rb_learner_train_step with the exchange in its place (sampler + noise, learn, exchange_rccl, clip + Adam): the replica
 * step as ONE C call.  Needs rb_learner_set_exchange to be armed for comm's world size.                                  */
int rb_learner_train_step_dist(rb_learner_t* l, const rb_train_step_t* a, rb_comm_t* comm, rb_stream_t stream);

/* Tell the library the caller changed grads_dev after rb_learner_learn (e.g. the RCCL
 * all-reduce of the replica path): the sum of squares the backward kernels accumulated on
 * the fly is then stale and rb_learner_clip_grad re-reads the gradient.                    */
int rb_learner_grads_modified(rb_learner_t* l);

/* Exact resume: the noise generator is Philox(seed, epoch, index) with a device-resident epoch that
 * every resample advances; a checkpoint that carries (seed, epoch) — next to parameters, Adam moments and the noise
 * buffers — continues the very same random stream.  Both calls synchronise `stream`.                               */
int rb_learner_get_rng(rb_learner_t* l, uint64_t* seed_host, uint64_t* epoch_host, rb_stream_t stream);
int rb_learner_set_rng(rb_learner_t* l, uint64_t seed, uint64_t epoch, rb_stream_t stream);

/* Agent.update_target_net (agent.py:102-103): params AND noise, device-to-device.   */
int rb_learner_sync_target(rb_learner_t* l, rb_stream_t stream);

/* White-box access for parity tests: copies an internal activation to out_dev.
 * what: 0 log_ps_a [B][atoms], 1 m (projected target) [B][atoms], 2 argmax a* i32[B],
 *       3 pns_a [B][atoms], 4 logits [3B][atoms*(actions+1)],
 *       5 hidden activations relu(fc_h_v | fc_h_a) of the differentiated forward [B][2*hidden] (model.py:72-73) — read WITHOUT
 *         running a pending optimiser pass (it is no parameter): h > 0 are the ReLU decisions the backward used.          */
int rb_learner_debug_read(rb_learner_t* l, int32_t what, void* out_dev, rb_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RAINBOW_HIP_H_ */
