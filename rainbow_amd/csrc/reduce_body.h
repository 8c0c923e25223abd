// reduce_body.h — the fixed-order sum of the conv weight-gradient slices (learner.hip k_reduce_conv_dw_all) as a device body that a
// foreign launch can host.  A learn call's conv weight-gradient launch (conv_lds.h k_conv_dw_all) leaves `slices` partial images per
// layer; their sum (slice 0 + 1 + 2 + ..., one thread per gradient element, ALL slice loads of an element in flight at once) is the
// gradient, and every 64 elements leave one sum-of-squares partial for clip_grad_norm_ (agent.py:97).  As a launch of its own this
// is a 6 us latency chain at the very end of the step; folded into the NEXT step's sampler launch (adam_body.h: the pending
// optimiser pass is hosted there, and it is the only consumer) it runs beside 30 us of streaming.
#pragma once
#include "rb_common.h"

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
  float* sq_part;         // optional: one slot per 64 elements = sum of squares of the gradients those threads produced
  // replica exchange: every reduced element is ALSO stored at copy_base + (its offset inside the flat gradient), i.e. into
  // the conv segment of this rank's exchange block
  const float* grads_base;
  float* copy_base;
  // tenant blocks behind the reduction's own (k_reduce_conv_dw_all only): copy snap_n floats (the learn call's online noise, for
  // the optimiser pass that forms the hidden layer's sigma gradient itself: the launch hosting that pass resamples the noise)
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

// The hosted form: element i of the fused index space by the calling thread (i >= total: nothing), every memory access through a
// buffer descriptor (the pointers come out of device memory: adam_body.h), results stored WRITE-THROUGH (the consumers are other
// workgroups of the same launch: rb_chain_signal<true> / sc1 loads).  Returns the square of the element (0 outside the range).
// Same arithmetic as k_reduce_conv_dw_all: ((0 + s0) + s1) + ...
#ifndef RB_RED_HOSTED_BATCH
#define RB_RED_HOSTED_BATCH 48
#endif
// `N` slices [s0, s0 + N) of element j4 / 4 added onto acc in slice order (clamped loads: always legal; all N in flight at once)
template <int N>
__device__ __forceinline__ float rb_add_slices_buf(float acc, const rb_buf& b, unsigned per4, unsigned j4, int s0, int slices) {
  float v[N];
#pragma unroll
  for (int u = 0; u < N; ++u) v[u] = rb_ld1_buf(b, j4, (unsigned)(s0 + u < slices ? s0 + u : slices - 1) * per4);
#pragma unroll
  for (int u = 0; u < N; ++u) acc += (s0 + u < slices) ? v[u] : 0.0f;
  return acc;
}
// `ra` points into a kernel argument's target (device memory, global address space): the layer's fields are read where they are
__device__ __forceinline__ float rb_reduce_conv_elem_hosted(const ReduceAllArgs* ra, int64_t i) {
  if (i >= ra->total) return 0.0f;
  const ReduceLayer* L = &ra->layer[0];
  if (ra->n_layers > 1 && i >= ra->layer[1].begin) L = &ra->layer[1];
  if (ra->n_layers > 2 && i >= ra->layer[2].begin) L = &ra->layer[2];
  const int slices = L->slices, K = L->K;
  const unsigned j = (unsigned)(i - L->begin);
  const unsigned per4 = 4u * (unsigned)L->cout * (unsigned)(K + 1);
  const rb_buf bp = rb_make_buf(L->part);
  // ((0 + s0) + s1) + ...: k_reduce_conv_dw_all's order.  Up to 32 slices in ONE batch of loads; beyond, batches of
  // RB_RED_HOSTED_BATCH (the hosting sampler kernel has 128 registers per lane: 64 or 96 loads in flight spilled)
  float acc = 0.0f;
  if (slices <= 32) acc = rb_add_slices_buf<32>(acc, bp, per4, 4u * j, 0, slices);
  else
    for (int s0 = 0; s0 < slices; s0 += RB_RED_HOSTED_BATCH) acc = rb_add_slices_buf<RB_RED_HOSTED_BATCH>(acc, bp, per4, 4u * j, s0, slices);
  const unsigned co = j / (unsigned)(K + 1), col = j - co * (unsigned)(K + 1);
  if (col < (unsigned)K) rb_st1_wt(L->gw, 4u * (co * (unsigned)K + col), acc);
  else rb_st1_wt(L->gb, 4u * co, acc);
  return acc * acc;
}
