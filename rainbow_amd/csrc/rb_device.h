// rb_device.h — the one place the kernels touch the platform.
//
// Product build: hipcc --offload-arch=gfx950 (CDNA4, wave64, MFMA, 160 KiB LDS).
// The RB_HOST_INTERP branch is NOT a portability layer: it binds the same kernel
// sources to tests/hipemu (a fiber-based host interpreter) so kernel logic can be
// unit-tested in the GPU-less build container.  Nothing in the shipped library or
// in the rainbow_amd package is ever built with RB_HOST_INTERP.
#pragma once

#if defined(RB_HOST_INTERP)
#include "hipemu.h"
#define RB_LAUNCH(kern, grid, block, stream, ...) \
  hipemu::launch([&]() { kern(__VA_ARGS__); }, (grid), (block))
#define RB_LAUNCH_T(tag, kern, grid, block, stream, ...) RB_LAUNCH(kern, grid, block, stream, __VA_ARGS__)
#else
#include <hip/hip_runtime.h>
typedef float rb_f32x16 __attribute__((ext_vector_type(16)));
typedef float rb_f32x4 __attribute__((ext_vector_type(4)));
// launch + optional event bracket (rb_profile_select, common.hip)
bool rb_prof_begin(const char* kernel_expr, hipStream_t stream);
void rb_prof_end(hipStream_t stream);
extern int g_rb_prof_on;
// RB_LAUNCH_T: same, with an explicit profiling tag (a kernel used for several layers gets one tag per layer)
// RB_HOST_TIMING builds: host time spent inside each launch call, per tag (rb_debug_host_timing, common.hip)
#if defined(RB_HOST_TIMING)
void rb_host_time_add(const char* tag, double us, double t0);
double rb_host_now_us();
#define RB_LAUNCH_T(tag, kern, grid, block, stream, ...)                                        \
  do {                                                                                          \
    const double rb_t0_ = rb_host_now_us();                                                     \
    hipLaunchKernelGGL(kern, (grid), (block), 0, (hipStream_t)(stream), __VA_ARGS__);           \
    rb_host_time_add(tag, rb_host_now_us() - rb_t0_, rb_t0_);                                   \
  } while (0)
#else
#define RB_LAUNCH_T(tag, kern, grid, block, stream, ...)                                        \
  do {                                                                                          \
    const bool rb_pf_ = g_rb_prof_on && rb_prof_begin(tag, (hipStream_t)(stream));              \
    hipLaunchKernelGGL(kern, (grid), (block), 0, (hipStream_t)(stream), __VA_ARGS__);           \
    if (rb_pf_) rb_prof_end((hipStream_t)(stream));                                             \
  } while (0)
#endif
#define RB_LAUNCH(kern, grid, block, stream, ...) RB_LAUNCH_T(#kern, kern, grid, block, stream, __VA_ARGS__)
#endif

#include <stdint.h>

#define RB_WAVE 64

// v_mfma_f32_32x32x2_f32: D(32x32) += A(32x2) * B(2x32), exact f32 fmaf chain.
// Lane l supplies A[l&31][l>>5] and B[l>>5][l&31]; D[r] sits at
// row (r&3) + 8*(r>>2) + 4*(l>>5), col l&31.
__device__ __forceinline__ rb_f32x16 rb_mfma32(float a, float b, rb_f32x16 c) {
#if defined(RB_HOST_INTERP)
  return hipemu_mfma_f32_32x32x2f32(a, b, c);
#else
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
#endif
}

// v_mfma_f32_16x16x4_f32: D(16x16) += A(16x4) * B(4x16).  Lane l supplies A[l&15][l>>4] and B[l>>4][l&15];
// D[r] sits at row 4*(l>>4) + r, col l&15.  32-cycle issue, 40-cycle dependent latency.
__device__ __forceinline__ rb_f32x4 rb_mfma16(float a, float b, rb_f32x4 c) {
#if defined(RB_HOST_INTERP)
  return hipemu_mfma_f32_16x16x4f32(a, b, c);
#else
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#endif
}

__device__ __forceinline__ int rb_mfma_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// Instruction-scheduler fence: the compiler may not move anything across it.  Used to pin the ISSUE ORDER of the
// prefetch loads in software-pipelined loops (left alone, the machine scheduler sinks all refill loads of an unrolled
// ring to the end of the loop body, which turns the ring into "burst, stall, compute").
#if defined(RB_HOST_INTERP)
#define RB_SCHED_FENCE() ((void)0)
#else
#define RB_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

// +1 on a word the HOST reads (pinned, device-mapped memory), by its only writer: a system-scope atomic load and store —
// global instructions when `p` is a kernel argument (a volatile read-modify-write through it was compiled to FLAT
// loads/stores), and no read-modify-write travels over PCIe
__device__ __forceinline__ void rb_atomic_inc_system(int32_t* p) {
#if defined(RB_HOST_INTERP)
  *p = *p + 1;
#else
  const int32_t c = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(p, c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
}

// ---- in-launch hand-off between workgroups of ONE launch (block-range "roles" with a partial dependency, e.g. the conv
// layers of one image).  The 8 XCDs have private, mutually non-coherent L2s and every CU a private L1, so a hand-off is an
// agent-scope release on the producer and an agent-scope acquire on EVERY consumer workgroup (cdna_hip_programming
// Guideline 16): producer = stores -> every wave drains (vmcnt 0) -> workgroup barrier -> one lane: release fence, drain,
// relaxed agent-scope add on the arrival counter; consumer = one lane polls the counter RELAXED (an acquire per poll would
// drop the CU's L1 every iteration), then ONE acquire fence, then the workgroup barrier, then plain loads.
// Counters are MONOTONIC across launches (target = launch number x arrivals per launch, passed by value): no memset
// between launches, no in-kernel reset.  Deadlock freedom: a consumer's producers always have LOWER block indices in a
// 1-D grid, workgroups are dispatched in index order, so every producer is resident or finished before a consumer can
// occupy a CU.  The spin is bounded all the same: on expiry the consumer flags `err` (read back by the host) and goes on.
// WT = the payload was stored write-through (sc1: rb_st1_wt / rb_st4_wt) — it is in memory once the stores have drained,
// no release fence (a fence writes back EVERY dirty line of the XCD's L2, not just this workgroup's).
// COH = the consumer reads the payload with agent-coherent loads (sc1: rb_ld*_buf_sc1) — no acquire fence (which drops the
// whole L1 of the CU, ~1.7 us).
template <bool WT>
__device__ __forceinline__ void rb_chain_signal(unsigned* counter) {       // all threads of the workgroup call
#if defined(RB_HOST_INTERP)
  __syncthreads();
  if (threadIdx.x == 0) *counter = *counter + 1u;
#else
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // this wave's stores have left the CU
  __syncthreads();
  if (threadIdx.x == 0) {
    if (!WT) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");                   // write the XCD's dirty lines back
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // (the fence's own wait, restated: Guideline 16 pitfall 12)
    }
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#endif
}
template <bool COH>
__device__ __forceinline__ void rb_chain_wait(const unsigned* counter, unsigned target, unsigned* err) {   // all threads call
#if defined(RB_HOST_INTERP)
  if (threadIdx.x == 0 && (int)(*counter - target) < 0) *err = 1u;         // blocks run in index order: producers are done
  __syncthreads();
#else
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    while ((int)(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
      __builtin_amdgcn_s_sleep(4);
      if (++spins > (1u << 22)) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }   // ~1 s
    }
    if (!COH) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
#endif
}

// The same hand-off for ALL-TO-ALL boundaries of a persistent launch (every workgroup arrives, every workgroup waits): one
// counter serialises its arrivals at ~12 ns each (256 workgroups: 3.3-4.4 us per boundary, tools/stamp/act_timeline.py), so
// the arrivals are SHARDED over 8 counters in 8 different 128-byte lines (shard = workgroup index mod 8, i.e. its XCD) and a
// waiter reads all eight in one batch.  `counters` = the boundary's 8 x 32 words; per_shard = arrivals per shard and launch.
#ifndef RB_FAN_SHARDS
#define RB_FAN_SHARDS 32           // (8 until round 6: with 256 producers — the hidden layer's phase — 32 arrivals per line still took 3 us)
#endif
#define RB_FAN_STRIDE 32            // words between shards (128 bytes)
__device__ __forceinline__ void rb_fan_signal(unsigned* counters, int wg) {           // all threads of the workgroup call
#if defined(RB_HOST_INTERP)
  __syncthreads();
  if (threadIdx.x == 0) counters[(wg % RB_FAN_SHARDS) * RB_FAN_STRIDE] += 1u;
#else
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // this wave's (write-through) stores have left the CU
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(counters + (wg % RB_FAN_SHARDS) * RB_FAN_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
// err_tag: what an expired wait stores into *err (the launch number: the reader compares with its own, so one failed launch
// does not poison the next)
__device__ __forceinline__ void rb_fan_wait(const unsigned* counters, unsigned target_per_shard, unsigned* err, unsigned err_tag = 1u) {   // all threads call
#if defined(RB_HOST_INTERP)
  __syncthreads();                                                          // (one launch per phase there: nothing to wait for)
  (void)counters; (void)target_per_shard; (void)err; (void)err_tag;
#else
  if (threadIdx.x < 64) {                                                   // wave 0: lane s polls shard s
    const int lane = (int)threadIdx.x;
    unsigned spins = 0;
    for (;;) {
      unsigned v = target_per_shard;
      if (lane < RB_FAN_SHARDS) v = __hip_atomic_load(counters + lane * RB_FAN_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__all((int)(v - target_per_shard) >= 0)) break;
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1u << 22)) { if (lane == 0) __hip_atomic_store(err, err_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
  }
  __syncthreads();
#endif
}
// The same wait where only the workgroups [0, producers) of the launch arrive at this boundary (shard = index mod 8, so shard s
// gets (producers - s + 7) / 8 arrivals per launch): `launches` = this launch's number, counters monotonic as above.
__device__ __forceinline__ void rb_fan_wait_first(const unsigned* counters, unsigned launches, int producers, unsigned* err, unsigned err_tag = 1u) {   // all threads call
#if defined(RB_HOST_INTERP)
  __syncthreads();
  (void)counters; (void)launches; (void)producers; (void)err; (void)err_tag;
#else
  if (threadIdx.x < 64) {
    const int lane = (int)threadIdx.x;
    const unsigned target = lane < RB_FAN_SHARDS ? launches * (unsigned)((producers - lane + RB_FAN_SHARDS - 1) / RB_FAN_SHARDS) : 0u;
    unsigned spins = 0;
    for (;;) {
      unsigned v = target;
      if (lane < RB_FAN_SHARDS && lane < producers) v = __hip_atomic_load(counters + lane * RB_FAN_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__all((int)(v - target) >= 0)) break;
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1u << 22)) { if (lane == 0) __hip_atomic_store(err, err_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
  }
  __syncthreads();
#endif
}

// A 4-byte load through a pointer the CALLER knows to be global memory: where a pointer is a runtime choice among several
// kernel arguments (the three frame sources of the first conv layer) the compiler may fall back to a generic pointer and a FLAT
// load, which also probes the LDS and scratch apertures; the address-space cast pins it to global_load.
__device__ __forceinline__ unsigned rb_ldg_u32(const void* p) {
#if defined(RB_HOST_INTERP)
  return *reinterpret_cast<const unsigned*>(p);
#else
  return *(const __attribute__((address_space(1))) unsigned*)(p);
#endif
}

// 16-byte global/LDS accesses (pointers must be 16-byte aligned)
__device__ __forceinline__ float4 rb_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void rb_st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// write-through 16-byte store (sc1): the line leaves the XCD's L2 as it is written instead of staying dirty until the
// end-of-kernel release, which otherwise flushes up to 8 x 4 MB behind a streaming kernel (MI355X_MICROARCH: stores).
// Issued as a raw buffer store (wave-uniform base, per-lane byte offset < 2 GB) so that it is a compiler-known
// instruction: a hand-written `global_store ... sc1` in inline asm escaped the hazard recogniser and corrupted data.
__device__ __forceinline__ void rb_st4_wt(float* base, unsigned byte_off, float4 v) {
#if defined(RB_HOST_INTERP)
  *reinterpret_cast<float4*>(reinterpret_cast<char*>(base) + byte_off) = v;
#else
  typedef unsigned int rb_v4u __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00027000);
  rb_v4u t;
  t.x = __float_as_uint(v.x); t.y = __float_as_uint(v.y); t.z = __float_as_uint(v.z); t.w = __float_as_uint(v.w);
  __builtin_amdgcn_raw_buffer_store_b128(t, r, (int)byte_off, 0, 16);   // aux bit 4 = sc1
#endif
}

// 4-byte write-through store (same idea for scattered element stores: a strided writer — the stride-2 phases of a
// transposed conv — leaves PARTIAL dirty lines in several XCDs' L2s, whose end-of-kernel write-back is slow and is charged
// to the next kernel's start)
__device__ __forceinline__ void rb_st1_wt(float* base, unsigned byte_off, float v) {
#if defined(RB_HOST_INTERP)
  *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off) = v;
#else
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00027000);
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)byte_off, 0, 16);   // aux bit 4 = sc1
#endif
}

// 16-byte load through a buffer descriptor: address = base + per-lane byte offset (VGPR) + wave-uniform byte offset (SGPR).
// ONE instruction and no address arithmetic in the loop — a streamed kernel whose loads are `pointer + 64-bit index` spends
// 5-6 VALU/SALU instructions per load on the address alone.  Offsets are 32-bit: buffers up to 2 GB.
struct rb_buf {
#if defined(RB_HOST_INTERP)
  const char* base;
  size_t nbytes;
#else
  __amdgpu_buffer_rsrc_t r;
#endif
};
__device__ __forceinline__ rb_buf rb_make_buf(const void* base) {
  rb_buf b;
#if defined(RB_HOST_INTERP)
  b.base = reinterpret_cast<const char*>(base);
  b.nbytes = (size_t)0x7fffffff;
#else
  b.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00027000);
#endif
  return b;
}
// ... with hardware range checking: a load whose byte range leaves [0, nbytes) returns zeros instead of faulting, so clamped
// "tail" addresses need no per-lane min() (the host interpreter checks the same bound).
__device__ __forceinline__ rb_buf rb_make_buf_n(const void* base, size_t nbytes) {
  rb_buf b;
#if defined(RB_HOST_INTERP)
  b.base = reinterpret_cast<const char*>(base);
  b.nbytes = nbytes;
#else
  b.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(nbytes > 0x7fffffffu ? 0x7fffffffu : nbytes), 0x00027000);
#endif
  return b;
}
__device__ __forceinline__ float4 rb_ld4_buf(const rb_buf& b, unsigned lane_off, unsigned uniform_off) {
#if defined(RB_HOST_INTERP)
  if ((size_t)lane_off + uniform_off + 16 > b.nbytes) return make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  return *reinterpret_cast<const float4*>(b.base + lane_off + uniform_off);
#else
  typedef unsigned int rb_v4u __attribute__((ext_vector_type(4)));
  const rb_v4u t = __builtin_amdgcn_raw_buffer_load_b128(b.r, (int)lane_off, (int)uniform_off, 0);
  float4 v;
  v.x = __uint_as_float(t.x); v.y = __uint_as_float(t.y); v.z = __uint_as_float(t.z); v.w = __uint_as_float(t.w);
  return v;
#endif
}

// agent-coherent variants (sc1): the load is served from the point where the XCDs agree (not from this CU's L1 / a stale
// line of this XCD's L2) — for data another workgroup of the SAME launch produced (rb_chain_*)
__device__ __forceinline__ float4 rb_ld4_buf_sc1(const rb_buf& b, unsigned lane_off, unsigned uniform_off) {
#if defined(RB_HOST_INTERP)
  return rb_ld4_buf(b, lane_off, uniform_off);
#else
  typedef unsigned int rb_v4u __attribute__((ext_vector_type(4)));
  const rb_v4u t = __builtin_amdgcn_raw_buffer_load_b128(b.r, (int)lane_off, (int)uniform_off, 16);   // aux bit 4 = sc1
  float4 v;
  v.x = __uint_as_float(t.x); v.y = __uint_as_float(t.y); v.z = __uint_as_float(t.z); v.w = __uint_as_float(t.w);
  return v;
#endif
}
__device__ __forceinline__ float rb_ld1_buf_sc1(const rb_buf& b, unsigned lane_off, unsigned uniform_off) {
#if defined(RB_HOST_INTERP)
  if ((size_t)lane_off + uniform_off + 4 > b.nbytes) return 0.0f;
  return *reinterpret_cast<const float*>(b.base + lane_off + uniform_off);
#else
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(b.r, (int)lane_off, (int)uniform_off, 16));
#endif
}

__device__ __forceinline__ float rb_ld1_buf(const rb_buf& b, unsigned lane_off, unsigned uniform_off) {
#if defined(RB_HOST_INTERP)
  if ((size_t)lane_off + uniform_off + 4 > b.nbytes) return 0.0f;
  return *reinterpret_cast<const float*>(b.base + lane_off + uniform_off);
#else
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(b.r, (int)lane_off, (int)uniform_off, 0));
#endif
}

__device__ __forceinline__ int rb_lane() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ int rb_wave() { return (int)(threadIdx.x >> 6); }

// a value the caller knows to be identical in all lanes of the wave: lets the compiler keep it (and everything derived
// from it: loop bounds, weight addresses) in scalar registers
__device__ __forceinline__ int rb_wave_uniform(int v) {
#if defined(RB_HOST_INTERP)
  return v;
#else
  return __builtin_amdgcn_readfirstlane(v);
#endif
}

// Ordering point for LDS traffic that stays inside one wave (a lane reads what another lane of the SAME wave wrote):
// the hardware executes a wave's LDS operations in order, so no s_barrier is needed — this only pins the compiler's
// (and the host interpreter's) ordering.
__device__ __forceinline__ void rb_wave_sync() {
#if defined(RB_HOST_INTERP)
  hipemu::wave_barrier();
#else
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// wave64 reductions (all 64 lanes must call; the result is returned to every lane).
// Six DPP steps — row_shr 1, 2, 4, 8 build each 16-lane row's total in its last lane, row_bcast:15 / :31 fold the rows
// into lane 63 — then one v_readlane.  The butterfly of __shfl_xor compiles to six ds_bpermute round trips through the
// LDS crossbar (~0.3-0.5 us per reduction: the head kernel's per-sample chain has ten of them, the sampler's three);
// a DPP operand costs an ordinary VALU slot.  Fixed summation order (deterministic), different from the butterfly's.
#if defined(RB_HOST_INTERP)
__device__ __forceinline__ float rb_wave_sum(float v) {
  // same association as the DPP sequence: prefix sums inside rows of 16, then row 0 + 1 and row 2 + 3, then the halves
  float x = v;
  for (int sh = 1; sh <= 8; sh <<= 1) {
    const float y = __shfl_up(x, (unsigned)sh, 64);
    if ((rb_lane() & 15) >= sh) x = x + y;
  }
  const float r0 = __shfl(x, 15, 64), r1 = __shfl(x, 31, 64), r2 = __shfl(x, 47, 64), r3 = __shfl(x, 63, 64);
  return (r3 + r2) + (r1 + r0);
}
__device__ __forceinline__ float rb_wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
  return v;
}
#else
#define RB_DPP_F(old, src, ctrl, rmask, bmask) \
  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float)(old)), __builtin_bit_cast(int, (float)(src)), ctrl, rmask, bmask, false))
__device__ __forceinline__ float rb_wave_sum(float v) {
  v += RB_DPP_F(0.0f, v, 0x111, 0xf, 0xf);       // row_shr:1
  v += RB_DPP_F(0.0f, v, 0x112, 0xf, 0xf);       // row_shr:2
  v += RB_DPP_F(0.0f, v, 0x114, 0xf, 0xf);       // row_shr:4
  v += RB_DPP_F(0.0f, v, 0x118, 0xf, 0xf);       // row_shr:8   -> lane 15 of every row holds the row's sum
  v += RB_DPP_F(0.0f, v, 0x142, 0xa, 0xf);       // row_bcast:15 into rows 1, 3
  v += RB_DPP_F(0.0f, v, 0x143, 0xc, 0xf);       // row_bcast:31 into rows 2, 3  -> lane 63 holds the wave's sum
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float rb_wave_max(float v) {
  const float ninf = -__builtin_inff();
  v = fmaxf(v, RB_DPP_F(ninf, v, 0x111, 0xf, 0xf));
  v = fmaxf(v, RB_DPP_F(ninf, v, 0x112, 0xf, 0xf));
  v = fmaxf(v, RB_DPP_F(ninf, v, 0x114, 0xf, 0xf));
  v = fmaxf(v, RB_DPP_F(ninf, v, 0x118, 0xf, 0xf));
  v = fmaxf(v, RB_DPP_F(ninf, v, 0x142, 0xa, 0xf));
  v = fmaxf(v, RB_DPP_F(ninf, v, 0x143, 0xc, 0xf));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
#endif
__device__ __forceinline__ double rb_wave_sum_f64(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
