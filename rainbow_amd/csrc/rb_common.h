// rb_common.h — host-side error plumbing + small device helpers shared by the kernels.
#pragma once
#include <stdarg.h>
#include <stdio.h>

#include "../../include/rainbow_hip.h"
#include "rb_device.h"

// ---- error reporting (never throws across the C ABI) -------------------------------
void rb_set_error(const char* fmt, ...);

#define RB_HIP_TRY(expr)                                                                    \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      rb_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
      return RB_ERR_HIP;                                                                    \
    }                                                                                       \
  } while (0)

#define RB_REQUIRE(cond, ...)     \
  do {                            \
    if (!(cond)) {                \
      rb_set_error(__VA_ARGS__);  \
      return RB_ERR_INVALID;      \
    }                             \
  } while (0)

#define RB_LAUNCH_CHECK()                                                              \
  do {                                                                                 \
    hipError_t e_ = hipGetLastError();                                                 \
    if (e_ != hipSuccess) {                                                            \
      rb_set_error("%s:%d: kernel launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return RB_ERR_HIP;                                                               \
    }                                                                                  \
  } while (0)

// ---- device allocations of the library (workspaces, the replay ring) ---------------
// hipMalloc / hipFree, except under RB_GUARD=1 (test runs): every block is then surrounded by two 4 KiB guard bands filled
// with a fixed byte, and rb_debug_check_guards (C ABI) reports any band a kernel has written into.
hipError_t rb_dev_malloc(void** p, size_t bytes);
void rb_dev_free(void* p);

static inline int64_t rb_div_up(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Is `stream` being captured into a hipGraph right now?  (A query error counts as "yes": the callers take the conservative path.)
// Whatever a captured launch bakes in is replayed verbatim: nothing in it may depend on work outside the graph or on a launch number.
static inline bool rb_stream_capturing(void* stream) {
#if defined(RB_HOST_INTERP)
  (void)stream;
  return false;
#else
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing((hipStream_t)stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone;
#endif
}

// ---- RB_OPTS="key=value,key=value": the library's ONE tuning / test-hook variable (common.hip; DESIGN.md §8 lists the keys).
// Returns `dflt` when the key is absent.  Read when a handle is created, never per launch.
int rb_opt(const char* key, int dflt);

// ---- Philox4x32-10 counter RNG (device sampler + noise) ------------------------------
struct rb_philox_out {
  uint32_t v[4];
};
__device__ __forceinline__ rb_philox_out rb_philox(uint64_t seed, uint64_t ctr_hi, uint64_t ctr_lo) {
  uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32), c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  rb_philox_out o;
  o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}
// 53-bit uniform in [0,1)
__device__ __forceinline__ double rb_u53(uint32_t hi, uint32_t lo) {
  const uint64_t x = (((uint64_t)hi << 32) | lo) >> 11;
  return (double)x * (1.0 / 9007199254740992.0);
}

// ---- block reductions (blockDim.x a multiple of 64, <= 1024; all threads call) -------
__device__ __forceinline__ float rb_block_max(float v, float* lds16) {
  v = rb_wave_max(v);
  __syncthreads();
  if (rb_lane() == 0) lds16[rb_wave()] = v;
  __syncthreads();
  const int nw = (int)(blockDim.x >> 6);
  float r = lds16[0];
  for (int w = 1; w < nw; ++w) r = fmaxf(r, lds16[w]);
  return r;
}
__device__ __forceinline__ float rb_block_sum(float v, float* lds16) {
  v = rb_wave_sum(v);
  __syncthreads();
  if (rb_lane() == 0) lds16[rb_wave()] = v;
  __syncthreads();
  const int nw = (int)(blockDim.x >> 6);
  float r = lds16[0];
  for (int w = 1; w < nw; ++w) r += lds16[w];
  return r;
}
__device__ __forceinline__ int rb_block_all(int pred, int* lds16) {
  const int wv = __all(pred);
  __syncthreads();
  if (rb_lane() == 0) lds16[rb_wave()] = wv;
  __syncthreads();
  const int nw = (int)(blockDim.x >> 6);
  int r = 1;
  for (int w = 0; w < nw; ++w) r &= lds16[w];
  return r;
}
