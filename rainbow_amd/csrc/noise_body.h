// noise_body.h — factorised-Gaussian noise resampling (model.py:32-40) as a device body that any launch can host:
// k_noise (learner.hip) runs it alone; k_sample (replay.hip) can carry it as extra workgroups so that the per-step
// noise draw costs no kernel boundary of its own.
#pragma once
#include "rb_common.h"

struct NoiseMap {
  // 32-bit on purpose: these 17 values live in SGPRs of every kernel that hosts the noise workgroups (as int64 they
  // helped push k_sample over the SGPR budget).  Draw counts and noise-buffer offsets are far below 2^31.
  int32_t seg_begin[9];   // prefix of draw counts: hv_in, hv_out, ha_in, ha_out, zv_in, zv_out, za_in, za_out
  int32_t dst[8];         // destination offsets in the noise buffer
};

// f(x) = sign(x) * sqrt(|x|)  (model.py:32-34).  raw == NULL: N(0,1) from Philox + Box-Muller.
// Draw order = the reference's: per layer randn(in) then randn(out); fc_h_v, fc_h_a, fc_z_v, fc_z_a
// (model.py:36-38, 82-85).  The Philox epoch is DEVICE state (ctr[0]) so that a captured hipGraph draws fresh noise
// on every replay; the last workgroup to finish (ticket in ctr[1]) advances it — every block has read the epoch
// before it takes a ticket.  `blk` / `nblk` = this workgroup's index / count among the noise workgroups of ONE net,
// `net` in [0, nets): net 1 resamples noise2 with epoch+1 — exactly the draws two single-net launches would make.
__device__ __forceinline__ void rb_noise_body(float* noise, float* noise2, const float* raw, const NoiseMap& map,
                                              uint64_t seed, unsigned long long* ctr, int blk, int nblk, int net, int nets) {
  const uint64_t epoch = ctr[0] + (uint64_t)net;
  if (net == 1) noise = noise2;
  const int total = map.seg_begin[8];
  for (int i = blk * (int)blockDim.x + (int)threadIdx.x; i < total; i += nblk * (int)blockDim.x) {
    float x;
    if (raw) {
      x = raw[i];
    } else {
      const rb_philox_out r = rb_philox(seed, epoch, (uint64_t)(i >> 1));
      const float u1 = ((float)(r.v[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
      const float u2 = ((float)(r.v[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
      const float rad = sqrtf(-2.0f * logf(u1));
      const float ang = 6.283185307179586f * u2;
      x = (i & 1) ? rad * sinf(ang) : rad * cosf(ang);
    }
    const float s = x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f);
    const float f = s * sqrtf(fabsf(x));
    int seg = 0;
#pragma unroll
    for (int q = 1; q < 8; ++q) seg += (i >= map.seg_begin[q]) ? 1 : 0;
    noise[map.dst[seg] + (i - map.seg_begin[seg])] = f;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const unsigned long long ticket = atomicAdd(&ctr[1], 1ull);
    if (ticket == (unsigned long long)(nblk * nets) - 1ull) {
      ctr[0] = ctr[0] + (unsigned long long)nets;
      ctr[1] = 0;
    }
  }
}

// everything a foreign launch needs to host the noise workgroups (filled by rb_learner_noise_job)
struct NoiseJob {
  float* noise;
  float* noise2;
  NoiseMap map;
  uint64_t seed;
  unsigned long long* ctr;
  int nblk, nets;
  const NoiseJob* dev;    // device-resident copy of this struct: a hosting kernel takes this pointer (8 bytes of kernel
                          // arguments instead of 120) and reads the fields inside its noise branch only
  // a second tenant for the same launch (adam_body.h): the previous learn call's deferred optimiser pass — device-resident
  // ClipAdamArgs and the number of 256-thread workgroups it needs (0 = none).  Filled in by rb_learner_train_step only.
  const void* adam_dev;
  int adam_blocks;
};
