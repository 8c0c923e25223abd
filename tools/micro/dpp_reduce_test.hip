// checks rb_wave_sum / rb_wave_max (rb_device.h, DPP form) against a serial reference on the device
#include "../../rainbow_amd/csrc/rb_device.h"
#include <cstdio>
#include <cmath>
__global__ void k(const float* in, float* out_sum, float* out_max) {
  const float v = in[blockIdx.x * 64 + threadIdx.x];
  out_sum[blockIdx.x * 64 + threadIdx.x] = rb_wave_sum(v);
  out_max[blockIdx.x * 64 + threadIdx.x] = rb_wave_max(v);
}
int main() {
  const int NB = 8;
  float h[NB * 64], hs[NB * 64], hm[NB * 64];
  for (int i = 0; i < NB * 64; ++i) h[i] = (float)((i * 7919) % 101) - 50.0f + 0.25f * (i % 3);
  float *d, *ds, *dm;
  hipMalloc(&d, sizeof(h)); hipMalloc(&ds, sizeof(h)); hipMalloc(&dm, sizeof(h));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(NB), dim3(64), 0, 0, d, ds, dm);
  hipMemcpy(hs, ds, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(hm, dm, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int b = 0; b < NB; ++b) {
    double s = 0; float m = -INFINITY;
    for (int i = 0; i < 64; ++i) { s += h[b * 64 + i]; m = fmaxf(m, h[b * 64 + i]); }
    for (int i = 0; i < 64; ++i) if (fabs(hs[b * 64 + i] - s) > 1e-3 || hm[b * 64 + i] != m) { if (bad < 5) printf("block %d lane %d: sum %f (want %f) max %f (want %f)\n", b, i, hs[b*64+i], s, hm[b*64+i], m); ++bad; }
  }
  printf("dpp reductions: %d mismatches\n", bad);
  return bad != 0;
}
