// Cold instruction-cache cost of a launch on MI355X: kernels whose body is N KB of straight-line SALU code (1 wave per CU),
// launched back to back (same kernel: warm after the first launch) against a rotation of distinct kernels whose combined
// code exceeds the 64 KB instruction cache (every launch starts cold).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/micro/icache_cold.hip -o /tmp/icache_cold && /tmp/icache_cold
#include <hip/hip_runtime.h>
#include <stdio.h>
#define BODY(N) asm volatile(".rept " #N "\n s_add_u32 s20, s20, 1\n .endr" ::: "s20", "scc")
#define KERN(name, N) __global__ void name(int* out) { BODY(N); if (out && threadIdx.x == 1234567) *out = 1; }
KERN(k8a, 2048)  KERN(k8b, 2048)  KERN(k8c, 2048)  KERN(k8d, 2048) KERN(k8e, 2048) KERN(k8f, 2048) KERN(k8g, 2048) KERN(k8h, 2048)
KERN(k8i, 2048)  KERN(k8j, 2048)  KERN(k8k, 2048)  KERN(k8l, 2048)
KERN(k32a, 8192) KERN(k32b, 8192) KERN(k32c, 8192) KERN(k32d, 8192)
KERN(k2a, 512)
typedef void (*kfn)(int*);
static double run(kfn* ks, int nk, int reps, int grid) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(ks[i % nk], dim3(grid), dim3(64), 0, 0, nullptr);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(ks[i % nk], dim3(grid), dim3(64), 0, 0, nullptr);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / reps;
}
int main() {
  kfn same8[1] = {k8a}, rot8[12] = {k8a, k8b, k8c, k8d, k8e, k8f, k8g, k8h, k8i, k8j, k8k, k8l};
  kfn same32[1] = {k32a}, rot32[4] = {k32a, k32b, k32c, k32d}, small[1] = {k2a};
  for (int grid : {256, 1024}) {
    printf("grid %d x 64 threads\n", grid);
    printf("  2 KB body, same kernel            %.2f us/launch\n", run(small, 1, 2000, grid));
    printf("  8 KB body, same kernel (warm)     %.2f us/launch\n", run(same8, 1, 2000, grid));
    printf("  8 KB body, 12 kernels in rotation %.2f us/launch   (96 KB working set: cold)\n", run(rot8, 12, 2000, grid));
    printf("  32 KB body, same kernel (warm)    %.2f us/launch\n", run(same32, 1, 2000, grid));
    printf("  32 KB body, 4 kernels in rotation %.2f us/launch   (128 KB working set: cold)\n", run(rot32, 4, 2000, grid));
  }
  return 0;
}
