// residency.hip — how many workgroups of a given size / LDS footprint does one CU hold at once?  Every workgroup spins ~20 us and
// records (start, end, XCC id, HW id); the host counts, per CU, the maximum number of workgroups whose intervals overlap.
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/residency.hip -o tools/micro/residency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <map>
#include <vector>
struct Rec { long long t0, t1; int xcc, hw; };
template <int LDSB, int SG = 0>
__global__ void k_spin(Rec* out, long long ticks) {
  __shared__ char lds[LDSB];
  if (SG == 1) asm volatile("s_mov_b32 s70, 0" ::: "s70");      // raises the kernel's SGPR count past 70 (+ VCC etc.)
  if (SG == 2) asm volatile("s_mov_b32 s90, 0" ::: "s90");
  if (SG == 3) asm volatile("v_mov_b32 v40, 0" ::: "v40");      // 41+ VGPRs
  if (SG == 4) asm volatile("v_mov_b32 v80, 0" ::: "v80");      // 81+ VGPRs
  lds[threadIdx.x] = (char)threadIdx.x;
  if (SG == 5) {                                                 // fill the whole allocation with 16-byte stores first
    float4* l4 = reinterpret_cast<float4*>(lds);
    for (int i = threadIdx.x; i < LDSB / 16; i += blockDim.x) l4[i] = make_float4((float)i, 1.0f, 2.0f, 3.0f);
  }
  if (SG == 6) {                                                 // ... with 4-byte stores
    float* l1 = reinterpret_cast<float*>(lds);
    for (int i = threadIdx.x; i < LDSB / 4; i += blockDim.x) l1[i] = (float)i;
  }
  if (SG == 7) {                                                 // a kernel that uses the matrix cores
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32((float)threadIdx.x, 1.0f, acc, 0, 0, 0);
    if (acc[0] == 123.0f) lds[1] = 1;
  }
  __syncthreads();
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) {
    Rec r; r.t0 = t0; r.t1 = wall_clock64();
    r.xcc = (int)__builtin_amdgcn_s_getreg(6164); r.hw = (int)(__builtin_amdgcn_s_getreg(63492) & 0xffff);
    out[blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)] = r;
  }
  if (lds[0] == 77) out[0].hw = 0;
}
template <int LDSB, int SG = 0>
static void run(int threads, int nblocks, long long ticks = 2000LL, dim3 grid3 = dim3(0)) {
  Rec* d; hipMalloc(&d, sizeof(Rec) * nblocks);
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k_spin<LDSB, SG>), grid3.x ? grid3 : dim3(nblocks), dim3(threads), 0, 0, d, ticks); hipDeviceSynchronize(); }
  std::vector<Rec> h(nblocks); hipMemcpy(h.data(), d, sizeof(Rec) * nblocks, hipMemcpyDeviceToHost);
  std::map<int, std::vector<Rec>> per;
  for (auto& r : h) per[r.xcc * 65536 + ((r.hw >> 8) & 0xff) + (((r.hw >> 13) & 7) << 8)].push_back(r);   // (xcc, cu id, se id)
  int worst = 0; long long tmin = h[0].t0, tmax = h[0].t1;
  for (auto& kv : per) {
    int best = 0;
    for (auto& a : kv.second) { int c = 0; for (auto& b : kv.second) if (b.t0 <= a.t0 && b.t1 > a.t0) ++c; best = std::max(best, c); }
    worst = std::max(worst, best);
  }
  for (auto& r : h) { tmin = std::min(tmin, r.t0); tmax = std::max(tmax, r.t1); }
  int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)(k_spin<LDSB, SG>), threads, 0);
  hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void*)k_spin<LDSB, SG>);
  printf("SG %d threads %4d  LDS %6d B  blocks %4d: distinct CUs %3zu, max co-resident per CU %d, API says %d, span %.1f us\n", SG, threads, LDSB, nblocks, per.size(), worst, occ, (tmax - tmin) * 0.01);
  hipFree(d);
}
int main() {
  for (int th : {256, 512, 576, 640, 704, 768, 1024}) run<60160>(th, 512);
  for (int th : {512, 640}) run<30000>(th, 768);
  for (int th : {256, 512, 640}) run<60160, 1>(th, 512);
  for (int th : {256, 512, 640}) run<60160, 2>(th, 512);
  printf("LDS filled before the spin (SG 5: 16-byte stores, 6: 4-byte stores), 5 us, 480 blocks:\n");
  for (int th : {512, 640}) run<60160, 5>(th, 480, 500LL);
  for (int th : {512, 640}) run<60160, 6>(th, 480, 500LL);
  printf("with an MFMA in the kernel (SG 7), 5 us, 480 blocks:\n");
  for (int th : {512, 640}) run<60160, 7>(th, 480, 500LL);
  printf("short kernels (5 us), 480 blocks:\n");
  for (int th : {512, 640}) run<60160, 3>(th, 480, 500LL);
  for (int th : {512, 640}) run<60160, 3>(th, 480, 500LL, dim3(96, 1, 5));
  for (int th : {512, 640}) run<60160, 3>(th, 480);
  for (int th : {512, 640}) run<60160, 4>(th, 480);
  return 0;
}
