// How far may the host run ahead of the GPU?  Launches N kernels of ~T us each and reports the host time per launch
// (steady state) and the wall time per kernel, for a small and a large kernel-argument block.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/launch_backpressure.hip -o gpurun_out/launch_bp && gpurun_out/launch_bp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
struct Small { long long cycles; float* out; };
struct Big { long long cycles; float* out; char pad[600]; };
template <class A>
__global__ void k_spin(A a) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < a.cycles) {}
  if (a.out && threadIdx.x == 0 && blockIdx.x == 0) a.out[0] = 1.0f;
}
template <class A>
static void run(const char* name, int us, int lds_bytes) {
  float* out; hipMalloc(&out, 4);
  A a{}; a.cycles = (long long)us * 100; a.out = out;     // wall_clock64: 100 MHz
  hipStream_t s; hipStreamCreate(&s);
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_spin<A>, dim3(256), dim3(256), lds_bytes, s, a);
  hipStreamSynchronize(s);
  const int N = 3000;
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_spin<A>, dim3(256), dim3(256), lds_bytes, s, a);
  auto t1 = std::chrono::steady_clock::now();
  hipStreamSynchronize(s);
  auto t2 = std::chrono::steady_clock::now();
  printf("%-28s kernel %3d us: host %.2f us/launch, wall %.2f us/kernel\n", name, us,
         std::chrono::duration<double, std::micro>(t1 - t0).count() / N, std::chrono::duration<double, std::micro>(t2 - t0).count() / N);
  hipStreamDestroy(s); hipFree(out);
}
__global__ void k_copy(const float4* a, float4* b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
static void run_copy(size_t mb) {
  const size_t n = mb * (1 << 20) / 16;
  float4 *a, *b; hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMemset(a, 0, n * 16);
  hipStream_t s; hipStreamCreate(&s);
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, s, a, b, n);
  hipStreamSynchronize(s);
  const int N = 2000;
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, s, a, b, n);
  auto t1 = std::chrono::steady_clock::now();
  hipStreamSynchronize(s);
  auto t2 = std::chrono::steady_clock::now();
  printf("copy %3zu MB (HBM-bound):      host %.2f us/launch, wall %.2f us/kernel\n", mb,
         std::chrono::duration<double, std::micro>(t1 - t0).count() / N, std::chrono::duration<double, std::micro>(t2 - t0).count() / N);
  hipStreamDestroy(s); hipFree(a); hipFree(b);
}
__global__ __launch_bounds__(512) void k_spin_lds(long long cycles, float* out) {       // big static LDS, 512 threads
  __shared__ float s[36000];
  s[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (out && threadIdx.x == 0 && blockIdx.x == 0) out[0] = s[1];
}
static void run_alternating(int lds_kernel) {
  float* out; hipMalloc(&out, 4);
  Small a{}; a.cycles = 1400; a.out = out;
  Big b{}; b.cycles = 1400; b.out = out;
  hipStream_t s; hipStreamCreate(&s);
  auto body = [&](int i) {
    if (i & 1) hipLaunchKernelGGL(k_spin<Small>, dim3(256), dim3(256), 0, s, a);
    else if (lds_kernel) hipLaunchKernelGGL(k_spin_lds, dim3(256), dim3(512), 0, s, 1400LL, out);
    else hipLaunchKernelGGL(k_spin<Big>, dim3(256), dim3(256), 0, s, b);
  };
  for (int i = 0; i < 200; ++i) body(i);
  hipStreamSynchronize(s);
  const int N = 3000;
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < N; ++i) body(i);
  auto t1 = std::chrono::steady_clock::now();
  hipStreamSynchronize(s);
  auto t2 = std::chrono::steady_clock::now();
  printf("alternating kernels (%s): host %.2f us/launch, wall %.2f us/kernel\n", lds_kernel ? "small / 144 KB-LDS 512-thread" : "small / big args",
         std::chrono::duration<double, std::micro>(t1 - t0).count() / N, std::chrono::duration<double, std::micro>(t2 - t0).count() / N);
  hipStreamDestroy(s); hipFree(out);
}
int main() {
  run_alternating(0);
  run_alternating(1);
  run_copy(32);
  run_copy(4);
  run<Small>("16-byte args", 1, 0);
  run<Small>("16-byte args", 14, 0);
  run<Big>("616-byte args", 14, 0);
  run<Small>("16-byte args, 64 KB LDS", 14, 65536);
  return 0;
}
