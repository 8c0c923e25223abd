// What does one dependent launch cost on this box?  Chains of N trivial kernels, wall time per kernel, under the
// conditions the learn step launches in: null stream vs a created stream, grid size, workgroup size, static LDS,
// kernel-argument size, pointer arguments that the runtime resolves to memory objects.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/launch_floor.hip -o tools/micro/launch_floor && tools/micro/launch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

struct Args8 { float* p; };
struct Args600 { float* p; char pad[592]; };

template <class A, int LDSW, int T>
__global__ __launch_bounds__(T) void k_nop(A a, int flag) {
  __shared__ float s[LDSW];
  if (LDSW > 1) { s[threadIdx.x % LDSW] = 1.0f; __syncthreads(); }
  if (flag == 12345) a.p[threadIdx.x] = LDSW > 1 ? s[(threadIdx.x + 1) % LDSW] : 1.0f;
}
// a kernel that touches memory: every workgroup writes 1 KB (dirty lines for the end-of-kernel release) and reads 1 KB
template <int T>
__global__ __launch_bounds__(T) void k_touch(float* a, const float* b) {
  const size_t i = (size_t)blockIdx.x * T + threadIdx.x;
  a[i] = b[i] + 1.0f;
}
// many pointer arguments (the runtime looks every one of them up)
__global__ void k_ptrs(float* a0, float* a1, float* a2, float* a3, float* a4, float* a5, float* a6, float* a7, float* a8, float* a9,
                       float* b0, float* b1, float* b2, float* b3, float* b4, float* b5, float* b6, float* b7, float* b8, float* b9, int flag) {
  if (flag == 12345) a0[0] = a1[0] + a2[0] + a3[0] + a4[0] + a5[0] + a6[0] + a7[0] + a8[0] + a9[0] + b0[0] + b1[0] + b2[0] + b3[0] + b4[0] + b5[0] + b6[0] + b7[0] + b8[0] + b9[0];
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class F>
static void chain(const char* name, hipStream_t s, F launch) {
  for (int i = 0; i < 300; ++i) launch();
  hipStreamSynchronize(s);
  const int N = 4000;
  const double t0 = now_us();
  for (int i = 0; i < N; ++i) launch();
  const double t1 = now_us();
  hipStreamSynchronize(s);
  const double t2 = now_us();
  printf("%-64s host %.2f us/launch   wall %.2f us/kernel\n", name, (t1 - t0) / N, (t2 - t0) / N);
}

int main() {
  float *a, *b;
  hipMalloc(&a, 64 << 20); hipMalloc(&b, 64 << 20);
  hipMemset(a, 0, 64 << 20); hipMemset(b, 0, 64 << 20);
  hipStream_t created, nonblocking;
  hipStreamCreate(&created);
  hipStreamCreateWithFlags(&nonblocking, hipStreamNonBlocking);
  struct { const char* name; hipStream_t s; } streams[3] = {{"null stream", nullptr}, {"created stream", created}, {"non-blocking stream", nonblocking}};
  for (auto& st : streams) {
    hipStream_t s = st.s;
    char nm[128];
    Args8 a8{a}; Args600 a600{}; a600.p = a;
    snprintf(nm, sizeof(nm), "%s: nop 1x64, 8 B args", st.name);
    chain(nm, s, [&] { hipLaunchKernelGGL((k_nop<Args8, 1, 64>), dim3(1), dim3(64), 0, s, a8, 0); });
    snprintf(nm, sizeof(nm), "%s: nop 256x256, 8 B args", st.name);
    chain(nm, s, [&] { hipLaunchKernelGGL((k_nop<Args8, 1, 256>), dim3(256), dim3(256), 0, s, a8, 0); });
    snprintf(nm, sizeof(nm), "%s: nop 256x512, 8 B args", st.name);
    chain(nm, s, [&] { hipLaunchKernelGGL((k_nop<Args8, 1, 512>), dim3(256), dim3(512), 0, s, a8, 0); });
    snprintf(nm, sizeof(nm), "%s: nop 256x512, 100 KB LDS", st.name);
    chain(nm, s, [&] { hipLaunchKernelGGL((k_nop<Args8, 25600, 512>), dim3(256), dim3(512), 0, s, a8, 0); });
    snprintf(nm, sizeof(nm), "%s: nop 480x512, 60 KB LDS", st.name);
    chain(nm, s, [&] { hipLaunchKernelGGL((k_nop<Args8, 15360, 512>), dim3(480), dim3(512), 0, s, a8, 0); });
    snprintf(nm, sizeof(nm), "%s: nop 256x256, 600 B args", st.name);
    chain(nm, s, [&] { hipLaunchKernelGGL((k_nop<Args600, 1, 256>), dim3(256), dim3(256), 0, s, a600, 0); });
    snprintf(nm, sizeof(nm), "%s: 20 pointer args, 1x64", st.name);
    chain(nm, s, [&] { hipLaunchKernelGGL(k_ptrs, dim3(1), dim3(64), 0, s, a, a + 64, a + 128, a + 192, a + 256, a + 320, a + 384, a + 448, a + 512, a + 576,
                                          b, b + 64, b + 128, b + 192, b + 256, b + 320, b + 384, b + 448, b + 512, b + 576, 0); });
    snprintf(nm, sizeof(nm), "%s: touch 256x256 (256 KB written)", st.name);
    chain(nm, s, [&] { hipLaunchKernelGGL(k_touch<256>, dim3(256), dim3(256), 0, s, a, b); });
    snprintf(nm, sizeof(nm), "%s: touch 4096x256 (4 MB written)", st.name);
    chain(nm, s, [&] { hipLaunchKernelGGL(k_touch<256>, dim3(4096), dim3(256), 0, s, a, b); });
  }
  // the same chain replayed from a hipGraph (created stream)
  {
    hipGraph_t g; hipGraphExec_t ge;
    Args8 a8{a};
    hipStreamBeginCapture(created, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < 14; ++i) hipLaunchKernelGGL((k_nop<Args8, 1, 256>), dim3(256), dim3(256), 0, created, a8, 0);
    hipStreamEndCapture(created, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 50; ++i) hipGraphLaunch(ge, created);
    hipStreamSynchronize(created);
    const int N = 500;
    const double t0 = now_us();
    for (int i = 0; i < N; ++i) hipGraphLaunch(ge, created);
    hipStreamSynchronize(created);
    const double t2 = now_us();
    printf("%-64s wall %.2f us/kernel (14 nop 256x256 per graph)\n", "hipGraph replay, created stream", (t2 - t0) / N / 14);
  }
  return 0;
}
