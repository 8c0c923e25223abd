// common.hip — error string + ABI version for librainbow_hip.so.
#include "rb_common.h"

#include <string.h>

static thread_local char g_rb_error[1024] = "";

void rb_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_rb_error, sizeof(g_rb_error), fmt, ap);
  va_end(ap);
}

// ---- live kernel timing (rb_profile_select / rb_profile_read) ---------------------------
#if !defined(RB_HOST_INTERP)
#include <vector>
int g_rb_prof_on = 0;
static char g_prof_pat[128] = "";
static std::vector<hipEvent_t> g_prof_events;   // pairs: [2i] before, [2i+1] after
static size_t g_prof_used = 0;
static int g_prof_stride = 1, g_prof_seen = 0;

bool rb_prof_begin(const char* kernel_expr, hipStream_t stream) {
  if (!strstr(kernel_expr, g_prof_pat)) return false;
  if (g_prof_stride > 1 && (g_prof_seen++ % g_prof_stride) != 0) return false;   // sample every stride-th matching launch
  if (g_prof_used + 2 > g_prof_events.size()) {
    if (g_prof_events.size() >= 65536) return false;
    for (int i = 0; i < 256; ++i) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return false;
      g_prof_events.push_back(e);
    }
  }
  return hipEventRecord(g_prof_events[g_prof_used++], stream) == hipSuccess;
}
void rb_prof_end(hipStream_t stream) { (void)hipEventRecord(g_prof_events[g_prof_used++], stream); }
#endif

#if defined(RB_HOST_TIMING)
#include <chrono>
#include <map>
#include <string>
struct HostT { double in_launch = 0, before = 0; long n = 0; };
static std::map<std::string, HostT> g_host_time;
static double g_host_last_end = 0;
double rb_host_now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
void rb_host_time_add(const char* tag, double us, double t0) {
  auto& e = g_host_time[tag];
  e.in_launch += us; e.n += 1;
  if (g_host_last_end > 0) e.before += t0 - g_host_last_end;     // host time since the previous launch returned
  g_host_last_end = t0 + us;
}
__global__ void k_debug_spin(long long cycles, float* out) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (out && threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1.0f;
}
// N launches of a ~us-long kernel on `stream`: host time per launch and wall time per kernel (is the host throttled?)
extern "C" int rb_debug_spin_launch(void* stream, int n, int us) {
  hipStream_t s = (hipStream_t)stream;
  for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_debug_spin, dim3(256), dim3(256), 0, s, (long long)us * 100, (float*)nullptr);
  hipStreamSynchronize(s);
  const double t0 = rb_host_now_us();
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_debug_spin, dim3(256), dim3(256), 0, s, (long long)us * 100, (float*)nullptr);
  const double t1 = rb_host_now_us();
  hipStreamSynchronize(s);
  const double t2 = rb_host_now_us();
  printf("spin %d us on stream %p: host %.2f us/launch, wall %.2f us/kernel\n", us, stream, (t1 - t0) / n, (t2 - t0) / n);
  return 0;
}
extern "C" int rb_debug_host_timing(int reset) {
  if (reset) { g_host_time.clear(); g_host_last_end = 0; return 0; }
  for (auto& kv : g_host_time)
    printf("  host %-34s n %6ld  in launch %7.2f us   since previous launch returned %7.2f us\n", kv.first.c_str(), kv.second.n,
           kv.second.in_launch / kv.second.n, kv.second.before / kv.second.n);
  return 0;
}
#endif

// ---- guarded device allocations (rb_common.h) -----------------------------------------
#include <stdlib.h>
#include <vector>
#define RB_GUARD_BYTES 4096
#define RB_GUARD_FILL 0xC5
struct GuardedBlock { char* raw; char* user; size_t bytes; };
static std::vector<GuardedBlock> g_guarded;
static int guard_on() {
  static const int on = getenv("RB_GUARD") && getenv("RB_GUARD")[0] == '1';
  return on;
}
hipError_t rb_dev_malloc(void** p, size_t bytes) {
  if (!guard_on()) return hipMalloc(p, bytes);
  const size_t padded = (bytes + 255) / 256 * 256;             // keep the user block's end 256-byte aligned like its start
  char* raw = nullptr;
  hipError_t e = hipMalloc((void**)&raw, padded + 2 * RB_GUARD_BYTES);
  if (e != hipSuccess) return e;
  e = hipMemset(raw, RB_GUARD_FILL, padded + 2 * RB_GUARD_BYTES);   // guards AND the alignment tail [bytes, padded)
  if (e != hipSuccess) { (void)hipFree(raw); return e; }
  if (bytes) {
    e = hipMemset(raw + RB_GUARD_BYTES, 0, bytes);
    if (e != hipSuccess) { (void)hipFree(raw); return e; }
  }
  g_guarded.push_back(GuardedBlock{raw, raw + RB_GUARD_BYTES, bytes});
  *p = raw + RB_GUARD_BYTES;
  return hipSuccess;
}
void rb_dev_free(void* p) {
  if (!p) return;
  for (size_t i = 0; i < g_guarded.size(); ++i)
    if (g_guarded[i].user == (char*)p) {
      (void)hipFree(g_guarded[i].raw);
      g_guarded.erase(g_guarded.begin() + (long)i);
      return;
    }
  (void)hipFree(p);
}

int rb_opt(const char* key, int dflt) {
  const char* s = getenv("RB_OPTS");
  if (!s || !key) return dflt;
  const size_t kl = strlen(key);
  while (*s) {
    const char* e = strchr(s, ',');
    const size_t n = e ? (size_t)(e - s) : strlen(s);
    if (n > kl + 1 && strncmp(s, key, kl) == 0 && s[kl] == '=') return atoi(s + kl + 1);
    if (!e) break;
    s = e + 1;
  }
  return dflt;
}

void rb_replay_note_device_write(void* dst_dev, const void* src_host, size_t nbytes);   // replay.hip

extern "C" {
int rb_profile_select(const char* kernel_substr) {
#if !defined(RB_HOST_INTERP)
  g_prof_used = 0;
  if (!kernel_substr || !kernel_substr[0]) { g_rb_prof_on = 0; g_prof_pat[0] = 0; return RB_OK; }
  snprintf(g_prof_pat, sizeof(g_prof_pat), "%s", kernel_substr);
  g_rb_prof_on = 1;
#else
  (void)kernel_substr;
#endif
  return RB_OK;
}

int rb_profile_stride(int32_t every) {
#if !defined(RB_HOST_INTERP)
  g_prof_stride = every > 1 ? every : 1;
  g_prof_seen = 0;
#else
  (void)every;
#endif
  return RB_OK;
}

int rb_profile_read(double* total_ms, int64_t* launches) {
  double total = 0.0;
  int64_t n = 0;
#if !defined(RB_HOST_INTERP)
  for (size_t i = 0; i + 1 < g_prof_used; i += 2) {
    RB_HIP_TRY(hipEventSynchronize(g_prof_events[i + 1]));
    float ms = 0.0f;
    RB_HIP_TRY(hipEventElapsedTime(&ms, g_prof_events[i], g_prof_events[i + 1]));
    total += ms;
    ++n;
  }
  g_prof_used = 0;
#endif
  if (total_ms) *total_ms = total;
  if (launches) *launches = n;
  return RB_OK;
}

int rb_profile_overhead(rb_stream_t stream, int32_t n, double* mean_ms) {
  RB_REQUIRE(mean_ms != nullptr && n >= 1 && n <= 4096, "rb_profile_overhead: bad argument");
  *mean_ms = 0.0;
#if !defined(RB_HOST_INTERP)
  std::vector<hipEvent_t> ev((size_t)2 * n);
  for (auto& e : ev) RB_HIP_TRY(hipEventCreate(&e));
  for (int i = 0; i < n; ++i) {
    RB_HIP_TRY(hipEventRecord(ev[2 * i], (hipStream_t)stream));
    RB_HIP_TRY(hipEventRecord(ev[2 * i + 1], (hipStream_t)stream));
  }
  RB_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  double total = 0.0;
  for (int i = 0; i < n; ++i) {
    float ms = 0.0f;
    RB_HIP_TRY(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
    total += ms;
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  *mean_ms = total / n;
#else
  (void)stream;
#endif
  return RB_OK;
}

int rb_debug_check_guards(int64_t* n_blocks, int64_t* n_bad) {
  RB_REQUIRE(n_blocks && n_bad, "rb_debug_check_guards: NULL argument");
  *n_blocks = (int64_t)g_guarded.size();
  *n_bad = 0;
  if (!guard_on()) return RB_OK;
  RB_HIP_TRY(hipDeviceSynchronize());
  std::vector<unsigned char> lo(RB_GUARD_BYTES), hi;
  for (const GuardedBlock& b : g_guarded) {
    const size_t padded = (b.bytes + 255) / 256 * 256;
    hi.resize(padded - b.bytes + RB_GUARD_BYTES);
    RB_HIP_TRY(hipMemcpy(lo.data(), b.raw, RB_GUARD_BYTES, hipMemcpyDeviceToHost));
    RB_HIP_TRY(hipMemcpy(hi.data(), b.user + b.bytes, hi.size(), hipMemcpyDeviceToHost));
    bool bad = false;
    for (unsigned char c : lo) bad |= c != RB_GUARD_FILL;
    for (unsigned char c : hi) bad |= c != RB_GUARD_FILL;
    if (bad) {
      if (*n_bad == 0) rb_set_error("rb_debug_check_guards: a guard band of the %lld-byte block at %p was overwritten", (long long)b.bytes, (void*)b.user);
      *n_bad += 1;
    }
  }
  return RB_OK;
}

#ifndef RB_SOURCE_HASH
#define RB_SOURCE_HASH ""
#endif
// (the marker lets build() read the hash of a library on disk without loading it)
__attribute__((used)) static const char g_rb_hash_marker[] = "@@RB_SOURCE_HASH=" RB_SOURCE_HASH "@@";
const char* rb_source_hash(void) { return RB_SOURCE_HASH; }
const char* rb_last_error(void) { return g_rb_error; }
int rb_abi_version(void) { return 1; }

int rb_copy_to_host(void* dst_host, const void* src_dev, size_t nbytes, rb_stream_t stream) {
  RB_REQUIRE(dst_host && src_dev, "rb_copy_to_host: NULL argument");
  RB_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  RB_HIP_TRY(hipMemcpy(dst_host, src_dev, nbytes, hipMemcpyDeviceToHost));
  return RB_OK;
}

int rb_copy_to_device(void* dst_dev, const void* src_host, size_t nbytes, rb_stream_t stream) {
  RB_REQUIRE(dst_dev && src_host, "rb_copy_to_device: NULL argument");
  RB_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  RB_HIP_TRY(hipMemcpy(dst_dev, src_host, nbytes, hipMemcpyHostToDevice));
  rb_replay_note_device_write(dst_dev, src_host, nbytes);   // a restored replay header also refreshes the host mirror
  return RB_OK;
}
}
