// Fourth micro-benchmark of the final Gram-Schmidt update: separate the reads from the stores in time. A workgroup owns G consecutive
// rounds of tiles, keeps the G results of each thread in registers (occupancy is one wave per SIMD: 512 VGPRs per thread) and stores them
// after its last load. G = 81 with a grid of 243 workgroups: the whole kernel reads first and stores last; G = 27, 9, 3: shorter phases.
// build: hipcc -O3 --offload-arch=gfx950 update_write4.hip -o update_write4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int KT = 30;

template <int G, bool STORE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_hold(const d2 *__restrict__ V, long long ld2, long long n2, d2 *v, const double *__restrict__ c, double *out, int rounds)
{
  double cc[KT];
#pragma unroll
  for (int i = 0; i < KT; i++) cc[i] = c[i];
  double sink = 0.0;
  for (int r0 = 0; r0 < rounds; r0 += G) {        // rounds is a multiple of G; tile of round r: r * gridDim.x + blockIdx.x
    d2 res[G];
#pragma unroll
    for (int u = 0; u < G; u++) {
      const long long j = ((long long)(r0 + u) * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
      d2 s = v[j];
      d2 x[KT];
#pragma unroll
      for (int i = 0; i < KT; i++) x[i] = __builtin_nontemporal_load(V + i * ld2 + j);
#pragma unroll
      for (int p = 0; p < 2; p++)
#pragma unroll
        for (int i = 0; i < KT; i++) { s.x = fma(cc[i], x[i].x, s.x); s.y = fma(cc[i], x[i].y, s.y); }
      res[u] = s;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < G; u++) {
      const long long j = ((long long)(r0 + u) * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
      if (STORE) v[j] = res[u]; else sink += res[u].x + res[u].y;
    }
  }
  if (sink == 12345.678) out[0] = sink;
}


// 81 rounds held as three register arrays of 27 (one array of 81 stays in scratch memory: the compiler does not promote it)
template <bool STORE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_hold81(const d2 *__restrict__ V, long long ld2, long long n2, d2 *v, const double *__restrict__ c, double *out)
{
  double cc[KT];
#pragma unroll
  for (int i = 0; i < KT; i++) cc[i] = c[i];
  double sink = 0.0;
  d2 ra[27], rb[27], rc[27];
#define PHASE(arr, base)                                                                                                   \
  _Pragma("unroll") for (int u = 0; u < 27; u++) {                                                                         \
    const long long jb = ((long long)((base) + u) * gridDim.x + blockIdx.x) * 256;    /* uniform: scalar base, one vector offset */ \
    d2 s = (v + jb)[threadIdx.x];                                                                                          \
    d2 x[KT];                                                                                                              \
    _Pragma("unroll") for (int i = 0; i < KT; i++) x[i] = __builtin_nontemporal_load(V + i * ld2 + jb + threadIdx.x);       \
    _Pragma("unroll") for (int p = 0; p < 2; p++)                                                                          \
      _Pragma("unroll") for (int i = 0; i < KT; i++) { s.x = fma(cc[i], x[i].x, s.x); s.y = fma(cc[i], x[i].y, s.y); }       \
    arr[u] = s;                                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                                     \
  }
  PHASE(ra, 0) PHASE(rb, 27) PHASE(rc, 54)
  __builtin_amdgcn_sched_barrier(0);
#define FLUSH(arr, base)                                                                                                   \
  _Pragma("unroll") for (int u = 0; u < 27; u++) {                                                                         \
    const long long jb = ((long long)((base) + u) * gridDim.x + blockIdx.x) * 256;                                         \
    if (STORE) (v + jb)[threadIdx.x] = arr[u]; else sink += arr[u].x + arr[u].y;                                           \
  }
  FLUSH(ra, 0) FLUSH(rb, 27) FLUSH(rc, 54)
  if (sink == 12345.678) out[0] = sink;
}


// results of G rounds parked in LDS (one workgroup per CU: the 160 KB are free), no unrolling needed; every thread reads back its own values
template <int G>
__global__ __launch_bounds__(256) void k_lds(const d2 *__restrict__ V, long long ld2, long long n2, d2 *v, const double *__restrict__ c, double *out, int rounds)
{
  extern __shared__ d2 park[];                   // [G][256]
  double cc[KT];
#pragma unroll
  for (int i = 0; i < KT; i++) cc[i] = c[i];
  int held = 0, first = 0;
  for (int r = 0; r < rounds; r++) {
    const long long j = ((long long)r * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
    d2 s = v[j];
    d2 x[KT];
#pragma unroll
    for (int i = 0; i < KT; i++) x[i] = __builtin_nontemporal_load(V + i * ld2 + j);
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
      for (int i = 0; i < KT; i++) { s.x = fma(cc[i], x[i].x, s.x); s.y = fma(cc[i], x[i].y, s.y); }
    park[held * 256 + threadIdx.x] = s;
    if (++held == G || r == rounds - 1) {
      for (int u = 0; u < held; u++) v[((long long)(first + u) * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = park[u * 256 + threadIdx.x];
      first = r + 1; held = 0;
    }
  }
}

int main()
{
  const long long n = 10077696, ld = n;            // 19683 tiles of 512 rows = 243 workgroups x 81 rounds
  d2 *V, *v; double *c, *out;
  CK(hipMalloc(&V, ld * 8 * KT)); CK(hipMemset(V, 0, ld * 8 * KT));
  CK(hipMalloc(&v, n * 8)); CK(hipMemset(v, 0, n * 8));
  CK(hipMalloc(&c, 8 * KT)); CK(hipMemset(c, 0, 8 * KT)); CK(hipMalloc(&out, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto launch, const char *name) {
    for (int r = 0; r < 3; r++) launch();
    CK(hipEventRecord(e0)); for (int r = 0; r < 10; r++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("%-72s %8.1f us   reads %7.1f GB/s\n", name, ms * 1e3, n * 8.0 * (KT + 1) / ms / 1e6);
  };
  char nm[160];
#define RUN(G, ST) do { snprintf(nm, 160, "grid 243, results of %2d rounds held, then %s", G, ST ? "stored" : "summed (no store)"); \
    time([&] { hipLaunchKernelGGL((k_hold<G, ST>), dim3(243), dim3(256), 0, 0, V, ld / 2, n / 2, v, c, out, 81); }, nm); } while (0)
  RUN(1, false); RUN(1, true); RUN(3, true); RUN(9, true); RUN(27, true);
  if (getenv("HOLD81")) time([&] { hipLaunchKernelGGL((k_hold81<true>), dim3(243), dim3(256), 0, 0, V, ld / 2, n / 2, v, c, out); }, "grid 243, results of all 81 rounds held in registers, then stored");
  if (getenv("HOLD81")) time([&] { hipLaunchKernelGGL((k_hold81<false>), dim3(243), dim3(256), 0, 0, V, ld / 2, n / 2, v, c, out); }, "grid 243, results of all 81 rounds held in registers, then summed");
  RUN(1, true);
#define RUNL(G) do { snprintf(nm, 160, "grid 243, results of %2d rounds parked in LDS, then stored", G); \
    CK(hipFuncSetAttribute((const void *)k_lds<G>, hipFuncAttributeMaxDynamicSharedMemorySize, G * 4096)); \
    time([&] { hipLaunchKernelGGL((k_lds<G>), dim3(243), dim3(256), G * 4096, 0, V, ld / 2, n / 2, v, c, out, 81); }, nm); } while (0)
  RUNL(1); RUNL(9); RUNL(27); RUNL(39);
  RUN(1, false);
  return 0;
}
