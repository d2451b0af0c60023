// Micro-benchmark: streaming KT columns at once (the access shape of the Gram-Schmidt sweeps): every lane reads VL
// 16-byte pieces of each of KT columns per tile. Which (grid, VL, tile order) reaches the pure-read rate?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <int KT, int VL>
__global__ __launch_bounds__(256) void k_cols(const d2 *__restrict__ a, long long ld2, long long n2, double *__restrict__ out)
{
  double s = 0.0;
  const long long tile = 256LL * VL, ntiles = (n2 + tile - 1) / tile;
  for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    d2 v[KT][VL];
#pragma unroll
    for (int i = 0; i < KT; i++)
#pragma unroll
      for (int u = 0; u < VL; u++) { const long long j = t * tile + u * 256 + threadIdx.x; v[i][u] = j < n2 ? __builtin_nontemporal_load(a + i * ld2 + j) : d2{0.0, 0.0}; }
#pragma unroll
    for (int i = 0; i < KT; i++)
#pragma unroll
      for (int u = 0; u < VL; u++) s += v[i][u].x + v[i][u].y;
  }
  if (s == 12345.678) out[0] = s;
}

// the access shape of the MFMA panel kernels: a 128-row tile, wave w streams columns w, w+4, ... (1 KB per column per tile)
template <int KT>
__global__ __launch_bounds__(256) void k_cols_wave(const d2 *__restrict__ a, long long ld2, long long n2, double *__restrict__ out)
{
  double s = 0.0;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long long ntiles = (n2 + 63) / 64;
  for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    d2 v[KT / 4];
#pragma unroll
    for (int q = 0; q < KT / 4; q++) { const long long j = t * 64 + lane; v[q] = j < n2 ? __builtin_nontemporal_load(a + (long long)(w + 4 * q) * ld2 + j) : d2{0.0, 0.0}; }
#pragma unroll
    for (int q = 0; q < KT / 4; q++) s += v[q].x + v[q].y;
  }
  if (s == 12345.678) out[0] = s;
}

int main()
{
  const long long ncol = 10077696, ld2 = ncol / 2, KTMAX = 28;
  d2 *a; double *out;
  CK(hipMalloc(&a, ncol * 8 * KTMAX)); CK(hipMalloc(&out, 8)); CK(hipMemset(a, 0, ncol * 8 * KTMAX));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto launch, const char *name, double bytes) {
    for (int w = 0; w < 2; w++) launch();
    CK(hipEventRecord(e0)); for (int r = 0; r < 5; r++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("%-40s %8.3f ms  %7.1f GB/s\n", name, ms, bytes / ms / 1e6);
  };
  char nm[96];
#define RUN(KT, VL, g) snprintf(nm, 96, "KT=%2d VL=%d grid %5d", KT, VL, g); time([&] { hipLaunchKernelGGL((k_cols<KT, VL>), dim3(g), dim3(256), 0, 0, a, ld2, ld2, out); }, nm, ncol * 8.0 * KT)
  for (int g : {256, 512, 768, 1024, 2048}) { RUN(28, 1, g); RUN(28, 2, g); RUN(14, 2, g); RUN(14, 4, g); RUN(7, 4, g); RUN(7, 8, g); RUN(4, 8, g); RUN(1, 8, g); }
#define RUNW(KT, g) snprintf(nm, 96, "wave-columns KT=%2d grid %5d", KT, g); time([&] { hipLaunchKernelGGL((k_cols_wave<KT>), dim3(g), dim3(256), 0, 0, a, ld2, ld2, out); }, nm, ncol * 8.0 * KT)
  for (int g : {512, 1024, 2048, 4096}) { RUNW(28, g); RUNW(16, g); RUNW(8, g); }
  return 0;
}
