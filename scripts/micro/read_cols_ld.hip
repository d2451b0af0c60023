// Micro-benchmark: does the distance between the columns of the basis (the leading dimension) matter for a sweep that streams 28 columns at once?
// ld = n + pad doubles for several pads. build: hipcc -O3 --offload-arch=gfx950 read_cols_ld.hip -o read_cols_ld
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
template <int KT>
__global__ __launch_bounds__(256) void k_cols(const d2 *__restrict__ a, long long ld2, long long n2, double *__restrict__ out)
{
  double s = 0.0;
  const long long ntiles = (n2 + 255) / 256;
  for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    d2 v[KT];
    const long long j = t * 256 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < KT; i++) v[i] = j < n2 ? __builtin_nontemporal_load(a + i * ld2 + j) : d2{0.0, 0.0};
#pragma unroll
    for (int i = 0; i < KT; i++) s += v[i].x + v[i].y;
  }
  if (s == 12345.678) out[0] = s;
}
int main()
{
  const long long n = 10077696, KT = 28;
  const long long pads[] = {0, 32, 64, 96, 160, 288, 544, 1056, 2080, 4128, 8224, 16416, 65568, 1048608};
  double *out; CK(hipMalloc(&out, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (long long pad : pads) {
    const long long ld = n + pad;
    d2 *a; CK(hipMalloc(&a, ld * 8 * KT)); CK(hipMemset(a, 0, ld * 8 * KT));
    for (int g : {256, 512}) {
      for (int w = 0; w < 2; w++) hipLaunchKernelGGL((k_cols<28>), dim3(g), dim3(256), 0, 0, a, ld / 2, n / 2, out);
      CK(hipEventRecord(e0)); for (int r = 0; r < 5; r++) hipLaunchKernelGGL((k_cols<28>), dim3(g), dim3(256), 0, 0, a, ld / 2, n / 2, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
      printf("pad %8lld doubles (column distance %% 4096 B = %5lld, %% 1 MiB = %8lld) grid %4d: %7.3f ms  %7.1f GB/s\n", pad, (ld * 8) % 4096, (ld * 8) % 1048576, g, ms, n * 8.0 * KT / ms / 1e6);
    }
    CK(hipFree(a));
  }
  return 0;
}
