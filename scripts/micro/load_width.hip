// Micro-benchmark: streaming-read rate of a 2.4 GB array by load width - 4, 8 or 16 bytes per lane (256 B / 512 B / 1 KiB per wave
// instruction), nontemporal, the same bytes in flight per wave (KB) in every shape. What the address path sustains per width decides
// how the CSR streams (4-byte columns, 8-byte values) should be loaded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <typename T> __device__ __forceinline__ double fold(T v);
template <> __device__ __forceinline__ double fold<int>(int v) { return (double)v; }
template <> __device__ __forceinline__ double fold<double>(double v) { return v; }
template <> __device__ __forceinline__ double fold<d2>(d2 v) { return v.x + v.y; }

template <typename T, int U>
__global__ __launch_bounds__(256) void k_read(const T *__restrict__ a, long long nT, double *__restrict__ out)
{
  double s = 0.0;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nT; i += stride * U) {
    T v[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const long long j = i + u * stride; v[u] = j < nT ? __builtin_nontemporal_load(a + j) : T{}; }
#pragma unroll
    for (int u = 0; u < U; u++) s += fold<T>(v[u]);
  }
  if (s == 12345.678) out[0] = s;
}

int main()
{
  const long long bytes = 2400LL * 1000 * 1000;
  void *a; double *out;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&out, 8));
  CK(hipMemset(a, 0, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto launch, const char *name) {
    for (int w = 0; w < 2; w++) launch();
    CK(hipEventRecord(e0)); for (int r = 0; r < 5; r++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("%-52s %8.3f ms  %7.1f GB/s\n", name, ms, (double)bytes / ms / 1e6);
  };
  for (int g : {256, 512, 1024, 2048, 4096}) {
    char nm[128];
#define RUN(T, U, label) snprintf(nm, 128, "%s  %2d loads in flight  grid %5d", label, U, g); time([&] { hipLaunchKernelGGL((k_read<T, U>), dim3(g), dim3(256), 0, 0, (const T *)a, (long long)(bytes / sizeof(T)), out); }, nm);
    RUN(int, 16, " 4 B/lane (4 KB/wave)") RUN(int, 32, " 4 B/lane (8 KB/wave)")
    RUN(double, 8, " 8 B/lane (4 KB/wave)") RUN(double, 16, " 8 B/lane (8 KB/wave)") RUN(double, 32, " 8 B/lane (16 KB/wave)")
    RUN(d2, 4, "16 B/lane (4 KB/wave)") RUN(d2, 8, "16 B/lane (8 KB/wave)") RUN(d2, 16, "16 B/lane (16 KB/wave)")
  }
  return 0;
}
