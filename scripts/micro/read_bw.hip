// Micro-benchmark: pure-read streaming rate (sum of a 2.4 GB array) for several loads-in-flight / grid shapes, and the
// copy rate for comparison. Prints GB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_read(const d2 *__restrict__ a, long long n2, double *__restrict__ out)
{
  double s = 0.0;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride * U) {
    d2 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const long long j = i + u * stride; v[u] = j < n2 ? (NT ? __builtin_nontemporal_load(a + j) : a[j]) : d2{0.0, 0.0}; }
#pragma unroll
    for (int u = 0; u < U; u++) s += v[u].x + v[u].y;
  }
  if (s == 12345.678) out[0] = s;
}
__global__ __launch_bounds__(256) void k_copy(const d2 *__restrict__ a, d2 *__restrict__ b, long long n2)
{
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long long)gridDim.x * 256) b[i] = a[i];
}

int main()
{
  const long long n = 300LL * 1000 * 1000, n2 = n / 2;
  d2 *a, *b; double *out;
  CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&out, 8));
  CK(hipMemset(a, 0, n * 8)); CK(hipMemset(b, 0, n * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto launch, const char *name, double bytes) {
    for (int w = 0; w < 2; w++) launch();
    CK(hipEventRecord(e0)); for (int r = 0; r < 5; r++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms, bytes / ms / 1e6);
  };
  for (int g : {256, 512, 1024, 2048, 4096, 16384}) {
    char nm[96];
    snprintf(nm, 96, "read  U=4 nt   grid %5d", g); time([&] { hipLaunchKernelGGL((k_read<4, true>), dim3(g), dim3(256), 0, 0, a, n2, out); }, nm, n * 8.0);
    snprintf(nm, 96, "read  U=8 nt   grid %5d", g); time([&] { hipLaunchKernelGGL((k_read<8, true>), dim3(g), dim3(256), 0, 0, a, n2, out); }, nm, n * 8.0);
    snprintf(nm, 96, "read  U=8      grid %5d", g); time([&] { hipLaunchKernelGGL((k_read<8, false>), dim3(g), dim3(256), 0, 0, a, n2, out); }, nm, n * 8.0);
    snprintf(nm, 96, "read  U=16 nt  grid %5d", g); time([&] { hipLaunchKernelGGL((k_read<16, true>), dim3(g), dim3(256), 0, 0, a, n2, out); }, nm, n * 8.0);
  }
  for (int g : {1024, 4096, 16384}) { char nm[96]; snprintf(nm, 96, "copy           grid %5d (read+write bytes)", g); time([&] { hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, a, b, n2); }, nm, 2 * n * 8.0); }
  time([&] { CK(hipMemcpyAsync(b, a, n * 8, hipMemcpyDeviceToDevice, 0)); }, "hipMemcpy D2D (read+write bytes)", 2 * n * 8.0);
  return 0;
}
