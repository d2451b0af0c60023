// Micro-benchmark: pure write bandwidth (1.25 GiB written by a grid-stride kernel, 16 B per lane), plain and nontemporal stores, several grids.
// build: hipcc -O3 --offload-arch=gfx950 write_bw.hip -o write_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
template <bool NT>
__global__ __launch_bounds__(256) void k_write(d2 *__restrict__ p, long long n2, double v)
{
  const d2 val = {v, v};
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long long)gridDim.x * 256) { if (NT) __builtin_nontemporal_store(val, p + i); else p[i] = val; }
}
__global__ __launch_bounds__(256) void k_copy(const d2 *__restrict__ a, d2 *__restrict__ p, long long n2)
{
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long long)gridDim.x * 256) __builtin_nontemporal_store(__builtin_nontemporal_load(a + i), p + i);
}
int main()
{
  const long long bytes = 1342177280LL, n2 = bytes / 16;
  d2 *p, *a; CK(hipMalloc(&p, bytes)); CK(hipMalloc(&a, bytes)); CK(hipMemset(a, 0, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto launch, const char *name, double b) {
    for (int r = 0; r < 2; r++) launch();
    CK(hipEventRecord(e0)); for (int r = 0; r < 5; r++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("%-40s %8.1f us  %7.1f GB/s\n", name, ms * 1e3, b / ms / 1e6);
  };
  char nm[96];
  for (int g : {256, 512, 1024, 2048, 8192}) {
    snprintf(nm, 96, "plain stores        grid %5d", g); time([&] { hipLaunchKernelGGL((k_write<false>), dim3(g), dim3(256), 0, 0, p, n2, 1.0); }, nm, (double)bytes);
    snprintf(nm, 96, "nontemporal stores  grid %5d", g); time([&] { hipLaunchKernelGGL((k_write<true>), dim3(g), dim3(256), 0, 0, p, n2, 1.0); }, nm, (double)bytes);
    snprintf(nm, 96, "copy (nt load+store) grid %5d", g); time([&] { hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, a, p, n2); }, nm, 2.0 * bytes);
  }
  return 0;
}
