// Micro-benchmark: random 8-byte gathers from a 40 MB vector, (a) uniformly over the whole vector from every XCD,
// (b) each XCD (blockIdx % 8) restricted to its own 1/8 slice (5 MB, about one L2). Prints ms and Ggathers/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_gather(const int *__restrict__ idx, const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ out, long long per_block)
{
  const long long base = (long long)blockIdx.x * per_block;
  double s = 0.0;
  for (long long e = threadIdx.x; e < per_block; e += 256 * 4) {
    int c[4]; double a[4], xv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { const long long ee = e + 256 * u; c[u] = ee < per_block ? __builtin_nontemporal_load(idx + base + ee) : -1; a[u] = ee < per_block ? __builtin_nontemporal_load(val + base + ee) : 0.0; }
#pragma unroll
    for (int u = 0; u < 4; u++) xv[u] = c[u] >= 0 ? x[c[u]] : 0.0;
#pragma unroll
    for (int u = 0; u < 4; u++) s = fma(a[u], xv[u], s);
  }
  out[(long long)blockIdx.x * 256 + threadIdx.x] = s;
}

static uint64_t sm(uint64_t &z) { z += 0x9E3779B97F4A7C15ULL; uint64_t x = z; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL; x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL; return x ^ (x >> 31); }

int main()
{
  const long long n = 5000000, nnz = 160LL * 1000 * 1000;
  const int blocks = 8192; const long long per_block = nnz / blocks;
  std::vector<int> ia(nnz), ib(nnz), ic(nnz);
  uint64_t z = 42;
  for (long long b = 0; b < blocks; b++) {
    const long long lo = (b % 8) * (n / 8);
    for (long long e = 0; e < per_block; e++) { const uint64_t r = sm(z); ia[b * per_block + e] = (int)(r % n); ib[b * per_block + e] = (int)(lo + (r >> 20) % (n / 8));
      const long long s16 = (b % 8) + 8 * (e >= per_block / 2); ic[b * per_block + e] = (int)(s16 * (n / 16) + (r >> 20) % (n / 16)); }
  }
  int *d_idx; double *d_val, *d_x, *d_out;
  CK(hipMalloc(&d_idx, nnz * 4)); CK(hipMalloc(&d_val, nnz * 8)); CK(hipMalloc(&d_x, n * 8)); CK(hipMalloc(&d_out, (size_t)blocks * 256 * 8));
  CK(hipMemset(d_val, 0, nnz * 8)); CK(hipMemset(d_x, 0, n * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 3; mode++) {
    CK(hipMemcpy(d_idx, mode == 2 ? ic.data() : (mode ? ib.data() : ia.data()), nnz * 4, hipMemcpyHostToDevice));
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, 0, d_idx, d_val, d_x, d_out, per_block);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 10; r++) hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, 0, d_idx, d_val, d_x, d_out, per_block);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("%s: %.3f ms per pass, %.1f Ggather/s, %.0f GB/s of streamed idx+val\n", mode == 2 ? "16 slices, two passes per XCD" : (mode ? "per-XCD slices (blockIdx%%8)" : "uniform over 40 MB"), ms, nnz / ms / 1e6, nnz * 12.0 / ms / 1e6);
  }
  return 0;
}
