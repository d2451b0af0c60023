// Micro-benchmark: how fast can a 21-column dot sweep read a basis that is resident in the Infinity Cache (config 2: n = 10^6, 168 MB)?
// The library's sweeps read it at 4.5-5.3 TB/s, slower than the same kernels stream config 3's basis from HBM (6.1-6.8 TB/s). Variants: workgroups per CU,
// cache policy of the loads (plain / nontemporal / sc1), one or two row tiles in flight per wave (software pipelining), rows per lane (2 = 16-byte loads,
// 4 = two 16-byte loads per column), tile order (interleaved over the grid / one contiguous range per workgroup).
// build: hipcc -O3 --offload-arch=gfx950 c2_sweep.hip -o c2_sweep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int KT = 21, BLOCK = 256;
typedef double d2 __attribute__((ext_vector_type(2)));
enum { PLAIN = 0, NT = 1, SC1 = 2, BUF = 3 /* buffer_load_dwordx4 ... sc1 through the builtin: the compiler tracks its completion */ };
typedef unsigned u4 __attribute__((ext_vector_type(4)));
template <int POL> __device__ __forceinline__ d2 ldp(const double *p)
{
  const d2 *q = reinterpret_cast<const d2 *>(p);
  if (POL == NT) return __builtin_nontemporal_load(q);
  if (POL == SC1) { d2 v; asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(q) : "memory"); return v; }
  return *q;
}
__device__ __forceinline__ double wave_sum(double v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }

// DEPTH row tiles (512 rows each) are loaded before the first product; CONTIG: a workgroup walks one contiguous range of tiles instead of every grid-th tile
template <int POL, int DEPTH, bool CONTIG>
__global__ __launch_bounds__(BLOCK) void k_dots(const double *__restrict__ V, long long ld, int n, const double *__restrict__ v, double *__restrict__ partials)
{
  const long long tile = 2LL * BLOCK, ntiles = (n + tile - 1) / tile;
  double acc[KT + 1];
#pragma unroll
  for (int c = 0; c <= KT; c++) acc[c] = 0.0;
  long long t0, t1, step;
  if (CONTIG) { const long long per = (ntiles + gridDim.x - 1) / gridDim.x; t0 = per * blockIdx.x; t1 = t0 + per < ntiles ? t0 + per : ntiles; step = 1; }
  else { t0 = blockIdx.x; t1 = ntiles; step = gridDim.x; }
  for (long long t = t0; t < t1; t += step * DEPTH) {
    d2 s[DEPTH], xv[DEPTH][KT];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
      const long long tt = t + d * step;
      const long long r = tt * tile + 2LL * threadIdx.x;
      const bool ok = tt < t1 && r + 1 < n;
      const long long rr = ok ? r : 0;
      s[d] = ldp<POL == BUF ? PLAIN : POL>(v + rr);
      if (!ok) { s[d].x = 0.0; s[d].y = 0.0; }
      if (POL == BUF) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)V, 0, (int)(ld * KT * 8), 0x00020000);
#pragma unroll
        for (int c = 0; c < KT; c++) xv[d][c] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(((long long)c * ld + rr) * 8), 0, 16));
        __builtin_amdgcn_sched_barrier(0);
      } else {
#pragma unroll
      for (int c = 0; c < KT; c++) xv[d][c] = ldp<POL>(V + (long long)c * ld + rr);
      if (POL == PLAIN || POL == NT) __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (POL == SC1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
#pragma unroll
      for (int c = 0; c < KT; c++) { acc[c] = fma(xv[d][c].x, s[d].x, acc[c]); acc[c] = fma(xv[d][c].y, s[d].y, acc[c]); }
      acc[KT] = fma(s[d].x, s[d].x, acc[KT]); acc[KT] = fma(s[d].y, s[d].y, acc[KT]);
    }
  }
  __shared__ double red[4][KT + 1];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c <= KT; c++) { const double q = wave_sum(acc[c]); if (lane == 0) red[w][c] = q; }
  __syncthreads();
  if (threadIdx.x <= KT) partials[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
int main()
{
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int ncu = 256; { hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); ncu = pr.multiProcessorCount; }
  double *partials; CK(hipMalloc(&partials, sizeof(double) * (KT + 1) * 8192));
  for (int n : {1000000, 10077696}) {
    const long long ld = n;
    double *V, *v; CK(hipMalloc(&V, sizeof(double) * ld * KT)); CK(hipMalloc(&v, sizeof(double) * ld));
    CK(hipMemset(V, 0, sizeof(double) * ld * KT)); CK(hipMemset(v, 0, sizeof(double) * ld));
    const double mb = 8.0 * n * (KT + 1) / 1e6;
    printf("n = %d: %d columns + the vector = %.0f MB per sweep (%s)\n", n, KT, mb, mb < 250 ? "resident in the Infinity Cache after the first sweep" : "streams from HBM");
    auto time = [&](auto fn, const char *name) {
      for (int r = 0; r < 5; r++) fn();
      CK(hipEventRecord(e0)); for (int r = 0; r < 50; r++) fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); const double us = 1e3 * ms / 50;
      printf("  %-62s %7.2f us  %5.2f TB/s\n", name, us, mb / us); fflush(stdout);
    };
    char nm[128];
#define RUN(POL, DEPTH, CONTIG, PC) do { int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_dots<POL, DEPTH, CONTIG>, BLOCK, 0)); \
    if ((PC) <= occ) { snprintf(nm, 128, "%-5s loads, %d tile(s) in flight, %-11s tiles, %d workgroups per CU", #POL, DEPTH, CONTIG ? "contiguous" : "interleaved", PC); \
      time([&] { hipLaunchKernelGGL((k_dots<POL, DEPTH, CONTIG>), dim3(ncu * (PC)), dim3(BLOCK), 0, 0, V, ld, n, v, partials); }, nm); } } while (0)
    for (int pc : {1, 2, 3, 4}) { RUN(PLAIN, 1, false, pc); RUN(NT, 1, false, pc); RUN(SC1, 1, false, pc); if (n <= 2000000) RUN(BUF, 1, false, pc); }
    for (int pc : {1, 2}) { RUN(PLAIN, 2, false, pc); RUN(NT, 2, false, pc); if (n <= 2000000) RUN(BUF, 2, false, pc); }
    for (int pc : {1, 2, 4}) { RUN(PLAIN, 1, true, pc); RUN(NT, 1, true, pc); }
    for (int pc : {1, 2}) { RUN(PLAIN, 2, true, pc); }
    CK(hipFree(V)); CK(hipFree(v));
  }
  return 0;
}
