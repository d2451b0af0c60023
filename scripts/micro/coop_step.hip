// Micro-benchmark: one Arnoldi step of config 2 (n = 10^6 rows, 21 basis columns = 168 MB, resident in the Infinity Cache) as THREE dependent launches
// (dot sweep; update in registers + next dots; final update + store) against ONE persistent launch with two grid barriers between the same three phases.
// Both forms run the same tile code and the same fixed-order reduction of the block partials at the head of phases 2 and 3 (what the library's update
// kernel does in its prologue). The barrier is the monotonic-counter form of the micro-architecture guide (lane-0 release fence, agent-scope add, sc1 poll
// with s_sleep, acquire fence); its bare cost is measured too (an otherwise empty persistent kernel with B barriers).
// build: hipcc -O3 --offload-arch=gfx950 coop_step.hip -o coop_step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int KT = 21, BLOCK = 256;
typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double wave_sum(double v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }

// block partial sums -> partials[c * grid + block]
__device__ __forceinline__ void put_partials(const double (&acc)[KT + 1], double *__restrict__ partials, int grid)
{
  __shared__ double red[4][KT + 1];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c <= KT; c++) { const double s = wave_sum(acc[c]); if (lane == 0) red[w][c] = s; }
  __syncthreads();
  if (threadIdx.x <= KT) partials[(size_t)threadIdx.x * grid + blockIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
  __syncthreads();
}
// every block: coefficients = fixed-order sum of the block partials (the update kernel's prologue)
__device__ __forceinline__ void get_coefs(const double *__restrict__ partials, int grid, double *c_lds)
{
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int c = w; c <= KT; c += 4) {
    double s = 0.0;
    for (int b = lane; b < grid; b += 64) s += __builtin_nontemporal_load(partials + (size_t)c * grid + b);
    s = wave_sum(s);
    if (lane == 0) c_lds[c] = s;
  }
  __syncthreads();
}
// PHASE 0: dots of v with the KT columns and itself. PHASE 1: v1 = v - V c (registers) and the dots of v1. PHASE 2: v = (v - V c) * alpha, stored.
template <int PHASE>
__device__ __forceinline__ void phase(const double *__restrict__ V, long long ld, int n, double *__restrict__ v, const double *c_lds, double *__restrict__ partials, int grid)
{
  const long long tile = 2LL * BLOCK, ntiles = (n + tile - 1) / tile;
  double acc[KT + 1];
#pragma unroll
  for (int c = 0; c <= KT; c++) acc[c] = 0.0;
  double cc[KT];
  if (PHASE > 0) {
    const double inv = 1.0 / (c_lds[KT] + 1.0);
#pragma unroll
    for (int c = 0; c < KT; c++) cc[c] = -c_lds[c] * inv * 1e-3;
  }
  for (long long t = blockIdx.x; t < ntiles; t += grid) {
    const long long r = t * tile + 2LL * threadIdx.x;
    if (r + 1 < n) {
      d2 s = *reinterpret_cast<const d2 *>(v + r);
      d2 xv[KT];
#pragma unroll
      for (int c = 0; c < KT; c++) xv[c] = *reinterpret_cast<const d2 *>(V + (long long)c * ld + r);
      if (PHASE > 0) {
#pragma unroll
        for (int c = 0; c < KT; c++) { s.x = fma(cc[c], xv[c].x, s.x); s.y = fma(cc[c], xv[c].y, s.y); }
      }
      if (PHASE == 2) { s.x *= 0.999; s.y *= 0.999; asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(reinterpret_cast<d2 *>(v + r)), "v"(s) : "memory"); }
      else {
#pragma unroll
        for (int c = 0; c < KT; c++) { acc[c] = fma(xv[c].x, s.x, acc[c]); acc[c] = fma(xv[c].y, s.y, acc[c]); }
        acc[KT] = fma(s.x, s.x, acc[KT]); acc[KT] = fma(s.y, s.y, acc[KT]);
      }
    }
  }
  if (PHASE < 2) put_partials(acc, partials, grid);
}
template <int PHASE>
__global__ __launch_bounds__(BLOCK) void k_phase(const double *__restrict__ V, long long ld, int n, double *__restrict__ v, const double *__restrict__ pin, double *__restrict__ pout)
{
  __shared__ double c_lds[KT + 1];
  if (PHASE > 0) get_coefs(pin, gridDim.x, c_lds);
  phase<PHASE>(V, ld, n, v, c_lds, pout, gridDim.x);
}
// grid barrier: monotonic counter, every block arrives once per barrier
__device__ __forceinline__ void grid_barrier(unsigned *counter, unsigned target)
{
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long long spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) { __builtin_amdgcn_s_sleep(4); if (++spins > 1000000LL) break; }   // bounded: never a hang
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}
__global__ __launch_bounds__(BLOCK) void k_step_persistent(const double *__restrict__ V, long long ld, int n, double *__restrict__ v, double *__restrict__ p0, double *__restrict__ p1,
                                                           unsigned *counter, unsigned base)
{
  __shared__ double c_lds[KT + 1];
  phase<0>(V, ld, n, v, c_lds, p0, gridDim.x);
  grid_barrier(counter, base + gridDim.x);
  get_coefs(p0, gridDim.x, c_lds);
  phase<1>(V, ld, n, v, c_lds, p1, gridDim.x);
  grid_barrier(counter, base + 2 * gridDim.x);
  get_coefs(p1, gridDim.x, c_lds);
  phase<2>(V, ld, n, v, c_lds, p0, gridDim.x);
}
__global__ __launch_bounds__(BLOCK) void k_barriers_only(unsigned *counter, unsigned base, int nb)
{
  for (int b = 1; b <= nb; b++) grid_barrier(counter, base + b * gridDim.x);
}
__global__ void k_empty() {}
int main()
{
  const int n = 1000000; const long long ld = 1000000;
  double *V, *v, *p0, *p1; unsigned *counter;
  CK(hipMalloc(&V, sizeof(double) * ld * KT)); CK(hipMalloc(&v, sizeof(double) * ld)); CK(hipMalloc(&p0, sizeof(double) * (KT + 1) * 4096)); CK(hipMalloc(&p1, sizeof(double) * (KT + 1) * 4096));
  CK(hipMalloc(&counter, 64)); CK(hipMemset(counter, 0, 64));
  std::vector<double> h((size_t)ld * KT); for (size_t i = 0; i < h.size(); i++) h[i] = 1e-3 * (double)((i * 2654435761ull) % 1000) - 0.5;
  CK(hipMemcpy(V, h.data(), sizeof(double) * ld * KT, hipMemcpyHostToDevice)); CK(hipMemcpy(v, h.data(), sizeof(double) * ld, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int ncu = 256; { hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); ncu = pr.multiProcessorCount; }
  unsigned base = 0;
  auto time = [&](auto fn, int reps) { for (int r = 0; r < 5; r++) fn(); CK(hipEventRecord(e0)); for (int r = 0; r < reps; r++) fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return 1e3 * ms / reps; };
  for (int per_cu : {1, 2}) {
    int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_step_persistent, BLOCK, 0));
    if (per_cu > occ) { printf("%d blocks per CU: the persistent kernel admits only %d\n", per_cu, occ); continue; }
    const int grid = ncu * per_cu;
    const double t3 = time([&] {
      hipLaunchKernelGGL(k_phase<0>, dim3(grid), dim3(BLOCK), 0, 0, V, ld, n, v, p0, p0);
      hipLaunchKernelGGL(k_phase<1>, dim3(grid), dim3(BLOCK), 0, 0, V, ld, n, v, p0, p1);
      hipLaunchKernelGGL(k_phase<2>, dim3(grid), dim3(BLOCK), 0, 0, V, ld, n, v, p1, p0); }, 200);
    const double t1 = time([&] { hipLaunchKernelGGL(k_step_persistent, dim3(grid), dim3(BLOCK), 0, 0, V, ld, n, v, p0, p1, counter, base); base += 2 * grid; }, 200);
    const double tb = time([&] { hipLaunchKernelGGL(k_barriers_only, dim3(grid), dim3(BLOCK), 0, 0, counter, base, 20); base += 20 * grid; }, 50);
    const double te = time([&] { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(BLOCK), 0, 0); }, 200);
    printf("grid %4d (%d per CU): three launches %7.2f us per step | one persistent launch, two grid barriers %7.2f us | a bare grid barrier %5.2f us (20 in an empty kernel: %6.1f us) | an empty launch %5.2f us\n",
           grid, per_cu, t3, t1, (tb - te) / 20.0, tb, te);
  }
  // the three phases one by one (events around 200 launches of each)
  const int grid = ncu * 2;
  const double a = time([&] { hipLaunchKernelGGL(k_phase<0>, dim3(grid), dim3(BLOCK), 0, 0, V, ld, n, v, p0, p0); }, 200);
  const double b = time([&] { hipLaunchKernelGGL(k_phase<1>, dim3(grid), dim3(BLOCK), 0, 0, V, ld, n, v, p0, p1); }, 200);
  const double c = time([&] { hipLaunchKernelGGL(k_phase<2>, dim3(grid), dim3(BLOCK), 0, 0, V, ld, n, v, p1, p0); }, 200);
  printf("phases alone at grid %d: dots %6.2f us (%.2f TB/s of 176 MB), update + dots %6.2f us, final update %6.2f us (184 MB: %.2f TB/s)\n", grid, a, 176.0 / a, b, c, 184.0 / c);
  return 0;
}
