// Micro-benchmark of the two Gram-Schmidt sweep shapes at n = 10 077 696, KT columns:
//   dot:    partial[i] += sum_r V(r,i) * y(r)                      (reads KT columns + y)
//   update: v(r) -= sum_i c_i V(r,i)  [and the fused dots]          (reads KT columns, reads + writes v)
// in the simplest possible form, to see what rate the access pattern itself allows for each grid size.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <int KT>
__global__ __launch_bounds__(256) void k_dot(const d2 *__restrict__ a, long long ld2, long long n2, const d2 *__restrict__ y, double *__restrict__ out)
{
  double acc[KT];
#pragma unroll
  for (int i = 0; i < KT; i++) acc[i] = 0.0;
  for (long long t = blockIdx.x; t * 256 < n2; t += gridDim.x) {
    const long long j = t * 256 + threadIdx.x;
    if (j < n2) {
      const d2 yv = y[j];
      d2 v[KT];
#pragma unroll
      for (int i = 0; i < KT; i++) v[i] = __builtin_nontemporal_load(a + i * ld2 + j);
#pragma unroll
      for (int i = 0; i < KT; i++) { acc[i] = fma(v[i].x, yv.x, acc[i]); acc[i] = fma(v[i].y, yv.y, acc[i]); }
    }
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < KT; i++) s += acc[i];
  if (s == 12345.678) out[0] = s;
}
template <int KT, bool FUSE, int NTV>
__global__ __launch_bounds__(256) void k_upd(const d2 *__restrict__ a, long long ld2, long long n2, d2 *__restrict__ vv, const double *__restrict__ c, double *__restrict__ out)
{
  double acc[KT + 1], cc[KT];
#pragma unroll
  for (int i = 0; i < KT; i++) { acc[i] = 0.0; cc[i] = c[i]; }
  acc[KT] = 0.0;
  for (long long t = blockIdx.x; t * 256 < n2; t += gridDim.x) {
    const long long j = t * 256 + threadIdx.x;
    if (j < n2) {
      d2 s = (NTV & 1) ? __builtin_nontemporal_load(vv + j) : vv[j];
      d2 v[KT];
#pragma unroll
      for (int i = 0; i < KT; i++) v[i] = __builtin_nontemporal_load(a + i * ld2 + j);
#pragma unroll
      for (int i = 0; i < KT; i++) { s.x = fma(cc[i], v[i].x, s.x); s.y = fma(cc[i], v[i].y, s.y); }
      if (NTV & 2) __builtin_nontemporal_store(s, vv + j); else vv[j] = s;
      if (FUSE) {
#pragma unroll
        for (int i = 0; i < KT; i++) { acc[i] = fma(v[i].x, s.x, acc[i]); acc[i] = fma(v[i].y, s.y, acc[i]); }
        acc[KT] = fma(s.x, s.x, acc[KT]); acc[KT] = fma(s.y, s.y, acc[KT]);
      }
    }
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i <= KT; i++) s += acc[i];
  if (s == 12345.678) out[0] = s;
}

int main()
{
  const long long ncol = 10077696, ld2 = ncol / 2; constexpr int KT = 28;
  d2 *a, *y; double *out, *c;
  CK(hipMalloc(&a, ncol * 8 * KT)); CK(hipMalloc(&y, ncol * 8)); CK(hipMalloc(&out, 8)); CK(hipMalloc(&c, 8 * 64));
  CK(hipMemset(a, 0, ncol * 8 * KT)); CK(hipMemset(y, 0, ncol * 8)); CK(hipMemset(c, 0, 8 * 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto launch, const char *name, double bytes) {
    for (int w = 0; w < 2; w++) launch();
    CK(hipEventRecord(e0)); for (int r = 0; r < 5; r++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("%-36s %8.3f ms  %7.1f GB/s\n", name, ms, bytes / ms / 1e6);
  };
  char nm[96];
  for (int g : {256, 512, 768, 1024, 2048}) {
    snprintf(nm, 96, "dot    KT=28 grid %5d", g); time([&] { hipLaunchKernelGGL((k_dot<KT>), dim3(g), dim3(256), 0, 0, a, ld2, ld2, y, out); }, nm, ncol * 8.0 * (KT + 1));
    snprintf(nm, 96, "update KT=28 grid %5d", g); time([&] { hipLaunchKernelGGL((k_upd<KT, false, 0>), dim3(g), dim3(256), 0, 0, a, ld2, ld2, y, c, out); }, nm, ncol * 8.0 * (KT + 2));
    snprintf(nm, 96, "update nt-store  grid %5d", g); time([&] { hipLaunchKernelGGL((k_upd<KT, false, 2>), dim3(g), dim3(256), 0, 0, a, ld2, ld2, y, c, out); }, nm, ncol * 8.0 * (KT + 2));
    snprintf(nm, 96, "update nt-ld+st  grid %5d", g); time([&] { hipLaunchKernelGGL((k_upd<KT, false, 3>), dim3(g), dim3(256), 0, 0, a, ld2, ld2, y, c, out); }, nm, ncol * 8.0 * (KT + 2));
    snprintf(nm, 96, "fused  nt-ld+st  grid %5d", g); time([&] { hipLaunchKernelGGL((k_upd<KT, true, 3>), dim3(g), dim3(256), 0, 0, a, ld2, ld2, y, c, out); }, nm, ncol * 8.0 * (KT + 2));
  }
  return 0;
}
