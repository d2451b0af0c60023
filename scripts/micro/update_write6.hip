// Sixth micro-benchmark (out-of-place variants added to the fifth): the stored column written to ANOTHER array than the one v is read from.
// Fifth micro-benchmark of the final update's stored column: every combination of the store's cache-policy bits (sc0, sc1, nt) through inline
// assembly, and the same for the load of v. 30 read columns (nontemporal), v read and written in place, one workgroup per CU.
// build: hipcc -O3 --offload-arch=gfx950 update_write5.hip -o update_write5
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int KT = 30;

template <int BITS> __device__ __forceinline__ void st(d2 *p, d2 v)
{
  if (BITS == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
  if (BITS == 1) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
  if (BITS == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
  if (BITS == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
  if (BITS == 4) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
  if (BITS == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" :: "v"(p), "v"(v) : "memory");
  if (BITS == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory");
  if (BITS == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
}

template <int BITS, bool STORE>
__global__ __launch_bounds__(256) void k_upd(const d2 *__restrict__ V, long long ld2, long long n2, const d2 *v, d2 *vout, const double *__restrict__ c, double *out)
{
  double cc[KT];
#pragma unroll
  for (int i = 0; i < KT; i++) cc[i] = c[i];
  const long long ntiles = (n2 + 255) / 256;
  double sink = 0.0;
  for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const long long j = t * 256 + threadIdx.x;
    if (j >= n2) continue;
    d2 s = v[j];
    d2 x[KT];
#pragma unroll
    for (int i = 0; i < KT; i++) x[i] = __builtin_nontemporal_load(V + i * ld2 + j);
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
      for (int i = 0; i < KT; i++) { s.x = fma(cc[i], x[i].x, s.x); s.y = fma(cc[i], x[i].y, s.y); }
    if (STORE) st<BITS>(vout + j, s); else sink += s.x + s.y;
  }
  if (sink == 12345.678) out[0] = sink;
}

int main()
{
  const long long n = 10077696, ld = n;
  d2 *V, *v, *w; double *c, *out;
  CK(hipMalloc(&V, ld * 8 * KT)); CK(hipMemset(V, 0, ld * 8 * KT));
  CK(hipMalloc(&v, n * 8)); CK(hipMemset(v, 0, n * 8)); CK(hipMalloc(&w, n * 8)); CK(hipMemset(w, 0, n * 8));
  CK(hipMalloc(&c, 8 * KT)); CK(hipMemset(c, 0, 8 * KT)); CK(hipMalloc(&out, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto launch, const char *name) {
    for (int r = 0; r < 3; r++) launch();
    CK(hipEventRecord(e0)); for (int r = 0; r < 10; r++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("%-40s %8.1f us\n", name, ms * 1e3);
  };
#define RUN(B, name) time([&] { hipLaunchKernelGGL((k_upd<B, true>), dim3(256), dim3(256), 0, 0, V, ld / 2, n / 2, v, v, c, out); }, name)
  time([&] { hipLaunchKernelGGL((k_upd<0, false>), dim3(256), dim3(256), 0, 0, V, ld / 2, n / 2, v, v, c, out); }, "no store");
  RUN(0, "store (no bits)"); RUN(1, "store sc0"); RUN(2, "store sc1"); RUN(3, "store sc0 sc1"); RUN(4, "store nt"); RUN(5, "store sc0 nt"); RUN(6, "store sc1 nt"); RUN(7, "store sc0 sc1 nt");
#define RUNO(B, name) time([&] { hipLaunchKernelGGL((k_upd<B, true>), dim3(256), dim3(256), 0, 0, V, ld / 2, n / 2, v, w, c, out); }, name)
  RUNO(0, "other array, store (no bits)"); RUNO(2, "other array, store sc1"); RUNO(3, "other array, store sc0 sc1"); RUNO(4, "other array, store nt");
  RUN(2, "store sc1"); RUNO(2, "other array, store sc1"); RUN(3, "store sc0 sc1"); RUNO(3, "other array, store sc0 sc1");
  time([&] { hipLaunchKernelGGL((k_upd<0, false>), dim3(256), dim3(256), 0, 0, V, ld / 2, n / 2, v, v, c, out); }, "no store");
  return 0;
}
