// Micro-benchmark: what does the one written column of the final Gram-Schmidt update cost next to its 30 read columns?
// v <- v - V(:,0:30) c on n = 216^3 rows, tile = 512 rows per workgroup-iteration, 1 or 2 workgroups per CU, variants of the store.
// build: hipcc -O3 --offload-arch=gfx950 update_write.hip -o update_write
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int KT = 30;
// MODE 0: no store; 1: plain store in place; 2: plain store to another array; 3: nt store in place; 4: in place, store of tile t-1 after the loads of tile t;
//      5: nt load of v + nt store; 6: store in place through a system-scope (write-through) store; 7: nt store to another array
template <int MODE>
__global__ __launch_bounds__(256) void k_upd(const d2 *__restrict__ V, long long ld2, long long n2, d2 *v, d2 *w, const double *__restrict__ c, double *out)
{
  double cc[KT];
#pragma unroll
  for (int i = 0; i < KT; i++) cc[i] = c[i];
  const long long ntiles = (n2 + 255) / 256;
  d2 prev = {0.0, 0.0}; long long prevj = -1;
  double sink = 0.0;
  for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const long long j = t * 256 + threadIdx.x;
    if (j >= n2) continue;
    d2 s = (MODE == 5) ? __builtin_nontemporal_load(v + j) : v[j];
    d2 x[KT];
#pragma unroll
    for (int i = 0; i < KT; i++) x[i] = __builtin_nontemporal_load(V + i * ld2 + j);
    if (MODE == 4 && prevj >= 0) v[prevj] = prev;
#pragma unroll
    for (int i = 0; i < KT; i++) { s.x = fma(cc[i], x[i].x, s.x); s.y = fma(cc[i], x[i].y, s.y); }
#pragma unroll
    for (int i = 0; i < KT; i++) { s.x = fma(cc[i], x[i].x, s.x); s.y = fma(cc[i], x[i].y, s.y); }
    if (MODE == 0) sink += s.x + s.y;
    else if (MODE == 1) v[j] = s;
    else if (MODE == 2) w[j] = s;
    else if (MODE == 3 || MODE == 5) __builtin_nontemporal_store(s, v + j);
    else if (MODE == 4) { prev = s; prevj = j; }
    else if (MODE == 6) { __hip_atomic_store((double *)(v + j), s.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __hip_atomic_store((double *)(v + j) + 1, s.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
    else if (MODE == 7) __builtin_nontemporal_store(s, w + j);
  }
  if (MODE == 4 && prevj >= 0) v[prevj] = prev;
  if (MODE == 0 && sink == 12345.678) out[0] = sink;
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
    printf("%-70s %8.1f us   reads %7.1f GB/s\n", name, ms * 1e3, n * 8.0 * (KT + 1) / ms / 1e6);
  };
  char nm[128];
#define RUN(MODE, g, what) do { snprintf(nm, 128, "mode %d grid %4d: %s", MODE, g, what); \
    time([&] { hipLaunchKernelGGL((k_upd<MODE>), dim3(g), dim3(256), 0, 0, V, ld / 2, n / 2, v, w, c, out); }, nm); } while (0)
  for (int g : {256, 512, 1024}) {
    RUN(0, g, "31 columns read, nothing written");
    RUN(1, g, "plain store in place");
    RUN(2, g, "plain store to another array");
    RUN(3, g, "nontemporal store in place");
    RUN(4, g, "plain store in place, one tile late");
    RUN(5, g, "nontemporal load of v + nontemporal store");
    RUN(6, g, "system-scope 8-byte stores in place");
    RUN(7, g, "nontemporal store to another array");
  }
  return 0;
}
