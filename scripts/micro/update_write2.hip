// Second micro-benchmark of the written column (see update_write.hip): is its cost linear in the bytes written, does it depend on the number of
// read streams beside it, and does a blocked tile assignment (every workgroup a contiguous range of rows) change it?
// build: hipcc -O3 --offload-arch=gfx950 update_write2.hip -o update_write2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
// every `wfrac`-th tile is stored (0: none); blocked: workgroup b owns tiles [b*T, (b+1)*T)
template <int KT>
__global__ __launch_bounds__(256) void k_upd(const d2 *__restrict__ V, long long ld2, long long n2, d2 *v, const double *__restrict__ c, double *out, int wfrac, int blocked)
{
  double cc[KT];
#pragma unroll
  for (int i = 0; i < KT; i++) cc[i] = c[i];
  const long long ntiles = (n2 + 255) / 256;
  const long long per = (ntiles + gridDim.x - 1) / gridDim.x;
  double sink = 0.0;
  for (long long it = 0; it < per; it++) {
    const long long t = blocked ? (long long)blockIdx.x * per + it : it * gridDim.x + blockIdx.x;
    if (t >= ntiles) break;
    const long long j = t * 256 + threadIdx.x;
    if (j >= n2) continue;
    d2 s = v[j];
    d2 x[KT];
#pragma unroll
    for (int i = 0; i < KT; i++) x[i] = __builtin_nontemporal_load(V + i * ld2 + j);
#pragma unroll
    for (int i = 0; i < KT; i++) { s.x = fma(cc[i], x[i].x, s.x); s.y = fma(cc[i], x[i].y, s.y); }
    if (wfrac && (t % wfrac) == 0) v[j] = s; else sink += s.x + s.y;
  }
  if (sink == 12345.678) out[0] = sink;
}

int main()
{
  const long long n = 10077696, ld = n;
  d2 *V, *v; double *c, *out;
  CK(hipMalloc(&V, ld * 8 * 30)); CK(hipMemset(V, 0, ld * 8 * 30));
  CK(hipMalloc(&v, n * 8)); CK(hipMemset(v, 0, n * 8));
  CK(hipMalloc(&c, 8 * 30)); CK(hipMemset(c, 0, 8 * 30)); CK(hipMalloc(&out, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto launch, const char *name, int kt) {
    for (int r = 0; r < 3; r++) launch();
    CK(hipEventRecord(e0)); for (int r = 0; r < 10; r++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("%-72s %8.1f us   reads %7.1f GB/s\n", name, ms * 1e3, n * 8.0 * (kt + 1) / ms / 1e6);
  };
  char nm[160];
#define RUN(KT, wfrac, blocked) do { snprintf(nm, 160, "%2d read columns + v, %s tiles, store %s", KT, blocked ? "blocked    " : "round-robin", wfrac == 0 ? "none" : wfrac == 1 ? "every tile" : wfrac == 2 ? "every 2nd tile" : "every 4th tile"); \
    time([&] { hipLaunchKernelGGL((k_upd<KT>), dim3(256), dim3(256), 0, 0, V, ld / 2, n / 2, v, c, out, wfrac, blocked); }, nm, KT); } while (0)
  for (int blocked = 0; blocked < 2; blocked++) {
    RUN(30, 0, blocked); RUN(30, 4, blocked); RUN(30, 2, blocked); RUN(30, 1, blocked);
    RUN(15, 0, blocked); RUN(15, 1, blocked);
    RUN(7, 0, blocked); RUN(7, 1, blocked);
    RUN(1, 0, blocked); RUN(1, 1, blocked);
  }
  return 0;
}
