// Third micro-benchmark of the final Gram-Schmidt update (30 read columns, v read and written, two passes of coefficients):
// does issuing the loads of tile t+1 before the arithmetic and the store of tile t (two register panels) recover the time the store costs?
// build: hipcc -O3 --offload-arch=gfx950 update_write3.hip -o update_write3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int KT = 30;

template <int NPASS, bool STORE>
__global__ __launch_bounds__(256) void k_plain(const d2 *__restrict__ V, long long ld2, long long n2, d2 *v, const double *__restrict__ c, double *out)
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
    for (int p = 0; p < NPASS; p++)
#pragma unroll
      for (int i = 0; i < KT; i++) { s.x = fma(cc[i], x[i].x, s.x); s.y = fma(cc[i], x[i].y, s.y); }
    if (STORE) v[j] = s; else sink += s.x + s.y;
  }
  if (sink == 12345.678) out[0] = sink;
}

// two register panels: the loads of the next tile are in flight while this tile is computed and stored
template <int NPASS, bool STORE>
__global__ __launch_bounds__(256) void k_pipe(const d2 *__restrict__ V, long long ld2, long long n2, d2 *v, const double *__restrict__ c, double *out)
{
  double cc[KT];
#pragma unroll
  for (int i = 0; i < KT; i++) cc[i] = c[i];
  const long long ntiles = n2 / 256;                 // (full tiles only in this benchmark)
  double sink = 0.0;
  d2 xa[KT], xb[KT], sa, sb;
  long long t = blockIdx.x;
  if (t >= ntiles) return;
  {
    const long long j = t * 256 + threadIdx.x;
    sa = v[j];
#pragma unroll
    for (int i = 0; i < KT; i++) xa[i] = __builtin_nontemporal_load(V + i * ld2 + j);
  }
  for (;;) {
    const long long tn = t + gridDim.x;
    const bool more = tn < ntiles;
    if (more) {
      const long long j = tn * 256 + threadIdx.x;
      sb = v[j];
#pragma unroll
      for (int i = 0; i < KT; i++) xb[i] = __builtin_nontemporal_load(V + i * ld2 + j);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < NPASS; p++)
#pragma unroll
      for (int i = 0; i < KT; i++) { sa.x = fma(cc[i], xa[i].x, sa.x); sa.y = fma(cc[i], xa[i].y, sa.y); }
    if (STORE) v[t * 256 + threadIdx.x] = sa; else sink += sa.x + sa.y;
    if (!more) break;
    t = tn;
    const long long tn2 = t + gridDim.x;
    const bool more2 = tn2 < ntiles;
    if (more2) {
      const long long j = tn2 * 256 + threadIdx.x;
      sa = v[j];
#pragma unroll
      for (int i = 0; i < KT; i++) xa[i] = __builtin_nontemporal_load(V + i * ld2 + j);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < NPASS; p++)
#pragma unroll
      for (int i = 0; i < KT; i++) { sb.x = fma(cc[i], xb[i].x, sb.x); sb.y = fma(cc[i], xb[i].y, sb.y); }
    if (STORE) v[t * 256 + threadIdx.x] = sb; else sink += sb.x + sb.y;
    if (!more2) break;
    t = tn2;
  }
  if (sink == 12345.678) out[0] = sink;
}


// update_write2's kernel (one pass), with its conditional store (VAR 0), with an unconditional store (VAR 1), and with the loop written as in k_plain (VAR 2)
template <int VAR>
__global__ __launch_bounds__(256) void k_w2(const d2 *__restrict__ V, long long ld2, long long n2, d2 *v, const double *__restrict__ c, double *out, int wfrac)
{
  double cc[KT];
#pragma unroll
  for (int i = 0; i < KT; i++) cc[i] = c[i];
  const long long ntiles = (n2 + 255) / 256;
  const long long per = (ntiles + gridDim.x - 1) / gridDim.x;
  double sink = 0.0;
  if (VAR == 2) {
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
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
  } else {
    for (long long it = 0; it < per; it++) {
      const long long t = it * gridDim.x + blockIdx.x;
      if (t >= ntiles) break;
      const long long j = t * 256 + threadIdx.x;
      if (j >= n2) continue;
      d2 s = v[j];
      d2 x[KT];
#pragma unroll
      for (int i = 0; i < KT; i++) x[i] = __builtin_nontemporal_load(V + i * ld2 + j);
#pragma unroll
      for (int i = 0; i < KT; i++) { s.x = fma(cc[i], x[i].x, s.x); s.y = fma(cc[i], x[i].y, s.y); }
      if (VAR == 0) { if (wfrac && (t % wfrac) == 0) v[j] = s; else sink += s.x + s.y; }
      else v[j] = s;
    }
  }
  if (sink == 12345.678) out[0] = sink;
}

int main()
{
  const long long n = 10077696, ld = n;
  d2 *V, *v; double *c, *out;
  CK(hipMalloc(&V, ld * 8 * KT)); CK(hipMemset(V, 0, ld * 8 * KT));
  CK(hipMalloc(&v, n * 8)); CK(hipMemset(v, 0, n * 8));
  CK(hipMalloc(&c, 8 * KT)); CK(hipMemset(c, 0, 8 * KT)); CK(hipMalloc(&out, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto launch, const char *name) {
    for (int r = 0; r < 3; r++) launch();
    CK(hipEventRecord(e0)); for (int r = 0; r < 10; r++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("%-64s %8.1f us   reads %7.1f GB/s\n", name, ms * 1e3, n * 8.0 * (KT + 1) / ms / 1e6);
  };
  char nm[160];
#define RUN(K, NP, ST, g) do { snprintf(nm, 160, "%-8s %d pass(es) %-9s grid %4d", #K, NP, ST ? "store" : "no store", g); \
    time([&] { hipLaunchKernelGGL((K<NP, ST>), dim3(g), dim3(256), 0, 0, V, ld / 2, n / 2, v, c, out); }, nm); } while (0)
#define RUNW(VAR, what) do { snprintf(nm, 160, "k_w2 variant %d: %s", VAR, what); \
    time([&] { hipLaunchKernelGGL((k_w2<VAR>), dim3(256), dim3(256), 0, 0, V, ld / 2, n / 2, v, c, out, 1); }, nm); } while (0)
  RUNW(0, "update_write2's kernel, conditional store"); RUNW(1, "same loop, unconditional store"); RUNW(2, "k_plain's loop, conditional store");
  RUNW(0, "update_write2's kernel, conditional store"); RUN(k_plain, 1, true, 256);
  {   // the same kernel with v at different places: right behind V's last column (the library's layout), and at offsets into a separate 1 GiB pool
    d2 *pool; CK(hipMalloc(&pool, (1LL << 30) + n * 8)); CK(hipMemset(pool, 0, (1LL << 30) + n * 8));
    d2 *Vb; CK(hipMalloc(&Vb, ld * 8 * (KT + 1))); CK(hipMemset(Vb, 0, ld * 8 * (KT + 1)));
    snprintf(nm, 160, "v = column 30 of a 31-column array, store");
    time([&] { hipLaunchKernelGGL((k_plain<2, true>), dim3(256), dim3(256), 0, 0, Vb, ld / 2, n / 2, Vb + KT * (ld / 2), c, out); }, nm);
    snprintf(nm, 160, "v = column 30 of a 31-column array, no store");
    time([&] { hipLaunchKernelGGL((k_plain<2, false>), dim3(256), dim3(256), 0, 0, Vb, ld / 2, n / 2, Vb + KT * (ld / 2), c, out); }, nm);
    const long long offs[] = {0, 4096, 65536, 1 << 20, 2 << 20, 16 << 20, 128 << 20, 512 << 20, 1000 << 20};
    for (long long off : offs) {
      snprintf(nm, 160, "v at pool + %lld KiB, store", off / 1024);
      time([&] { hipLaunchKernelGGL((k_plain<2, true>), dim3(256), dim3(256), 0, 0, V, ld / 2, n / 2, pool + off / 16, c, out); }, nm);
    }
    printf("addresses: V %p  v %p  pool %p  Vb %p\n", (void *)V, (void *)v, (void *)pool, (void *)Vb);
  }
  for (int g : {256}) {
    RUN(k_plain, 1, false, g); RUN(k_plain, 1, true, g); RUN(k_plain, 2, false, g); RUN(k_plain, 2, true, g);
    RUN(k_pipe, 1, false, g); RUN(k_pipe, 1, true, g); RUN(k_pipe, 2, false, g); RUN(k_pipe, 2, true, g);
  }
  return 0;
}
