// Micro-benchmark: does a write -> read round trip of a PANEL of the binned product's gathered-x array stay in the 256 MB Infinity Cache?
// Phase W writes B bytes of G (16 B per lane) next to a read of B/4 bytes (the 16-bit columns); phase R reads the same B bytes of G, B bytes of values and
// B/4 bytes of rows. Total G = 1.25 GiB in panels of B bytes; serial (W(p) R(p) on one stream) and overlapped (W on one stream, R on another behind an
// event, so W(p+1) runs beside R(p)). Store / load policy of G varied; the value / row / column streams are always nontemporal.
// build: hipcc -O3 --offload-arch=gfx950 mall_roundtrip.hip -o mall_roundtrip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
enum { PLAIN = 0, NT = 1, SC1 = 2 };
template <int POL> __device__ inline void st(d2 *p, d2 v)
{
  if (POL == NT) __builtin_nontemporal_store(v, p);
  else if (POL == SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
  else *p = v;
}
template <int POL> __device__ inline d2 ld(const d2 *p)
{
  if (POL == NT) return __builtin_nontemporal_load(p);
  return *p;
}
// W: G[i] = f(C[i]) for i in the panel: lane <-> one 16-byte pair of G and the 4-byte pair of column codes it comes from (coalesced, 4 per lane in flight)
template <int POL>
__global__ __launch_bounds__(256) void k_w(d2 *__restrict__ G, const unsigned *__restrict__ C, long long n2)
{
  const long long stride = (long long)gridDim.x * 256;
  for (long long i0 = (long long)blockIdx.x * 256 + threadIdx.x; i0 < n2; i0 += 4 * stride) {
    unsigned c[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { const long long i = i0 + k * stride; c[k] = i < n2 ? __builtin_nontemporal_load(C + i) : 0u; }
#pragma unroll
    for (int k = 0; k < 4; k++) { const long long i = i0 + k * stride; if (i < n2) { const d2 v = {(double)(c[k] & 0xffffu), (double)(c[k] >> 16)}; st<POL>(G + i, v); } }
  }
}
template <int POL>
__global__ __launch_bounds__(256) void k_r(const d2 *__restrict__ G, const d2 *__restrict__ V, const unsigned *__restrict__ R, long long n2, double *__restrict__ sink)
{
  double s = 0;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i0 = (long long)blockIdx.x * 256 + threadIdx.x; i0 < n2; i0 += 4 * stride) {
    d2 g[4], v[4]; unsigned r[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { const long long i = i0 + k * stride < n2 ? i0 + k * stride : n2 - 1; g[k] = ld<POL>(G + i); v[k] = __builtin_nontemporal_load(V + i); r[k] = __builtin_nontemporal_load(R + i); }
#pragma unroll
    for (int k = 0; k < 4; k++) s += g[k].x * v[k].x + g[k].y * v[k].y + (double)r[k];
  }
  if (s == 12345.678) sink[0] = s;
}
int main(int argc, char **argv)
{
  const long long bytes = 1342177280LL;
  d2 *G, *V; unsigned *C, *R; double *sink;
  CK(hipMalloc(&G, bytes)); CK(hipMalloc(&V, bytes)); CK(hipMalloc(&C, bytes / 4 + 64)); CK(hipMalloc(&R, bytes / 4 + 64)); CK(hipMalloc(&sink, 8));
  CK(hipMemset(G, 0, bytes)); CK(hipMemset(V, 0, bytes)); CK(hipMemset(C, 0, bytes / 4 + 64)); CK(hipMemset(R, 0, bytes / 4 + 64));
  hipStream_t sa, sb; CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<hipEvent_t> ev(256); for (auto &evt : ev) CK(hipEventCreateWithFlags(&evt, hipEventDisableTiming));
  const int grid = argc > 1 ? atoi(argv[1]) : 2048;
  auto run = [&](int polw, int polr, long long B, bool overlap) {
    const int np = (int)((bytes + B - 1) / B);
    auto product = [&] {
      for (int p = 0; p < np; p++) {
        const long long o = (long long)p * B, b = std::min(B, bytes - o), n2 = b / 16;
        hipStream_t sw = sa, sr = overlap ? sb : sa;
        d2 *g = G + o / 16; const d2 *v = V + o / 16; const unsigned *c = C + o / 16, *r = R + o / 16;
        if (polw == PLAIN) hipLaunchKernelGGL(k_w<PLAIN>, dim3(grid), dim3(256), 0, sw, g, c, n2);
        else if (polw == NT) hipLaunchKernelGGL(k_w<NT>, dim3(grid), dim3(256), 0, sw, g, c, n2);
        else hipLaunchKernelGGL(k_w<SC1>, dim3(grid), dim3(256), 0, sw, g, c, n2);
        if (overlap) { CK(hipEventRecord(ev[p % 256], sw)); CK(hipStreamWaitEvent(sr, ev[p % 256], 0)); }
        if (polr == PLAIN) hipLaunchKernelGGL(k_r<PLAIN>, dim3(grid), dim3(256), 0, sr, g, v, r, n2, sink);
        else hipLaunchKernelGGL(k_r<NT>, dim3(grid), dim3(256), 0, sr, g, v, r, n2, sink);
      }
      if (overlap) { CK(hipEventRecord(ev[255], sb)); CK(hipStreamWaitEvent(sa, ev[255], 0)); }
    };
    for (int r = 0; r < 2; r++) product();
    CK(hipEventRecord(e0, sa)); for (int r = 0; r < 4; r++) product(); CK(hipEventRecord(e1, sa)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 4;
    static const char *pn[] = {"plain", "nt", "sc1"};
    printf("panel %5lld MB x %3d  G store %-5s load %-5s %-8s %8.1f us per product  (%.2f TB/s of 3.5 x G bytes)\n", B >> 20, np, pn[polw], pn[polr], overlap ? "overlap" : "serial", ms * 1e3,
           3.5 * bytes / ms / 1e9);
    fflush(stdout);
  };
  for (long long mb : {1280, 256, 128, 96, 64, 32})
    for (int polw : {NT, PLAIN, SC1})
      for (int polr : {NT, PLAIN})
        for (int ov = 0; ov < 2; ov++) {
          if (mb == 1280 && ov) continue;
          run(polw, polr, (long long)mb << 20, ov != 0);
        }
  return 0;
}
