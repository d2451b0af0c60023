// Micro-benchmark for a two-phase ("propagation blocking") product of a uniformly random matrix, config-5 shape: n = 5e6, 33 entries per row.
//   phase 1: one workgroup per column slice (NS = 512 slices of 9766 columns): x slice in LDS, the slice's entries (16-bit local column,
//            ordered by destination wave-bin) gather from LDS and write G in bin-major order (segments of ~157 entries)
//   phase 2: one wave per wave-bin (WB = 2048 bins of 2442 rows): streams its contiguous G, val and 16-bit local row, adds into LDS
//            accumulators (ds_add_f64), writes its rows of y
// Synthetic entry streams (uniform segment length, random 16-bit indices): what the two kernels cost, not a correct product.
// build: hipcc -O3 --offload-arch=gfx950 binned_spmv.hip -o binned_spmv
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int NS = 512, WB = 2048, CS = 9766, WR = 2442, SEG = 158;

// phase 1: grid NS, 1024 threads. Segments start at even positions in both orders (one padding entry where needed), so a lane takes a PAIR of
// consecutive entries: one 4-byte load of two column indices, two LDS gathers, one 16-byte store. A wave takes 1024 consecutive entries of its slice
// at a time; the segment its window starts in comes from a table built with the layout (wseg), the next MAXSEG boundaries are read once per window.
constexpr int MAXSEG = 12;
typedef double d2v __attribute__((ext_vector_type(2)));
template <int DIAG>
__global__ __launch_bounds__(1024) void k_phase1(const unsigned short *__restrict__ col16, const int *__restrict__ off1, const int *__restrict__ off2t, const int *__restrict__ wseg, int nwin,
                                                 const double *__restrict__ x, double *__restrict__ G, long long per_slice, const double *__restrict__ val1)
{
  extern __shared__ double lds[];
  double *xs = lds;                       // CS doubles
  int *o1 = (int *)(lds + CS);            // WB + 1
  int *o2 = o1 + WB + 1;                  // WB
  const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = blockDim.x >> 6;
  for (int i = tid; i < CS; i += blockDim.x) xs[i] = x[(long long)s * CS + i];
  for (int i = tid; i <= WB; i += blockDim.x) o1[i] = off1[(long long)s * (WB + 1) + i];
  for (int i = tid; i < WB; i += blockDim.x) o2[i] = off2t[(long long)s * WB + i];
  __syncthreads();
  const unsigned *cs = (const unsigned *)(col16 + (long long)s * per_slice);
  const int total = o1[WB];               // even
  double sink = 0.0;
  for (int win = w; win * 1024 < total; win += nw) {
    const int base = win * 1024;
    unsigned c[8]; d2v a[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { const int e = base + k * 128 + 2 * lane; c[k] = e < total ? __builtin_nontemporal_load(cs + (e >> 1)) : 0u;
      if (DIAG == 5) a[k] = e < total ? __builtin_nontemporal_load((const d2v *)(val1 + (long long)s * per_slice + e)) : d2v{0.0, 0.0}; }
    const int lo = wseg[(long long)s * nwin + win];
    int bnd[MAXSEG], dlt[MAXSEG];
#pragma unroll
    for (int j = 0; j < MAXSEG; j++) { const int sg = min(lo + j, WB - 1); bnd[j] = o1[sg + 1]; dlt[j] = o2[sg] - o1[sg]; }
    const bool fits = bnd[MAXSEG - 1] >= min(base + 1024, total);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int e = base + k * 128 + 2 * lane;
      if (e < total) {
        int d;
        if (fits) {
          d = dlt[0];
#pragma unroll
          for (int j = 1; j < MAXSEG; j++) d = (e >= bnd[j - 1]) ? dlt[j] : d;
        } else { int sg = lo; while (e >= o1[sg + 1]) sg++; d = o2[sg] - o1[sg]; }
        d2v g = {xs[c[k] & 0xffffu], xs[c[k] >> 16]};
        if (DIAG == 5) { g.x *= a[k].x; g.y *= a[k].y; }
        if (DIAG == 1) sink += g.x + g.y + d;
        else if (DIAG == 6) *(d2v *)(G + (e + d)) = g;
        else if (DIAG == 7) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"((d2v *)(G + (e + d))), "v"(g) : "memory");
        else if (DIAG == 8) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"((d2v *)(G + (e + d))), "v"(g) : "memory");
        else __builtin_nontemporal_store(g, (d2v *)(G + (e + d)));
      }
    }
  }
  if (DIAG == 1 && sink == 1.2345e300) G[0] = sink;
}

// phase 2: grid WB / 4, 256 threads, one wave per wave-bin
template <bool VALS>
__global__ __launch_bounds__(256) void k_phase2(const double *__restrict__ G, const double *__restrict__ val, const unsigned short *__restrict__ row16,
                                                const long long *__restrict__ binstart, double *__restrict__ y, int n)
{
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wb = blockIdx.x * 4 + w;
  double *acc = lds + w * WR;
  for (int i = lane; i < WR; i += 64) acc[i] = 0.0;
  const long long e0 = binstart[wb], e1 = binstart[wb + 1];
  for (long long b = e0; b < e1; b += 512) {
    double g[8], a[8]; unsigned short r[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { const long long e = b + k * 64 + lane; const bool ok = e < e1; g[k] = ok ? __builtin_nontemporal_load(G + e) : 0.0; a[k] = VALS ? (ok ? __builtin_nontemporal_load(val + e) : 0.0) : 1.0; r[k] = ok ? __builtin_nontemporal_load(row16 + e) : 0; }
#pragma unroll
    for (int k = 0; k < 8; k++) { const long long e = b + k * 64 + lane; if (e < e1) __hip_atomic_fetch_add(acc + r[k], a[k] * g[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  }
  for (int i = lane; i < WR; i += 64) { const long long row = (long long)wb * WR + i; if (row < n) y[row] = acc[i]; }
}

int main()
{
  const long long per_slice = (long long)WB * SEG, nnz = per_slice * NS;
  const int n = 5000000;
  printf("entries %lld (%.1f per row)\n", nnz, (double)nnz / n);
  unsigned short *col16, *row16; double *val, *G, *x, *y; int *off1, *off2t, *wseg; long long *binstart;
  const int nwin = (int)((per_slice + 1023) / 1024);
  CK(hipMalloc(&col16, nnz * 2)); CK(hipMalloc(&row16, nnz * 2)); CK(hipMalloc(&val, nnz * 8)); CK(hipMalloc(&G, nnz * 8));
  CK(hipMalloc(&x, (long long)NS * CS * 8)); CK(hipMalloc(&y, (long long)WB * WR * 8));
  CK(hipMalloc(&off1, (long long)NS * (WB + 1) * 4)); CK(hipMalloc(&off2t, (long long)NS * WB * 4)); CK(hipMalloc(&binstart, (WB + 1) * 8));
  {
    std::vector<unsigned short> h(nnz);
    unsigned long long st = 88172645463325252ULL;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    for (long long i = 0; i < nnz; i++) h[i] = (unsigned short)(rnd() % CS);
    CK(hipMemcpy(col16, h.data(), nnz * 2, hipMemcpyHostToDevice));
    // bin-major: rows sorted inside each (wb, s) segment
    for (long long i = 0; i < nnz; i++) h[i] = (unsigned short)(((i % SEG) * WR) / SEG);
    CK(hipMemcpy(row16, h.data(), nnz * 2, hipMemcpyHostToDevice));
    std::vector<int> o1((long long)NS * (WB + 1)), o2((long long)NS * WB);
    const bool contiguous = getenv("CONTIG") != nullptr;      // diagnostic: phase 1 writes G in its own (slice-major) order
    const int grp = getenv("GROUP") ? atoi(getenv("GROUP")) : 1;   // GROUP=g: the segments of g consecutive wave-bins are adjacent per slice ([wb / g][slice][wb % g]); phase 2's timing is then meaningless
    printf("group %d\n", grp);
    for (int s = 0; s < NS; s++) { for (int wb = 0; wb <= WB; wb++) o1[(long long)s * (WB + 1) + wb] = wb * SEG; for (int wb = 0; wb < WB; wb++) o2[(long long)s * WB + wb] = contiguous ? (int)(((long long)s * WB + wb) * SEG) : (int)((((long long)(wb / grp) * NS + s) * grp + wb % grp) * SEG); }
    CK(hipMemcpy(off1, o1.data(), o1.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(off2t, o2.data(), o2.size() * 4, hipMemcpyHostToDevice));
    std::vector<int> ws((long long)NS * nwin); for (int sl = 0; sl < NS; sl++) for (int wi = 0; wi < nwin; wi++) ws[(long long)sl * nwin + wi] = (wi * 1024) / SEG;
    CK(hipMalloc(&wseg, ws.size() * 4)); CK(hipMemcpy(wseg, ws.data(), ws.size() * 4, hipMemcpyHostToDevice));
    std::vector<long long> bs(WB + 1); for (int wb = 0; wb <= WB; wb++) bs[wb] = (long long)wb * NS * SEG;
    CK(hipMemcpy(binstart, bs.data(), bs.size() * 8, hipMemcpyHostToDevice));
  }
  CK(hipMemset(val, 0, nnz * 8)); CK(hipMemset(x, 0, (long long)NS * CS * 8));
  const size_t lds1 = CS * 8 + (2 * WB + 1) * 4, lds2 = 4 * WR * 8;
  CK(hipFuncSetAttribute((const void *)k_phase1<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
  CK(hipFuncSetAttribute((const void *)k_phase1<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));

  CK(hipFuncSetAttribute((const void *)k_phase2<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2)); CK(hipFuncSetAttribute((const void *)k_phase2<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2)); CK(hipFuncSetAttribute((const void *)k_phase1<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1)); CK(hipFuncSetAttribute((const void *)k_phase1<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1)); CK(hipFuncSetAttribute((const void *)k_phase1<7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1)); CK(hipFuncSetAttribute((const void *)k_phase1<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
  hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  for (int rep = 0; rep < 6; rep++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_phase1<0>, dim3(NS), dim3(1024), lds1, 0, col16, off1, off2t, wseg, nwin, x, G, per_slice, (const double *)nullptr);
    CK(hipEventRecord(e1));
    hipLaunchKernelGGL(k_phase2<true>, dim3(WB / 4), dim3(256), lds2, 0, G, val, row16, binstart, y, n);
    CK(hipEventRecord(e2)); CK(hipEventSynchronize(e2));
    CK(hipGetLastError());
    float m1, m2; CK(hipEventElapsedTime(&m1, e0, e1)); CK(hipEventElapsedTime(&m2, e1, e2));
    if (rep >= 2) printf("phase 1 %7.1f us (%6.1f GB/s of 10 B/entry)   phase 2 %7.1f us (%6.1f GB/s of 18 B/entry)   total %7.1f us\n", m1 * 1e3, nnz * 10.0 / m1 / 1e6, m2 * 1e3, nnz * 18.0 / m2 / 1e6, (m1 + m2) * 1e3);
  }
  auto diag = [&](auto kern, const char *name) {
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(kern, dim3(NS), dim3(1024), lds1, 0, col16, off1, off2t, wseg, nwin, x, G, per_slice, (const double *)nullptr);
    CK(hipEventRecord(e0)); for (int r = 0; r < 4; r++) hipLaunchKernelGGL(kern, dim3(NS), dim3(1024), lds1, 0, col16, off1, off2t, wseg, nwin, x, G, per_slice, (const double *)nullptr);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); printf("phase 1, %-44s %7.1f us\n", name, ms / 4 * 1e3);
  };
  diag(k_phase1<0>, "as it is (nt store)"); diag(k_phase1<1>, "no store"); diag(k_phase1<6>, "plain store"); diag(k_phase1<7>, "sc1 store"); diag(k_phase1<8>, "sc0 sc1 nt store"); diag(k_phase1<0>, "as it is (nt store)");
  for (int rep = 0; rep < 4; rep++) {     // the values on phase 1's side: P = val * x[col] out, phase 2 reads P and the row only
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_phase1<5>, dim3(NS), dim3(1024), lds1, 0, col16, off1, off2t, wseg, nwin, x, G, per_slice, val);
    CK(hipEventRecord(e1));
    hipLaunchKernelGGL(k_phase2<false>, dim3(WB / 4), dim3(256), lds2, 0, G, val, row16, binstart, y, n);
    CK(hipEventRecord(e2)); CK(hipEventSynchronize(e2));
    float m1, m2; CK(hipEventElapsedTime(&m1, e0, e1)); CK(hipEventElapsedTime(&m2, e1, e2));
    if (rep >= 1) printf("values in phase 1: phase 1 %7.1f us (18 B/entry)   phase 2 %7.1f us (10 B/entry)   total %7.1f us\n", m1 * 1e3, m2 * 1e3, (m1 + m2) * 1e3);
  }
  for (int nt : {256, 512, 768, 1024}) {
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k_phase1<0>, dim3(NS), dim3(nt), lds1, 0, col16, off1, off2t, wseg, nwin, x, G, per_slice, (const double *)nullptr);
    CK(hipEventRecord(e0)); for (int r = 0; r < 4; r++) hipLaunchKernelGGL(k_phase1<0>, dim3(NS), dim3(nt), lds1, 0, col16, off1, off2t, wseg, nwin, x, G, per_slice, (const double *)nullptr);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); printf("phase 1 with %4d threads per workgroup: %7.1f us\n", nt, ms / 4 * 1e3);
  }
  return 0;
}
