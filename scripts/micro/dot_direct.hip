// Micro-benchmark for the LDS-free BVDot kernel: which per-wave access shape streams 64 columns (two 32-column panels) of
// n = 216^3 rows fastest, and what the MFMA work costs on top.
//   pat 0: lane (c = l&15, q = l>>4) loads rows r0+2q, +1 of column 16*t + c      (16 columns x 64 B per wave-instruction: MFMA operand layout)
//   pat 1: lane (c = l&7,  p = l>>3) loads rows r0+2p, +1 of column 8*t + c       (8 columns x 128 B per wave-instruction)
//   pat 2: lane l loads rows r0+2l, +1 of column t                                 (1 column x 1 KiB: the row sweeps' shape)
// build: hipcc -O3 --offload-arch=gfx950 dot_direct.hip -o dot_direct
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ d2 ldv(const double *p) { return NT ? __builtin_nontemporal_load((const d2 *)p) : *(const d2 *)p; }

// NCOL columns in all, U row groups per chunk. PAT 0: groups of 8 rows, PAT 1: 16 rows, PAT 2: 128 rows.
template <int PAT, int NCOL, int U, bool NT, bool MFMA>
__global__ __launch_bounds__(256) void k_stream(const double *__restrict__ a, long long ld, long long n, double *__restrict__ out)
{
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  constexpr int CPI = PAT == 0 ? 16 : (PAT == 1 ? 8 : 1);       // columns per wave-instruction
  constexpr int RPG = PAT == 0 ? 8 : (PAT == 1 ? 16 : 128);     // rows per group
  constexpr int NI = NCOL / CPI;                                // instructions per row group
  const int c = PAT == 0 ? (lane & 15) : (PAT == 1 ? (lane & 7) : 0);
  const int roff = PAT == 0 ? 2 * (lane >> 4) : (PAT == 1 ? 2 * (lane >> 3) : 2 * lane);
  const long long CH = (long long)RPG * U, nch = n / CH;
  const long long gw = (long long)blockIdx.x * 4 + w, GW = (long long)gridDim.x * 4;
  double s = 0.0;
  d4 acc[2][2];
  for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) acc[i][j] = (d4){0, 0, 0, 0};
  for (long long ch = gw; ch < nch; ch += GW) {
    const long long r0 = ch * CH + roff;
    d2 v[U][NI];
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int t = 0; t < NI; t++) v[u][t] = ldv<NT>(a + (long long)(t * CPI + c) * ld + r0 + (long long)u * RPG);
    __builtin_amdgcn_sched_barrier(0);
    if (MFMA && PAT == 0 && NI == 4) {
#pragma unroll
      for (int u = 0; u < U; u++)
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int j = 0; j < 2; j++) {
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[u][i].x, v[u][2 + j].x, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[u][i].y, v[u][2 + j].y, acc[i][j], 0, 0, 0);
          }
    } else {
#pragma unroll
      for (int u = 0; u < U; u++)
#pragma unroll
        for (int t = 0; t < NI; t++) s += v[u][t].x + v[u][t].y;
    }
  }
  if (MFMA) for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (s == 12345.678) out[0] = s;
}

int main()
{
  const long long n = 10077696, ld = n, NCOL = 64;
  double *a, *out;
  CK(hipMalloc(&a, n * 8 * NCOL)); CK(hipMalloc(&out, 8)); CK(hipMemset(a, 0, n * 8 * NCOL));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto launch, const char *name, double bytes) {
    for (int w = 0; w < 2; w++) launch();
    CK(hipEventRecord(e0)); for (int r = 0; r < 5; r++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("%-64s %8.3f ms  %7.1f GB/s\n", name, ms, bytes / ms / 1e6);
  };
  char nm[128];
#define RUN(PAT, U, NTL, MF, g) do { snprintf(nm, 128, "pat %d U=%d %s %s grid %5d", PAT, U, NTL ? "nt   " : "plain", MF ? "mfma" : "sum ", g); \
    time([&] { hipLaunchKernelGGL((k_stream<PAT, 64, U, NTL, MF>), dim3(g), dim3(256), 0, 0, a, ld, n, out); }, nm, n * 8.0 * NCOL); } while (0)
  for (int g : {256, 512, 768, 1024, 2048}) {
    RUN(0, 2, true, false, g); RUN(0, 4, true, false, g); RUN(0, 8, true, false, g); RUN(0, 4, false, false, g);
    RUN(0, 2, true, true, g); RUN(0, 4, true, true, g); RUN(0, 4, false, true, g);
    RUN(1, 2, true, false, g); RUN(1, 4, true, false, g); RUN(1, 4, false, false, g);
    RUN(2, 1, true, false, g);
  }
  return 0;
}
