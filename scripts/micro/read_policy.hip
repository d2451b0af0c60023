// Micro-benchmark: the cache-policy bits of the basis loads. 31 columns of 216^3 doubles read through buffer loads (raw_buffer_load_b128, one
// descriptor over the whole 2.5 GB array) with every combination of sc0 / nt / sc1, against the global nontemporal load the library uses.
// build: hipcc -O3 --offload-arch=gfx950 read_policy.hip -o read_policy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
constexpr int KT = 31;

template <int AUX>
__global__ __launch_bounds__(256) void k_read_buf(const d2 *__restrict__ V, long long ld2, long long n2, double *out)
{
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)V, 0, 0xffffffffu, 0x00020000);
  const long long ntiles = (n2 + 255) / 256;
  double sink = 0.0;
  for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const long long j = t * 256 + threadIdx.x;
    if (j >= n2) continue;
    u4 x[KT];
#pragma unroll
    for (int i = 0; i < KT; i++) x[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(j * 16), (unsigned)((unsigned long long)i * ld2 * 16), AUX);
#pragma unroll
    for (int i = 0; i < KT; i++) { d2 v = __builtin_bit_cast(d2, x[i]); sink += v.x + v.y; }
  }
  if (sink == 12345.678) out[0] = sink;
}
template <bool NT>
__global__ __launch_bounds__(256) void k_read_glob(const d2 *__restrict__ V, long long ld2, long long n2, double *out)
{
  const long long ntiles = (n2 + 255) / 256;
  double sink = 0.0;
  for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const long long j = t * 256 + threadIdx.x;
    if (j >= n2) continue;
    d2 x[KT];
#pragma unroll
    for (int i = 0; i < KT; i++) x[i] = NT ? __builtin_nontemporal_load(V + i * ld2 + j) : V[i * ld2 + j];
#pragma unroll
    for (int i = 0; i < KT; i++) sink += x[i].x + x[i].y;
  }
  if (sink == 12345.678) out[0] = sink;
}
int main()
{
  const long long n = 10077696, ld = n;
  d2 *V; double *out;
  CK(hipMalloc(&V, ld * 8 * KT)); CK(hipMemset(V, 0, ld * 8 * KT)); CK(hipMalloc(&out, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto launch, const char *name) {
    for (int r = 0; r < 3; r++) launch();
    CK(hipEventRecord(e0)); for (int r = 0; r < 10; r++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("%-44s %8.1f us  %7.1f GB/s\n", name, ms * 1e3, n * 8.0 * KT / ms / 1e6);
  };
  time([&] { hipLaunchKernelGGL((k_read_glob<true>), dim3(256), dim3(256), 0, 0, V, ld / 2, n / 2, out); }, "global load, nt (the library's)");
  time([&] { hipLaunchKernelGGL((k_read_glob<false>), dim3(256), dim3(256), 0, 0, V, ld / 2, n / 2, out); }, "global load, plain");
#define RUN(A, name) time([&] { hipLaunchKernelGGL((k_read_buf<A>), dim3(256), dim3(256), 0, 0, V, ld / 2, n / 2, out); }, name)
  RUN(0, "buffer load, no bits"); RUN(1, "buffer load, sc0"); RUN(2, "buffer load, nt"); RUN(3, "buffer load, sc0 nt");
  RUN(16, "buffer load, sc1"); RUN(17, "buffer load, sc0 sc1"); RUN(18, "buffer load, sc1 nt"); RUN(19, "buffer load, sc0 sc1 nt");
  time([&] { hipLaunchKernelGGL((k_read_glob<true>), dim3(256), dim3(256), 0, 0, V, ld / 2, n / 2, out); }, "global load, nt (the library's)");
  return 0;
}
