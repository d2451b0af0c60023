// Device side of the one-shot allreduce (ks_ctx.hip explains the protocol): used by the stand-alone kernel there and by the
// Gram-Schmidt slot's reduce kernel, which sends the block sums straight from LDS.
#pragma once
#include "ksgpu_internal.h"

// in: count doubles (LDS or global), out: count doubles (may alias in); one workgroup, all threads
__device__ __forceinline__ void ks_oneshot_sum(const double *in, double *out, int count, const KsOneShotArgs &o,
                                               unsigned (*sh)[2 * KS_ONESHOT_MAX_COUNT], int *failed)
{
  const int tid = threadIdx.x, nw = 2 * count, total = o.size * nw;
  const size_t par = (size_t)(o.par & 1u) * KS_ONESHOT_MAX_RANKS;
  const unsigned seq = o.seq;
  if (tid == 0) *failed = *(volatile int *)o.err_local;      // an earlier call of this rank gave up: send (the peers may still be fine), do not wait again
  for (int idx = tid; idx < total; idx += blockDim.x) {
    const int p = idx / nw, w = idx - p * nw;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(in[w >> 1]);
    const unsigned half = (w & 1) ? (unsigned)(bits >> 32) : (unsigned)bits;
    __hip_atomic_store(o.peer[p] + (par + o.me) * (2 * KS_ONESHOT_MAX_COUNT) + w, ((unsigned long long)seq << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __syncthreads();
  const long long t0 = wall_clock64();
  for (int idx = tid; idx < total; idx += blockDim.x) {
    const int src = idx / nw, w = idx - src * nw;
    const unsigned long long *slot = o.mine + (par + src) * (2 * KS_ONESHOT_MAX_COUNT) + w;
    unsigned long long pk;
    unsigned spins = 0;
    while (((pk = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) >> 32) != seq) {
      if (*(volatile int *)failed) { pk = 0; break; }
      if ((++spins & 255u) == 0 && wall_clock64() - t0 > o.timeout_ticks) { *failed = 1; pk = 0; break; }
      __builtin_amdgcn_s_sleep(2);
    }
    sh[src][w] = (unsigned)pk;
  }
  __syncthreads();
  if (*failed) {
    if (tid == 0 && *(volatile int *)o.err_local == 0) {
      *(volatile int *)o.err_local = (int)seq;
      __hip_atomic_store(o.err, (int)seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (int e = tid; e < count; e += blockDim.x) out[e] = __longlong_as_double(0x7ff8000000000000LL);
    return;
  }
  for (int e = tid; e < count; e += blockDim.x) {
    double acc = 0.0;
    for (int r = 0; r < o.size; r++) acc += __longlong_as_double((long long)(((unsigned long long)sh[r][2 * e + 1] << 32) | sh[r][2 * e]));
    out[e] = acc;
  }
}
