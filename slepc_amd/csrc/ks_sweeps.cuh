// Row-sweep device kernels shared by the BV ops and the fused Gram-Schmidt (gfx950, wave64).
//
// All BV kernels on the Krylov path are HBM-bound tall-skinny sweeps over an n x k column-major
// panel (n ~ 1e7 rows, k <= 64 columns, ld >= n). The mapping is the same everywhere:
//   lane  <-> row  (VEC=2: two consecutive rows per lane, one 16-byte load per column per lane, so
//                   one wave-instruction reads 1 KiB contiguous of ONE column),
//   tile  = 256 threads x VEC rows, tiles dealt to blocks grid-stride,
//   all k columns of the tile are loaded by independent loads issued back to back (k x 1 KiB in
//   flight per wave -> deep memory-level parallelism at low occupancy),
//   per-thread partial sums live in registers for the whole sweep, are combined once per block with
//   wave64 shuffles + LDS, and leave the block as one row of the `partials` array; a 1-block kernel
//   sums the partials in a fixed order (run-to-run deterministic, independent of scheduling).
// Column counts are compile-time (KT) so accumulators stay in VGPRs; a launch uses the smallest
// KT >= ncols and clamps the column index for the tail (duplicate loads hit L1, results discarded).
#pragma once
#include "ksgpu_internal.h"

namespace ksk {

// Streaming (read-once) 16-byte load of a basis column: `global_load_dwordx4 ... nt`. The panel V is far larger
// than L2 + Infinity Cache and every element is used once per sweep, so the nontemporal hint keeps the stream from
// displacing the vector being updated; measured +6 % steps/s on MI355X (n = 1e7).
typedef double ks_d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 ldcol2(const double *p) { const ks_d2v t = __builtin_nontemporal_load(reinterpret_cast<const ks_d2v *>(p)); double2 v; v.x = t.x; v.y = t.y; return v; }
__device__ __forceinline__ double ldstream(const double *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ int ldstream(const int *p) { return __builtin_nontemporal_load(p); }
typedef unsigned ks_u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ldstream4(const uint4 *p) { const ks_u4v t = __builtin_nontemporal_load(reinterpret_cast<const ks_u4v *>(p)); uint4 v; v.x = t.x; v.y = t.y; v.z = t.z; v.w = t.w; return v; }

// A basis that fits the 256 MB Infinity Cache (config 2: 21 columns x 1e6 rows = 168 MB) is re-read from there by every sweep:
// plain loads then, the nontemporal hint only costs (measured +4.5 % steps/s on config 2, -7 % on config 3 with plain loads,
// profiles/r01e_other_configs_probe.txt). Chosen per launch from the size of the BV's storage.
__device__ __forceinline__ double2 ldplain2(const double *p) { return *reinterpret_cast<const double2 *>(p); }
template <bool PLAIN> __device__ __forceinline__ double2 ldbasis2(const double *p) { return PLAIN ? ldplain2(p) : ldcol2(p); }
static inline int ks_basis_is_cache_resident(size_t columns, size_t ld) { return columns * ld * sizeof(double) <= (size_t)200 * 1000 * 1000; }

constexpr int SW_BLOCK = 256;
constexpr int SW_WAVES = SW_BLOCK / 64;

// Sum over the 64 lanes with DPP moves (VALU cross-lane reads: quad permutes, row rotations, the two row broadcasts) instead of __shfl_down, which
// goes through the LDS crossbar (ds_bpermute, two per double and step): a sweep ends with KT + 1 of these per wave, 4 waves sharing one LDS pipe.
// The total forms in lane 63 and is handed to every lane.
template <int CTRL> __device__ __forceinline__ double dpp_mov(double v)
{
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v)
{
  v += dpp_mov<0xb1>(v);       // quad_perm [1,0,3,2]
  v += dpp_mov<0x4e>(v);       // quad_perm [2,3,0,1]
  v += dpp_mov<0x124>(v);      // row_ror 4
  v += dpp_mov<0x128>(v);      // row_ror 8: every lane of a 16-lane row holds the row's sum
  v += dpp_mov<0x142>(v);      // row_bcast 15: rows 1..3 add the sum of the row before
  v += dpp_mov<0x143>(v);      // row_bcast 31: rows 2, 3 add lane 31 (rows 0 + 1); lane 63 = all four rows
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// Block-combine KT per-thread accumulators and write them to partials[i*gridDim.x + blockIdx.x], i<ncols.
template <int KT>
__device__ __forceinline__ void block_write_partials(double (&acc)[KT], int ncols, double *__restrict__ partials)
{
  __shared__ double red[SW_WAVES][KT];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < KT; i++) {
    double s = wave_sum(acc[i]);
    if (lane == 0) red[w][i] = s;
  }
  __syncthreads();
  if ((int)threadIdx.x < ncols) {
    double s = red[0][threadIdx.x];
#pragma unroll
    for (int ww = 1; ww < SW_WAVES; ww++) s += red[ww][threadIdx.x];
    partials[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = s;
  }
}

// partials <- A(:,0:ncols)^T y        (gemv-C of BVDotVec_BLAS_Private, bvblas.c:240-261)
template <int KT, int VEC, bool PLAIN>
__device__ __forceinline__ void dot_tiles(const double *__restrict__ A, long long lda, int n, int ncols, const double *__restrict__ y, int rev, double (&acc)[KT])
{
  const long long tile = (long long)SW_BLOCK * VEC;
  const long long ntiles = ((long long)n + tile - 1) / tile;
  for (long long t0 = blockIdx.x; t0 < ntiles; t0 += gridDim.x) {
    const long long t = rev ? ntiles - 1 - t0 : t0;
    const long long r = t * tile + (long long)threadIdx.x * VEC;
    if (VEC == 2) {
      if (r + 1 < n) {
        // all KT column loads of the tile are issued back to back (KT KiB in flight per wave), then the products
        double2 xv[KT];
#pragma unroll
        for (int i = 0; i < KT; i++) { const int ii = i < ncols ? i : ncols - 1; xv[i] = ldbasis2<PLAIN>(A + (long long)ii * lda + r); }
        const double2 yv = *reinterpret_cast<const double2 *>(y + r);
        __builtin_amdgcn_sched_barrier(0);              // keep the scheduler from folding the loads back into a 2-deep load/fma chain
#pragma unroll
        for (int i = 0; i < KT; i++) { acc[i] = fma(xv[i].x, yv.x, acc[i]); acc[i] = fma(xv[i].y, yv.y, acc[i]); }
      } else if (r < n) {
        const double yv = y[r];
#pragma unroll
        for (int i = 0; i < KT; i++) { const int ii = i < ncols ? i : ncols - 1; acc[i] = fma(A[(long long)ii * lda + r], yv, acc[i]); }
      }
    } else {
      if (r < n) {
        const double yv = y[r];
#pragma unroll
        for (int i = 0; i < KT; i++) { const int ii = i < ncols ? i : ncols - 1; acc[i] = fma(A[(long long)ii * lda + r], yv, acc[i]); }
      }
    }
  }
}

template <int KT, int VEC>
__global__ __launch_bounds__(SW_BLOCK) void k_dot_sweep(const double *__restrict__ A, long long lda, int n, int ncols,
                                                        const double *__restrict__ y, double *__restrict__ partials,
                                                        const KsGsState *__restrict__ gate, int *__restrict__ pgrid, int rev, int plain)
{
  if (gate && !gate->active) return;
  if (pgrid && blockIdx.x == 0 && threadIdx.x == 0) *pgrid = gridDim.x;     // the partials' stride travels with them
  double acc[KT];
#pragma unroll
  for (int i = 0; i < KT; i++) acc[i] = 0.0;
  if (VEC == 2 && plain) dot_tiles<KT, VEC, true>(A, lda, n, ncols, y, rev, acc);
  else dot_tiles<KT, VEC, false>(A, lda, n, ncols, y, rev, acc);
  block_write_partials<KT>(acc, ncols, partials);
}

// y = beta*y + alpha*A(:,0:ncols) q   (gemv-N of BVMultVec_BLAS_Private, bvblas.c:56-67); q on device
template <int VEC>
__global__ __launch_bounds__(SW_BLOCK) void k_multvec(const double *__restrict__ A, long long lda, int n, int ncols, double alpha, double beta,
                                                      const double *__restrict__ q, double *__restrict__ y, const KsGsState *__restrict__ gate)
{
  if (gate && !gate->do_update) return;           // a chunk of a device-resident Gram-Schmidt update (wide bases): runs only when the bookkeeping asked for it
  const long long tile = (long long)SW_BLOCK * VEC;
  const long long ntiles = ((long long)n + tile - 1) / tile;
  for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const long long r = t * tile + (long long)threadIdx.x * VEC;
    if (VEC == 2 && r + 1 < n) {
      double2 s;
      if (beta == 0.0) { s.x = 0.0; s.y = 0.0; }
      else { s = *reinterpret_cast<const double2 *>(y + r); if (beta != 1.0) { s.x *= beta; s.y *= beta; } }
      int i = 0;
      for (; i + 8 <= ncols; i += 8) {
        double2 xv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) xv[u] = *reinterpret_cast<const double2 *>(A + (long long)(i + u) * lda + r);
#pragma unroll
        for (int u = 0; u < 8; u++) { const double c = alpha * q[i + u]; s.x = fma(c, xv[u].x, s.x); s.y = fma(c, xv[u].y, s.y); }
      }
      for (; i < ncols; i++) { const double2 xv = *reinterpret_cast<const double2 *>(A + (long long)i * lda + r); const double c = alpha * q[i]; s.x = fma(c, xv.x, s.x); s.y = fma(c, xv.y, s.y); }
      *reinterpret_cast<double2 *>(y + r) = s;
    } else {
      for (int v = 0; v < VEC; v++) {
        const long long rr = r + v;
        if (rr < n) {
          double s = (beta == 0.0) ? 0.0 : beta * y[rr];
          for (int i = 0; i < ncols; i++) s = fma(alpha * q[i], A[(long long)i * lda + rr], s);
          y[rr] = s;
        }
      }
    }
  }
}

// 1-block reduction of block partials: out[i] = sum_b partials[i*nblocks + b], i < ncols (fixed order).
// Each wave owns columns w, w+nw, ...; a lane first gathers its <= KS_MAX_BLOCKS/64 strided partials with
// independent loads (all in flight together: the dependent-load chain was the whole cost of this kernel),
// then sums them in index order, then the wave combines with a fixed shuffle tree.
__device__ __forceinline__ void reduce_partials_to_lds(const double *__restrict__ partials, int nblocks, int ncols, double *c_lds)
{
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  constexpr int PER_LANE = 16;
  if (nblocks <= 64 * 8) {
    // Grids of up to 512 workgroups (every sweep of a basis that is not huge, and all of them at 2 workgroups per CU): a wave takes FOUR of its
    // columns per round - 32 independent loads in flight per lane, one memory round trip for four columns instead of four. In the update
    // kernel this runs in EVERY workgroup's prologue, ahead of the sweep: at n = 1e6 (config 2) the column-at-a-time form was a sixth of the
    // kernel. Same order of additions per column as the general form below: the same bits.
    constexpr int G = 4, PL = 8;
    for (int i0 = w; i0 < ncols; i0 += nw * G) {
      double v[G][PL];
#pragma unroll
      for (int g = 0; g < G; g++) {
        const int i = i0 + g * nw;
        const double *p = partials + (size_t)(i < ncols ? i : i0) * nblocks;
#pragma unroll
        for (int u = 0; u < PL; u++) { const int b = lane + 64 * u; v[g][u] = (b < nblocks) ? p[b] : 0.0; }
      }
#pragma unroll
      for (int g = 0; g < G; g++) {
        const int i = i0 + g * nw;
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < PL; u++) s += v[g][u];
        // (the general form adds 16 values per lane and pass; the upper 8 are zeros here: s + 0.0 == s)
        s = wave_sum(s);
        if (lane == 0 && i < ncols) c_lds[i] = s;
      }
    }
    __syncthreads();
    return;
  }
  for (int i = w; i < ncols; i += nw) {
    const double *p = partials + (size_t)i * nblocks;
    double s = 0.0;
    for (int b0 = 0; b0 < nblocks; b0 += 64 * PER_LANE) {
      double v[PER_LANE];
#pragma unroll
      for (int u = 0; u < PER_LANE; u++) { const int b = b0 + lane + 64 * u; v[u] = (b < nblocks) ? p[b] : 0.0; }
#pragma unroll
      for (int u = 0; u < PER_LANE; u++) s += v[u];
    }
    s = wave_sum(s);
    if (lane == 0) c_lds[i] = s;
  }
  __syncthreads();
}

} // namespace ksk

// compiled column-tile sizes: every count up to 32 (a padded column would be a full extra HBM read: the streaming loads
// bypass the caches), then steps of 8
static inline int ks_kt_for(int ncols)
{
  if (ncols <= 1) return 1;
  if (ncols <= 32) return ncols;
  return (ncols + 7) / 8 * 8 > 64 ? 64 : (ncols + 7) / 8 * 8;
}

// dispatch helper: smallest compiled KT >= ncols
#define KS_KT_CASE(v, MACRO) case v: { MACRO(v); } break;
#define KS_KT_DISPATCH(ncols, MACRO)                                              \
  do {                                                                            \
    switch (ks_kt_for(ncols)) {                                                   \
      KS_KT_CASE(1, MACRO) KS_KT_CASE(2, MACRO) KS_KT_CASE(3, MACRO) KS_KT_CASE(4, MACRO) KS_KT_CASE(5, MACRO) KS_KT_CASE(6, MACRO) KS_KT_CASE(7, MACRO) KS_KT_CASE(8, MACRO) \
      KS_KT_CASE(9, MACRO) KS_KT_CASE(10, MACRO) KS_KT_CASE(11, MACRO) KS_KT_CASE(12, MACRO) KS_KT_CASE(13, MACRO) KS_KT_CASE(14, MACRO) KS_KT_CASE(15, MACRO) KS_KT_CASE(16, MACRO) \
      KS_KT_CASE(17, MACRO) KS_KT_CASE(18, MACRO) KS_KT_CASE(19, MACRO) KS_KT_CASE(20, MACRO) KS_KT_CASE(21, MACRO) KS_KT_CASE(22, MACRO) KS_KT_CASE(23, MACRO) KS_KT_CASE(24, MACRO) \
      KS_KT_CASE(25, MACRO) KS_KT_CASE(26, MACRO) KS_KT_CASE(27, MACRO) KS_KT_CASE(28, MACRO) KS_KT_CASE(29, MACRO) KS_KT_CASE(30, MACRO) KS_KT_CASE(31, MACRO) KS_KT_CASE(32, MACRO) \
      KS_KT_CASE(40, MACRO) KS_KT_CASE(48, MACRO) KS_KT_CASE(56, MACRO) default: { MACRO(64); } break;                                                                              \
    }                                                                             \
  } while (0)
