// MatMult(AIJ): CSR sparse matrix-vector product for gfx950.
//
// Replaces PETSc MatMult_SeqAIJ / MatMult_MPIAIJ as reached from BVMatMultColumn
// (src/sys/classes/bv/interface/bvops.c:862-885) through MatMult_STOperator / STApply_Generic
// (src/sys/classes/st/interface/stsolve.c:16-25,244-259).
//
// Layout in HBM (PETSc MPIAIJ style): the rank's row block is split into a "diagonal" block whose
// columns are owned by this rank (stored with LOCAL column indices) and an "off-diagonal" block
// whose columns live on other ranks (stored with compressed GHOST indices). rowptr int32[n+1],
// col int32[nnz], val f64[nnz]. x, y are columns of the BV (contiguous, stride 1).
//
// Kernel: "CSR-vector with sub-wave row groups". G = 2^g lanes cooperate on one row (G chosen from
// the mean row length: 8 for the 7-point Laplacian, 32 for ~32 nnz/row), so a 64-wide wavefront
// streams 64/G consecutive rows whose val/col entries are contiguous in memory: the val (8 B/lane)
// and col (4 B/lane) loads of a wave are one coalesced segment. Partial products are combined with
// DPP/shuffle butterflies inside the group. Each thread keeps UNROLL independent rows in flight
// so that rowptr -> (col,val) -> x[col] dependent chains of different rows overlap.
// Algorithmic bytes per call: 12*nnz + 4*(n+1) + 16*n.
#include "ks_sweeps.cuh"
#include "ks_csr.h"
#include <algorithm>
#include <numeric>
#include <thread>
#include <system_error>
#include <sched.h>
#include <new>
#include <type_traits>

namespace {

constexpr int SPMV_BLOCK = 256;

template <int G>
__device__ __forceinline__ double group_reduce(double v)
{
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// y[row] (+)= sum_p val[p] * x[col[p]]  over rows handled through an optional compressed row list
template <int G, int UNROLL, bool ACCUM, bool ROWLIST>
__global__ __launch_bounds__(SPMV_BLOCK) void k_spmv_csr(int nrows, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                          const double *__restrict__ val, const double *__restrict__ x,
                                                          double *__restrict__ y, const int *__restrict__ rowlist)
{
  constexpr int GPW = 64 / G;                       // row groups per wavefront
  constexpr int RW = GPW * UNROLL;                  // rows per wavefront per sweep
  const int lane = threadIdx.x & 63;
  const int gi = lane / G, lane_g = lane % G;
  const long long wave = ((long long)blockIdx.x * SPMV_BLOCK + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * SPMV_BLOCK) >> 6;
  // for a fixed u the GPW groups of a wave own GPW CONSECUTIVE rows, so one wave-instruction
  // reads one contiguous run of val/col entries
  for (long long wb = wave * RW; wb < nrows; wb += nwaves * RW) {
    int p0[UNROLL], p1[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      long long r = wb + u * GPW + gi;
      if (r < nrows) { p0[u] = rowptr[r]; p1[u] = rowptr[r + 1]; } else { p0[u] = 0; p1[u] = 0; }
    }
    double s[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      double acc = 0.0;
      for (int p = p0[u] + lane_g; p < p1[u]; p += G) acc = fma(val[p], x[col[p]], acc);
      s[u] = acc;
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      double t = group_reduce<G>(s[u]);
      long long r = wb + u * GPW + gi;
      if (lane_g == 0 && r < nrows) {
        long long out = ROWLIST ? rowlist[r] : r;
        if (ACCUM) y[out] += t; else __builtin_nontemporal_store(t, y + out);
      }
    }
  }
}


// ---- CSR, row blocks streamed through LDS ("coalesced CSR row-block loads") -----------------------------------------------------
// A workgroup takes 256 consecutive rows; their entries are ONE contiguous run of col / val, which it streams in chunks of 1024
// with fully coalesced nontemporal loads (lane e of a chunk loads entry e: no lane idles on a short row, no row length is a
// multiple of anything), gathers x for each entry (4 independent gathers per lane in flight) and parks value and x in LDS;
// then every row (= thread) runs the reference's own loop over its segment: acc = fma(val, x, acc) in entry order, so the
// result has the bits of the SELL / dictionary kernels. LDS slots are skewed by one per 32 so that rows whose length is a
// multiple of 32 do not put a whole wave on one bank. The general-matrix kernel: ragged rows, empty rows, rows longer than a
// chunk; the CSR-vector kernel above keeps the small matrices and the compressed off-diagonal block.
constexpr int CS_EPT = 4, CS_CHUNK = 256 * CS_EPT;    // 1024-entry chunks: 212 us on the 216^3 Laplacian (512: 224, 2048: 252)
__device__ __forceinline__ int cs_slot(int e) { return e + (e >> 5); }
__global__ __launch_bounds__(256) void k_spmv_csr_stream(int n, const int *__restrict__ rp, const int *__restrict__ col, const double *__restrict__ val,
                                                         const double *__restrict__ x, double *__restrict__ y)
{
  __shared__ double sa[CS_CHUNK + CS_CHUNK / 32], sx[CS_CHUNK + CS_CHUNK / 32];
  __shared__ int erange[2];
  const int tid = threadIdx.x;
  for (long long R0 = (long long)blockIdx.x * 256; R0 < n; R0 += (long long)gridDim.x * 256) {
    const long long r = R0 + tid;
    const bool has = r < n;
    const int p0 = has ? ksk::ldstream(rp + r) : 0, p1 = has ? ksk::ldstream(rp + r + 1) : 0;
    if (tid == 0) erange[0] = p0;
    if (has && (r == n - 1 || tid == 255)) erange[1] = p1;
    __syncthreads();
    const int E0 = erange[0], E1 = erange[1];
    double acc = 0.0;
    for (int e0 = E0; e0 < E1; e0 += CS_CHUNK) {
      int c[CS_EPT]; double a[CS_EPT];
#pragma unroll
      for (int u = 0; u < CS_EPT; u++) { const int e = e0 + u * 256 + tid; const bool ok = e < E1; c[u] = ok ? ksk::ldstream(col + e) : -1; a[u] = ok ? ksk::ldstream(val + e) : 0.0; }
#pragma unroll
      for (int u = 0; u < CS_EPT; u++) { const int sl = cs_slot(u * 256 + tid); sa[sl] = a[u]; sx[sl] = c[u] >= 0 ? x[c[u]] : 0.0; }
      __syncthreads();
      const int lo = max(p0, e0), hi = min(p1, e0 + CS_CHUNK);
      for (int p = lo; p < hi; p++) { const int sl = cs_slot(p - e0); acc = fma(sa[sl], sx[sl], acc); }
      __syncthreads();
    }
    if (has) __builtin_nontemporal_store(acc, y + r);
    __syncthreads();                       // erange is rewritten by the next row block
  }
}

// ---- CSR, row blocks streamed through LDS, one wave per 64 rows (no workgroup barrier) -------------------------------------------
// The same idea with the wave as the unit. A wave takes 64 consecutive rows; their entries are ONE contiguous run of col / val, which it
// streams in chunks of 512 with fully coalesced nontemporal loads (lane l of step u loads entry 64 u + l) and parks in a wave-private piece
// of LDS; then lane = row: every lane runs the reference's loop over its own row's segment, acc = fma(val, x, acc) in entry order - the
// bits of the SELL / dictionary kernels. Nothing waits for another wave: LDS operations of one wave execute in order, so the hand-over
// from the loading lanes to the row lanes needs no barrier (the workgroup form above spends four fifths of its wave cycles parked at
// barriers and s_waitcnt, profiles/r03_pmc_csr_kernels.txt), and the col / val loads of the next chunk - of the same rows or of the
// wave's next 64, whose row pointers were loaded one group ahead - are issued before the row sums of this one. Workgroups on one XCD
// (blockIdx % 8) take one contiguous eighth of the rows, so an XCD's L2 holds the part of x its rows gather from.
// ROWSIDE: where x is gathered. false: by the loading lanes (lane = entry), (value, x) pairs go through LDS - every lane has work whatever
// the row lengths. true: by the row lanes (lane = row) - (value, column) go through LDS and the 64 lanes of a gather instruction ask for the
// same entry position of 64 consecutive rows, which for a banded or stencil-like matrix is a few cache lines where the entry-side gather
// touches two to three times as many (216^3 Laplacian: the entry-side gathers cost 35 of 211 us, profiles/r03_csr_wave_variants.txt);
// pays only while a chunk spans most of the wave's rows, i.e. for short rows: chosen at assembly from the mean row length.
// (16-byte loads of four consecutive entries per lane were tried for the streams: fewer instructions, no faster, and the gathers of such a
// lane assignment touch still more lines.)
typedef __attribute__((address_space(3))) void ks_lds_void;          // operands of __builtin_amdgcn_global_load_lds (LDS-DMA)
typedef const __attribute__((address_space(1))) void ks_glb_void;
constexpr int CW_PAD = 8;                                  // col / val allocations are this much longer than nnz
// CW_STEPS: 64 entries per step; 8 steps = chunks of 512 (row side with 256-entry chunks: 62 registers, 8 waves per SIMD, and 237 us instead of 199)
__device__ __forceinline__ int cw_slot(int e) { return e + (e >> 5); }     // one slot of skew per 32 entries (rows whose length is a multiple of 32)
// Load width matters more than instruction count here: 4-byte-per-lane streaming loads top out at 0.7 - 2.5 TB/s on this part, 8- and 16-byte
// ones at 7 (scripts/micro/load_width.hip, profiles/r03_micro_load_width.txt). So the 4-byte column indices are loaded two per lane (a chunk
// starts on an even entry: aligned 8-byte loads; lane l of step u holds entries 128 u + 2 l, + 1), the values one per lane (entry 64 u + l),
// and a row's two row pointers come as one 8-byte load.
typedef int ks_i2v __attribute__((ext_vector_type(2)));
template <int CW_STEPS> struct CwRegs { ks_i2v c[CW_STEPS / 2]; double a[CW_STEPS]; };
template <int CW_STEPS>
__device__ __forceinline__ void cw_load(CwRegs<CW_STEPS> &r, const int *__restrict__ col, const double *__restrict__ val, int e0, int E1, int lane)
{
#pragma unroll
  for (int u = 0; u < CW_STEPS / 2; u++) {
    const int e = e0 + u * 128 + 2 * lane;
    r.c[u] = e < E1 ? __builtin_nontemporal_load(reinterpret_cast<const ks_i2v *>(col + e)) : ks_i2v{-1, -1};      // may take one entry past E1: CW_PAD
  }
#pragma unroll
  for (int u = 0; u < CW_STEPS; u++) {
    const int e = e0 + u * 64 + lane;
    r.a[u] = e < E1 ? ksk::ldstream(val + e) : 0.0;
  }
}
struct CwRows { int p0, p1, E0, E1; long long r; bool has; };
__device__ __forceinline__ CwRows cw_rows(int n, const int *__restrict__ rp, int g, int w, int lane)
{
  CwRows q; q.p0 = q.p1 = q.E0 = q.E1 = 0; q.has = false;
  const long long r0 = (long long)g * 256 + (long long)w * 64;
  q.r = r0 + lane;
  if (r0 >= n) return q;
  q.has = q.r < n;
  if (q.has) { ks_i2v pp; __builtin_memcpy(&pp, rp + q.r, sizeof(pp)); q.p0 = pp.x; q.p1 = pp.y; }       // rp[r], rp[r + 1]: one 8-byte load (4-byte aligned)
  q.E0 = rp[r0]; q.E1 = rp[r0 + 64 < n ? r0 + 64 : n];                 // the wave's run of entries (uniform: scalar loads)
  return q;
}
constexpr int CW_U = 4;                                    // row side: gathers in flight per lane (8: 110 registers, 4 waves per SIMD, 204 us; 2 or 3 at 6 waves per SIMD still spill)
template <bool ROWSIDE, int CW_STEPS>
__global__ __launch_bounds__(256, ROWSIDE ? 5 : 4) void k_spmv_csr_wave(int n, const int *__restrict__ rp, const int *__restrict__ col, const double *__restrict__ val,
                                                       const double *__restrict__ x, double *__restrict__ y, int xcd_remap)
{
  constexpr int CW_CHUNK = 64 * CW_STEPS;
  __shared__ double sa_all[4][CW_CHUNK + CW_CHUNK / 32];
  __shared__ double sb_all[4][ROWSIDE ? (CW_CHUNK + CW_CHUNK / 32) / 2 : CW_CHUNK + CW_CHUNK / 32];       // x values, or the columns (4 bytes each)
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double *sa = sa_all[w], *sx = sb_all[w];
  int *sc = reinterpret_cast<int *>(sb_all[w]);
  const int NG = (n + 255) / 256;                          // groups of 256 rows: one per workgroup and iteration, 64 rows per wave
  int g, gend, gstep;
  if (xcd_remap) {
    const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3, lc = gridDim.x >> 3;
    g = (int)((long long)NG * xcd / 8) + li; gend = (int)((long long)NG * (xcd + 1) / 8); gstep = lc;
  } else { g = blockIdx.x; gend = NG; gstep = gridDim.x; }
  if (g >= gend) return;
  // Software pipeline over (row group, chunk): while the rows of one chunk are summed, the col / val loads of the NEXT chunk are in
  // flight - the next chunk of the same rows, or the first chunk of the wave's next 64 rows.
  CwRows cu = cw_rows(n, rp, g, w, lane);
  CwRows nx = g + gstep < gend ? cw_rows(n, rp, g + gstep, w, lane) : CwRows{0, 0, 0, 0, 0, false};
  CwRegs<CW_STEPS> cur, nxt;
  int e0 = cu.E0 & ~1;
  if (e0 < cu.E1) cw_load(cur, col, val, e0, cu.E1, lane);
  bool nxt_loaded = false;                                 // the first chunk of group nx is already in `nxt`
  double acc = 0.0;
  for (;;) {
    if (e0 < cu.E1) {
#pragma unroll
      for (int u = 0; u < CW_STEPS; u++) sa[cw_slot(u * 64 + lane)] = cur.a[u];
      if (ROWSIDE) {
#pragma unroll
        for (int u = 0; u < CW_STEPS / 2; u++) { const int sl = cw_slot(u * 128 + 2 * lane); sc[sl] = cur.c[u].x; sc[sl + 1] = cur.c[u].y; }      // 2 l, 2 l + 1 never straddle a skew step
      } else {
        double xg[CW_STEPS];
#pragma unroll
        for (int u = 0; u < CW_STEPS / 2; u++) {
          const int e = e0 + u * 128 + 2 * lane;
          xg[2 * u] = e < cu.E1 ? x[cur.c[u].x] : 0.0; xg[2 * u + 1] = e + 1 < cu.E1 ? x[cur.c[u].y] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < CW_STEPS / 2; u++) { const int sl = cw_slot(u * 128 + 2 * lane); sx[sl] = xg[2 * u]; sx[sl + 1] = xg[2 * u + 1]; }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int en = e0 + CW_CHUNK;
      if (en < cu.E1) cw_load(nxt, col, val, en, cu.E1, lane);
      else if ((nx.E0 & ~1) < nx.E1) { cw_load(nxt, col, val, nx.E0 & ~1, nx.E1, lane); nxt_loaded = true; }
      const int lo = max(cu.p0, e0), hi = min(cu.p1, en);
      if (ROWSIDE) {
        for (int p = lo; __builtin_amdgcn_ballot_w64(p < hi) != 0; p += CW_U) {
          double av[CW_U], xv[CW_U];
#pragma unroll
          for (int j = 0; j < CW_U; j++) {
            const bool ok = p + j < hi;
            const int sl = cw_slot(ok ? p + j - e0 : 0);
            av[j] = sa[sl];
            xv[j] = ok ? x[sc[sl]] : 0.0;
          }
#pragma unroll
          for (int j = 0; j < CW_U; j++) if (p + j < hi) acc = fma(av[j], xv[j], acc);
        }
      } else {
        for (int p = lo; p < hi; p++) { const int sl = cw_slot(p - e0); acc = fma(sa[sl], sx[sl], acc); }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (en < cu.E1) { cur = nxt; e0 = en; continue; }
    }
    // this wave's 64 rows are complete
    if (cu.has) __builtin_nontemporal_store(acc, y + cu.r);
    acc = 0.0;
    g += gstep;
    if (g >= gend) break;
    cu = nx;
    nx = g + gstep < gend ? cw_rows(n, rp, g + gstep, w, lane) : CwRows{0, 0, 0, 0, 0, false};
    e0 = cu.E0 & ~1;
    if (nxt_loaded) cur = nxt;
    else if (e0 < cu.E1) cw_load(cur, col, val, e0, cu.E1, lane);
    nxt_loaded = false;
  }
}

// ---- the same with the col / val streams going STRAIGHT into LDS (global_load_lds_dwordx4: no register staging, no ds_write pass) ----
// Row side only (short rows). A chunk of 512 entries is six LDS-DMA instructions per wave (four for the values: lane l of instruction i brings
// entries 128 i + 2 l, + 1; two for the columns: 256 i + 4 l .. + 3) into a lane-linear image - the DMA's destination is base + lane x 16, so
// the image cannot be skewed; rows whose length is a multiple of 16 would meet on one bank, which is why the register-staged form above stays
// for those (chosen at assembly). The registers the staged form spends on two chunks in flight (48 of its 96) are free here: more waves per
// SIMD take over the latency hiding. A chunk starts on a multiple of four entries (16-byte aligned in both streams; up to three entries of
// the rows before it are fetched and ignored).
template <int CW_STEPS, int WPS, int GU>
__global__ __launch_bounds__(256, WPS) void k_spmv_csr_wave_dma(int n, const int *__restrict__ rp, const int *__restrict__ col, const double *__restrict__ val,
                                                                const double *__restrict__ x, double *__restrict__ y, int xcd_remap)
{
  constexpr int CH = 64 * CW_STEPS;
  __shared__ __attribute__((aligned(16))) double sa_all[4][CH];
  __shared__ __attribute__((aligned(16))) int sc_all[4][CH];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double *sa = sa_all[w];
  int *sc = sc_all[w];
  const int NG = (n + 255) / 256;
  int g, gend, gstep;
  if (xcd_remap) {
    const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3, lc = gridDim.x >> 3;
    g = (int)((long long)NG * xcd / 8) + li; gend = (int)((long long)NG * (xcd + 1) / 8); gstep = lc;
  } else { g = blockIdx.x; gend = NG; gstep = gridDim.x; }
  if (g >= gend) return;
  CwRows cu = cw_rows(n, rp, g, w, lane);
  CwRows nx = g + gstep < gend ? cw_rows(n, rp, g + gstep, w, lane) : CwRows{0, 0, 0, 0, 0, false};
  for (;;) {
    double acc = 0.0;
    for (int e0 = cu.E0 & ~3; e0 < cu.E1; e0 += CH) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // the row lanes' reads of the previous chunk are done before this one may land
#pragma unroll
      for (int i = 0; i < CH / 128; i++) {
        const int e = e0 + 128 * i + 2 * lane;
        if (e < cu.E1) __builtin_amdgcn_global_load_lds((ks_glb_void *)(val + e), (ks_lds_void *)(sa + 128 * i), 16, 0, 2);       // aux 2 = nt: the default policy cost 10 % (profiles/r04_csr_lds_dma.txt)      // may take one entry past E1: CW_PAD
      }
#pragma unroll
      for (int i = 0; i < CH / 256; i++) {
        const int e = e0 + 256 * i + 4 * lane;
        if (e < cu.E1) __builtin_amdgcn_global_load_lds((ks_glb_void *)(col + e), (ks_lds_void *)(sc + 256 * i), 16, 0, 2);      // up to three past E1: CW_PAD
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // an LDS-DMA is a pending LDS write on the VM counter
      const int lo = max(cu.p0, e0), hi = min(cu.p1, e0 + CH);
      for (int p = lo; __builtin_amdgcn_ballot_w64(p < hi) != 0; p += GU) {
        double av[GU], xv[GU];
#pragma unroll
        for (int j = 0; j < GU; j++) {
          const bool ok = p + j < hi;
          const int sl = ok ? p + j - e0 : 0;
          av[j] = sa[sl];
          xv[j] = ok ? x[sc[sl]] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < GU; j++) if (p + j < hi) acc = fma(av[j], xv[j], acc);
      }
    }
    if (cu.has) __builtin_nontemporal_store(acc, y + cu.r);
    g += gstep;
    if (g >= gend) break;
    cu = nx;
    nx = g + gstep < gend ? cw_rows(n, rp, g + gstep, w, lane) : CwRows{0, 0, 0, 0, 0, false};
  }
}

// ---- sliced ELL (SELL-64) ---------------------------------------------------------------------------
// lane <-> row: every val/col load of a wavefront is one contiguous run, the x gather of a
// stencil matrix is contiguous too (consecutive rows -> consecutive columns), y is stored 512 B per wave,
// no cross-lane reduction. Row lengths (int32 per row, the same 4n bytes CSR spends on rowptr) mask the
// padding, so padded slots are never multiplied (no 0*NaN pollution, empty rows give exactly 0).
// Round 4: entries are stored in PAIRS - entries 2q and 2q + 1 of a lane's row side by side, 128 entries per pair index q - so that a lane
// fetches two columns with one 8-byte load and two values with one 16-byte load (a wave: 512 B and 1 KB contiguous): 4-byte-per-lane streaming
// loads top out at 0.7 - 2.4 TB/s on this part (profiles/r03_micro_load_width.txt), and a third of this kernel's load instructions were such.
// A slice of odd width keeps its last entry as a column of singles behind its pairs: same storage as before, same entry order, same fma chain
// - the same bits (asserted against the CSR kernels and the dictionary layouts).
__device__ __forceinline__ long long sell_pos(long long sbase, int w, int j, int lane)
{
  return (j | 1) < w ? sbase + (long long)(j >> 1) * 128 + lane * 2 + (j & 1) : sbase + (long long)(w >> 1) * 128 + lane;
}
template <int UNR>           // pairs in flight per lane
__global__ __launch_bounds__(SPMV_BLOCK) void k_spmv_sell(int nrows, int nslices, const int *__restrict__ sp, const int *__restrict__ rlen,
                                                          const int *__restrict__ col, const double *__restrict__ val,
                                                          const double *__restrict__ x, double *__restrict__ y, int xcd_remap)
{
  const int lane = threadIdx.x & 63;
  const int wpb = SPMV_BLOCK / 64;
  long long nblk = gridDim.x;
  long long b = blockIdx.x;
  if (xcd_remap) {                     // blocks b, b+8, ... share an XCD: give each XCD one contiguous range of slices
    const long long per = nblk / 8;
    if (b < per * 8) b = (b % 8) * per + b / 8;
  }
  const long long nsb = ((long long)nslices + wpb - 1) / wpb;     // slice groups
  for (long long g = b; g < nsb; g += nblk) {
    const long long s = g * wpb + (threadIdx.x >> 6);
    if (s >= nslices) continue;
    const long long r = s * 64 + lane;
    const int w = sp[s + 1] - sp[s], wp = w >> 1;
    const int len = (r < nrows) ? rlen[r] : 0;
    const long long sb = (long long)sp[s] * 64;
    double acc = 0.0;
    for (int q = 0; q < wp; q += UNR) {         // fully predicated batches: all loads of a batch are independent
      ks_i2v c[UNR]; ksk::ks_d2v a[UNR]; double x0[UNR], x1[UNR];
#pragma unroll
      for (int u = 0; u < UNR; u++) {
        const int j = 2 * (q + u);
        const bool ok = q + u < wp && j < len;
        const long long p = sb + (long long)(q + u) * 128 + lane * 2;
        c[u] = ok ? __builtin_nontemporal_load(reinterpret_cast<const ks_i2v *>(col + p)) : ks_i2v{-1, -1};
        a[u] = ok ? __builtin_nontemporal_load(reinterpret_cast<const ksk::ks_d2v *>(val + p)) : ksk::ks_d2v{0.0, 0.0};
        if (j + 1 >= len) { c[u].y = -1; a[u].y = 0.0; }         // the pair's second slot is padding: never gathered, never multiplied
      }
#pragma unroll
      for (int u = 0; u < UNR; u++) { x0[u] = c[u].x >= 0 ? x[c[u].x] : 0.0; x1[u] = c[u].y >= 0 ? x[c[u].y] : 0.0; }
#pragma unroll
      for (int u = 0; u < UNR; u++) { acc = fma(a[u].x, x0[u], acc); acc = fma(a[u].y, x1[u], acc); }
    }
    if (w & 1) {                                 // the slice's last entry slot: singles
      const bool ok = w - 1 < len;
      const long long p = sb + (long long)wp * 128 + lane;
      const int c = ok ? ksk::ldstream(col + p) : -1;
      const double a = ok ? ksk::ldstream(val + p) : 0.0;
      acc = fma(a, c >= 0 ? x[c] : 0.0, acc);
    }
    if (r < nrows) __builtin_nontemporal_store(acc, y + r);
  }
}

// ---- dictionary ELL ----------------------------------------------------------------------------------
// Stencil and graph matrices repeat a handful of values at a handful of column offsets (the 7-point Laplacian: 2 values,
// 7 offsets). When a matrix has at most 255 distinct values (compared bit for bit), at most 256 distinct offsets
// col - row and rows of at most 32 entries, every entry is stored as two bytes (offset code, value code; value code 255
// marks padding): 16 or 32 bytes per row instead of 12 per entry, one 16-byte load per lane. The dictionaries sit in
// LDS. Entries keep their CSR order and the products are accumulated in that order with fma, exactly as k_spmv_sell
// does, so y is bit-identical to the SELL / CSR result.
template <int W>
__global__ __launch_bounds__(SPMV_BLOCK) void k_spmv_dict(int nrows, const uint4 *__restrict__ codes, const double *__restrict__ dval, int nval, const int *__restrict__ doff, int noff,
                                                          const double *__restrict__ x, double *__restrict__ y, int xcd_remap)
{
  __shared__ double sv[256];
  __shared__ int so[256];
  for (int i = threadIdx.x; i < nval; i += SPMV_BLOCK) sv[i] = dval[i];
  for (int i = threadIdx.x; i < noff; i += SPMV_BLOCK) so[i] = doff[i];
  __syncthreads();
  constexpr int Q = W / 8;                                  // uint4 (8 entries) per row
  // Workgroups b, b+8, b+16, ... run on the same XCD (round-robin dispatch). With xcd_remap (grid a multiple of 8) each
  // XCD walks ONE contiguous eighth of the row groups, so that the x entries its rows share (the +-nx, +-nx*ny
  // neighbours of a stencil) are fetched into that XCD's L2 once instead of into all eight.
  const long long groups = ((long long)nrows + SPMV_BLOCK - 1) / SPMV_BLOCK;
  long long g0 = 0, g1 = groups, lb = blockIdx.x, nb = gridDim.x;
  if (xcd_remap) {
    const long long gper = (groups + 7) / 8;
    g0 = (blockIdx.x % 8) * gper; g1 = g0 + gper < groups ? g0 + gper : groups;
    lb = blockIdx.x / 8; nb = gridDim.x / 8;
  }
  for (long long g = g0 + lb; g < g1; g += nb) {
    const long long r = g * SPMV_BLOCK + threadIdx.x;
    if (r >= nrows) break;
    uint4 c[Q];
#pragma unroll
    for (int q = 0; q < Q; q++) c[q] = ksk::ldstream4(codes + r * Q + q);
    double a[W], xv[W];
#pragma unroll
    for (int q = 0; q < Q; q++) {
      const unsigned wds[4] = {c[q].x, c[q].y, c[q].z, c[q].w};
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const unsigned code = (wds[e >> 1] >> ((e & 1) * 16)) & 0xffffu;
        const unsigned oc = code & 0xffu, vc = code >> 8;
        const bool ok = vc != 255u;
        a[q * 8 + e] = ok ? sv[vc] : 0.0;
        xv[q * 8 + e] = ok ? x[r + so[oc]] : 0.0;
      }
    }
    double acc = 0.0;
#pragma unroll
    for (int e = 0; e < W; e++) acc = fma(a[e], xv[e], acc);
    // Nontemporal, as in every product kernel below: a plain store leaves the 80 MB of y dirty in the L2s, and their write-back then mixes into the
    // read streams of the dot sweep that follows (15 us of its 303 on the 216^3 workload: profiles/r02_ab_spmv_store_nt.txt)
    __builtin_nontemporal_store(acc, y + r);
  }
}

// ---- dictionary product fused into the dot sweep that follows it --------------------------------------------------------------------
// A Krylov step is y = A x followed by the dot products of y with the basis columns (and itself). While the basis is resident in the
// Infinity Cache (config 1 and 2: a step is five dependent launches of 5 - 30 us each) the product is worth a launch of its own no longer:
// the dictionary layout is row-local (lane = row), so the dot sweep's tile loop computes its two rows of y itself - same entry order and fma
// chain as k_spmv_dict, same bits -, stores them, and uses them from registers as the vector of the dots and as the last "column". One
// launch, one kernel boundary and one read of y less per step. Not for bases that stream from HBM: the sweep's tiles are interleaved over
// the XCDs, so the stencil's far neighbours (+-nx*ny) would miss the tile's L2 where k_spmv_dict, which gives every XCD one contiguous
// range of rows, hits it.
template <int W>
__device__ __forceinline__ double dict_row(const uint4 *__restrict__ codes, long long r, const double *sv, const int *so, const double *__restrict__ x)
{
  constexpr int Q = W / 8;
  uint4 c[Q];
#pragma unroll
  for (int q = 0; q < Q; q++) c[q] = codes[r * Q + q];
  double a[W], xv[W];
#pragma unroll
  for (int q = 0; q < Q; q++) {
    const unsigned wds[4] = {c[q].x, c[q].y, c[q].z, c[q].w};
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const unsigned code = (wds[e >> 1] >> ((e & 1) * 16)) & 0xffffu;
      const unsigned oc = code & 0xffu, vc = code >> 8;
      const bool ok = vc != 255u;
      a[q * 8 + e] = ok ? sv[vc] : 0.0;
      xv[q * 8 + e] = ok ? x[r + so[oc]] : 0.0;
    }
  }
  double acc = 0.0;
#pragma unroll
  for (int e = 0; e < W; e++) acc = fma(a[e], xv[e], acc);
  return acc;
}
template <int KT, int W>
__global__ __launch_bounds__(ksk::SW_BLOCK) void k_dot_spmv_dict(const double *__restrict__ Vb, long long ld, int n, int ncols, const uint4 *__restrict__ codes,
                                                                 const double *__restrict__ dval, int nval, const int *__restrict__ doff, int noff,
                                                                 const double *__restrict__ x, double *__restrict__ y, double *__restrict__ partials,
                                                                 const KsGsState *__restrict__ gate, int *__restrict__ pgrid, int rev)
{
  using namespace ksk;
  if (gate && !gate->active) return;
  __shared__ double sv[256];
  __shared__ int so[256];
  for (int i = threadIdx.x; i < nval; i += SW_BLOCK) sv[i] = dval[i];
  for (int i = threadIdx.x; i < noff; i += SW_BLOCK) so[i] = doff[i];
  __syncthreads();
  if (pgrid && blockIdx.x == 0 && threadIdx.x == 0) *pgrid = gridDim.x;
  double acc[KT];
#pragma unroll
  for (int i = 0; i < KT; i++) acc[i] = 0.0;
  const int nprev = ncols - 1;                               // basis columns in memory; the last "column" is y itself
  const long long tile = (long long)SW_BLOCK * 2, ntiles = ((long long)n + tile - 1) / tile;
  for (long long t0 = blockIdx.x; t0 < ntiles; t0 += gridDim.x) {
    const long long t = rev ? ntiles - 1 - t0 : t0;
    const long long r = t * tile + (long long)threadIdx.x * 2;
    if (r + 1 < n) {
      double2 xv[KT];
#pragma unroll
      for (int i = 0; i < KT; i++) { const int ii = i < nprev ? i : (nprev > 0 ? nprev - 1 : 0); if (nprev > 0) xv[i] = ldplain2(Vb + (long long)ii * ld + r); else xv[i] = double2{0.0, 0.0}; }
      double2 yv;
      yv.x = dict_row<W>(codes, r, sv, so, x); yv.y = dict_row<W>(codes, r + 1, sv, so, x);
      *reinterpret_cast<double2 *>(y + r) = yv;
#pragma unroll
      for (int i = 0; i < KT; i++) {
        const double2 c = (i == nprev) ? yv : xv[i];
        acc[i] = fma(c.x, yv.x, acc[i]); acc[i] = fma(c.y, yv.y, acc[i]);
      }
    } else if (r < n) {
      const double yv = dict_row<W>(codes, r, sv, so, x);
      y[r] = yv;
#pragma unroll
      for (int i = 0; i < KT; i++) {
        const int ii = i < nprev ? i : (nprev > 0 ? nprev - 1 : 0);
        const double c = (i == nprev) ? yv : (nprev > 0 ? Vb[(long long)ii * ld + r] : 0.0);
        acc[i] = fma(c, yv, acc[i]);
      }
    }
  }
  block_write_partials<KT>(acc, ncols, partials);
}

// Offset-dictionary ELL: the matrix has arbitrary values (variable-coefficient stencils) but still only a few distinct column
// offsets. The index of an entry shrinks from 4 bytes to 1 (code 255 = padding); the values stay full doubles, stored
// slice-column-major like SELL-64 so that a wave reads 512 contiguous bytes per entry slot. 7-point stencil: 80 bytes per row
// instead of 104. Same entry order and fma chain as the other layouts.
template <int W>
__global__ __launch_bounds__(SPMV_BLOCK) void k_spmv_odict(int nrows, const unsigned char *__restrict__ codes, const double *__restrict__ vals, const int *__restrict__ doff, int noff,
                                                           const double *__restrict__ x, double *__restrict__ y, int xcd_remap)
{
  __shared__ int so[256];
  for (int i = threadIdx.x; i < noff; i += SPMV_BLOCK) so[i] = doff[i];
  __syncthreads();
  const long long groups = ((long long)nrows + SPMV_BLOCK - 1) / SPMV_BLOCK;
  long long g0 = 0, g1 = groups, lb = blockIdx.x, nb = gridDim.x;
  if (xcd_remap) {
    const long long gper = (groups + 7) / 8;
    g0 = (blockIdx.x % 8) * gper; g1 = g0 + gper < groups ? g0 + gper : groups;
    lb = blockIdx.x / 8; nb = gridDim.x / 8;
  }
  for (long long g = g0 + lb; g < g1; g += nb) {
    const long long r = g * SPMV_BLOCK + threadIdx.x;
    if (r >= nrows) break;
    unsigned wds[W / 4];
    if (W == 8) { const uint2 c = *reinterpret_cast<const uint2 *>(codes + r * 8); wds[0] = c.x; wds[1] = c.y; }
    else {
#pragma unroll
      for (int q = 0; q < W / 16; q++) {
        const uint4 c = ksk::ldstream4(reinterpret_cast<const uint4 *>(codes + r * W) + q);
        wds[(4 * q) % (W / 4)] = c.x; wds[(4 * q + 1) % (W / 4)] = c.y; wds[(4 * q + 2) % (W / 4)] = c.z; wds[(4 * q + 3) % (W / 4)] = c.w;
      }
    }
    const double *vb = vals + ((r >> 6) * W) * 64 + (r & 63);
    double a[W], xv[W];
#pragma unroll
    for (int e = 0; e < W; e++) {
      const unsigned oc = (wds[e >> 2] >> ((e & 3) * 8)) & 0xffu;
      const bool ok = oc != 255u;
      a[e] = ok ? ksk::ldstream(vb + (long long)e * 64) : 0.0;
      xv[e] = ok ? x[r + so[oc]] : 0.0;
    }
    double acc = 0.0;
#pragma unroll
    for (int e = 0; e < W; e++) acc = fma(a[e], xv[e], acc);
    __builtin_nontemporal_store(acc, y + r);
  }
}

// one thread per row: encode the row's entries against the sorted candidate dictionaries (binary search); entries that
// are not covered are counted and the first `cap` of them recorded so that the host can extend the dictionaries
__global__ void k_dict_encode(int n, int W, const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val,
                              const long long *__restrict__ dbits, int nv, const int *__restrict__ doffs, int no,
                              unsigned short *__restrict__ codes, int *miss, long long *miss_bits, int *miss_off, int cap,
                              unsigned char *__restrict__ codes8, double *__restrict__ vals_out)
{
  // nv < 0: offsets-only mode (values kept in full): 1-byte codes into codes8, values into vals_out in slice-column-major order
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int p0 = rowptr[r], len = rowptr[r + 1] - p0;
  for (int j = 0; j < W; j++) {
    unsigned short code = 0xff00u;                                       // padding
    if (nv < 0) {
      unsigned char c8 = 255; double v = 0.0;
      if (j < len) {
        const int off = col[p0 + j] - (int)r;
        int lo = 0, hi = no; while (lo < hi) { const int m = (lo + hi) >> 1; if (doffs[m] < off) lo = m + 1; else hi = m; }
        if (lo < no && doffs[lo] == off) c8 = (unsigned char)lo;
        else { if (*(volatile int *)miss < cap) { const int idx = atomicAdd(miss, 1); if (idx < cap) { miss_bits[idx] = 0; miss_off[idx] = off; } } else atomicAdd(miss + 1, 1); c8 = 0; }
        v = val[p0 + j];
      }
      codes8[r * W + j] = c8;
      vals_out[((r >> 6) * W + j) * 64 + (r & 63)] = v;
      continue;
    }
    if (j < len) {
      const long long bits = __double_as_longlong(val[p0 + j]);
      const int off = col[p0 + j] - (int)r;
      int lo = 0, hi = nv; while (lo < hi) { const int m = (lo + hi) >> 1; if (dbits[m] < bits) lo = m + 1; else hi = m; }
      const int vi = (lo < nv && dbits[lo] == bits) ? lo : -1;
      lo = 0; hi = no; while (lo < hi) { const int m = (lo + hi) >> 1; if (doffs[m] < off) lo = m + 1; else hi = m; }
      const int oi = (lo < no && doffs[lo] == off) ? lo : -1;
      if (vi < 0 || oi < 0) {
        if (*(volatile int *)miss < cap) { const int idx = atomicAdd(miss, 1); if (idx < cap) { miss_bits[idx] = bits; miss_off[idx] = off; } }
        else atomicAdd(miss + 1, 1);
        code = 0;
      } else code = (unsigned short)((vi << 8) | oi);
    }
    codes[r * W + j] = code;
  }
}
__global__ void k_max_rowlen(int n, const int *__restrict__ rowptr, int *out)
{
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  int len = (r < n) ? rowptr[r + 1] - rowptr[r] : 0;
  for (int off = 32; off > 0; off >>= 1) len = max(len, __shfl_xor(len, off, 64));
  if ((threadIdx.x & 63) == 0 && len > 0) atomicMax(out, len);
}

__global__ void k_sell_widths(int n, int nslices, const int *__restrict__ rowptr, int *__restrict__ width, int *__restrict__ rlen)
{
  const long long s = (long long)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
  if (s >= nslices) return;
  const int lane = threadIdx.x & 63;
  const long long r = s * 64 + lane;
  int len = (r < n) ? rowptr[r + 1] - rowptr[r] : 0;
  if (r < n) rlen[r] = len;
  for (int off = 32; off > 0; off >>= 1) len = max(len, __shfl_xor(len, off, 64));
  if (lane == 0) width[s] = len;
}
__global__ void k_sell_fill(int n, int nslices, const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val,
                            const int *__restrict__ sp, int *__restrict__ scol, double *__restrict__ sval)
{
  const long long s = (long long)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
  if (s >= nslices) return;
  const int lane = threadIdx.x & 63;
  const long long r = s * 64 + lane;
  const int w = sp[s + 1] - sp[s];
  const int p0 = (r < n) ? rowptr[r] : 0, len = (r < n) ? rowptr[r + 1] - p0 : 0;
  const long long sb = (long long)sp[s] * 64;
  for (int j = 0; j < w; j++) {
    const long long p = sell_pos(sb, w, j, lane);
    scol[p] = (j < len) ? col[p0 + j] : 0;
    sval[p] = (j < len) ? val[p0 + j] : 0.0;
  }
}

template <int G, bool ACCUM, bool ROWLIST>
void launch_spmv_g(hipStream_t st, int num_cu, int nrows, const int *rowptr, const int *col, const double *val, const double *x, double *y, const int *rowlist)
{
  constexpr int UNROLL = 4;
  constexpr int RW = (64 / G) * UNROLL;                        // rows per wavefront per sweep
  long long waves = ((long long)nrows + RW - 1) / RW;
  long long blocks = (waves * 64 + SPMV_BLOCK - 1) / SPMV_BLOCK;
  long long maxb = (long long)num_cu * 32;
  if (blocks > maxb) blocks = maxb;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL((k_spmv_csr<G, UNROLL, ACCUM, ROWLIST>), dim3((unsigned)blocks), dim3(SPMV_BLOCK), 0, st, nrows, rowptr, col, val, x, y, rowlist);
}

template <bool ACCUM, bool ROWLIST>
void launch_spmv(hipStream_t st, int num_cu, int lanes, int nrows, const int *rowptr, const int *col, const double *val, const double *x, double *y, const int *rowlist)
{
  switch (lanes) {
    case 2:  launch_spmv_g<2, ACCUM, ROWLIST>(st, num_cu, nrows, rowptr, col, val, x, y, rowlist); break;
    case 4:  launch_spmv_g<4, ACCUM, ROWLIST>(st, num_cu, nrows, rowptr, col, val, x, y, rowlist); break;
    case 8:  launch_spmv_g<8, ACCUM, ROWLIST>(st, num_cu, nrows, rowptr, col, val, x, y, rowlist); break;
    case 16: launch_spmv_g<16, ACCUM, ROWLIST>(st, num_cu, nrows, rowptr, col, val, x, y, rowlist); break;
    case 32: launch_spmv_g<32, ACCUM, ROWLIST>(st, num_cu, nrows, rowptr, col, val, x, y, rowlist); break;
    default: launch_spmv_g<64, ACCUM, ROWLIST>(st, num_cu, nrows, rowptr, col, val, x, y, rowlist); break;
  }
}

int pick_lanes(long long nnz, int n)
{
  double mean = n > 0 ? (double)nnz / n : 1.0;
  int g = 2;
  while (g < 64 && g < mean) g <<= 1;      // smallest power of two >= mean row length
  return g;
}

__global__ void k_pack(int n, const int *__restrict__ idx, const double *__restrict__ x, double *__restrict__ out)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = x[idx[i]];
}

// ---- synthetic generators, built directly in device memory -------------------------------------
// 3-D 7-point Laplacian, ex19.c:47-78: diag 6, off -1, natural ordering (x fastest), Dirichlet.
// Local rows = planes [z0,z0+nzl). Entries whose column is owned by another slab go to the
// off-diagonal block with ghost index: lower plane -> [0,plane), upper plane -> [nlow, nlow+plane).
__global__ void k_lap3d_count(int nx, int ny, int nz, int z0, int nzl, int *cnt_d, int *cnt_o)
{
  long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long n = (long long)nx * ny * nzl;
  if (r >= n) return;
  int i = (int)(r % nx); long long t = r / nx; int j = (int)(t % ny); int kl = (int)(t / ny); int k = z0 + kl;
  int cd = 1, co = 0;
  if (k > 0) { if (kl > 0) cd++; else co++; }
  if (j > 0) cd++;
  if (i > 0) cd++;
  if (i < nx - 1) cd++;
  if (j < ny - 1) cd++;
  if (k < nz - 1) { if (kl < nzl - 1) cd++; else co++; }
  cnt_d[r] = cd; cnt_o[r] = co;
}

__global__ void k_lap3d_fill(int nx, int ny, int nz, int z0, int nzl, const int *rp_d, int *col_d, double *val_d,
                             const int *rp_o, int *col_o, double *val_o)
{
  long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long plane = (long long)nx * ny, n = plane * nzl;
  if (r >= n) return;
  int i = (int)(r % nx); long long t = r / nx; int j = (int)(t % ny); int kl = (int)(t / ny); int k = z0 + kl;
  int p = rp_d[r], q = rp_o[r];
  const int nlow = (z0 > 0) ? (int)plane : 0;
  if (k > 0) { if (kl > 0) { col_d[p] = (int)(r - plane); val_d[p++] = -1.0; } else { col_o[q] = (int)(r); val_o[q++] = -1.0; } }
  if (j > 0) { col_d[p] = (int)(r - nx); val_d[p++] = -1.0; }
  if (i > 0) { col_d[p] = (int)(r - 1); val_d[p++] = -1.0; }
  col_d[p] = (int)r; val_d[p++] = 6.0;
  if (i < nx - 1) { col_d[p] = (int)(r + 1); val_d[p++] = -1.0; }
  if (j < ny - 1) { col_d[p] = (int)(r + nx); val_d[p++] = -1.0; }
  if (k < nz - 1) { if (kl < nzl - 1) { col_d[p] = (int)(r + plane); val_d[p++] = -1.0; } else { col_o[q] = nlow + (int)(r - (n - plane)); val_o[q++] = -1.0; } }
}

// 2-D 5-point Laplacian, ex2.c:44-51 (single slab): diag 4, off -1, II=i*n+j
__global__ void k_lap2d_count(int n, int m, int *cnt)
{
  long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= (long long)n * m) return;
  int i = (int)(r / n), j = (int)(r % n);
  cnt[r] = 1 + (i > 0) + (i < m - 1) + (j > 0) + (j < n - 1);
}
__global__ void k_lap2d_fill(int n, int m, const int *rp, int *col, double *val)
{
  long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= (long long)n * m) return;
  int i = (int)(r / n), j = (int)(r % n);
  int p = rp[r];
  if (i > 0) { col[p] = (int)(r - n); val[p++] = -1.0; }
  if (j > 0) { col[p] = (int)(r - 1); val[p++] = -1.0; }
  col[p] = (int)r; val[p++] = 4.0;
  if (j < n - 1) { col[p] = (int)(r + 1); val[p++] = -1.0; }
  if (i < m - 1) { col[p] = (int)(r + n); val[p++] = -1.0; }
}

__global__ void k_rows_with_entries(int n, const int *rp, int *flag)
{
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) flag[r] = (rp[r + 1] > rp[r]) ? 1 : 0;
}
__global__ void k_compact_rows(int n, const int *rp, const int *pos, int *rows, int *rp_c)
{
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n && rp[r + 1] > rp[r]) { rows[pos[r]] = r; rp_c[pos[r]] = rp[r]; }
}

// Exclusive prefix sum of ints (assembly only: row pointers from row counts). Three launches: sums of 2048-item tiles, an exclusive scan
// of the tile sums by one workgroup, then every tile scans itself (wave shuffles, 8 items per thread) starting from its offset.
constexpr int SCAN_TILE = 2048;
__device__ __forceinline__ int scan_wave_incl(int v)
{
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(v, d, 64); if ((int)(threadIdx.x & 63) >= d) v += t; }
  return v;
}
__global__ __launch_bounds__(256) void k_scan_tile_sums(const int *__restrict__ in, long long n, int *__restrict__ sums)
{
  __shared__ int ws[4];
  const long long base = (long long)blockIdx.x * SCAN_TILE;
  int s = 0;
  for (int i = threadIdx.x; i < SCAN_TILE; i += 256) { const long long g = base + i; if (g < n) s += in[g]; }
  for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ __launch_bounds__(1024) void k_scan_sums(int *__restrict__ sums, int ntiles)
{
  __shared__ int ws[16];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int b0 = 0; b0 < ntiles; b0 += 1024) {
    const int i = b0 + threadIdx.x;
    const int v = i < ntiles ? sums[i] : 0;
    int inc = scan_wave_incl(v);
    if ((threadIdx.x & 63) == 63) ws[threadIdx.x >> 6] = inc;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) woff += ws[w];
    const int c = carry;
    if (i < ntiles) sums[i] = c + woff + inc - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = c + woff + inc;
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void k_scan_tiles(const int *__restrict__ in, int *__restrict__ out, long long n, const int *__restrict__ offs)
{
  __shared__ int ws[4];
  const long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * 8;
  int v[8], t = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) { const long long g = base + j; v[j] = g < n ? in[g] : 0; t += v[j]; }
  const int inc = scan_wave_incl(t);
  if ((threadIdx.x & 63) == 63) ws[threadIdx.x >> 6] = inc;
  __syncthreads();
  int run = offs[blockIdx.x] + inc - t;
  for (int w = 0; w < (int)(threadIdx.x >> 6); w++) run += ws[w];
#pragma unroll
  for (int j = 0; j < 8; j++) { const long long g = base + j; if (g < n) out[g] = run; run += v[j]; }
}
int exclusive_scan_int(hipStream_t st, const int *in, int *out, long long nitems)
{
  if (nitems <= 0) return KS_SUCCESS;
  const int ntiles = (int)((nitems + SCAN_TILE - 1) / SCAN_TILE);
  int *sums = nullptr;
  KS_HIP(hipMalloc(&sums, sizeof(int) * (size_t)ntiles));
  hipLaunchKernelGGL(k_scan_tile_sums, dim3(ntiles), dim3(256), 0, st, in, nitems, sums);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, st, sums, ntiles);
  hipLaunchKernelGGL(k_scan_tiles, dim3(ntiles), dim3(256), 0, st, in, out, nitems, sums);
  hipError_t e = hipGetLastError();
  hipStreamSynchronize(st);
  hipFree(sums);
  KS_HIP(e);
  return KS_SUCCESS;
}

// Build the halo plan from the sorted list of needed global columns (garray, host).
// Owners are found from the allgathered row_start array; every rank tells its owners which rows it needs.
int build_halo_plan(ks_mat A, const std::vector<int> &garray)
{
  ks_ctx ctx = A->ctx;
  const int size = ctx->comm.size, rank = ctx->comm.rank;
  A->nghost = (int)garray.size();
  if (size == 1) { KS_CHECK(garray.empty(), KS_ERR_ARG_OUTOFRANGE, "column index outside [0,n) on a single rank"); return KS_SUCCESS; }
  // 1. ownership ranges
  std::vector<int> starts(size + 1);
  KS_CALL(ks_comm_allgather_host(ctx, &A->row_start, sizeof(int), starts.data()));
  starts[size] = A->n_global;
  for (int p = 0; p < size; p++) KS_CHECK(starts[p] <= starts[p + 1], KS_ERR_ARG_WRONG, "row blocks must be contiguous and ordered by rank");
  // 2. how many entries I need from every owner; 3. everybody learns everybody's needs
  std::vector<int> recv_cnt(size, 0), all_cnt((size_t)size * size, 0), send_cnt(size, 0);
  { int p = 0; for (int g : garray) { while (p + 1 < size && g >= starts[p + 1]) p++; KS_CHECK(p != rank, KS_ERR_PLIB, "ghost column owned by self"); recv_cnt[p]++; } }
  KS_CALL(ks_comm_allgather_host(ctx, recv_cnt.data(), (int)(sizeof(int) * size), all_cnt.data()));
  for (int p = 0; p < size; p++) send_cnt[p] = (p == rank) ? 0 : all_cnt[(size_t)p * size + rank];
  // 4. exchange index lists: I send my garray segments (global ids) to their owners and receive the ids I must serve
  int nsend = 0; for (int p = 0; p < size; p++) nsend += send_cnt[p];
  int *d_g = nullptr, *d_sidx = nullptr;
  KS_HIP(hipMalloc(&d_g, sizeof(int) * std::max<size_t>(garray.size(), 1)));
  KS_HIP(hipMalloc(&d_sidx, sizeof(int) * std::max(nsend, 1)));
  KS_HIP(hipMemcpyAsync(d_g, garray.data(), sizeof(int) * garray.size(), hipMemcpyHostToDevice, ctx->stream));
  A->peers.clear(); A->send_cnt.clear(); A->recv_cnt.clear(); A->send_off.clear(); A->recv_off.clear();
  int roff = 0, soff = 0;
  for (int p = 0; p < size; p++) {
    if (p == rank || (recv_cnt[p] == 0 && send_cnt[p] == 0)) continue;
    A->peers.push_back(p); A->recv_cnt.push_back(recv_cnt[p]); A->send_cnt.push_back(send_cnt[p]); A->recv_off.push_back(roff); A->send_off.push_back(soff);
    roff += recv_cnt[p]; soff += send_cnt[p];
  }
  // in this exchange the roles are swapped: what I will RECEIVE during SpMV (ghost segments) is what I SEND now (their ids)
  KS_CALL(ks_comm_exchange(ctx, (int)A->peers.size(), A->peers.data(), d_g, A->recv_off.data(), A->recv_cnt.data(),
                           d_sidx, A->send_off.data(), A->send_cnt.data(), (int)sizeof(int)));
  std::vector<int> sidx(std::max(nsend, 1));
  KS_HIP(hipMemcpyAsync(sidx.data(), d_sidx, sizeof(int) * nsend, hipMemcpyDeviceToHost, ctx->stream));
  KS_HIP(ks_sync(ctx));
  for (int i = 0; i < nsend; i++) { sidx[i] -= A->row_start; KS_CHECK(sidx[i] >= 0 && sidx[i] < A->n, KS_ERR_PLIB, "peer requested a row this rank does not own"); }
  KS_HIP(hipMemcpy(d_sidx, sidx.data(), sizeof(int) * nsend, hipMemcpyHostToDevice));
  A->send_idx = d_sidx; A->nsend = nsend;
  KS_HIP(hipMalloc(&A->send_buf, sizeof(double) * std::max(nsend, 1)));
  KS_HIP(hipMalloc(&A->ghost, sizeof(double) * std::max(A->nghost, 1)));
  hipFree(d_g);
  return KS_SUCCESS;
}

int compact_offdiag_rows(ks_mat A)
{
  // rows with off-diagonal entries -> compressed row list + compressed rowptr (PETSc "compressed row" AIJ)
  ks_ctx ctx = A->ctx;
  if (A->nnz_o == 0) { A->n_orows = 0; return KS_SUCCESS; }
  int *flag = nullptr, *pos = nullptr;
  KS_HIP(hipMalloc(&flag, sizeof(int) * (A->n + 1))); KS_HIP(hipMalloc(&pos, sizeof(int) * (A->n + 1)));
  KS_HIP(hipMemsetAsync(flag, 0, sizeof(int) * (A->n + 1), ctx->stream));
  hipLaunchKernelGGL(k_rows_with_entries, dim3((A->n + 255) / 256), dim3(256), 0, ctx->stream, A->n, A->o_rowptr, flag);
  KS_CALL(exclusive_scan_int(ctx->stream, flag, pos, A->n + 1));
  int norows = 0; KS_HIP(hipMemcpy(&norows, pos + A->n, sizeof(int), hipMemcpyDeviceToHost));
  int *rows = nullptr, *rp_c = nullptr;
  KS_HIP(hipMalloc(&rows, sizeof(int) * std::max(norows, 1))); KS_HIP(hipMalloc(&rp_c, sizeof(int) * (norows + 1)));
  hipLaunchKernelGGL(k_compact_rows, dim3((A->n + 255) / 256), dim3(256), 0, ctx->stream, A->n, A->o_rowptr, pos, rows, rp_c);
  int last = (int)A->nnz_o; KS_HIP(hipMemcpyAsync(rp_c + norows, &last, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  KS_HIP(ks_sync(ctx));
  hipFree(flag); hipFree(pos); hipFree(A->o_rowptr);
  A->o_rowptr = rp_c; A->o_rows = rows; A->n_orows = norows;
  return KS_SUCCESS;
}

// Build the SELL-64 copy of the diagonal block when its padding is small (<= 12.5 % extra entries).
// ---- XCD-sliced layout ---------------------------------------------------------------------------------------------
// Measured on MI355X (scripts/micro/gather_xcd.hip): 1.6e8 random 8-byte gathers from a 40 MB vector take 2.83 ms when
// every XCD gathers from all of it (each one a 128-B line from the Infinity Cache) and 1.23 ms when the workgroups of
// XCD i (blockIdx % 8 == i) only touch the i-th eighth (L2 hits).
__global__ void k_slice_count(int n, int nslice, int slice_cols, const int *__restrict__ rp, const int *__restrict__ col, int *__restrict__ cnt)
{
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r > n) return;
  for (int s = 0; s < nslice; s++) cnt[(size_t)s * (n + 1) + r] = 0;
  if (r == n) return;
  for (int p = rp[r]; p < rp[r + 1]; p++) cnt[(size_t)(col[p] / slice_cols) * (n + 1) + r]++;
}
__global__ void k_slice_fill(int n, int nslice, int slice_cols, const int *__restrict__ rp, const int *__restrict__ col, const double *__restrict__ val,
                             const int *__restrict__ srp, const long long *__restrict__ base, int *__restrict__ scol, double *__restrict__ sval)
{
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  for (int s = 0; s < nslice; s++) {                    // stable: entries keep their order inside (row, slice)
    long long pos = base[s] + srp[(size_t)s * (n + 1) + r];
    for (int p = rp[r]; p < rp[r + 1]; p++) if (col[p] / slice_cols == s) { scol[pos] = col[p]; sval[pos] = val[p]; pos++; }
  }
}
__global__ void k_far_entries(int n, int far, const int *__restrict__ rp, const int *__restrict__ col, unsigned long long *__restrict__ count)
{
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long c = 0;
  if (r < n) for (int p = rp[r]; p < rp[r + 1]; p++) { const long long d = (long long)col[p] - r; if (d > far || d < -far) c++; }
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}
// Partial y of XCD x = blockIdx % 8 over its slices. A workgroup takes 256 consecutive rows; their entries are one
// contiguous run of the slice's arrays, which the workgroup streams in chunks of 1024 with fully coalesced, nontemporal
// loads (every lane has 4 independent gathers in flight), parks the products in LDS, and then every row (= thread) adds up
// its own segment in entry order.
constexpr int SL_EPT = 4;
__global__ __launch_bounds__(256) void k_spmv_sliced(int n, int nslice, const int *__restrict__ srp, const long long *__restrict__ base,
                                                     const int *__restrict__ col, const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ ypart)
{
  __shared__ double prod[256 * SL_EPT];
  __shared__ int erange[2];
  const int xcd = blockIdx.x & 7, tid = threadIdx.x;
  const long long bx = blockIdx.x >> 3, nbx = gridDim.x >> 3;
  double *yp = ypart + (size_t)xcd * n;
  for (int s = xcd; s < nslice; s += 8) {               // one column range at a time, so that it stays in this XCD's L2
    const int *rp = srp + (size_t)s * (n + 1);
    const int *cs = col + base[s];
    const double *vs = val + base[s];
    for (long long R0 = bx * 256; R0 < n; R0 += nbx * 256) {
      const long long r = R0 + tid;
      const bool has = r < n;
      const int p0 = has ? ksk::ldstream(rp + r) : 0, p1 = has ? ksk::ldstream(rp + r + 1) : 0;
      if (tid == 0) erange[0] = p0;
      if (has && (r == n - 1 || tid == 255)) erange[1] = p1;
      __syncthreads();
      const int E0 = erange[0], E1 = erange[1];
      double acc = (s == xcd || !has) ? 0.0 : yp[r];
      for (int e0 = E0; e0 < E1; e0 += 256 * SL_EPT) {
        int c[SL_EPT]; double a[SL_EPT];
#pragma unroll
        for (int u = 0; u < SL_EPT; u++) { const int e = e0 + u * 256 + tid; const bool ok = e < E1; c[u] = ok ? ksk::ldstream(cs + e) : -1; a[u] = ok ? ksk::ldstream(vs + e) : 0.0; }
#pragma unroll
        for (int u = 0; u < SL_EPT; u++) prod[u * 256 + tid] = c[u] >= 0 ? a[u] * x[c[u]] : 0.0;
        __syncthreads();
        const int lo = max(p0, e0), hi = min(p1, e0 + 256 * SL_EPT);
        for (int p = lo; p < hi; p++) acc += prod[p - e0];
        __syncthreads();
      }
      if (has) __builtin_nontemporal_store(acc, yp + r);
    }
  }
}
__global__ void k_sum_parts(int n, const double *__restrict__ ypart, double *__restrict__ y)
{
  for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (long long)gridDim.x * blockDim.x) {
    double s = ypart[r];
#pragma unroll
    for (int x = 1; x < 8; x++) s += ypart[(size_t)x * n + r];
    __builtin_nontemporal_store(s, y + r);
  }
}


// ---- binned product (ksgpu_internal.h: use_binned) ---------------------------------------------------------------------------------
// Why: the product of a uniformly random matrix in a row-ordered layout pulls a 128-byte line of x through L2 -> L1 for every nonzero and
// runs at that line rate (1.3 ms for config 5), however x is cut for the L2s. Here no random access leaves the CU: phase 1
// gathers from a piece of x in LDS and streams the gathered values out in the order phase 2 wants them; phase 2 streams them back in with
// the values and adds into rows of y in LDS. 28 bytes per nonzero of pure streams instead of 12 bytes + a line (profiles/r02_micro_binned_spmv*).
constexpr int BN_MAXSEG = 12;            // segments a 1024-entry window may touch on the fast path
constexpr int BN_CS_MAX = 9984;          // columns of a slice: 78 KB of LDS next to the two offset rows
constexpr int BN_SEG_PAD = 8;            // a (slice, wave-bin) segment holds a multiple of 8 entries (padding: value 0 into the spare accumulator): every segment then
                                         // starts on a 64-byte boundary of G / the values in both orders. Against padding to pairs only, same box: gather 385 -> 343-357 us,
                                         // reduce 511 -> 493 us with 2.5 % more entries (profiles/r03_ab_binned.txt); 4: 370-377 / 508, 16: 346-362 / 497
// phase 1: grid = slices, 1024 threads; LDS: x piece [cs], off1 row [wb + 1], off2t row [wb] (+ DMA: 2 KB per wave for the window's column codes)
// DMA (round 4): the window's 1024 column codes come by two global_load_lds_dwordx4 per wave (16 bytes per lane) into a wave-private piece of LDS and the
// lanes read their pairs from there. The register form (DMA = false, kept for reference) loads a pair per lane and instruction - 4 bytes per lane, and
// 4-byte-per-lane streaming loads top out at 0.7 - 2.4 TB/s on this part (profiles/r03_micro_load_width.txt). Worth 2 - 8 % of this kernel depending on the
// box (334-350 against 363-365 us; 336-341 against 343-351): it stays bound by its store. A second buffer with the next window's codes in flight (counted
// vmcnt behind the window's eight stores) measured no better than the register form: profiles/r04_binned_gather_dma.txt.
// Slices start on multiples of 8 entries (every segment is padded to 8): the 16-byte DMA is aligned.
template <bool DMA>
__global__ __launch_bounds__(1024) void k_binned_gather(int n, int cs, int wb, int nwin, const long long *__restrict__ sbase, const unsigned short *__restrict__ col16,
                                                        const int *__restrict__ off1, const int *__restrict__ off2t, const int *__restrict__ wseg,
                                                        const double *__restrict__ x, double *__restrict__ G)
{
  extern __shared__ double bn_lds[];
  double *xs = bn_lds;
  int *o1 = (int *)(bn_lds + cs);
  int *o2 = o1 + wb + 1;
  const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = blockDim.x >> 6;
  const long long c0 = (long long)s * cs;
  for (int i = tid; i < cs; i += blockDim.x) xs[i] = (c0 + i < n) ? x[c0 + i] : 0.0;
  for (int i = tid; i <= wb; i += blockDim.x) o1[i] = off1[(size_t)s * (wb + 1) + i];
  for (int i = tid; i < wb; i += blockDim.x) o2[i] = off2t[(size_t)s * wb + i];
  __syncthreads();
  const unsigned *cp = reinterpret_cast<const unsigned *>(col16 + sbase[s]);     // slices start at even positions: 4-byte aligned
  const unsigned short *cg = col16 + sbase[s];
  unsigned *cw = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(bn_lds) + ((((size_t)cs * 8 + (size_t)(2 * wb + 1) * 4) + 15) & ~(size_t)15)) + (size_t)w * 512;
  const int total = o1[wb];                                                       // a multiple of 8
  for (int win = w; win * 1024 < total; win += nw) {
    const int base = win * 1024;
    unsigned c[8];
    if (DMA) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // the reads of the previous window's codes are done before the next ones may land
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const int e8 = base + 512 * i + 8 * lane;
        if (e8 < total) __builtin_amdgcn_global_load_lds((ks_glb_void *)(cg + e8), (ks_lds_void *)(cw + 256 * i), 16, 0, 2);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < 8; k++) { const int e = base + k * 128 + 2 * lane; c[k] = e < total ? cw[k * 64 + lane] : 0u; }
    } else {
#pragma unroll
    for (int k = 0; k < 8; k++) { const int e = base + k * 128 + 2 * lane; c[k] = e < total ? __builtin_nontemporal_load(cp + (e >> 1)) : 0u; }
    }
    const int lo = wseg[(size_t)s * nwin + win];
    int bnd[BN_MAXSEG], dlt[BN_MAXSEG];
#pragma unroll
    for (int j = 0; j < BN_MAXSEG; j++) { const int sg = min(lo + j, wb - 1); bnd[j] = o1[sg + 1]; dlt[j] = o2[sg] - o1[sg]; }
    const bool fits = bnd[BN_MAXSEG - 1] >= min(base + 1024, total);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int e = base + k * 128 + 2 * lane;
      if (e < total) {
        int d;
        if (fits) {
          d = dlt[0];
#pragma unroll
          for (int j = 1; j < BN_MAXSEG; j++) d = (e >= bnd[j - 1]) ? dlt[j] : d;
        } else { int sg = lo; while (e >= o1[sg + 1]) sg++; d = o2[sg] - o1[sg]; }      // many short segments: look each pair up
        const ksk::ks_d2v g = {xs[c[k] & 0xffffu], xs[c[k] >> 16]};
        __builtin_nontemporal_store(g, reinterpret_cast<ksk::ks_d2v *>(G + ((long long)e + d)));   // pairs never straddle a segment: both orders keep segments even
      }
    }
  }
}
// phase 2: grid = wave-bins / 4, 256 threads: a wave per wave-bin; LDS: 4 x (wr + 1) accumulators (the last one takes the padding entries).
// The wave-bin's entries are ns pieces (one per slice, about 157 entries each, interleaved with the pieces of the workgroup's other three
// waves); the wave takes them in slice order, THREE pieces at a time, each as up to four 64-entry steps (masked beyond the piece's end), so
// that 12 steps of (G, value, row) loads are in flight before the first add. Where a piece is and how long comes from the wave-bin's rows
// of two small tables (wave-uniform: scalar loads) - no per-entry search. (A walk over 512-entry logical windows with a compare chain per
// entry, the mirror of phase 1, cost 70 us more than the contiguous stream it replaced; this form: profiles/r03_ab_binned.txt.)
constexpr int BN_P2 = 3;
__global__ __launch_bounds__(256) void k_binned_reduce(int n, int wr, int ns, const int *__restrict__ log2, const int *__restrict__ off2,
                                                       const double *__restrict__ G, const double *__restrict__ val,
                                                       const unsigned short *__restrict__ row16, double *__restrict__ y, const double *__restrict__ rowscale)
{
  extern __shared__ double bn_lds[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x * 4 + w;
  double *acc = bn_lds + (size_t)w * (wr + 1);
  for (int i = lane; i <= wr; i += 64) acc[i] = 0.0;
  const int *lg = log2 + (size_t)b * (ns + 1), *ph = off2 + (size_t)b * ns;
  for (int s0 = 0; s0 < ns; s0 += BN_P2) {
    int pb[BN_P2], pl[BN_P2], maxl = 0;
#pragma unroll
    for (int j = 0; j < BN_P2; j++) {
      const int sg = s0 + j < ns ? s0 + j : ns - 1;
      pb[j] = ph[sg]; pl[j] = s0 + j < ns ? lg[sg + 1] - lg[sg] : 0;
      maxl = max(maxl, pl[j]);
    }
    for (int o = 0; o < maxl; o += 256) {                   // one round unless a piece is longer than 256 entries
      double g[BN_P2][4], a[BN_P2][4]; unsigned short r[BN_P2][4];
#pragma unroll
      for (int j = 0; j < BN_P2; j++)
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int e = o + k * 64 + lane; const bool ok = e < pl[j];
          const long long p = (long long)pb[j] + e;
          g[j][k] = ok ? __builtin_nontemporal_load(G + p) : 0.0; a[j][k] = ok ? __builtin_nontemporal_load(val + p) : 0.0; r[j][k] = ok ? __builtin_nontemporal_load(row16 + p) : (unsigned short)wr;
        }
      // the adds of one instruction go lane by lane, instructions in program order (piece after piece, entry after entry): a fixed order per
      // row, run after run
#pragma unroll
      for (int j = 0; j < BN_P2; j++)
#pragma unroll
        for (int k = 0; k < 4; k++) if (o + k * 64 + lane < pl[j]) __hip_atomic_fetch_add(acc + r[j][k], a[j][k] * g[j][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  for (int i = lane; i < wr; i += 64) { const long long row = (long long)b * wr + i; if (row < n) __builtin_nontemporal_store(rowscale ? rowscale[row] * acc[i] : acc[i], y + row); }
}


// Host-side build of the binned layout from the device CSR of the diagonal block (copied back once; the counting sort per wave-bin is cache
// friendly and runs on a few threads).
static int build_binned(ks_mat A)
{
  ks_ctx ctx = A->ctx;
  const char *force = getenv("KSGPU_SPMV");
  if (force && strcmp(force, "binned")) return KS_SUCCESS;
  const int n = A->n;
  if (n < 4096 || A->nnz_d == 0) return KS_SUCCESS;
  if (!force) {
    // the same matrices the XCD-sliced layout was built for: x well beyond an L2 and most entries far from the diagonal
    if ((double)n * 8.0 < 6.0 * 1048576.0 || A->nnz_d < 8LL * n) return KS_SUCCESS;
    unsigned long long *cnt = nullptr, h = 0;
    KS_HIP(hipMalloc(&cnt, sizeof(unsigned long long))); KS_HIP(hipMemsetAsync(cnt, 0, sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(k_far_entries, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, n / 16, A->d_rowptr, A->d_col, cnt);
    KS_HIP(hipMemcpyAsync(&h, cnt, sizeof(h), hipMemcpyDeviceToHost, ctx->stream)); KS_HIP(ks_sync(ctx)); hipFree(cnt);
    if ((double)h < 0.5 * (double)A->nnz_d) return KS_SUCCESS;
  }
  // slices: a multiple of the CU count (a workgroup per slice, one resident per CU), each at most BN_CS_MAX columns
  const int ncu = std::max(ctx->num_cu, 1);
  const long long per_round = (long long)ncu * BN_CS_MAX;
  const int ns = (int)(ncu * ((n + per_round - 1) / per_round));
  const int cs = (n + ns - 1) / ns;
  const int wb = 4 * ns, wr = (n + wb - 1) / wb;
  if (cs > 65535 || wr + 1 > 65535) return KS_SUCCESS;
  if ((size_t)cs * 8 + (size_t)(2 * wb + 1) * 4 + 16 + 16 * 2048 > 156 * 1024 || (size_t)4 * (wr + 1) * 8 > 156 * 1024) return KS_SUCCESS;   // the offset rows of more than ~20 M local rows no longer fit LDS next to the piece of x: the XCD-sliced layout takes those
  if (A->nnz_d + (long long)ns * wb * (BN_SEG_PAD - 1) >= 2147483647LL) return KS_SUCCESS;          // bin-major positions are 32-bit
  const long long nnz = A->nnz_d;
  try {                                               // the build holds about 25 bytes per nonzero in host memory: without it the sliced layout takes the matrix
  std::vector<int> rp(n + 1), col(nnz); std::vector<double> val(nnz);
  KS_HIP(hipMemcpyAsync(rp.data(), A->d_rowptr, sizeof(int) * (n + 1), hipMemcpyDeviceToHost, ctx->stream));
  KS_HIP(hipMemcpyAsync(col.data(), A->d_col, sizeof(int) * nnz, hipMemcpyDeviceToHost, ctx->stream));
  KS_HIP(hipMemcpyAsync(val.data(), A->d_val, sizeof(double) * nnz, hipMemcpyDeviceToHost, ctx->stream));
  KS_HIP(ks_sync(ctx));
  // helper threads: the CPUs this process may run on (affinity mask: what a cgroup / taskset leaves), at most 16; a thread that cannot
  // be started (pids limit) is simply not used - the calling thread takes every stride that has no thread of its own
  unsigned ncpu = std::thread::hardware_concurrency();
  { cpu_set_t cs_; CPU_ZERO(&cs_); if (sched_getaffinity(0, sizeof(cs_), &cs_) == 0 && CPU_COUNT(&cs_) > 0) ncpu = (unsigned)CPU_COUNT(&cs_); }
  const unsigned nthr = std::max(1u, std::min(16u, ncpu));
  auto parallel_bins = [&](auto fn) {
    std::vector<std::thread> th;
    unsigned started = 1;                                   // stride 0 belongs to the calling thread
    for (unsigned t = 1; t < nthr; t++) {
      try { th.emplace_back([&, t] { for (int b = (int)t; b < wb; b += (int)nthr) fn(b); }); started = t + 1; }
      catch (const std::system_error &) { break; }
    }
    for (int b = 0; b < wb; b += (int)nthr) fn(b);
    for (unsigned t = started; t < nthr; t++) for (int b = (int)t; b < wb; b += (int)nthr) fn(b);      // strides whose thread did not start
    for (auto &x : th) x.join();
  };
  // segment lengths (padded to BN_SEG_PAD entries), bin-major [wb][ns]
  std::vector<int> len((size_t)wb * ns, 0);
  parallel_bins([&](int b) {
    int *L = len.data() + (size_t)b * ns;
    const int r0 = std::min((long long)b * wr, (long long)n), r1 = std::min((long long)(b + 1) * wr, (long long)n);
    for (int p = rp[r0]; p < rp[r1]; p++) L[col[p] / cs]++;
    for (int s = 0; s < ns; s++) L[s] = (L[s] + BN_SEG_PAD - 1) / BN_SEG_PAD * BN_SEG_PAD;
  });
  // Bin-major order, GROUPED: `grp` consecutive wave-bins are interleaved slice by slice - [group][slice][wave-bin of the group]
  // - so that the segments a slice's workgroup writes in phase 1 for `grp` consecutive wave-bins are one contiguous run (80 KB instead of 64
  // pieces of 1.26 KB, each 0.6 MB from the next; the gather is store-bound: 436 -> 316-331 us in the stand-alone benchmark with uniform
  // segments, profiles/r03_micro_binned_group.txt).
  // A wave-bin's own entries are then no longer one stream but ns pieces: phase 2 walks them through a LOGICAL position (piece after piece)
  // that a per-wave-bin table log2[wb][ns + 1] maps to the physical one (off2), window by window, the way phase 1 finds its destinations.
  std::vector<int> off2((size_t)wb * ns);                 // physical start of segment (wb, s)
  std::vector<int> log2((size_t)wb * (ns + 1));           // logical start of segment (wb, s) inside wave-bin wb; [ns]: the wave-bin's entry count
  long long run = 0;
  // wave-bins whose segments are adjacent per slice. Measured on config 5's matrix, one box (profiles/r03_ab_binned_groups.txt), phase 1 +
  // phase 2: 1 (the old order): 445 + 488 us; 4: 404 + 507; 16: 390 + 511; 64: 386 + 500; 256: 388 + 511 - the gather gains what its
  // stores gain from longer runs, the reduce pays a little for reading its pieces further apart.
  int grp = 64; while (grp > 1 && wb % grp) grp /= 2;
  for (int g = 0; g < wb / grp; g++)
    for (int s = 0; s < ns; s++)
      for (int wl = 0; wl < grp; wl++) { const int b = grp * g + wl; off2[(size_t)b * ns + s] = (int)run; run += len[(size_t)b * ns + s]; }
  const long long entries = run;
  for (int b = 0; b < wb; b++) {
    int lrun = 0;
    for (int s = 0; s < ns; s++) { log2[(size_t)b * (ns + 1) + s] = lrun; lrun += len[(size_t)b * ns + s]; }
    log2[(size_t)b * (ns + 1) + ns] = lrun;
  }
  std::vector<int> off1((size_t)ns * (wb + 1)), off2t((size_t)ns * wb);
  std::vector<long long> sbase(ns + 1);
  long long srun = 0; int nwin = 1;
  for (int s = 0; s < ns; s++) {
    sbase[s] = srun;
    int lrun = 0;
    for (int b = 0; b < wb; b++) { off1[(size_t)s * (wb + 1) + b] = lrun; off2t[(size_t)s * wb + b] = off2[(size_t)b * ns + s]; lrun += len[(size_t)b * ns + s]; }
    off1[(size_t)s * (wb + 1) + wb] = lrun;
    srun += lrun;
    nwin = std::max(nwin, (lrun + 1023) / 1024);
  }
  sbase[ns] = srun;
  KS_CHECK(srun == entries, KS_ERR_PLIB, "binned layout: the two orders disagree (%lld vs %lld entries)", srun, entries);
  std::vector<int> wseg((size_t)ns * nwin, 0);
  for (int s = 0; s < ns; s++) {
    const int *o1 = off1.data() + (size_t)s * (wb + 1);
    int sg = 0;
    for (int wdw = 0; wdw < nwin; wdw++) {
      const int base = wdw * 1024;
      while (sg < wb - 1 && o1[sg + 1] <= base) sg++;
      wseg[(size_t)s * nwin + wdw] = sg;
    }
  }
  std::vector<double> val2(entries, 0.0);
  std::vector<unsigned short> row16(entries, (unsigned short)wr), col16(entries, 0);      // padding: value 0 into the spare accumulator, column 0 of its slice
  parallel_bins([&](int b) {
    std::vector<int> cur(ns, 0);
    const int r0 = std::min((long long)b * wr, (long long)n), r1 = std::min((long long)(b + 1) * wr, (long long)n);
    for (int r = r0; r < r1; r++)
      for (int p = rp[r]; p < rp[r + 1]; p++) {
        const int s = col[p] / cs, i = cur[s]++;
        const long long p2 = (long long)off2[(size_t)b * ns + s] + i;
        val2[p2] = val[p]; row16[p2] = (unsigned short)(r - r0);
        col16[sbase[s] + off1[(size_t)s * (wb + 1) + b] + i] = (unsigned short)(col[p] - s * cs);
      }
  });
  std::vector<int>().swap(col); std::vector<double>().swap(val);
  auto up = [&](auto **dev, const auto &host) -> int {
    using T = typename std::remove_reference<decltype(host)>::type::value_type;
    KS_HIP(hipMalloc((void **)dev, sizeof(T) * std::max<size_t>(host.size(), 1)));
    KS_HIP(hipMemcpy(*dev, host.data(), sizeof(T) * host.size(), hipMemcpyHostToDevice));
    return KS_SUCCESS;
  };
  KS_CALL(up(&A->bn_col16, col16)); KS_CALL(up(&A->bn_row16, row16)); KS_CALL(up(&A->bn_val, val2));
  KS_CALL(up(&A->bn_off1, off1)); KS_CALL(up(&A->bn_off2t, off2t)); KS_CALL(up(&A->bn_wseg, wseg));
  KS_CALL(up(&A->bn_sbase, sbase)); KS_CALL(up(&A->bn_off2, off2)); KS_CALL(up(&A->bn_log2, log2));
  KS_HIP(hipMalloc(&A->bn_g, sizeof(double) * std::max<long long>(entries, 1)));
  KS_HIP(hipMemset(A->bn_g, 0, sizeof(double) * std::max<long long>(entries, 1)));
  const int lds1 = cs * 8 + (2 * wb + 1) * 4 + 16 + 16 * 2048, lds2 = 4 * (wr + 1) * 8;      // (+ 2 KB of column codes per wave)
  KS_HIP(hipFuncSetAttribute((const void *)k_binned_gather<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds1));
  KS_HIP(hipFuncSetAttribute((const void *)k_binned_reduce, hipFuncAttributeMaxDynamicSharedMemorySize, lds2));
  // diagonal and infinity norm are taken from the CSR arrays before they are released
  KS_HIP(hipMalloc(&A->diag_cache, sizeof(double) * n));
  KS_CALL(ks_mat_get_diagonal_internal(A, A->diag_cache));
  KS_HIP(ks_sync(ctx));
  double nrm = 0.0;
  KS_CALL(ks_mat_norm_inf_local(A, &nrm));
  A->norm_inf_cache = nrm;
  A->use_binned = true; A->bn_ns = ns; A->bn_cs = cs; A->bn_wb = wb; A->bn_wr = wr; A->bn_nwin = nwin; A->bn_entries = entries;
  hipFree(A->d_col); hipFree(A->d_val); A->d_col = nullptr; A->d_val = nullptr;
  } catch (const std::exception &) {                    // out of host memory (or anything else the build throws): the other layouts take the matrix
    hipFree(A->bn_col16); hipFree(A->bn_row16); hipFree(A->bn_val); hipFree(A->bn_g); hipFree(A->bn_off1); hipFree(A->bn_off2t); hipFree(A->bn_wseg); hipFree(A->bn_sbase); hipFree(A->bn_off2); hipFree(A->bn_log2);
    A->bn_col16 = A->bn_row16 = nullptr; A->bn_val = A->bn_g = nullptr; A->bn_off1 = A->bn_off2t = A->bn_wseg = nullptr; A->bn_sbase = nullptr; A->bn_off2 = A->bn_log2 = nullptr;
    hipFree(A->diag_cache); A->diag_cache = nullptr;
    (void)hipGetLastError();
  }
  return KS_SUCCESS;
}

int build_sliced(ks_mat A)
{
  if (A->use_binned) return KS_SUCCESS;
  ks_ctx ctx = A->ctx;
  const char *force = getenv("KSGPU_SPMV");
  if (force && strcmp(force, "sliced")) return KS_SUCCESS;
  const int n = A->n;
  if (n < 4096 || A->nnz_d == 0) return KS_SUCCESS;
  if (!force) {
    // automatic choice: x larger than one XCD's L2 can hold next to the streamed entries and most entries far from the
    // diagonal (nothing a row-ordered sweep could reuse). Measured, 33 nnz/row uniformly random: x = 4 MB CSR 0.118 ms /
    // sliced 0.128 ms; 8 MB 0.379 / 0.247; 16 MB 1.97 / 0.49; 40 MB 2.90 / 1.37.
    if ((double)n * 8.0 < 6.0 * 1048576.0 || A->nnz_d < 8LL * n) return KS_SUCCESS;
    unsigned long long *cnt = nullptr, h = 0;
    KS_HIP(hipMalloc(&cnt, sizeof(unsigned long long))); KS_HIP(hipMemsetAsync(cnt, 0, sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(k_far_entries, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, n / 16, A->d_rowptr, A->d_col, cnt);
    KS_HIP(hipMemcpyAsync(&h, cnt, sizeof(h), hipMemcpyDeviceToHost, ctx->stream)); KS_HIP(ks_sync(ctx)); hipFree(cnt);
    if ((double)h < 0.5 * (double)A->nnz_d) return KS_SUCCESS;
  }
  const int max_slice_rows = 786432;   // 6 MiB of x per slice: the 5 MiB slices of the 40 MB probe ran at the L2 rate
  int P = (int)(((long long)n + 8LL * max_slice_rows - 1) / (8LL * max_slice_rows)); if (P < 1) P = 1;
  KS_CHECK(P <= 8, KS_ERR_SUP, "sliced SpMV layout supports up to %d local rows", 64 * max_slice_rows);
  const int S = 8 * P, sc = (n + S - 1) / S;
  int *cnt = nullptr;
  KS_HIP(hipMalloc(&cnt, sizeof(int) * (size_t)S * (n + 1)));
  KS_HIP(hipMalloc(&A->sl_rowptr, sizeof(int) * (size_t)S * (n + 1)));
  hipLaunchKernelGGL(k_slice_count, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, ctx->stream, n, S, sc, A->d_rowptr, A->d_col, cnt);
  std::vector<long long> base(S + 1, 0);
  for (int s = 0; s < S; s++) {
    KS_CALL(exclusive_scan_int(ctx->stream, cnt + (size_t)s * (n + 1), A->sl_rowptr + (size_t)s * (n + 1), n + 1));
    int tot = 0;
    KS_HIP(hipMemcpyAsync(&tot, A->sl_rowptr + (size_t)s * (n + 1) + n, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    KS_HIP(ks_sync(ctx));
    base[s + 1] = base[s] + tot;
  }
  hipFree(cnt);
  KS_CHECK(base[S] == A->nnz_d, KS_ERR_PLIB, "slice counts do not add up (%lld vs %lld)", base[S], A->nnz_d);
  KS_HIP(hipMalloc(&A->sl_base, sizeof(long long) * (S + 1)));
  KS_HIP(hipMemcpyAsync(A->sl_base, base.data(), sizeof(long long) * (S + 1), hipMemcpyHostToDevice, ctx->stream));
  KS_HIP(ks_sync(ctx));
  KS_HIP(hipMalloc(&A->sl_col, sizeof(int) * A->nnz_d)); KS_HIP(hipMalloc(&A->sl_val, sizeof(double) * A->nnz_d));
  hipLaunchKernelGGL(k_slice_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, S, sc, A->d_rowptr, A->d_col, A->d_val, A->sl_rowptr, A->sl_base, A->sl_col, A->sl_val);
  KS_HIP(hipMalloc(&A->ypart, sizeof(double) * 8 * (size_t)n));
  // diagonal and infinity norm are taken from the CSR arrays before they are released
  KS_HIP(hipMalloc(&A->diag_cache, sizeof(double) * n));
  KS_CALL(ks_mat_get_diagonal_internal(A, A->diag_cache));
  KS_HIP(ks_sync(ctx));
  KS_HIP(hipGetLastError());
  double nrm = 0.0;
  KS_CALL(ks_mat_norm_inf_local(A, &nrm));                 // local rows only: no collective inside the (per-rank) layout choice
  A->norm_inf_cache = nrm;
  A->use_sliced = true; A->nslice = S; A->slice_cols = sc;
  hipFree(A->d_col); hipFree(A->d_val); A->d_col = nullptr; A->d_val = nullptr;
  return KS_SUCCESS;
}

// Try the dictionary layout (see k_spmv_dict). Needs the CSR arrays of the diagonal block on the device.
int build_dict(ks_mat A)
{
  ks_ctx ctx = A->ctx;
  const char *force = getenv("KSGPU_SPMV");
  if (force && strcmp(force, "dict") && strcmp(force, "odict")) return KS_SUCCESS;       // any other forced layout
  bool value_mode = !(force && !strcmp(force, "odict"));                 // "odict" forces the offsets-only form
  const int n = A->n;
  int *d_int = nullptr;
  KS_HIP(hipMalloc(&d_int, sizeof(int) * 4));
  KS_HIP(hipMemsetAsync(d_int, 0, sizeof(int) * 4, ctx->stream));
  hipLaunchKernelGGL(k_max_rowlen, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, A->d_rowptr, d_int + 2);
  int maxlen = 0;
  KS_HIP(hipMemcpyAsync(&maxlen, d_int + 2, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  KS_HIP(ks_sync(ctx));
  if (maxlen > 32 || maxlen == 0) { hipFree(d_int); return KS_SUCCESS; }
  const int W = maxlen <= 8 ? 8 : (maxlen <= 16 ? 16 : 32);                // 32: 27-point stencils
  if ((double)W * n > 4.0 * (double)A->nnz_d + 4096.0) { hipFree(d_int); return KS_SUCCESS; }   // mostly padding: nothing to gain
  const int cap = 4096;
  unsigned short *codes = nullptr; long long *d_bits = nullptr, *m_bits = nullptr; int *d_offs = nullptr, *m_off = nullptr;
  KS_HIP(hipMalloc(&codes, sizeof(unsigned short) * (size_t)n * W));
  KS_HIP(hipMalloc(&d_bits, sizeof(long long) * 256)); KS_HIP(hipMalloc(&d_offs, sizeof(int) * 256));
  KS_HIP(hipMalloc(&m_bits, sizeof(long long) * cap)); KS_HIP(hipMalloc(&m_off, sizeof(int) * cap));
  unsigned char *codes8 = nullptr; double *vals_out = nullptr;
  auto cleanup = [&]() { hipFree(d_int); hipFree(codes); hipFree(d_bits); hipFree(d_offs); hipFree(m_bits); hipFree(m_off); hipFree(codes8); hipFree(vals_out); };
  const size_t nslot = (size_t)((n + 63) / 64) * 64 * W;
  std::vector<long long> vals; std::vector<int> offs;                   // sorted candidate dictionaries
  bool done = false;
  for (int round = 0; round < 8 && !done; round++) {
    KS_HIP(hipMemsetAsync(d_int, 0, sizeof(int) * 2, ctx->stream));
    if (!vals.empty()) KS_HIP(hipMemcpyAsync(d_bits, vals.data(), sizeof(long long) * vals.size(), hipMemcpyHostToDevice, ctx->stream));
    if (!offs.empty()) KS_HIP(hipMemcpyAsync(d_offs, offs.data(), sizeof(int) * offs.size(), hipMemcpyHostToDevice, ctx->stream));
    if (!value_mode && !codes8) {
      KS_HIP(hipMalloc(&codes8, nslot)); KS_HIP(hipMalloc(&vals_out, sizeof(double) * nslot));
      KS_HIP(hipMemsetAsync(vals_out, 0, sizeof(double) * nslot, ctx->stream));
    }
    hipLaunchKernelGGL(k_dict_encode, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, W, A->d_rowptr, A->d_col, A->d_val,
                       d_bits, value_mode ? (int)vals.size() : -1, d_offs, (int)offs.size(), codes, d_int, m_bits, m_off, cap, codes8, vals_out);
    int miss[2] = {0, 0};
    KS_HIP(hipMemcpyAsync(miss, d_int, sizeof(int) * 2, hipMemcpyDeviceToHost, ctx->stream));
    KS_HIP(ks_sync(ctx));
    if (miss[0] == 0) { done = true; break; }
    const int got = std::min(miss[0], cap);
    std::vector<long long> mb(got); std::vector<int> mo(got);
    KS_HIP(hipMemcpy(mb.data(), m_bits, sizeof(long long) * got, hipMemcpyDeviceToHost));
    KS_HIP(hipMemcpy(mo.data(), m_off, sizeof(int) * got, hipMemcpyDeviceToHost));
    if (value_mode) { vals.insert(vals.end(), mb.begin(), mb.end()); std::sort(vals.begin(), vals.end()); vals.erase(std::unique(vals.begin(), vals.end()), vals.end()); }
    offs.insert(offs.end(), mo.begin(), mo.end()); std::sort(offs.begin(), offs.end()); offs.erase(std::unique(offs.begin(), offs.end()), offs.end());
    if (value_mode && vals.size() > 255) { value_mode = false; vals.clear(); }        // too many values: keep them in full, compress the indices only
    if (offs.size() > (value_mode ? 256u : 255u)) break;                 // not a dictionary matrix
  }
  if (!done) { cleanup(); return KS_SUCCESS; }
  if (!value_mode) {
    std::vector<int> dof(256, 0);
    for (size_t i = 0; i < offs.size(); i++) dof[i] = offs[i];
    KS_HIP(hipMalloc(&A->dc_off, sizeof(int) * 256));
    KS_HIP(hipMemcpy(A->dc_off, dof.data(), sizeof(int) * 256, hipMemcpyHostToDevice));
    A->dc_codes8 = codes8; codes8 = nullptr; A->dc_vals = vals_out; vals_out = nullptr;
    A->use_odict = true; A->dict_w = W; A->dict_nval = 0; A->dict_noff = (int)offs.size();
    cleanup();
    return KS_SUCCESS;
  }
  std::vector<double> dv(256, 0.0); std::vector<int> dof(256, 0);
  for (size_t i = 0; i < vals.size(); i++) memcpy(&dv[i], &vals[i], sizeof(double));
  for (size_t i = 0; i < offs.size(); i++) dof[i] = offs[i];
  KS_HIP(hipMalloc(&A->dc_val, sizeof(double) * 256)); KS_HIP(hipMalloc(&A->dc_off, sizeof(int) * 256));
  KS_HIP(hipMemcpy(A->dc_val, dv.data(), sizeof(double) * 256, hipMemcpyHostToDevice));
  KS_HIP(hipMemcpy(A->dc_off, dof.data(), sizeof(int) * 256, hipMemcpyHostToDevice));
  A->dc_codes = codes; codes = nullptr;
  A->use_dict = true; A->dict_w = W; A->dict_nval = (int)vals.size(); A->dict_noff = (int)offs.size();
  cleanup();
  return KS_SUCCESS;
}

int build_sell(ks_mat A)
{
  if (A->use_sliced || A->use_binned) return KS_SUCCESS;
  ks_ctx ctx = A->ctx;
  const char *force = getenv("KSGPU_SPMV");
  if (force && !strcmp(force, "csrvec")) { A->force_csr_vector = true; return KS_SUCCESS; }     // the CSR-vector kernel at any size (A/B against the row-block kernel)
  if (force && !strcmp(force, "csrblock")) { A->force_csr_block = true; return KS_SUCCESS; }    // the workgroup-per-256-rows form of the row-block kernel (A/B against the wave form)
  if (force && !strcmp(force, "csrregs")) { A->force_csr_regs = true; return KS_SUCCESS; }      // the register-staged form of the wave kernel also for short rows (A/B against the LDS-DMA form)
  if (force && !strcmp(force, "csr")) return KS_SUCCESS;
  if (A->n == 0 || A->nnz_d == 0) return KS_SUCCESS;
  KS_CALL(build_dict(A));                                   // independent of the SELL decision below; needs the CSR arrays
  if (A->use_dict || A->use_odict) {
    // the dictionary form is the only copy kept: diagonal and infinity norm are taken from the CSR arrays before they go
    KS_HIP(hipMalloc(&A->diag_cache, sizeof(double) * A->n));
    KS_CALL(ks_mat_get_diagonal_internal(A, A->diag_cache));
    double nrm = 0.0;
    KS_CALL(ks_mat_norm_inf_local(A, &nrm));
    KS_HIP(ks_sync(ctx));
    A->norm_inf_cache = nrm; A->have_cache = true;
    hipFree(A->d_col); hipFree(A->d_val); A->d_col = nullptr; A->d_val = nullptr;
    return KS_SUCCESS;
  }
  const int ns = (A->n + 63) / 64;
  int *width = nullptr;
  KS_HIP(hipMalloc(&width, sizeof(int) * (ns + 1)));
  KS_HIP(hipMalloc(&A->s_len, sizeof(int) * A->n));
  KS_HIP(hipMalloc(&A->s_ptr, sizeof(int) * (ns + 1)));
  KS_HIP(hipMemsetAsync(width + ns, 0, sizeof(int), ctx->stream));
  hipLaunchKernelGGL(k_sell_widths, dim3((ns + 3) / 4), dim3(256), 0, ctx->stream, A->n, ns, A->d_rowptr, width, A->s_len);
  KS_CALL(exclusive_scan_int(ctx->stream, width, A->s_ptr, ns + 1));
  int total = 0;
  KS_HIP(hipMemcpy(&total, A->s_ptr + ns, sizeof(int), hipMemcpyDeviceToHost));
  hipFree(width);
  const long long entries = (long long)total * 64;
  const bool ok = (force && !strcmp(force, "sell")) || (double)entries <= 1.125 * (double)A->nnz_d + 64.0 * 64.0;
  if (!ok || entries <= 0) { hipFree(A->s_len); hipFree(A->s_ptr); A->s_len = A->s_ptr = nullptr; return KS_SUCCESS; }
  KS_HIP(hipMalloc(&A->s_col, sizeof(int) * entries));
  KS_HIP(hipMalloc(&A->s_val, sizeof(double) * entries));
  hipLaunchKernelGGL(k_sell_fill, dim3((ns + 3) / 4), dim3(256), 0, ctx->stream, A->n, ns, A->d_rowptr, A->d_col, A->d_val, A->s_ptr, A->s_col, A->s_val);
  KS_HIP(ks_sync(ctx));
  KS_HIP(hipGetLastError());
  A->use_sell = true; A->nslices = ns; A->s_entries = entries;
  // the CSR copy of the diagonal block is no longer needed on the device
  hipFree(A->d_col); hipFree(A->d_val); A->d_col = nullptr; A->d_val = nullptr;
  return KS_SUCCESS;
}

} // namespace

// y = A x fused with the dot sweep of y against the ncols - 1 columns in front of it (and itself), where that pays: a single rank, the
// dictionary layout, no B-inner product, a basis that lives in the Infinity Cache, two-row tiles. *done = false: the caller runs the
// product and the dots as separate launches. y must be column jy of bv; the dots are those ksk_dot(bv, col(-nc), ld, nc + jy + 1, y) leaves.
int ks_mat_mult_dot_fused(ks_mat A, ks_bv bv, const double *x, int jy, bool gate, bool *done)
{
  *done = false;
  ks_ctx ctx = A->ctx;
  const int ncols = bv->nc + jy + 1;
  const double *Vb = ks_bv_col(bv, -bv->nc);
  double *y = ks_bv_col(bv, jy);
  if (!A->use_dict || A->shell_mult || ks_is_multi(ctx) || A->n_orows > 0 || bv->matrix || A->n != bv->n) return KS_SUCCESS;
  if (ncols < 1 || ncols > KS_MAX_COLS || bv->ld % 2 || (((uintptr_t)Vb) & 15) || (((uintptr_t)y) & 15)) return KS_SUCCESS;
  if (!ksk::ks_basis_is_cache_resident((size_t)(bv->nc + bv->m), (size_t)bv->ld)) return KS_SUCCESS;
  if (ctx->dbg.no_spmv_dot) return KS_SUCCESS;        // test hook: the separate launches, to compare bits
  const int dot_per_cu = std::max(1, std::min(4, (30 + ncols - 1) / ncols));          // the grid ksk_dot would use: the partials, and so the bits, are the same
  const int grid = ks_sweep_grid_for(ctx, bv->n, 2, nullptr, dot_per_cu);
  const int rev = bv->sweep_dir; bv->sweep_dir ^= 1;
  bv->spec.valid = false; bv->last_grid = grid;
  const KsGsState *g = gate ? bv->gs : nullptr;
  KsProfScope ps(ctx, KS_K_SPMVDOT, 8.0 * bv->n * ncols + 12.0 * A->nnz + 4.0 * (A->n + 1) + 16.0 * A->n, ks_kt_for(ncols), 8.0 * bv->n * ncols + (2.0 * A->dict_w + 8.0) * A->n);
#define LAUNCH_FD(KT)                                                                                                                                      \
  do {                                                                                                                                                     \
    if (A->dict_w == 8) hipLaunchKernelGGL((k_dot_spmv_dict<KT, 8>), dim3(grid), dim3(ksk::SW_BLOCK), 0, ctx->stream, Vb, (long long)bv->ld, bv->n, ncols, (const uint4 *)A->dc_codes, A->dc_val, A->dict_nval, A->dc_off, A->dict_noff, x, y, bv->partials, g, &bv->gs->pgrid, rev); \
    else if (A->dict_w == 16) hipLaunchKernelGGL((k_dot_spmv_dict<KT, 16>), dim3(grid), dim3(ksk::SW_BLOCK), 0, ctx->stream, Vb, (long long)bv->ld, bv->n, ncols, (const uint4 *)A->dc_codes, A->dc_val, A->dict_nval, A->dc_off, A->dict_noff, x, y, bv->partials, g, &bv->gs->pgrid, rev); \
    else return KS_SUCCESS;                                                                                                                                \
  } while (0)
  if (A->dict_w != 8 && A->dict_w != 16) return KS_SUCCESS;
  KS_KT_DISPATCH(ncols, LAUNCH_FD);
#undef LAUNCH_FD
  KS_HIP(hipGetLastError());
  *done = true;
  return KS_SUCCESS;
}

extern "C" int ks_mat_create_csr(ks_ctx ctx, int n_local, int row_start, int n_global, const int *rowptr, const int *col, const double *val, ks_mat *out)
{
  return ks_mat_create_csr_flags(ctx, n_local, row_start, n_global, rowptr, col, val, 0u, out);
}

extern "C" int ks_mat_create_csr_flags(ks_ctx ctx, int n_local, int row_start, int n_global, const int *rowptr, const int *col, const double *val, unsigned flags, ks_mat *out)
{
  KS_CHECK(ctx && out, KS_ERR_ARG_NULL, "ctx/out is NULL");
  KS_CHECK((flags & ~KS_MAT_KEEP_CSR) == 0, KS_ERR_ARG_OUTOFRANGE, "unknown matrix creation flags 0x%x", flags);
  KS_CHECK(n_local >= 0 && row_start >= 0 && row_start + n_local <= n_global, KS_ERR_ARG_OUTOFRANGE, "bad row range [%d,%d) of %d", row_start, row_start + n_local, n_global);
  KS_CHECK(rowptr && (rowptr[n_local] == 0 || (col && val)), KS_ERR_ARG_NULL, "CSR arrays are NULL");
  KS_CHECK(rowptr[0] == 0, KS_ERR_ARG_WRONG, "rowptr[0] must be 0");
  KS_HIP(hipSetDevice(ctx->device));
  const long long nnz = rowptr[n_local];
  ks_mat A = new ks_mat_s(); A->ctx = ctx; A->n = n_local; A->row_start = row_start; A->n_global = n_global; A->nnz = nnz;
  // split diag / off-diag on the host (setup path)
  std::vector<int> rp_d(n_local + 1, 0), rp_o(n_local + 1, 0), cd, co; std::vector<double> vd, vo;
  cd.reserve(nnz); vd.reserve(nnz);
  for (int r = 0; r < n_local; r++) {
    KS_CHECK(rowptr[r + 1] >= rowptr[r], KS_ERR_ARG_WRONG, "rowptr not monotone at row %d", r);
    for (int p = rowptr[r]; p < rowptr[r + 1]; p++) {
      int c = col[p];
      if (c < 0 || c >= n_global) { delete A; KS_FAIL(KS_ERR_ARG_OUTOFRANGE, "column %d out of range at row %d", c, r); }
      if (c >= row_start && c < row_start + n_local) { cd.push_back(c - row_start); vd.push_back(val[p]); }
      else { co.push_back(c); vo.push_back(val[p]); }
    }
    rp_d[r + 1] = (int)cd.size(); rp_o[r + 1] = (int)co.size();
  }
  std::vector<int> garray(co);
  std::sort(garray.begin(), garray.end()); garray.erase(std::unique(garray.begin(), garray.end()), garray.end());
  for (auto &c : co) c = (int)(std::lower_bound(garray.begin(), garray.end(), c) - garray.begin());
  A->nnz_d = (long long)cd.size(); A->nnz_o = (long long)co.size();
  KS_HIP(hipMalloc(&A->d_rowptr, sizeof(int) * (n_local + 1)));
  KS_HIP(hipMalloc(&A->d_col, sizeof(int) * (cd.size() + CW_PAD)));
  KS_HIP(hipMalloc(&A->d_val, sizeof(double) * (vd.size() + CW_PAD)));
  KS_HIP(hipMemcpy(A->d_rowptr, rp_d.data(), sizeof(int) * (n_local + 1), hipMemcpyHostToDevice));
  KS_HIP(hipMemcpy(A->d_col, cd.data(), sizeof(int) * cd.size(), hipMemcpyHostToDevice));
  KS_HIP(hipMemcpy(A->d_val, vd.data(), sizeof(double) * vd.size(), hipMemcpyHostToDevice));
  A->lanes_per_row = pick_lanes(A->nnz_d, n_local);
  if (A->nnz_o) {
    KS_HIP(hipMalloc(&A->o_rowptr, sizeof(int) * (n_local + 1)));
    KS_HIP(hipMalloc(&A->o_col, sizeof(int) * co.size()));
    KS_HIP(hipMalloc(&A->o_val, sizeof(double) * vo.size()));
    KS_HIP(hipMemcpy(A->o_rowptr, rp_o.data(), sizeof(int) * (n_local + 1), hipMemcpyHostToDevice));
    KS_HIP(hipMemcpy(A->o_col, co.data(), sizeof(int) * co.size(), hipMemcpyHostToDevice));
    KS_HIP(hipMemcpy(A->o_val, vo.data(), sizeof(double) * vo.size(), hipMemcpyHostToDevice));
  }
  int rc = build_halo_plan(A, garray);
  if (!rc) rc = compact_offdiag_rows(A);
  if (!rc) rc = build_binned(A);
  if (!rc) rc = build_sliced(A);
  if (!rc) rc = build_sell(A);
  if (rc) { ks_mat_destroy(A); return rc; }
  if (flags & KS_MAT_KEEP_CSR) {
    try { A->k_rowptr.assign(rowptr, rowptr + n_local + 1); A->k_col.assign(col, col + nnz); A->k_val.assign(val, val + nnz); }
    catch (const std::exception &e) { ks_mat_destroy(A); KS_FAIL(KS_ERR_MEM, "KS_MAT_KEEP_CSR: %s", e.what()); }
    A->keep_csr = true;
  }
  *out = A;
  return KS_SUCCESS;
}

// MatDuplicate + MatAXPY / MatShift on the kept CSR arrays (ks_csr.cpp), then the ordinary assembly of the result
extern "C" int ks_mat_create_axpy(ks_mat A, double alpha, ks_mat B, unsigned flags, ks_mat *out)
{
  KS_CHECK(A && out, KS_ERR_ARG_NULL, "A/out is NULL");
  KS_CHECK(!A->shell_mult && (!B || !B->shell_mult), KS_ERR_SUP, "MatAXPY of a shell matrix");
  KS_CHECK(A->keep_csr && (!B || B->keep_csr), KS_ERR_ORDER, "MatAXPY needs the CSR arrays of its operands: create them with KS_MAT_KEEP_CSR");
  KS_CHECK(!B || (B->n == A->n && B->row_start == A->row_start && B->n_global == A->n_global && B->ctx == A->ctx), KS_ERR_ARG_INCOMP, "Mismatching row blocks of A (%d rows from %d) and B (%d rows from %d)", A->n, A->row_start, B ? B->n : 0, B ? B->row_start : 0);
  std::vector<int> rp, col; std::vector<double> val;
  bool fits = false;
  try { fits = ksc::csr_axpy(A->n, A->row_start, A->k_rowptr.data(), A->k_col.data(), A->k_val.data(), alpha, B ? B->k_rowptr.data() : nullptr, B ? B->k_col.data() : nullptr, B ? B->k_val.data() : nullptr, rp, col, val); }
  catch (const std::exception &e) { KS_FAIL(KS_ERR_MEM, "MatAXPY on the host: %s", e.what()); }
  KS_CHECK(fits, KS_ERR_ARG_OUTOFRANGE, "the sum exceeds 32-bit PetscInt indices");
  return ks_mat_create_csr_flags(A->ctx, A->n, A->row_start, A->n_global, rp.data(), col.data(), val.data(), flags, out);
}

extern "C" int ks_mat_create_laplacian3d(ks_ctx ctx, int nx, int ny, int nz, int z0, int nzl, ks_mat *out)
{
  KS_CHECK(ctx && out, KS_ERR_ARG_NULL, "ctx/out is NULL");
  KS_CHECK(nx > 0 && ny > 0 && nz > 0 && z0 >= 0 && nzl > 0 && z0 + nzl <= nz, KS_ERR_ARG_OUTOFRANGE, "bad grid %dx%dx%d planes [%d,%d)", nx, ny, nz, z0, z0 + nzl);
  const long long plane = (long long)nx * ny, n = plane * nzl, N = plane * nz;
  KS_CHECK(N * 7 < 2147483647LL && n < 2147483647LL, KS_ERR_ARG_OUTOFRANGE, "problem exceeds 32-bit PetscInt indices");
  KS_HIP(hipSetDevice(ctx->device));
  ks_mat A = new ks_mat_s(); A->ctx = ctx; A->n = (int)n; A->row_start = (int)(plane * z0); A->n_global = (int)N;
  int *cnt_d = nullptr, *cnt_o = nullptr;
  KS_HIP(hipMalloc(&cnt_d, sizeof(int) * (n + 1))); KS_HIP(hipMalloc(&cnt_o, sizeof(int) * (n + 1)));
  KS_HIP(hipMemsetAsync(cnt_d + n, 0, sizeof(int), ctx->stream)); KS_HIP(hipMemsetAsync(cnt_o + n, 0, sizeof(int), ctx->stream));
  const unsigned nb = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_lap3d_count, dim3(nb), dim3(256), 0, ctx->stream, nx, ny, nz, z0, nzl, cnt_d, cnt_o);
  KS_HIP(hipMalloc(&A->d_rowptr, sizeof(int) * (n + 1))); KS_HIP(hipMalloc(&A->o_rowptr, sizeof(int) * (n + 1)));
  KS_CALL(exclusive_scan_int(ctx->stream, cnt_d, A->d_rowptr, n + 1));
  KS_CALL(exclusive_scan_int(ctx->stream, cnt_o, A->o_rowptr, n + 1));
  int nnzd = 0, nnzo = 0;
  KS_HIP(hipMemcpy(&nnzd, A->d_rowptr + n, sizeof(int), hipMemcpyDeviceToHost));
  KS_HIP(hipMemcpy(&nnzo, A->o_rowptr + n, sizeof(int), hipMemcpyDeviceToHost));
  hipFree(cnt_d); hipFree(cnt_o);
  A->nnz_d = nnzd; A->nnz_o = nnzo; A->nnz = (long long)nnzd + nnzo;
  KS_HIP(hipMalloc(&A->d_col, sizeof(int) * (nnzd + CW_PAD))); KS_HIP(hipMalloc(&A->d_val, sizeof(double) * (nnzd + CW_PAD)));
  KS_HIP(hipMalloc(&A->o_col, sizeof(int) * std::max(nnzo, 1))); KS_HIP(hipMalloc(&A->o_val, sizeof(double) * std::max(nnzo, 1)));
  hipLaunchKernelGGL(k_lap3d_fill, dim3(nb), dim3(256), 0, ctx->stream, nx, ny, nz, z0, nzl, A->d_rowptr, A->d_col, A->d_val, A->o_rowptr, A->o_col, A->o_val);
  KS_HIP(ks_sync(ctx));
  A->lanes_per_row = pick_lanes(A->nnz_d, A->n);
  // ghost columns: the plane below (owned by the previous slab) then the plane above
  std::vector<int> garray;
  if (z0 > 0) for (long long c = 0; c < plane; c++) garray.push_back((int)(plane * (z0 - 1) + c));
  if (z0 + nzl < nz) for (long long c = 0; c < plane; c++) garray.push_back((int)(plane * (z0 + nzl) + c));
  int rc = KS_SUCCESS;
  if (ctx->comm.size == 1 && !garray.empty()) { ks_mat_destroy(A); KS_FAIL(KS_ERR_ARG_INCOMP, "a partial slab needs a multi-rank communicator"); }
  rc = build_halo_plan(A, garray);
  if (!rc) rc = compact_offdiag_rows(A);
  if (!rc) rc = build_sell(A);
  if (rc) { ks_mat_destroy(A); return rc; }
  *out = A;
  return KS_SUCCESS;
}

extern "C" int ks_mat_create_laplacian2d(ks_ctx ctx, int n, int m, ks_mat *out)
{
  KS_CHECK(ctx && out, KS_ERR_ARG_NULL, "ctx/out is NULL");
  KS_CHECK(n > 0 && m > 0 && (long long)n * m * 5 < 2147483647LL, KS_ERR_ARG_OUTOFRANGE, "bad grid %dx%d", n, m);
  KS_CHECK(ctx->comm.size == 1, KS_ERR_SUP, "2-D generator is single-rank");
  KS_HIP(hipSetDevice(ctx->device));
  const long long N = (long long)n * m;
  ks_mat A = new ks_mat_s(); A->ctx = ctx; A->n = (int)N; A->row_start = 0; A->n_global = (int)N;
  int *cnt = nullptr; KS_HIP(hipMalloc(&cnt, sizeof(int) * (N + 1)));
  KS_HIP(hipMemsetAsync(cnt + N, 0, sizeof(int), ctx->stream));
  const unsigned nb = (unsigned)((N + 255) / 256);
  hipLaunchKernelGGL(k_lap2d_count, dim3(nb), dim3(256), 0, ctx->stream, n, m, cnt);
  KS_HIP(hipMalloc(&A->d_rowptr, sizeof(int) * (N + 1)));
  KS_CALL(exclusive_scan_int(ctx->stream, cnt, A->d_rowptr, N + 1));
  int nnz = 0; KS_HIP(hipMemcpy(&nnz, A->d_rowptr + N, sizeof(int), hipMemcpyDeviceToHost));
  hipFree(cnt);
  A->nnz = A->nnz_d = nnz;
  KS_HIP(hipMalloc(&A->d_col, sizeof(int) * (nnz + CW_PAD))); KS_HIP(hipMalloc(&A->d_val, sizeof(double) * (nnz + CW_PAD)));
  hipLaunchKernelGGL(k_lap2d_fill, dim3(nb), dim3(256), 0, ctx->stream, n, m, A->d_rowptr, A->d_col, A->d_val);
  KS_HIP(ks_sync(ctx));
  A->lanes_per_row = pick_lanes(A->nnz_d, A->n);
  { int rc = build_sell(A); if (rc) { ks_mat_destroy(A); return rc; } }
  *out = A;
  return KS_SUCCESS;
}

extern "C" int ks_mat_destroy(ks_mat A)
{
  if (!A) return KS_SUCCESS;
  if (A->At) { ks_mat_destroy(A->At); A->At = nullptr; }
  hipSetDevice(A->ctx->device);
  ks_sync(A->ctx);
  hipFree(A->d_rowptr); hipFree(A->d_col); hipFree(A->d_val);
  hipFree(A->o_rowptr); hipFree(A->o_col); hipFree(A->o_val); hipFree(A->o_rows);
  if (A->ctx->halo_stream) hipStreamSynchronize(A->ctx->halo_stream);
  ks_halo_release(A);                 // not collective: waits for the neighbours' last acknowledgements (ks_halo.hip); ks_mat_set_halo(A, KS_HALO_PROVIDER) first is the collective way
  hipFree(A->ghost); hipFree(A->send_idx); hipFree(A->send_buf);
  hipFree(A->s_ptr); hipFree(A->s_len); hipFree(A->s_col); hipFree(A->s_val);
  hipFree(A->dc_codes); hipFree(A->dc_val); hipFree(A->dc_off); hipFree(A->dc_codes8); hipFree(A->dc_vals);
  hipFree(A->sl_rowptr); hipFree(A->sl_col); hipFree(A->sl_val); hipFree(A->sl_base); hipFree(A->ypart); hipFree(A->diag_cache);
  hipFree(A->bn_col16); hipFree(A->bn_row16); hipFree(A->bn_val); hipFree(A->bn_g); hipFree(A->bn_off1); hipFree(A->bn_off2t); hipFree(A->bn_wseg); hipFree(A->bn_sbase); hipFree(A->bn_off2); hipFree(A->bn_log2);
  delete A;
  return KS_SUCCESS;
}

// A matrix-free operator whose callback only enqueues work on the context's stream (no host synchronisation, no host reads of
// device results) lets BVMatLanczos / BVMatArnoldi enqueue the whole run ahead, as they do for assembled matrices.
extern "C" int ks_mat_shell_set_enqueue_only(ks_mat A, int flag)
{
  KS_CHECK(A, KS_ERR_ARG_NULL, "Mat is NULL");
  KS_CHECK(A->shell_mult, KS_ERR_ARG_WRONGSTATE, "not a matrix-free operator");
  A->shell_nosync = flag != 0;
  return KS_SUCCESS;
}
extern "C" int ks_mat_get_layout(ks_mat A, int *layout)     // storage of the diagonal block: KS_MAT_LAYOUT_*
{
  KS_CHECK(A && layout, KS_ERR_ARG_NULL, "NULL argument");
  *layout = A->shell_mult ? KS_MAT_LAYOUT_SHELL : A->use_binned ? KS_MAT_LAYOUT_BINNED : (A->use_sliced ? KS_MAT_LAYOUT_SLICED : (A->use_dict ? KS_MAT_LAYOUT_DICT : (A->use_odict ? KS_MAT_LAYOUT_ODICT : (A->use_sell ? KS_MAT_LAYOUT_SELL : KS_MAT_LAYOUT_CSR))));
  return KS_SUCCESS;
}
extern "C" int ks_mat_get_sizes(ks_mat A, int *n_local, int *n_global, long long *nnz_local)
{
  KS_CHECK(A, KS_ERR_ARG_NULL, "Mat is NULL");
  if (n_local) *n_local = A->n;
  if (n_global) *n_global = A->n_global;
  if (nnz_local) *nnz_local = A->nnz;
  return KS_SUCCESS;
}

// y = A x.  Multi-rank: pack boundary entries, exchange with the neighbours (RCCL send/recv over xGMI),
// diagonal block product, then the off-diagonal rows add their ghost contributions.
// a row scaling can ride in the product's last pass: the binned layout on a rank without off-diagonal rows (their contribution is added afterwards)
bool ks_mat_can_rowscale(ks_mat A) { return A && !A->shell_mult && A->use_binned && A->n_orows == 0; }

int ks_mat_mult_internal(ks_mat A, const double *x, double *y, const double *rowscale)
{
  ks_ctx ctx = A->ctx;
  if (rowscale && !ks_mat_can_rowscale(A)) KS_FAIL(KS_ERR_PLIB, "row scaling asked of a product that cannot fold it in");
  if (A->shell_mult) return A->shell_mult(A->shell_user, x, y);
  const bool multi = ctx->comm.size > 1 && (A->nsend > 0 || A->nghost > 0);
  // Halo under the diagonal-block product (PETSc: VecScatterBegin / local product / VecScatterEnd in MatMult_MPIAIJ): pack and
  // neighbour exchange go to the halo stream once x is complete on the main stream; the main stream runs the diagonal block
  // and only the off-diagonal rows wait for the ghosts. The next product's pack is ordered after this one's off-diagonal rows
  // through ev_x (recorded on the main stream), so send_buf / ghost are never overwritten while still being read.
  const bool overlap = multi && ctx->halo_overlap;
  if (multi) {
    hipStream_t hs = ctx->stream;
    if (overlap) {
      KS_CALL(ks_ctx_halo_stream(ctx));
      hs = ctx->halo_stream;
      KS_HIP(hipEventRecord(ctx->ev_x, ctx->stream));
      KS_HIP(hipStreamWaitEvent(hs, ctx->ev_x, 0));
    }
    KsProfScope ps(ctx, KS_K_HALO, 8.0 * (A->nsend + A->nghost));      // (events on the main stream: with the overlap this times the enqueue only)
    if (A->hp.enabled) KS_CALL(ks_halo_peer_exchange(A, x, hs));      // straight into the neighbours' ghost mailboxes: no library call
    else {
      if (A->nsend) hipLaunchKernelGGL(k_pack, dim3((A->nsend + 255) / 256), dim3(256), 0, hs, A->nsend, A->send_idx, x, A->send_buf);
      KS_CALL(ks_comm_exchange(ctx, (int)A->peers.size(), A->peers.data(), A->send_buf, A->send_off.data(), A->send_cnt.data(),
                               A->ghost, A->recv_off.data(), A->recv_cnt.data(), (int)sizeof(double), hs));
    }
    if (overlap) KS_HIP(hipEventRecord(ctx->ev_halo, hs));
  }
  {
    const double csr_bytes = 12.0 * A->nnz + 4.0 * (A->n + 1) + 16.0 * A->n;                    // what the CSR algorithm moves
    KsProfScope ps(ctx, KS_K_SPMV, csr_bytes, A->use_binned ? 18 : A->use_dict ? 16 : (A->use_odict ? 17 : (A->use_sell ? 8 : 0)),   // variant 16: k_spmv_dict, 17: k_spmv_odict, 8: k_spmv_sell<8>, 0: k_spmv_csr
                   A->use_binned ? 28.0 * A->bn_entries + 16.0 * A->n + 12.0 * A->nnz_o
                   : A->use_dict ? (2.0 * A->dict_w + 16.0) * A->n + 12.0 * A->nnz_o
                   : (A->use_odict ? 8.0 * A->nnz_d + (A->dict_w + 16.0) * A->n + 12.0 * A->nnz_o : -1.0));   // the dictionary layouts' own compulsory bytes
    if (A->use_binned) {
      hipLaunchKernelGGL(k_binned_gather<true>, dim3((unsigned)A->bn_ns), dim3(1024), (size_t)A->bn_cs * 8 + (size_t)(2 * A->bn_wb + 1) * 4 + 16 + 16 * 2048, ctx->stream,
                         A->n, A->bn_cs, A->bn_wb, A->bn_nwin, A->bn_sbase, A->bn_col16, A->bn_off1, A->bn_off2t, A->bn_wseg, x, A->bn_g);
      hipLaunchKernelGGL(k_binned_reduce, dim3((unsigned)(A->bn_wb / 4)), dim3(256), (size_t)4 * (A->bn_wr + 1) * 8, ctx->stream, A->n, A->bn_wr, A->bn_ns, A->bn_log2, A->bn_off2,
                         A->bn_g, A->bn_val, A->bn_row16, y, rowscale);
    } else if (A->use_sliced) {
      const int per_xcd = std::max(1, std::min((A->n + 255) / 256, (ctx->num_cu / 8) * 8));       // 8 resident workgroups per CU of the XCD
      hipLaunchKernelGGL(k_spmv_sliced, dim3((unsigned)(8 * per_xcd)), dim3(256), 0, ctx->stream, A->n, A->nslice, A->sl_rowptr, A->sl_base, A->sl_col, A->sl_val, x, A->ypart);
      hipLaunchKernelGGL(k_sum_parts, dim3((unsigned)std::min((A->n + 255) / 256, ctx->num_cu * 8)), dim3(256), 0, ctx->stream, A->n, A->ypart, y);
    } else if (A->use_dict) {
      const int dremap_env = 1, dmul = 64;
      const long long groups = ((long long)A->n + SPMV_BLOCK - 1) / SPMV_BLOCK;
      long long nblk = std::max<long long>(1, std::min<long long>(groups, (long long)ctx->num_cu * dmul));
      const int dremap = (dremap_env && nblk >= 64) ? 1 : 0;               // small matrices: nothing to pin
      if (dremap) nblk = std::min<long long>((nblk + 7) / 8, (groups + 7) / 8) * 8;
      const dim3 gr((unsigned)nblk);
      if (A->dict_w == 8) hipLaunchKernelGGL((k_spmv_dict<8>), gr, dim3(SPMV_BLOCK), 0, ctx->stream, A->n, (const uint4 *)A->dc_codes, A->dc_val, A->dict_nval, A->dc_off, A->dict_noff, x, y, dremap);
      else if (A->dict_w == 32) hipLaunchKernelGGL((k_spmv_dict<32>), gr, dim3(SPMV_BLOCK), 0, ctx->stream, A->n, (const uint4 *)A->dc_codes, A->dc_val, A->dict_nval, A->dc_off, A->dict_noff, x, y, dremap);
      else hipLaunchKernelGGL((k_spmv_dict<16>), gr, dim3(SPMV_BLOCK), 0, ctx->stream, A->n, (const uint4 *)A->dc_codes, A->dc_val, A->dict_nval, A->dc_off, A->dict_noff, x, y, dremap);
    } else if (A->use_odict) {
      const int oremap_env = 1, omul = 64;
      const long long groups = ((long long)A->n + SPMV_BLOCK - 1) / SPMV_BLOCK;
      long long nblk = std::max<long long>(1, std::min<long long>(groups, (long long)ctx->num_cu * omul));
      const int oremap = (oremap_env && nblk >= 64) ? 1 : 0;
      if (oremap) nblk = std::min<long long>((nblk + 7) / 8, (groups + 7) / 8) * 8;
      if (A->dict_w == 8) hipLaunchKernelGGL((k_spmv_odict<8>), dim3((unsigned)nblk), dim3(SPMV_BLOCK), 0, ctx->stream, A->n, A->dc_codes8, A->dc_vals, A->dc_off, A->dict_noff, x, y, oremap);
      else if (A->dict_w == 32) hipLaunchKernelGGL((k_spmv_odict<32>), dim3((unsigned)nblk), dim3(SPMV_BLOCK), 0, ctx->stream, A->n, A->dc_codes8, A->dc_vals, A->dc_off, A->dict_noff, x, y, oremap);
      else hipLaunchKernelGGL((k_spmv_odict<16>), dim3((unsigned)nblk), dim3(SPMV_BLOCK), 0, ctx->stream, A->n, A->dc_codes8, A->dc_vals, A->dc_off, A->dict_noff, x, y, oremap);
    } else if (A->use_sell) {
      const int remap_env = 1;   // each XCD one contiguous range of slices: 179 -> 172 us on the 216^3 Laplacian
      const long long groups = ((long long)A->nslices + 3) / 4;
      long long blocks = std::min<long long>(groups, (long long)ctx->num_cu * 16);
      const int bmul = 4096;   // one 256-row group per block measured fastest
      blocks = std::min<long long>(groups, (long long)ctx->num_cu * bmul);
      const dim3 gr((unsigned)std::max<long long>(blocks, 1));
      const int remap = (remap_env && blocks == groups && blocks >= 64) ? 1 : 0;     // only with one slice group per workgroup (a strided loop would interleave the ranges again)
      hipLaunchKernelGGL((k_spmv_sell<4>), gr, dim3(SPMV_BLOCK), 0, ctx->stream, A->n, A->nslices, A->s_ptr, A->s_len, A->s_col, A->s_val, x, y, remap);
    } else if (A->n >= 2048 && !A->force_csr_vector && !A->force_csr_block) {
      // 4 or 5 workgroups of 4 waves per CU (registers; forcing 6 spills: 259 us); a multiple of 8 so that every XCD gets its eighth of the rows
      const long long NG = ((long long)A->n + 255) / 256;
      const bool rowside = A->nnz_d <= 12LL * A->n;          // short rows: gather on the row side (register-staged form)
      const bool dma = A->nnz_d <= 16LL * A->n && !A->force_csr_regs;      // the LDS-DMA form gathers on the row side up to 16 entries per row on average
      long long nb = std::min<long long>(NG, (long long)ctx->num_cu * (rowside ? 5 : 4));
      const int remap = nb >= 64 ? 1 : 0;
      if (remap) nb = (nb / 8) * 8;
      if (dma) {
        // short rows: the LDS-DMA form (six workgroups of four waves per CU: 144 KB of LDS; 33 registers). 216^3 Laplacian: 191 us against the
        // register-staged form's 207 on the same box (profiles/r04_csr_lds_dma.txt)
        // 64 rows of up to 8 entries fit one 512-entry chunk; up to 12 entries one of 768 (9 KB of LDS per wave: four workgroups per CU), up to 16 one of
        // 1024 (three per CU) - a second chunk per row group is a second serialised DMA wait (rows of 9: 56.6 -> 48.2 us with the wider chunk; beyond
        // 16 entries per row the entry-side register form is as fast or faster: profiles/r04_csr_lds_dma.txt)
        const bool wide = A->nnz_d > 8LL * A->n;
        const bool wider = A->nnz_d > 12LL * A->n;
        long long nd = std::min<long long>(NG, (long long)ctx->num_cu * (wider ? 3 : wide ? 4 : 6));
        const int rd = nd >= 64 ? 1 : 0;
        if (rd) nd = (nd / 8) * 8;
        if (wider) hipLaunchKernelGGL((k_spmv_csr_wave_dma<16, 3, 8>), dim3((unsigned)nd), dim3(256), 0, ctx->stream, A->n, A->d_rowptr, A->d_col, A->d_val, x, y, rd);
        else if (wide) hipLaunchKernelGGL((k_spmv_csr_wave_dma<12, 4, 8>), dim3((unsigned)nd), dim3(256), 0, ctx->stream, A->n, A->d_rowptr, A->d_col, A->d_val, x, y, rd);
        else hipLaunchKernelGGL((k_spmv_csr_wave_dma<8, 6, 8>), dim3((unsigned)nd), dim3(256), 0, ctx->stream, A->n, A->d_rowptr, A->d_col, A->d_val, x, y, rd);
      } else if (rowside) hipLaunchKernelGGL((k_spmv_csr_wave<true, 8>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, A->n, A->d_rowptr, A->d_col, A->d_val, x, y, remap);
      else hipLaunchKernelGGL((k_spmv_csr_wave<false, 8>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, A->n, A->d_rowptr, A->d_col, A->d_val, x, y, remap);
    } else if (A->n >= 2048 && !A->force_csr_vector) {
      const unsigned nb = (unsigned)std::min<long long>(((long long)A->n + 255) / 256, (long long)ctx->num_cu * 8);
      hipLaunchKernelGGL(k_spmv_csr_stream, dim3(nb), dim3(256), 0, ctx->stream, A->n, A->d_rowptr, A->d_col, A->d_val, x, y);
    } else
      launch_spmv<false, false>(ctx->stream, ctx->num_cu, A->lanes_per_row, A->n, A->d_rowptr, A->d_col, A->d_val, x, y, nullptr);
    if (overlap) KS_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_halo, 0));          // also when this rank has no off-diagonal rows: keeps the two streams in step
    if (A->n_orows > 0)
      launch_spmv<true, true>(ctx->stream, ctx->num_cu, 2, A->n_orows, A->o_rowptr, A->o_col, A->o_val, A->ghost, y, A->o_rows);
  }
  KS_HIP(hipGetLastError());
  return KS_SUCCESS;
}

// MatGetDiagonal: entries (i,i) of the diagonal block, 0 where the pattern has none
__global__ void k_diag_csr(int n, const int *__restrict__ rp, const int *__restrict__ col, const double *__restrict__ val, double *__restrict__ d)
{
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  double v = 0.0;
  for (int p = rp[r]; p < rp[r + 1]; p++) if (col[p] == r) v += val[p];
  d[r] = v;
}
__global__ void k_diag_sell(int n, const int *__restrict__ sp, const int *__restrict__ rlen, const int *__restrict__ col, const double *__restrict__ val, double *__restrict__ d)
{
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const long long s = r >> 6, sb = (long long)sp[s] * 64;
  const int w = sp[s + 1] - sp[s], lane = (int)(r & 63);
  double v = 0.0;
  for (int j = 0; j < rlen[r]; j++) { const long long p = sell_pos(sb, w, j, lane); if (col[p] == r) v += val[p]; }
  d[r] = v;
}
int ks_mat_get_diagonal_internal(ks_mat A, double *d)
{
  ks_ctx ctx = A->ctx;
  KS_CHECK(!A->shell_mult, KS_ERR_SUP, "a matrix-free operator has no stored diagonal");
  if (A->n == 0) return KS_SUCCESS;
  if (A->use_sliced || A->use_binned || A->have_cache) { KS_HIP(hipMemcpyAsync(d, A->diag_cache, sizeof(double) * A->n, hipMemcpyDeviceToDevice, ctx->stream)); return KS_SUCCESS; }
  const unsigned nb = (unsigned)((A->n + 255) / 256);
  if (A->use_sell) hipLaunchKernelGGL(k_diag_sell, dim3(nb), dim3(256), 0, ctx->stream, A->n, A->s_ptr, A->s_len, A->s_col, A->s_val, d);
  else hipLaunchKernelGGL(k_diag_csr, dim3(nb), dim3(256), 0, ctx->stream, A->n, A->d_rowptr, A->d_col, A->d_val, d);
  KS_HIP(hipGetLastError());
  return KS_SUCCESS;
}
// MatNorm(A,NORM_INFINITY): max over rows of the sum of |a_ij| (diagonal and off-diagonal blocks)
__global__ void k_rowabs_csr(int n, const int *__restrict__ rp, const double *__restrict__ val, double *__restrict__ out, int accumulate)
{
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  double v = 0.0;
  for (int p = rp[r]; p < rp[r + 1]; p++) v += fabs(val[p]);
  out[r] = accumulate ? out[r] + v : v;
}
__global__ void k_rowabs_sell(int n, const int *__restrict__ sp, const int *__restrict__ rlen, const double *__restrict__ val, double *__restrict__ out)
{
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const long long s = r >> 6, sb = (long long)sp[s] * 64;
  const int w = sp[s + 1] - sp[s], lane = (int)(r & 63);
  double v = 0.0;
  for (int j = 0; j < rlen[r]; j++) v += fabs(val[sell_pos(sb, w, j, lane)]);
  out[r] = v;
}
__global__ void k_rowabs_rows(int nrows, const int *__restrict__ rows, const int *__restrict__ rp, const double *__restrict__ val, double *__restrict__ out)
{
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nrows) return;
  double v = 0.0;
  for (int p = rp[i]; p < rp[i + 1]; p++) v += fabs(val[p]);
  out[rows[i]] += v;
}
int ks_mat_norm_inf_local(ks_mat A, double *val)          // this rank's rows only
{
  ks_ctx ctx = A->ctx;
  if (A->use_sliced || A->use_binned || A->have_cache) { *val = A->norm_inf_cache; return KS_SUCCESS; }     // taken before the CSR arrays were released
  double local = 0.0;
  if (A->n > 0) {
    double *w = nullptr;
    KS_HIP(hipMalloc(&w, sizeof(double) * A->n));
    const unsigned nb = (unsigned)((A->n + 255) / 256);
    if (A->use_sell) hipLaunchKernelGGL(k_rowabs_sell, dim3(nb), dim3(256), 0, ctx->stream, A->n, A->s_ptr, A->s_len, A->s_val, w);
    else hipLaunchKernelGGL(k_rowabs_csr, dim3(nb), dim3(256), 0, ctx->stream, A->n, A->d_rowptr, A->d_val, w, 0);
    if (A->n_orows > 0) hipLaunchKernelGGL(k_rowabs_rows, dim3((unsigned)((A->n_orows + 255) / 256)), dim3(256), 0, ctx->stream, A->n_orows, A->o_rows, A->o_rowptr, A->o_val, w);
    std::vector<double> h(A->n);
    int rc = hipGetLastError() == hipSuccess && hipMemcpyAsync(h.data(), w, sizeof(double) * A->n, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess && ks_sync(ctx) == hipSuccess ? 0 : 1;
    hipFree(w);
    KS_CHECK(!rc, KS_ERR_LIB, "row-sum kernel failed");
    for (double v : h) local = std::max(local, v);
  }
  *val = local;
  return KS_SUCCESS;
}
extern "C" int ks_mat_norm_inf(ks_mat A, double *val)
{
  KS_CHECK(A && val, KS_ERR_ARG_NULL, "NULL argument");
  KS_CHECK(!A->shell_mult, KS_ERR_SUP, "a matrix-free operator has no norm operation");      // MatHasOperation(A,MATOP_NORM) epssolve.c:786
  ks_ctx ctx = A->ctx;
  KS_HIP(hipSetDevice(ctx->device));
  double local = 0.0;
  KS_CALL(ks_mat_norm_inf_local(A, &local));
  if (ctx->comm.size > 1) {
    std::vector<double> all(ctx->comm.size);
    KS_CALL(ks_comm_allgather_host(ctx, &local, (int)sizeof(double), all.data()));
    for (double v : all) local = std::max(local, v);
  }
  *val = local;
  return KS_SUCCESS;
}

extern "C" int ks_mat_get_diagonal(ks_mat A, double *d_dev)
{
  KS_CHECK(A && d_dev, KS_ERR_ARG_NULL, "NULL argument");
  KS_HIP(hipSetDevice(A->ctx->device));
  return ks_mat_get_diagonal_internal(A, d_dev);
}

// MatLoad of a PETSc binary viewer file (the format of share/slepc/datafiles/matrices/*.petsc): big-endian int32
// header {MAT_FILE_CLASSID = 1211216, rows, cols, nnz}, int32 row lengths, int32 column indices, float64 values.
// Each rank keeps the row block PETSC_DECIDE would give it (n/size rows, the first n%size ranks one more).
namespace {
inline uint32_t be32(const unsigned char *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3]; }
inline double be64f(const unsigned char *p) { uint64_t v = 0; for (int i = 0; i < 8; i++) v = (v << 8) | p[i]; double d; memcpy(&d, &v, 8); return d; }
}
extern "C" int ks_mat_load_petsc_binary(ks_ctx ctx, const char *path, ks_mat *out)
{
  KS_CHECK(ctx && path && out, KS_ERR_ARG_NULL, "NULL argument");
  FILE *f = fopen(path, "rb");
  KS_CHECK(f, KS_ERR_FILE_OPEN, "Cannot open file %s", path);
  std::vector<unsigned char> buf;
  fseek(f, 0, SEEK_END); const long sz = ftell(f); fseek(f, 0, SEEK_SET);
  buf.resize(sz > 0 ? (size_t)sz : 0);
  const size_t got = buf.empty() ? 0 : fread(buf.data(), 1, buf.size(), f);
  fclose(f);
  KS_CHECK(got == buf.size() && buf.size() >= 16, KS_ERR_FILE_UNEXPECTED, "Short read on %s", path);
  KS_CHECK(be32(buf.data()) == 1211216u, KS_ERR_FILE_UNEXPECTED, "Not a Mat object in %s (classid %u)", path, be32(buf.data()));
  const long long rows = (int32_t)be32(buf.data() + 4), cols = (int32_t)be32(buf.data() + 8), nnz = (int32_t)be32(buf.data() + 12);
  KS_CHECK(rows >= 0 && cols == rows && nnz >= 0, KS_ERR_FILE_UNEXPECTED, "Unsupported matrix shape %lld x %lld (nnz %lld) in %s: square sparse matrices only", rows, cols, nnz, path);
  KS_CHECK((long long)buf.size() >= 16 + 4 * rows + 12 * nnz, KS_ERR_FILE_UNEXPECTED, "File %s is truncated", path);
  const unsigned char *pl = buf.data() + 16, *pc = pl + 4 * rows, *pv = pc + 4 * nnz;
  std::vector<long long> start(rows + 1, 0);
  for (long long i = 0; i < rows; i++) {
    const long long len = (int32_t)be32(pl + 4 * i);
    KS_CHECK(len >= 0 && len <= cols, KS_ERR_FILE_UNEXPECTED, "Row %lld of %s has length %lld", i, path, len);
    start[i + 1] = start[i] + len;
  }
  KS_CHECK(start[rows] == nnz, KS_ERR_FILE_UNEXPECTED, "Row lengths of %s do not add up to its nnz", path);
  const int size = ctx->comm.size, rank = ctx->comm.rank;
  const long long base = rows / size, rem = rows % size;
  const long long r0 = rank * base + std::min<long long>(rank, rem), nloc = base + (rank < rem ? 1 : 0);
  std::vector<int> rp(nloc + 1), ci((size_t)(start[r0 + nloc] - start[r0]));
  std::vector<double> va(ci.size());
  for (long long i = 0; i <= nloc; i++) rp[i] = (int)(start[r0 + i] - start[r0]);
  for (size_t e = 0; e < ci.size(); e++) { ci[e] = (int32_t)be32(pc + 4 * (start[r0] + e)); va[e] = be64f(pv + 8 * (start[r0] + e)); }
  return ks_mat_create_csr_flags(ctx, (int)nloc, (int)r0, (int)rows, rp.data(), ci.data(), va.data(), KS_MAT_KEEP_CSR, out);
}

// MatCreateShell + MatShellSetOperation(MATOP_MULT) (the matrix-free route of src/eps/tutorials/ex3.c)
extern "C" int ks_mat_create_shell(ks_ctx ctx, int n_local, int row_start, int n_global, ks_shell_mult_fn mult, void *user, ks_mat *out)
{
  KS_CHECK(ctx && out && mult, KS_ERR_ARG_NULL, "NULL argument");
  KS_CHECK(n_local >= 0 && n_global >= n_local && row_start >= 0, KS_ERR_ARG_OUTOFRANGE, "bad sizes n_local=%d n_global=%d row_start=%d", n_local, n_global, row_start);
  ks_mat A = new ks_mat_s(); A->ctx = ctx; A->n = n_local; A->row_start = row_start; A->n_global = n_global;
  A->shell_mult = mult; A->shell_user = user;
  *out = A;
  return KS_SUCCESS;
}

extern "C" int ks_mat_mult(ks_mat A, const double *x_dev, double *y_dev)
{
  KS_CHECK(A && x_dev && y_dev, KS_ERR_ARG_NULL, "NULL argument");
  KS_CHECK(x_dev != y_dev, KS_ERR_ARG_WRONG, "x and y must be different vectors");   // MatMult requirement
  KS_HIP(hipSetDevice(A->ctx->device));
  return ks_mat_mult_internal(A, x_dev, y_dev);
}

// MatMultTranspose: through the transposed matrix, built once (MatTranspose on the host, ks_csr.cpp) and multiplied like any other
int ks_mat_mult_transpose_internal(ks_mat A, const double *x, double *y)
{
  if (A->shell_mult) {
    KS_CHECK(A->shell_mult_t, KS_ERR_SUP, "the shell matrix has no MATOP_MULT_TRANSPOSE (ks_mat_shell_set_mult_transpose)");
    const int rc = A->shell_mult_t(A->shell_user, x, y);
    KS_CHECK(rc == 0, rc > 0 ? rc : KS_ERR_LIB, "the shell matrix's transposed product returned %d", rc);
    return KS_SUCCESS;
  }
  if (!A->At) {
    KS_CHECK(A->ctx->comm.size == 1 && A->n == A->n_global, KS_ERR_SUP, "MatMultTranspose of a row-sharded matrix is not built (the transpose is a redistribution)");
    KS_CHECK(A->keep_csr, KS_ERR_ORDER, "MatMultTranspose builds the transpose from the CSR arrays of the matrix: create it with KS_MAT_KEEP_CSR");
    std::vector<int> rp, col; std::vector<double> val;
    try { ksc::csr_transpose(A->n, A->n_global, A->k_rowptr.data(), A->k_col.data(), A->k_val.data(), rp, col, val); }
    catch (const std::exception &e) { KS_FAIL(KS_ERR_MEM, "MatTranspose on the host: %s", e.what()); }
    KS_CALL(ks_mat_create_csr_flags(A->ctx, A->n, 0, A->n_global, rp.data(), col.data(), val.data(), 0u, &A->At));
  }
  return ks_mat_mult_internal(A->At, x, y);
}
extern "C" int ks_mat_mult_transpose(ks_mat A, const double *x_dev, double *y_dev)
{
  KS_CHECK(A && x_dev && y_dev, KS_ERR_ARG_NULL, "NULL argument");
  KS_CHECK(x_dev != y_dev, KS_ERR_ARG_WRONG, "x and y must be different vectors");
  KS_HIP(hipSetDevice(A->ctx->device));
  return ks_mat_mult_transpose_internal(A, x_dev, y_dev);
}
extern "C" int ks_mat_shell_set_mult_transpose(ks_mat A, ks_shell_mult_fn mult_transpose)
{
  KS_CHECK(A, KS_ERR_ARG_NULL, "A is NULL");
  KS_CHECK(A->shell_mult, KS_ERR_ARG_WRONG, "not a shell matrix");
  A->shell_mult_t = mult_transpose;
  return KS_SUCCESS;
}

extern "C" int ks_mat_mult_host(ks_mat A, const double *x_host, double *y_host)
{
  KS_CHECK(A && x_host && y_host, KS_ERR_ARG_NULL, "NULL argument");
  KS_CHECK(A->ctx->comm.size == 1, KS_ERR_SUP, "host convenience wrapper is single-rank");
  KS_HIP(hipSetDevice(A->ctx->device));
  double *x = nullptr, *y = nullptr;
  KS_HIP(hipMalloc(&x, sizeof(double) * std::max(A->n_global, 1))); KS_HIP(hipMalloc(&y, sizeof(double) * std::max(A->n, 1)));
  KS_HIP(hipMemcpy(x, x_host, sizeof(double) * A->n_global, hipMemcpyHostToDevice));
  int rc = ks_mat_mult_internal(A, x, y);
  if (!rc) { ks_sync(A->ctx); hipMemcpy(y_host, y, sizeof(double) * A->n, hipMemcpyDeviceToHost); }
  hipFree(x); hipFree(y);
  return rc;
}
