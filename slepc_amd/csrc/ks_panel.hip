// Dense tall-skinny panel contractions on the FP64 matrix cores (gfx950 v_mfma_f64_16x16x4_f64):
//   BVDot        M = Y^H X            (src/sys/classes/bv/interface/bvglobal.c:86, BVDot_BLAS_Private bvblas.c:199-233)
//   BVMult       Y = beta*Y + alpha*X*Q   (bvops.c:49,   BVMult_BLAS_Private bvblas.c:24-49)
//   BVMultInPlace V(:,s:e) = V*Q(:,s:e)    (bvops.c:220,  BVMultInPlace_BLAS_Private bvblas.c:74-106)
// These are the only GEMM-shaped operations of the path (n x <=64 panels, n ~ 1e7): 2*n*k1*k2 flop over
// 8n(k1+k2) bytes, i.e. 4-16 flop/byte - still HBM-bound on MI355X but needing ~20 TFLOP/s of FP64, which is
// what the MFMA pipe is for (the VALU stays free for address/staging work).
//
// Layout of one MFMA (wave64): D(16x16) += A(16x4) * B(4x16); lane l supplies A[i=l&15][k=l>>4] and
// B[k=l>>4][j=l&15] (one f64 each) and owns D[row=(l>>4)+4*reg][col=l&15], reg=0..3.
//
// No LDS staging: both kernels load their operands straight into the MFMA fragment layout with 16-byte loads (which rows a
// k-step contracts, and which rows a lane owns, is chosen to fit the loads), so every wave streams its own row chunks with
// many KiB in flight, like the row sweeps.
#include "ks_sweeps.cuh"
#include <algorithm>

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int PB = 256;          // threads per block (4 waves)

// resident blocks per CU of one kernel symbol (cached: the query costs microseconds per call)
int occ_blocks(const void *kernel, size_t lds_bytes)
{
  static thread_local std::vector<std::pair<const void *, int>> cache;
  for (auto &e : cache) if (e.first == kernel) return e.second;
  const int nb = std::min(ks_occupancy(kernel, PB, lds_bytes, 1), 8);
  cache.push_back({kernel, nb});
  return nb;
}

// SAME (both kernels below): Y and X are the same columns (a Gram matrix, BVDot(X,X,M) and the CHOL / SVQB orthogonalisations): the panel
// is read once and serves as both operands - half the HBM bytes, the same MFMA sequence, the same bits.
// ---- BVDot without LDS: loads land directly in MFMA operand layout ------------------------------------------------------------
// M = Y^T X is a sum over rows, so WHICH rows a k-step contracts is free as long as both operands use the same ones. Lane l
// (c = l & 15, q = l >> 4) loads the two rows r0+2q, r0+2q+1 of column c of a 16-column tile with one 16-byte load: the .x
// halves of the wave are exactly the A / B fragments of one v_mfma_f64_16x16x4 (rows r0+{0,2,4,6}), the .y halves those of a
// second one (rows r0+{1,3,5,7}). One wave-instruction reads 16 columns x 64 contiguous bytes; the next row group reads the
// other half of the same 128-byte lines, which is why these are plain (L1-allocating) loads: with the nontemporal hint each
// half line is its own L2 request and the shape reads at 5.4 TB/s, plain at 6.06 (8 columns x 128 B: 6.8, scripts/micro/
// dot_direct.hip, profiles/r02_micro_dot_direct.txt; the MFMAs cost nothing on top). Nothing is staged, nothing synchronises:
// every wave streams its own row chunks with U x (MT+NT) KiB in flight, like the row sweeps (the LDS-staged form it replaces
// could keep only one tile per block in flight and ran at 4 TB/s, profiles/r01e_pmc_panel_kernels_wait_lds.txt). Row order
// inside a block and the block combine are fixed: deterministic.
template <int MT, int NT, bool SAME, int U>
__global__ __launch_bounds__(PB) void k_panel_dot_direct(const double *__restrict__ Y, long long ldy, int my, const double *__restrict__ X, long long ldx, int nx,
                                                          int n, double *__restrict__ partialsM)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];     // MC*NC doubles: the block combine
  constexpr int MC = MT * 16, NC = NT * 16;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, q = lane >> 4;
  const double *py[MT], *px[NT];
#pragma unroll
  for (int mt = 0; mt < MT; mt++) { const int col = mt * 16 + c; py[mt] = Y + (long long)(col < my ? col : my - 1) * ldy + 2 * q; }   // a padded column repeats the last one: it only
#pragma unroll
  for (int nt = 0; nt < NT; nt++) { const int col = nt * 16 + c; px[nt] = X + (long long)(col < nx ? col : nx - 1) * ldx + 2 * q; }   // reaches rows / columns of M that are dropped
  d4 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; a++)
#pragma unroll
    for (int b = 0; b < NT; b++) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
  constexpr int CH = 8 * U;                                        // rows per chunk
  const long long nch = ((long long)n + CH - 1) / CH;
  const long long gw = (long long)blockIdx.x * (PB / 64) + w, GW = (long long)gridDim.x * (PB / 64);
  for (long long ch = gw; ch < nch; ch += GW) {
    const long long r0 = ch * CH;
    double2 a[U][MT], b[U][SAME ? 1 : NT];
    if (r0 + CH <= n) {
#pragma unroll
      for (int u = 0; u < U; u++) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++) a[u][mt] = *reinterpret_cast<const double2 *>(py[mt] + r0 + 8 * u);
        if (!SAME) {
#pragma unroll
          for (int nt = 0; nt < NT; nt++) b[u][nt] = *reinterpret_cast<const double2 *>(px[nt] + r0 + 8 * u);
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; u++) {
        const long long r = r0 + 8 * u + 2 * q;
#pragma unroll
        for (int mt = 0; mt < MT; mt++) { double2 v; v.x = r < n ? py[mt][r0 + 8 * u] : 0.0; v.y = r + 1 < n ? py[mt][r0 + 8 * u + 1] : 0.0; a[u][mt] = v; }
        if (!SAME) {
#pragma unroll
          for (int nt = 0; nt < NT; nt++) { double2 v; v.x = r < n ? px[nt][r0 + 8 * u] : 0.0; v.y = r + 1 < n ? px[nt][r0 + 8 * u + 1] : 0.0; b[u][nt] = v; }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);                             // all loads of the chunk in flight before the first product waits
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
          const double2 bb = SAME ? a[u][nt] : b[u][SAME ? 0 : nt];
          acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][mt].x, bb.x, acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][mt].y, bb.y, acc[mt][nt], 0, 0, 0);
        }
  }
  // combine the four waves in wave order through one MC x NC buffer, then one block partial to HBM
  for (int ww = 0; ww < PB / 64; ww++) {
    if (w == ww) {
#pragma unroll
      for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
          for (int reg = 0; reg < 4; reg++) {
            const int i = mt * 16 + q + 4 * reg, j = nt * 16 + c;
            if (ww == 0) lds[i * NC + j] = acc[mt][nt][reg]; else lds[i * NC + j] += acc[mt][nt][reg];
          }
    }
    __syncthreads();
  }
  for (int e = threadIdx.x; e < MC * NC; e += PB) partialsM[(size_t)blockIdx.x * MC * NC + e] = lds[e];
}

// out (my x nx, column-major, contiguous) = sum over blocks of the padded MC x NC row-major block partials
__global__ void k_reduce_blocks(const double *__restrict__ partialsM, int nblocks, int MC, int NC, int my, int nx, double *__restrict__ out)
{
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= MC * NC) return;
  const int i = e / NC, j = e % NC;
  if (i >= my || j >= nx) return;
  double s = 0.0;
  for (int b = 0; b < nblocks; b++) s += partialsM[(size_t)b * MC * NC + e];
  out[i + (size_t)j * my] = s;
}

// C(:,0:nout) = beta*C + alpha*A(:,0:kin)*Q, computed transposed on the matrix cores and without LDS:
//   D[i][j] = sum_k Q[k][i] * A[row j][k]   (i = output column, j = row slot)
// Lane l (j = l & 15, kq = l >> 4) loads the two rows R+2j, R+2j+1 of column 4*ks + kq with one 16-byte load: a wave-instruction
// reads 4 columns x 256 contiguous bytes (whole 128-byte lines), its .x halves are the B fragment of the k-step for the even
// rows of a 32-row group, its .y halves the one for the odd rows. The Q fragments live in registers for the whole sweep. A lane
// ends up owning rows R+2j, R+2j+1 of output columns i = kq + 4*reg (+16 per tile): one 16-byte store each, 256 contiguous
// bytes per column and instruction. A 32-row group is private to one wave and all its loads are complete before its first
// store (the products need them), so C may alias columns of A (BVMultInPlace). U groups per iteration keep U x KS4 KiB in flight.
template <int KS4, int NT, int U>
__global__ __launch_bounds__(PB) void k_panel_mult_direct(const double *A, long long lda, int n, int kin, const double *__restrict__ Q, int qsk, int qsi, int nout,
                                                           double alpha, double beta, double *C, long long ldc, int vecC)
{
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, j = lane & 15, kq = lane >> 4;
  // A-operand fragments of Q^T: lane supplies Qt[i=j][k=kq] = Q[k][i]
  double qf[KS4][NT];
#pragma unroll
  for (int ks = 0; ks < KS4; ks++)
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
      const int k = ks * 4 + kq, i = nt * 16 + j;
      qf[ks][nt] = (k < kin && i < nout) ? Q[(size_t)k * qsk + (size_t)i * qsi] : 0.0;   // Q(k,i): strides (1,ldq), or (ldq,1) for the transposed form
    }
  const double *pa[KS4];
#pragma unroll
  for (int ks = 0; ks < KS4; ks++) { const int k = ks * 4 + kq; pa[ks] = A + (long long)(k < kin ? k : kin - 1) * lda + 2 * j; }   // a padded column repeats the last one; its Q entries are zero
  constexpr int CH = 32 * U;
  const long long nch = ((long long)n + CH - 1) / CH;
  const long long gw = (long long)blockIdx.x * (PB / 64) + w, GW = (long long)gridDim.x * (PB / 64);
  for (long long ch = gw; ch < nch; ch += GW) {
    const long long r0 = ch * CH;
    double2 v[U][KS4];
    if (r0 + CH <= n) {
#pragma unroll
      for (int u = 0; u < U; u++)
#pragma unroll
        for (int ks = 0; ks < KS4; ks++) v[u][ks] = ksk::ldcol2(pa[ks] + r0 + 32 * u);
    } else {
#pragma unroll
      for (int u = 0; u < U; u++) {
        const long long r = r0 + 32 * u + 2 * j;
#pragma unroll
        for (int ks = 0; ks < KS4; ks++) { double2 t; t.x = r < n ? pa[ks][r0 + 32 * u] : 0.0; t.y = r + 1 < n ? pa[ks][r0 + 32 * u + 1] : 0.0; v[u][ks] = t; }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; u++) {
      d4 e[NT], o[NT];
#pragma unroll
      for (int nt = 0; nt < NT; nt++) { e[nt] = (d4){0.0, 0.0, 0.0, 0.0}; o[nt] = (d4){0.0, 0.0, 0.0, 0.0}; }
#pragma unroll
      for (int ks = 0; ks < KS4; ks++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
          e[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(qf[ks][nt], v[u][ks].x, e[nt], 0, 0, 0);
          o[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(qf[ks][nt], v[u][ks].y, o[nt], 0, 0, 0);
        }
      const long long row = r0 + 32 * u + 2 * j;
      if (row < n) {
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
          for (int reg = 0; reg < 4; reg++) {
            const int col = nt * 16 + kq + 4 * reg;
            if (col < nout) {
              double *c = C + (long long)col * ldc + row;
              if (vecC && row + 1 < n) {
                double2 r;
                if (beta == 0.0) { r.x = alpha * e[nt][reg]; r.y = alpha * o[nt][reg]; }
                else { const double2 old = *reinterpret_cast<const double2 *>(c); r.x = fma(alpha, e[nt][reg], beta * old.x); r.y = fma(alpha, o[nt][reg], beta * old.y); }
                ksk::ks_d2v rv; rv.x = r.x; rv.y = r.y;
                asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(reinterpret_cast<ksk::ks_d2v *>(c)), "v"(rv) : "memory");      // written through: 665 instead of 710 us per restart on the 216^3 workload (nontemporal before)      // written once, re-read much later: keep it out of the way of the input stream
              } else {
                c[0] = (beta == 0.0) ? alpha * e[nt][reg] : fma(alpha, e[nt][reg], beta * c[0]);
                if (row + 1 < n) c[1] = (beta == 0.0) ? alpha * o[nt][reg] : fma(alpha, o[nt][reg], beta * c[1]);
              }
            }
          }
      }
    }
  }
}

} // namespace

// ---- launchers -------------------------------------------------------------------------------------------------
// M_dev (my x nx, column-major contiguous, i.e. M_dev[i + j*my]) = Y(:,0:my)^T X(:,0:nx); my,nx <= 64
int ksp_dot_mfma(ks_bv bv, const double *Y, int ldy, int my, const double *X, int ldx, int nx, int n, double *M_dev)
{
  ks_ctx ctx = bv->ctx;
  KS_CHECK(my >= 1 && my <= 64 && nx >= 1 && nx <= 64, KS_ERR_PLIB, "panel dot %dx%d", my, nx);
  KS_CHECK(ldy % 2 == 0 && ldx % 2 == 0 && (((uintptr_t)Y | (uintptr_t)X) & 15) == 0, KS_ERR_SUP, "MFMA panel kernels need 16-byte aligned columns");
  const int MT = (my + 15) / 16, NT = (nx + 15) / 16, MC = MT * 16, NC = NT * 16;
  const bool same = (Y == X && ldy == ldx && my == nx);
  int grid = 1;
  {
    // row chunks of 8U rows per wave; U so that one wave keeps about 16 KiB in flight whatever the tile counts
    const int streams = same ? MT : MT + NT;
    const int U = streams <= 2 ? 8 : (streams <= 4 ? 4 : 2);
    const long long nch = ((long long)n + 8 * U - 1) / (8 * U);
    const size_t lds_bytes = (size_t)MC * NC * sizeof(double);
    KsProfScope ps(ctx, KS_K_BVDOT, 8.0 * n * (my + nx), 0, same ? 8.0 * n * my : -1.0);
#define DOTD_LAUNCH(M_, N_, S_, U_) do { \
      const int nb = occ_blocks((const void *)k_panel_dot_direct<M_, N_, S_, U_>, lds_bytes); \
      grid = (int)std::max<long long>(1, std::min<long long>((nch + 3) / 4, (long long)ctx->num_cu * nb)); \
      const size_t need = (size_t)grid * MC * NC; \
      if (bv->panel_len < need) { if (bv->panel) hipFree(bv->panel); bv->panel = nullptr; bv->panel_len = 0; KS_HIP(hipMalloc(&bv->panel, need * sizeof(double))); bv->panel_len = need; } \
      hipLaunchKernelGGL((k_panel_dot_direct<M_, N_, S_, U_>), dim3(grid), dim3(PB), lds_bytes, ctx->stream, Y, (long long)ldy, my, X, (long long)ldx, nx, n, bv->panel); } while (0)
#define DOTD_CASE(M_, N_, U_) case (M_) * 8 + (N_): DOTD_LAUNCH(M_, N_, false, U_); break;
    if (same) {
      switch (MT) { case 1: DOTD_LAUNCH(1, 1, true, 8); break; case 2: DOTD_LAUNCH(2, 2, true, 8); break; case 3: DOTD_LAUNCH(3, 3, true, 4); break; default: DOTD_LAUNCH(4, 4, true, 4); break; }
    } else
    switch (MT * 8 + NT) {
      DOTD_CASE(1, 1, 8) DOTD_CASE(1, 2, 4) DOTD_CASE(1, 3, 4) DOTD_CASE(1, 4, 2)
      DOTD_CASE(2, 1, 4) DOTD_CASE(2, 2, 4) DOTD_CASE(2, 3, 2) DOTD_CASE(2, 4, 2)
      DOTD_CASE(3, 1, 4) DOTD_CASE(3, 2, 2) DOTD_CASE(3, 3, 2) DOTD_CASE(3, 4, 2)
      DOTD_CASE(4, 1, 2) DOTD_CASE(4, 2, 2) DOTD_CASE(4, 3, 2) DOTD_CASE(4, 4, 2)
      default: KS_FAIL(KS_ERR_PLIB, "bad tile counts");
    }
#undef DOTD_CASE
#undef DOTD_LAUNCH
    (void)U;
  }
  KS_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_reduce_blocks, dim3((MC * NC + 255) / 256), dim3(256), 0, ctx->stream, bv->panel, grid, MC, NC, my, nx, M_dev);
  KS_HIP(hipGetLastError());
  return KS_SUCCESS;
}

// C(:,0:nout) = beta*C + alpha*A(:,0:kin)*Q(0:kin,0:nout) on the matrix cores; Q on the device with element
// strides (qsk,qsi); kin,nout <= 64; C may alias columns of A (BVMultInPlace).
int ksp_mult_mfma(ks_ctx ctx, int kclass, const double *A, int lda, int n, int kin, const double *Qdev, int qsk, int qsi, int nout,
                  double alpha, double beta, double *C, int ldc)
{
  KS_CHECK(kin >= 1 && kin <= 64 && nout >= 1 && nout <= 64, KS_ERR_PLIB, "panel mult %dx%d", kin, nout);
  const int KS4 = ((kin + 15) / 16) * 4, NT = (nout + 15) / 16;      // k-steps rounded to 16 inner columns
  const int vecC = (ldc % 2 == 0 && (((uintptr_t)C) & 15) == 0) ? 1 : 0;
  int grid = 1;
  KsProfScope ps(ctx, kclass, 8.0 * n * (kin + nout * (beta == 0.0 ? 1 : 2)), KS4 * 4);
#define MULT_CASE(K_, N_, U_) case (K_) * 8 + (N_): { \
    const int nb = occ_blocks((const void *)k_panel_mult_direct<K_, N_, U_>, 0); \
    const long long nch = ((long long)n + 32 * (U_) - 1) / (32 * (U_)); \
    grid = (int)std::max<long long>(1, std::min<long long>((nch + 3) / 4, (long long)ctx->num_cu * nb)); \
    hipLaunchKernelGGL((k_panel_mult_direct<K_, N_, U_>), dim3(grid), dim3(PB), 0, ctx->stream, A, (long long)lda, n, kin, Qdev, qsk, qsi, nout, alpha, beta, C, (long long)ldc, vecC); } break;
  switch (KS4 * 8 + NT) {
    MULT_CASE(4, 1, 4) MULT_CASE(4, 2, 4) MULT_CASE(4, 3, 4) MULT_CASE(4, 4, 2)
    MULT_CASE(8, 1, 2) MULT_CASE(8, 2, 2) MULT_CASE(8, 3, 2) MULT_CASE(8, 4, 2)
    MULT_CASE(12, 1, 2) MULT_CASE(12, 2, 2) MULT_CASE(12, 3, 1) MULT_CASE(12, 4, 1)
    MULT_CASE(16, 1, 1) MULT_CASE(16, 2, 1) MULT_CASE(16, 3, 1) MULT_CASE(16, 4, 1)
    default: KS_FAIL(KS_ERR_PLIB, "bad tile counts");
  }
#undef MULT_CASE
  KS_HIP(hipGetLastError());
  return KS_SUCCESS;
}
