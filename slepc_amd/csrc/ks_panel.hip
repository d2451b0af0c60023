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
// Staging: a row tile (128 rows) of the panels is copied HBM -> LDS with one coalesced 16-byte load per lane
// (a wave-instruction moves 128 rows of ONE column = 1 KiB contiguous), stored column-major with a row pitch of
// 130 doubles. With that pitch the MFMA operand reads - 16 different columns x 4 consecutive rows per wave
// (Dot), or 16 consecutive rows x 4 columns (Mult) - are bank-conflict free for ds_read_b64.
#include "ks_sweeps.cuh"
#include <algorithm>

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int PB = 256;          // threads per block (4 waves)
constexpr int TR = 128;          // rows per tile
constexpr int RP = 130;          // LDS row pitch in doubles (== 2 mod 32: conflict-free Dot operand reads)
constexpr int RPM = 144;         // pitch for the Mult operand reads (== 16 mod 32)

// 128 rows of one column: lane handles rows r0+2*lane, +1 (zero beyond n or for a padding column)
__device__ __forceinline__ double2 load_column(const double *__restrict__ src, long long r0, int n, int lane)
{
  const long long r = r0 + 2 * lane;
  double2 v; v.x = 0.0; v.y = 0.0;
  if (src) {
    if (r + 1 < n) v = ksk::ldcol2(src + r);
    else if (r < n) v.x = src[r];
  }
  return v;
}
__device__ __forceinline__ void stage_column(const double *__restrict__ src, long long r0, int n, double *dst, int lane)
{
  *reinterpret_cast<double2 *>(dst + 2 * lane) = load_column(src, r0, n, lane);
}

// partialsM[b][i*NT16 + j] = sum over the block's rows of Y(r,i) X(r,j)
// SAME: Y and X are the same columns (a Gram matrix, BVDot(X,X,M) and the CHOL / SVQB orthogonalisations): the panel is
// read and staged once and serves as both operands - half the HBM bytes, the same MFMA sequence, the same bits.
template <int MT, int NT, bool SAME>
__global__ __launch_bounds__(PB) void k_panel_dot_mfma(const double *__restrict__ Y, long long ldy, int my, const double *__restrict__ X, long long ldx, int nx,
                                                        int n, double *__restrict__ partialsM)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int MC = MT * 16, NC = NT * 16;
  constexpr int SC = SAME ? MC : MC + NC;               // staged columns
  double *ldsY = lds, *ldsX = SAME ? lds : lds + MC * RP;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  d4 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; a++)
#pragma unroll
    for (int b = 0; b < NT; b++) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
  const long long ntiles = ((long long)n + TR - 1) / TR;
  // software pipeline: the NEXT tile's columns are already in flight (registers) while this tile is multiplied
  constexpr int CW = SC / (PB / 64);                 // columns staged by one wave
  constexpr bool PREF = SC <= 64;                    // register budget: prefetch for up to 64 staged columns
  double2 pre[PREF ? CW : 1];
  auto colsrc = [&](int c) -> const double * {
    return (c < MC) ? (c < my ? Y + (long long)c * ldy : nullptr) : (c - MC < nx ? X + (long long)(c - MC) * ldx : nullptr);
  };
  if (PREF && (long long)blockIdx.x < ntiles) {
#pragma unroll
    for (int q = 0; q < CW; q++) pre[q] = load_column(colsrc(w + q * (PB / 64)), (long long)blockIdx.x * TR, n, lane);
  }
  for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const long long r0 = t * TR;
    if (PREF) {
#pragma unroll
      for (int q = 0; q < CW; q++) *reinterpret_cast<double2 *>(lds + (w + q * (PB / 64)) * RP + 2 * lane) = pre[q];
    } else {
      for (int c = w; c < SC; c += PB / 64) stage_column(colsrc(c), r0, n, lds + c * RP, lane);
    }
    __syncthreads();
    if (PREF && t + gridDim.x < ntiles) {
#pragma unroll
      for (int q = 0; q < CW; q++) pre[q] = load_column(colsrc(w + q * (PB / 64)), (t + gridDim.x) * TR, n, lane);
    }
#pragma unroll
    for (int ks = 0; ks < TR / 4 / 4; ks++) {          // wave w owns rows [32w, 32w+32) of the tile: 8 k-steps of 4 rows
      const int rr = w * (TR / 4) + ks * 4 + (lane >> 4);
      double a[MT], b[NT];
#pragma unroll
      for (int mt = 0; mt < MT; mt++) a[mt] = ldsY[(mt * 16 + (lane & 15)) * RP + rr];
#pragma unroll
      for (int nt = 0; nt < NT; nt++) b[nt] = ldsX[(nt * 16 + (lane & 15)) * RP + rr];
#pragma unroll
      for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
    }
    __syncthreads();
  }
  // combine the four waves through LDS (reuse the staging area), then one block partial to HBM
  double *red = lds;            // [w][MC*NC]
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
      for (int reg = 0; reg < 4; reg++) {
        const int i = mt * 16 + (lane >> 4) + 4 * reg, j = nt * 16 + (lane & 15);
        red[(w * MC + i) * NC + j] = acc[mt][nt][reg];
      }
  __syncthreads();
  for (int e = threadIdx.x; e < MC * NC; e += PB) {
    double s = red[e];
#pragma unroll
    for (int ww = 1; ww < PB / 64; ww++) s += red[ww * MC * NC + e];
    partialsM[(size_t)blockIdx.x * MC * NC + e] = s;
  }
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

// C(:,0:nout) = beta*C + alpha*A(:,0:kin)*Q, computed transposed on the matrix cores:
//   D[i][j] = sum_k Q[k][i] * A[row j][k]   (i = output column, j = row inside a 16-row group)
// so that a store instruction writes, for each of 4 output columns, 16 consecutive rows (one 128-byte line).
// KT4 = ceil(kin/4) k-steps, NT = ceil(nout/16) output column tiles. Q fragments live in registers for the whole
// sweep. The A tile is complete in LDS before any output row of the tile is written: in-place safe.
template <int KS4, int NT>
__global__ __launch_bounds__(PB) void k_panel_mult_mfma(const double *A, long long lda, int n, int kin, const double *__restrict__ Q, int qsk, int qsi, int nout,
                                                         double alpha, double beta, double *C, long long ldc)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];   // KS4*4 columns x RPM
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // A-operand fragments of Q^T: lane supplies Qt[i=l&15][k=l>>4] = Q[k][i]
  double qf[KS4][NT];
#pragma unroll
  for (int ks = 0; ks < KS4; ks++)
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
      const int k = ks * 4 + (lane >> 4), i = nt * 16 + (lane & 15);
      qf[ks][nt] = (k < kin && i < nout) ? Q[(size_t)k * qsk + (size_t)i * qsi] : 0.0;   // Q(k,i): strides (1,ldq), or (ldq,1) for the transposed form
    }
  const long long ntiles = ((long long)n + TR - 1) / TR;
  // software pipeline: the next tile's KS4 columns of this wave are in flight while the current tile is multiplied
  double2 pre[KS4];
  if ((long long)blockIdx.x < ntiles) {
#pragma unroll
    for (int q = 0; q < KS4; q++) { const int c = w + q * (PB / 64); pre[q] = load_column(c < kin ? A + (long long)c * lda : nullptr, (long long)blockIdx.x * TR, n, lane); }
  }
  for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const long long r0 = t * TR;
#pragma unroll
    for (int q = 0; q < KS4; q++) *reinterpret_cast<double2 *>(lds + (w + q * (PB / 64)) * RPM + 2 * lane) = pre[q];
    __syncthreads();
    if (t + gridDim.x < ntiles) {
#pragma unroll
      for (int q = 0; q < KS4; q++) { const int c = w + q * (PB / 64); pre[q] = load_column(c < kin ? A + (long long)c * lda : nullptr, (t + gridDim.x) * TR, n, lane); }
    }
    // wave w owns row groups 2w, 2w+1 (16 rows each) of the 128-row tile
#pragma unroll
    for (int g = 0; g < 2; g++) {
      const int rg = (2 * w + g) * 16;
      d4 acc[NT];
#pragma unroll
      for (int nt = 0; nt < NT; nt++) acc[nt] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < KS4; ks++) {
        const double bfrag = lds[(ks * 4 + (lane >> 4)) * RPM + rg + (lane & 15)];     // B[k][j] = A[row j][k]
#pragma unroll
        for (int nt = 0; nt < NT; nt++) acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(qf[ks][nt], bfrag, acc[nt], 0, 0, 0);
      }
      const long long row = r0 + rg + (lane & 15);
      if (row < n) {
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
          for (int reg = 0; reg < 4; reg++) {
            const int col = nt * 16 + (lane >> 4) + 4 * reg;
            if (col < nout) {
              double *c = C + (long long)col * ldc + row;
              *c = (beta == 0.0) ? alpha * acc[nt][reg] : fma(alpha, acc[nt][reg], beta * (*c));
            }
          }
      }
    }
    __syncthreads();
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
  const long long ntiles = ((long long)n + TR - 1) / TR;
  const bool same = (Y == X && ldy == ldx && my == nx);
  const size_t lds_bytes = std::max<size_t>((size_t)(same ? MC : MC + NC) * RP, (size_t)4 * MC * NC) * sizeof(double);
  const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, (150 * 1024) / lds_bytes));
  const int grid = (int)std::max<long long>(1, std::min<long long>(ntiles, (long long)ctx->num_cu * per_cu));
  const size_t need = (size_t)grid * MC * NC;
  if (bv->panel_len < need) {
    if (bv->panel) hipFree(bv->panel);
    bv->panel = nullptr; bv->panel_len = 0;
    KS_HIP(hipMalloc(&bv->panel, need * sizeof(double)));
    bv->panel_len = need;
  }
  KsProfScope ps(ctx, KS_K_BVDOT, 8.0 * n * (my + nx), 0, same ? 8.0 * n * my : -1.0);
#define DOT_LAUNCH(M_, N_, S_) do { \
    KS_HIP(hipFuncSetAttribute((const void *)k_panel_dot_mfma<M_, N_, S_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)); \
    hipLaunchKernelGGL((k_panel_dot_mfma<M_, N_, S_>), dim3(grid), dim3(PB), lds_bytes, ctx->stream, Y, (long long)ldy, my, X, (long long)ldx, nx, n, bv->panel); } while (0)
#define DOT_CASE(M_, N_) case (M_) * 8 + (N_): DOT_LAUNCH(M_, N_, false); break;
  if (same) {
    switch (MT) { case 1: DOT_LAUNCH(1, 1, true); break; case 2: DOT_LAUNCH(2, 2, true); break; case 3: DOT_LAUNCH(3, 3, true); break; default: DOT_LAUNCH(4, 4, true); break; }
  } else
  switch (MT * 8 + NT) {
    DOT_CASE(1, 1) DOT_CASE(1, 2) DOT_CASE(1, 3) DOT_CASE(1, 4)
    DOT_CASE(2, 1) DOT_CASE(2, 2) DOT_CASE(2, 3) DOT_CASE(2, 4)
    DOT_CASE(3, 1) DOT_CASE(3, 2) DOT_CASE(3, 3) DOT_CASE(3, 4)
    DOT_CASE(4, 1) DOT_CASE(4, 2) DOT_CASE(4, 3) DOT_CASE(4, 4)
    default: KS_FAIL(KS_ERR_PLIB, "bad tile counts");
  }
#undef DOT_CASE
#undef DOT_LAUNCH
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
  const long long ntiles = ((long long)n + TR - 1) / TR;
  const size_t lds_bytes = (size_t)KS4 * 4 * RPM * sizeof(double);
  const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, (150 * 1024) / lds_bytes));
  const int grid = (int)std::max<long long>(1, std::min<long long>(ntiles, (long long)ctx->num_cu * per_cu));
  KsProfScope ps(ctx, kclass, 8.0 * n * (kin + nout * (beta == 0.0 ? 1 : 2)), KS4 * 4);
#define MULT_CASE(K_, N_) case (K_) * 8 + (N_): \
    KS_HIP(hipFuncSetAttribute((const void *)k_panel_mult_mfma<K_, N_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)); \
    hipLaunchKernelGGL((k_panel_mult_mfma<K_, N_>), dim3(grid), dim3(PB), lds_bytes, ctx->stream, A, (long long)lda, n, kin, Qdev, qsk, qsi, nout, alpha, beta, C, (long long)ldc); break;
  switch (KS4 * 8 + NT) {
    MULT_CASE(4, 1) MULT_CASE(4, 2) MULT_CASE(4, 3) MULT_CASE(4, 4)
    MULT_CASE(8, 1) MULT_CASE(8, 2) MULT_CASE(8, 3) MULT_CASE(8, 4)
    MULT_CASE(12, 1) MULT_CASE(12, 2) MULT_CASE(12, 3) MULT_CASE(12, 4)
    MULT_CASE(16, 1) MULT_CASE(16, 2) MULT_CASE(16, 3) MULT_CASE(16, 4)
    default: KS_FAIL(KS_ERR_PLIB, "bad tile counts");
  }
#undef MULT_CASE
  KS_HIP(hipGetLastError());
  return KS_SUCCESS;
}
