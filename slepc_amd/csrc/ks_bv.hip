// BV object and the generic _BVOps slots (everything except the fused Gram-Schmidt, see ks_gs.hip).
// Storage mirrors BVSVEC (src/sys/classes/bv/impls/svec/svec.c): one device array of m*ld doubles,
// column-major; ops act on the active window [l,k) at array+(nc+l)*ld.
#include "ks_sweeps.cuh"
#include <algorithm>

using namespace ksk;

namespace {

__global__ void k_scale(double *__restrict__ x, size_t n, double alpha)
{
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  const size_t stride = (size_t)gridDim.x * blockDim.x * 2;
  for (; i + 1 < n; i += stride) { double2 v = *reinterpret_cast<double2 *>(x + i); v.x *= alpha; v.y *= alpha; *reinterpret_cast<double2 *>(x + i) = v; }
  if (i < n) x[i] *= alpha;
}

// splitmix64 stream shared verbatim with the CPU oracle (oracle/ks_oracle.c orc_random_value)
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x)
{
  x += 0x9E3779B97F4A7C15ULL; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL; x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL; return x ^ (x >> 31);
}
__global__ void k_random_column(double *__restrict__ x, int n, unsigned long long seed, unsigned long long col, unsigned long long row0)
{
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long z = splitmix64(seed ^ splitmix64(col * 0x100000001B3ULL + (row0 + (unsigned long long)i) + 0x12345678ULL * (col + 1)));
  x[i] = (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

__global__ void k_reduce_only(const double *__restrict__ partials, int nblocks, int ncols, double *__restrict__ out)
{
  __shared__ double c_lds[KS_MAX_COLS + 8];
  reduce_partials_to_lds(partials, nblocks, ncols, c_lds);
  if ((int)threadIdx.x < ncols) out[threadIdx.x] = c_lds[threadIdx.x];
}

// per-column partial reductions for norms: mode 0: sum of squares, 1: sum |x|, 2: sum of (x*inv)^2, 3: max |x| ; partials[c*G + b]
template <int MODE>
__global__ __launch_bounds__(SW_BLOCK) void k_colsum(const double *__restrict__ A, long long lda, int n, int ncols, double *__restrict__ partials, double inv = 1.0)
{
  __shared__ double red[SW_WAVES];
  for (int c = 0; c < ncols; c++) {
    double s = 0.0;
    for (long long r = (long long)blockIdx.x * SW_BLOCK + threadIdx.x; r < n; r += (long long)gridDim.x * SW_BLOCK) {
      const double v = A[(long long)c * lda + r];
      if (MODE == 0) s += v * v; else if (MODE == 1) s += fabs(v); else if (MODE == 2) s += (v * inv) * (v * inv); else s = fmax(s, fabs(v));
    }
    if (MODE == 3) { for (int off = 32; off > 0; off >>= 1) s = fmax(s, __shfl_xor(s, off, 64)); } else s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double t = red[0]; for (int w = 1; w < SW_WAVES; w++) t = MODE == 3 ? fmax(t, red[w]) : t + red[w]; partials[(size_t)c * gridDim.x + blockIdx.x] = t; }
    __syncthreads();
  }
}
// infinity norm: max over rows of sum_c |A(r,c)| ; one partial (max) per block
__global__ __launch_bounds__(SW_BLOCK) void k_rowsum_max(const double *__restrict__ A, long long lda, int n, int ncols, double *__restrict__ partials)
{
  __shared__ double red[SW_WAVES];
  double mx = 0.0;
  for (long long r = (long long)blockIdx.x * SW_BLOCK + threadIdx.x; r < n; r += (long long)gridDim.x * SW_BLOCK) {
    double s = 0.0;
    for (int c = 0; c < ncols; c++) s += fabs(A[(long long)c * lda + r]);
    mx = fmax(mx, s);
  }
  for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_down(mx, off, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) { double t = red[0]; for (int w = 1; w < SW_WAVES; w++) t = fmax(t, red[w]); partials[blockIdx.x] = t; }
}

// C(:,0:nout) = beta*C + alpha * A(:,0:kin) * Q(0:kin,0:nout).  One row per thread, the row of A lives in
// registers (so C may alias columns of A: in-place BVMultInPlace), Q is staged in LDS as [nout][KT]
// (zero padded), read with wave-uniform (broadcast) addresses.   bvblas.c:24-49 and :74-106.
template <int KT, bool TRANSQ>
__global__ __launch_bounds__(SW_BLOCK) void k_panel_mult(const double *A, long long lda, int n, int kin, const double *__restrict__ Q, int ldq,
                                                         int nout, double alpha, double beta, double *C, long long ldc)
{
  extern __shared__ __attribute__((aligned(16))) double q_lds[];   // nout*KT
  for (int idx = threadIdx.x; idx < nout * KT; idx += SW_BLOCK) {
    const int j = idx / KT, i = idx % KT;
    q_lds[idx] = (i < kin) ? (TRANSQ ? Q[j + (size_t)i * ldq] : Q[i + (size_t)j * ldq]) : 0.0;
  }
  __syncthreads();
  for (long long r = (long long)blockIdx.x * SW_BLOCK + threadIdx.x; r < n; r += (long long)gridDim.x * SW_BLOCK) {
    double x[KT];
#pragma unroll
    for (int i = 0; i < KT; i++) { const int ii = i < kin ? i : kin - 1; x[i] = A[(long long)ii * lda + r]; }
    for (int j = 0; j < nout; j++) {
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < KT; i++) s = fma(x[i], q_lds[j * KT + i], s);
      double *c = C + (long long)j * ldc + r;
      *c = (beta == 0.0) ? alpha * s : fma(alpha, s, beta * (*c));
    }
  }
}

// B = alpha*A + beta*B   (BVAXPY_BLAS_Private bvblas.c:163-192)
__global__ void k_axpby(const double *__restrict__ A, long long lda, double *__restrict__ B, long long ldb, int n, int ncols, double alpha, double beta)
{
  for (int c = 0; c < ncols; c++)
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (long long)gridDim.x * blockDim.x) {
      double *b = B + (long long)c * ldb + r;
      *b = (beta != 1.0) ? alpha * A[(long long)c * lda + r] + beta * (*b) : fma(alpha, A[(long long)c * lda + r], *b);
    }
}

int sweep_grid(ks_ctx ctx, int n, int vec) { return ks_sweep_grid(ctx, n, vec); }

bool aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

} // namespace

// ---- launchers shared with ks_gs.hip -----------------------------------------------------------
// Grid of a persistent row sweep: exactly the number of blocks that are resident at once (CUs x blocks per CU
// admitted by the kernel's VGPR/LDS budget), so every block streams the same number of tiles with no second
// round of dispatch. Measured on MI355X: +5 % on the register-heavy update kernel versus a fixed 4 blocks/CU.
int ks_sweep_grid_for(ks_ctx ctx, int n, int vec, const void *kernel, int force_per_cu)
{
  const int bmul = force_per_cu;
  int per_cu = 4;
  if (bmul > 0) per_cu = bmul;
  else if (kernel) {
    static thread_local std::vector<std::pair<const void *, int>> cache;       // per thread: contexts may live on different threads
    bool found = false;
    for (auto &e : cache) if (e.first == kernel) { per_cu = e.second; found = true; break; }
    if (!found) {
      const int nb = ks_occupancy(kernel, SW_BLOCK, 0, 4);
      per_cu = std::min(nb, 8);
      cache.push_back({kernel, per_cu});
    }
  }
  long long tile = (long long)SW_BLOCK * vec;
  long long ntiles = ((long long)n + tile - 1) / tile;
  long long g = std::min<long long>(std::max<long long>(ntiles, 1), (long long)ctx->num_cu * per_cu);
  return (int)std::min<long long>(g, KS_MAX_BLOCKS);
}
int ks_sweep_grid(ks_ctx ctx, int n, int vec) { return ks_sweep_grid_for(ctx, n, vec, nullptr, 0); }

int ksk_dot(ks_bv bv, const double *A, int lda, int ncols, const double *y, bool gate)
{
  ks_ctx ctx = bv->ctx;
  KS_CHECK(ncols >= 1 && ncols <= KS_MAX_COLS, KS_ERR_PLIB, "dot sweep with %d columns", ncols);
  const bool v2 = (lda % 2 == 0) && aligned16(A) && aligned16(y);
  int grid = 1;
  // Every wave keeps all ncols column loads of its tile in flight (ncols KiB); about 120 KiB per CU saturate the HBM path,
  // more resident blocks only add concurrent DRAM streams: 1 block per CU at 30 columns (6.34 TB/s, 4 blocks: 6.20), more for
  // narrow sweeps.
  const int dot_per_cu = std::max(1, std::min(4, (30 + ncols - 1) / ncols));
  const KsGsState *g = gate ? bv->gs : nullptr;
  bv->spec.valid = false;                                       // the partials a chained Gram-Schmidt pass would have read are rewritten
  const int plain = ks_basis_is_cache_resident((size_t)(bv->nc + bv->m), (size_t)bv->ld);
  const int rev = bv->sweep_dir; bv->sweep_dir ^= 1;            // snake over the basis with the sweeps before and after (ks_gs.hip launch_update)
  KsProfScope ps(ctx, KS_K_DOT, 8.0 * bv->n * (ncols + (y >= A && y < A + (size_t)ncols * lda ? 0 : 1)), ks_kt_for(ncols));
#define LAUNCH_DOT(KT)                                                                                                                        \
  do {                                                                                                                                        \
    if (v2) { grid = ks_sweep_grid_for(ctx, bv->n, 2, (const void *)k_dot_sweep<KT, 2>, dot_per_cu); bv->last_grid = grid;                                \
      hipLaunchKernelGGL((k_dot_sweep<KT, 2>), dim3(grid), dim3(SW_BLOCK), 0, ctx->stream, A, (long long)lda, bv->n, ncols, y, bv->partials, g, &bv->gs->pgrid, rev, plain); } \
    else { grid = ks_sweep_grid_for(ctx, bv->n, 1, (const void *)k_dot_sweep<KT, 1>, dot_per_cu); bv->last_grid = grid;                                   \
      hipLaunchKernelGGL((k_dot_sweep<KT, 1>), dim3(grid), dim3(SW_BLOCK), 0, ctx->stream, A, (long long)lda, bv->n, ncols, y, bv->partials, g, &bv->gs->pgrid, rev, plain); }   \
  } while (0)
  KS_KT_DISPATCH(ncols, LAUNCH_DOT);
#undef LAUNCH_DOT
  KS_HIP(hipGetLastError());
  return KS_SUCCESS;
}

int ksk_reduce_partials(ks_bv bv, int ncols, double *out_dev)
{
  ks_ctx ctx = bv->ctx;
  hipLaunchKernelGGL(k_reduce_only, dim3(1), dim3(1024), 0, ctx->stream, bv->partials, bv->last_grid, ncols, out_dev);
  KS_HIP(hipGetLastError());
  return KS_SUCCESS;
}

int ksk_multvec(ks_bv bv, const double *A, int lda, int ncols, double alpha, double beta, const double *q_dev, double *y, const KsGsState *gate)
{
  ks_ctx ctx = bv->ctx;
  const bool v2 = (lda % 2 == 0) && aligned16(A) && aligned16(y);
  const int grid = sweep_grid(ctx, bv->n, v2 ? 2 : 1);
  bv->spec.valid = false;                                       // y may be a column of this BV
  KsProfScope ps(ctx, KS_K_UPD, 8.0 * bv->n * (ncols + 2));
  if (v2) hipLaunchKernelGGL((k_multvec<2>), dim3(grid), dim3(SW_BLOCK), 0, ctx->stream, A, (long long)lda, bv->n, ncols, alpha, beta, q_dev, y, gate);
  else hipLaunchKernelGGL((k_multvec<1>), dim3(grid), dim3(SW_BLOCK), 0, ctx->stream, A, (long long)lda, bv->n, ncols, alpha, beta, q_dev, y, gate);
  KS_HIP(hipGetLastError());
  return KS_SUCCESS;
}

int ksk_scale(ks_ctx ctx, double *x, size_t n, double alpha)
{
  if (n == 0 || alpha == 1.0) return KS_SUCCESS;
  KsProfScope ps(ctx, KS_K_SCALE, 16.0 * n);
  if (alpha == 0.0) { KS_HIP(hipMemsetAsync(x, 0, n * sizeof(double), ctx->stream)); return KS_SUCCESS; }   // bvblas.c:271
  if (!aligned16(x)) { hipLaunchKernelGGL(k_scale, dim3(1), dim3(1), 0, ctx->stream, x, (size_t)1, alpha); x++; n--; if (!n) return KS_SUCCESS; }
  size_t blocks = std::min<size_t>((n / 2 + 255) / 256 + 1, (size_t)ctx->num_cu * 8);
  hipLaunchKernelGGL(k_scale, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, x, n, alpha);
  KS_HIP(hipGetLastError());
  return KS_SUCCESS;
}

int ksk_copy(ks_ctx ctx, const double *src, double *dst, size_t n)
{
  if (!n || src == dst) return KS_SUCCESS;
  KsProfScope ps(ctx, KS_K_COPY, 16.0 * n);
  KS_HIP(hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  return KS_SUCCESS;
}

// dst (device) = src (pinned host memory as the device sees it): a few KB of coefficients
__global__ void k_copy_mapped(const double *__restrict__ src, double *__restrict__ dst, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// stage host coefficients into the BV's device scratch (async from pinned memory when they fit)
static int stage_coefs(ks_bv bv, const double *host, size_t len, double **dev)
{
  ks_ctx ctx = bv->ctx;
  if (len > bv->coef_len) {                                 // a wider coefficient block than this BV's own m x m: grow the scratch
    KS_HIP(ks_sync(ctx));
    hipFree(bv->coef); bv->coef = nullptr; bv->coef_len = 0;
    KS_HIP(hipMalloc(&bv->coef, (len + 64) * sizeof(double)));
    bv->coef_len = len + 64;
  }
  if (len <= KS_PINNED_H2D_DOUBLES) {
    // through one of two pinned halves, no host wait
    const int h = ctx->h2d_next; ctx->h2d_next ^= 1;
    double *pin = ctx->h_pinned + KS_PINNED_D2H_BYTES / sizeof(double) + (size_t)h * KS_PINNED_H2D_DOUBLES;
    if (ctx->h_pinned_dev) {
      // A kernel reads the mapped half instead of the runtime's host-to-device copy, whose call alone held the host for tens of microseconds per
      // restart with the GPU idle all the while (config 2, profiles/r03_config2_restart_host_time.txt). The half is free again once the host has
      // seen the stamp of a results kernel enqueued after that reader (every restart cycle waits for one); otherwise wait for the stream.
      if (ctx->h2d_busy[h] && ctx->fetch_waited <= ctx->h2d_seq[h]) KS_HIP(ks_sync(ctx));
      memcpy(pin, host, len * sizeof(double));
      const double *pin_dev = (const double *)ctx->h_pinned_dev + (pin - ctx->h_pinned);
      hipLaunchKernelGGL(k_copy_mapped, dim3((unsigned)std::min<size_t>(8, (len + 511) / 512)), dim3(256), 0, ctx->stream, pin_dev, bv->coef, len);
      KS_HIP(hipGetLastError());
      ctx->h2d_busy[h] = true; ctx->h2d_seq[h] = ctx->fetch_seq;
    } else {
      // the event of a half says when the upload that last used it has left
      if (!ctx->ev_h2d[h]) KS_HIP(hipEventCreateWithFlags(&ctx->ev_h2d[h], hipEventDisableTiming));
      else KS_HIP(hipEventSynchronize(ctx->ev_h2d[h]));
      memcpy(pin, host, len * sizeof(double));
      KS_HIP(hipMemcpyAsync(bv->coef, pin, len * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
      KS_HIP(hipEventRecord(ctx->ev_h2d[h], ctx->stream));
    }
  } else {
    KS_HIP(hipMemcpyAsync(bv->coef, host, len * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    KS_HIP(ks_sync(ctx));                               // pageable source: the caller may reuse it at once
  }
  *dev = bv->coef;
  return KS_SUCCESS;
}

// ---- lifecycle -----------------------------------------------------------------------------------
extern "C" int ks_bv_create(ks_ctx ctx, int n_local, int n_global, int m, int ld, ks_bv *out)
{
  KS_CHECK(ctx && out, KS_ERR_ARG_NULL, "ctx/out is NULL");
  KS_CHECK(n_local >= 0 && n_global >= n_local && m >= 1, KS_ERR_ARG_OUTOFRANGE, "bad sizes n=%d N=%d m=%d", n_local, n_global, m);
  KS_HIP(hipSetDevice(ctx->device));
  ks_bv bv = new ks_bv_s();
  bv->ctx = ctx; bv->n = n_local; bv->N = n_global; bv->m = m; bv->l = 0; bv->k = m; bv->nc = 0;
  bv->fused_gs = !ctx->dbg.no_fused_gs;
  if (ld) {   // BV_SetDefaultLD bvimpl.h:471-484: a user value must be >= n
    if (ld < n_local) { delete bv; KS_FAIL(KS_ERR_USER_INPUT, "The leading dimension %d should be larger or equal to the local number of rows %d", ld, n_local); }
    bv->ld = ld;
  } else bv->ld = std::max(32, ((n_local + 31) / 32) * 32);   // default: 256-byte aligned columns (reference: 16-byte)
  const size_t cells = (size_t)m * bv->ld;
  bv->coef_len = std::max<size_t>((size_t)m * m, (size_t)KS_MAX_COLS * KS_MAX_COLS) + 64;
  hipError_t e = hipMalloc(&bv->array, cells * sizeof(double));
  if (e != hipSuccess) { delete bv; KS_FAIL(KS_ERR_MEM, "hipMalloc of %zu bytes for the BV failed", cells * sizeof(double)); }
  KS_HIP(hipMemsetAsync(bv->array, 0, cells * sizeof(double), ctx->stream));
  KS_HIP(hipMalloc(&bv->buffer, (size_t)m * m * sizeof(double)));
  KS_HIP(hipMemsetAsync(bv->buffer, 0, (size_t)m * m * sizeof(double), ctx->stream));
  KS_HIP(hipMalloc(&bv->partials_base, (size_t)2 * KS_MAX_BLOCKS * KS_PSTRIDE * sizeof(double)));
  bv->partials = bv->partials_base; bv->partials_alt = bv->partials_base + (size_t)KS_MAX_BLOCKS * KS_PSTRIDE;
  KS_HIP(hipMalloc(&bv->coef, bv->coef_len * sizeof(double)));
  KS_HIP(hipMalloc(&bv->hc, (size_t)2 * (m + 8) * sizeof(double)));
  KS_HIP(hipMalloc(&bv->cred, (size_t)KS_PSTRIDE * sizeof(double)));
  KS_HIP(hipMalloc(&bv->cw, (size_t)(m + 8) * sizeof(double)));
  KS_HIP(hipMalloc(&bv->pend, sizeof(double) * 3 * KS_PSTRIDE)); KS_HIP(hipMemsetAsync(bv->pend, 0, sizeof(double) * 3 * KS_PSTRIDE, ctx->stream));
  KS_HIP(hipMalloc(&bv->gs_base, 2 * sizeof(KsGsState)));
  KS_HIP(hipMemsetAsync(bv->gs_base, 0, 2 * sizeof(KsGsState), ctx->stream));
  bv->gs = bv->gs_base; bv->gs_alt = bv->gs_base + 1;
  KS_HIP(hipMalloc(&bv->recs, (size_t)(m + 1) * sizeof(KsStepRec)));
  KS_HIP(hipMemsetAsync(bv->recs, 0, (size_t)(m + 1) * sizeof(KsStepRec), ctx->stream));
  KS_HIP(ks_sync(ctx));
  *out = bv;
  return KS_SUCCESS;
}

extern "C" int ks_bv_destroy(ks_bv bv)
{
  if (!bv) return KS_SUCCESS;
  hipSetDevice(bv->ctx->device);
  ks_sync(bv->ctx);
  hipFree(bv->array); hipFree(bv->own_buffer ? bv->buffer : bv->buffer_own); hipFree(bv->partials_base); hipFree(bv->coef); hipFree(bv->hc); hipFree(bv->cred); hipFree(bv->cw); hipFree(bv->gs_base); hipFree(bv->recs); hipFree(bv->panel); hipFree(bv->Bx); hipFree(bv->pend);
  delete bv;
  return KS_SUCCESS;
}

// BVResize bvbasic.c:190-260: change the number of columns, optionally keeping the first min(m_old, m_new) of them.
// The column storage and everything sized by m are re-created; settings (orthogonalisation, inner-product matrix,
// ownership start) and the leading dimension stay. Active columns are reset to [0, m) as in the reference.
extern "C" int ks_bv_resize(ks_bv bv, int m, int copy)
{
  KS_CHECK(bv, KS_ERR_ARG_NULL, "BV is NULL");
  KS_CHECK(m > 0, KS_ERR_ARG_OUTOFRANGE, "Number of columns %d must be positive", m);
  if (m == bv->m) return KS_SUCCESS;
  KS_CHECK(!bv->nc, KS_ERR_ARG_WRONGSTATE, "Cannot resize a BV with constraints");                    // bvbasic.c:350
  ks_bv nb = nullptr;
  KS_CALL(ks_bv_create(bv->ctx, bv->n, bv->N, m, bv->ld, &nb));
  if (copy) {
    const int mc = std::min(m, bv->m);
    for (int j = 0; j < mc; j++) { int rc = ksk_copy(bv->ctx, ks_bv_col(bv, j), ks_bv_col(nb, j), bv->n); if (rc) { ks_bv_destroy(nb); return rc; } }
    KS_HIP(ks_sync(bv->ctx));
  }
  // swap the storage of the two objects, keep the caller's handle and settings
  std::swap(bv->array, nb->array); std::swap(bv->buffer, nb->buffer); std::swap(bv->own_buffer, nb->own_buffer); std::swap(bv->buffer_own, nb->buffer_own); std::swap(bv->coef, nb->coef); std::swap(bv->coef_len, nb->coef_len);
  std::swap(bv->hc, nb->hc); std::swap(bv->cw, nb->cw); std::swap(bv->recs, nb->recs); std::swap(bv->panel, nb->panel); std::swap(bv->panel_len, nb->panel_len);
  std::swap(bv->m, nb->m);
  bv->l = 0; bv->k = m;
  return ks_bv_destroy(nb);
}

// BVSetRandom bvops.c:380-407: every active column, reproducible (value depends on seed, column and global row only)
extern "C" int ks_bv_set_random(ks_bv bv, uint64_t seed)
{
  KS_CHECK(bv, KS_ERR_ARG_NULL, "BV is NULL");
  for (int j = bv->l; j < bv->k; j++) KS_CALL(ks_bv_set_random_column(bv, j, seed));
  return KS_SUCCESS;
}

// BVInsertVec bvops.c:568 / BVCopyVec bvops.c:484 on device vectors of n_local doubles
extern "C" int ks_bv_insert_vec(ks_bv bv, int j, const double *w_dev)
{
  KS_CHECK(bv && w_dev, KS_ERR_ARG_NULL, "NULL argument");
  KS_CHECK(j >= 0 && j < bv->m, KS_ERR_ARG_OUTOFRANGE, "Argument j has wrong value %d, the number of columns is %d", j, bv->m);
  KS_HIP(hipSetDevice(bv->ctx->device));
  return ksk_copy(bv->ctx, w_dev, ks_bv_col(bv, j), bv->n);
}
extern "C" int ks_bv_copy_vec(ks_bv bv, int j, double *w_dev)
{
  KS_CHECK(bv && w_dev, KS_ERR_ARG_NULL, "NULL argument");
  KS_CHECK(j >= 0 && j < bv->m, KS_ERR_ARG_OUTOFRANGE, "Argument j has wrong value %d, the number of columns is %d", j, bv->m);
  KS_HIP(hipSetDevice(bv->ctx->device));
  return ksk_copy(bv->ctx, ks_bv_col(bv, j), w_dev, bv->n);
}

// BVInsertVecs bvfunc.c:331-375: copy the device vectors W[0..*m) into columns s.., one at a time; with orth each is
// orthogonalised against everything before it (constraints included), normalised, or dropped when dependent.
extern "C" int ks_bv_insert_vecs(ks_bv bv, int s, int *m, const double *const *W_dev, int orth)
{
  KS_CHECK(bv && m, KS_ERR_ARG_NULL, "NULL argument");
  if (!*m) return KS_SUCCESS;
  KS_CHECK(*m > 0, KS_ERR_ARG_OUTOFRANGE, "Number of vectors (given %d) cannot be negative", *m);
  KS_CHECK(W_dev, KS_ERR_ARG_NULL, "NULL argument");
  KS_CHECK(s >= 0 && s < bv->m, KS_ERR_ARG_OUTOFRANGE, "Argument s has wrong value %d, should be between 0 and %d", s, bv->m - 1);
  KS_CHECK(s + *m <= bv->m, KS_ERR_ARG_OUTOFRANGE, "Too many vectors provided, there is only room for %d", bv->m);
  KS_HIP(hipSetDevice(bv->ctx->device));
  int ndep = 0;
  for (int i = 0; i < *m; i++) {
    KS_CHECK(W_dev[i], KS_ERR_ARG_NULL, "vector %d is NULL", i);
    KS_CALL(ksk_copy(bv->ctx, W_dev[i], ks_bv_col(bv, s + i - ndep), bv->n));
    if (orth) {
      double norm = 0.0; int lindep = 0;
      KS_CALL(ks_bv_orthogonalizecolumn(bv, s + i - ndep, nullptr, &norm, &lindep));
      if (norm == 0.0 || lindep) ndep++;                                             // "Removing linearly dependent vector"
      else KS_CALL(ks_bv_scalecolumn(bv, s + i - ndep, 1.0 / norm));
    }
  }
  *m -= ndep;
  return KS_SUCCESS;
}

// BVInsertConstraints bvfunc.c:411-439. DESTRUCTIVE: the storage is re-created with *nc + m columns (BVResize without
// copy), the vectors are orthonormalised into its leading columns, which from then on are columns -nc..-1: every
// Gram-Schmidt sweep starts there, everything else keeps addressing the m regular columns.
extern "C" int ks_bv_insert_constraints(ks_bv bv, int *nc, const double *const *C_dev)
{
  KS_CHECK(bv && nc, KS_ERR_ARG_NULL, "NULL argument");
  if (!*nc) return KS_SUCCESS;
  KS_CHECK(*nc > 0, KS_ERR_ARG_OUTOFRANGE, "Number of constraints (given %d) cannot be negative", *nc);
  KS_CHECK(C_dev, KS_ERR_ARG_NULL, "NULL argument");
  KS_CHECK(!bv->nc, KS_ERR_ARG_WRONGSTATE, "Constraints already present in this BV object");
  const int msave = bv->m;
  KS_CALL(ks_bv_resize(bv, *nc + msave, 0));
  int rc = ks_bv_insert_vecs(bv, 0, nc, C_dev, 1);
  if (rc) { ks_bv_resize(bv, msave, 0); return rc; }
  bv->nc = *nc; bv->m = msave; bv->l = 0; bv->k = msave;
  return KS_SUCCESS;
}

// BVSetNumConstraints bvbasic.c:260-294: fewer constraints shift the regular columns down and turn the freed ones
// into regular columns at the end (EPSSolve drops its deflation space this way, epssolve.c:201-205)
extern "C" int ks_bv_set_num_constraints(ks_bv bv, int nc)
{
  KS_CHECK(bv, KS_ERR_ARG_NULL, "BV is NULL");
  KS_CHECK(nc >= 0, KS_ERR_ARG_OUTOFRANGE, "Number of constraints (given %d) cannot be negative", nc);
  const int diff = nc - bv->nc, total = bv->nc + bv->m;
  if (!diff) return KS_SUCCESS;
  KS_CHECK(total - nc > 0, KS_ERR_ARG_OUTOFRANGE, "Not enough columns for the given nc value");
  KS_HIP(hipSetDevice(bv->ctx->device));
  if (diff < 0) for (int i = 0; i < bv->m; i++) KS_CALL(ksk_copy(bv->ctx, ks_bv_col(bv, i), ks_bv_col(bv, i + diff), bv->n));
  bv->nc = nc; bv->m = total - nc;
  bv->l = std::min(bv->l, bv->m); bv->k = std::min(bv->k, bv->m);
  return KS_SUCCESS;
}
extern "C" int ks_bv_get_num_constraints(ks_bv bv, int *nc) { KS_CHECK(bv && nc, KS_ERR_ARG_NULL, "NULL argument"); *nc = bv->nc; return KS_SUCCESS; }

extern "C" int ks_bv_duplicate(ks_bv bv, ks_bv *out)
{
  KS_CHECK(bv && out, KS_ERR_ARG_NULL, "NULL argument");
  KS_CALL(ks_bv_create(bv->ctx, bv->n, bv->N, bv->m, bv->ld, out));
  (*out)->orthog_type = bv->orthog_type; (*out)->orthog_ref = bv->orthog_ref; (*out)->orthog_eta = bv->orthog_eta;
  (*out)->l = bv->l; (*out)->k = bv->k; (*out)->row_start = bv->row_start;
  return KS_SUCCESS;
}

extern "C" int ks_bv_get_sizes(ks_bv bv, int *n, int *N, int *m, int *ld)
{
  KS_CHECK(bv, KS_ERR_ARG_NULL, "BV is NULL");
  if (n) *n = bv->n; if (N) *N = bv->N; if (m) *m = bv->m; if (ld) *ld = bv->ld;
  return KS_SUCCESS;
}

extern "C" int ks_bv_set_ownership_start(ks_bv bv, int row_start)
{
  KS_CHECK(bv, KS_ERR_ARG_NULL, "BV is NULL");
  KS_CHECK(row_start >= 0 && row_start + bv->n <= bv->N, KS_ERR_ARG_OUTOFRANGE, "rows [%d,%d) outside the global size %d", row_start, row_start + bv->n, bv->N);
  bv->row_start = row_start;
  return KS_SUCCESS;
}

extern "C" int ks_bv_set_active_columns(ks_bv bv, int l, int k)   // bvbasic.c:421-440
{
  KS_CHECK(bv, KS_ERR_ARG_NULL, "BV is NULL");
  if (k < 0) k = bv->m;     // PETSC_DETERMINE
  if (l < 0) l = 0;
  KS_CHECK(k <= bv->m, KS_ERR_ARG_OUTOFRANGE, "Illegal value of k. Must be between 0 and m");
  KS_CHECK(l <= k, KS_ERR_ARG_OUTOFRANGE, "Illegal value of l. Must be less than k");
  bv->l = l; bv->k = k;
  return KS_SUCCESS;
}

extern "C" int ks_bv_get_active_columns(ks_bv bv, int *l, int *k)
{
  KS_CHECK(bv, KS_ERR_ARG_NULL, "BV is NULL");
  if (l) *l = bv->l; if (k) *k = bv->k;
  return KS_SUCCESS;
}

extern "C" int ks_bv_set_orthogonalization(ks_bv bv, int type, int refine, double eta)   // bvorthog / bvbasic.c BVSetOrthogonalization
{
  KS_CHECK(bv, KS_ERR_ARG_NULL, "BV is NULL");
  KS_CHECK(type == KS_BV_ORTHOG_CGS || type == KS_BV_ORTHOG_MGS, KS_ERR_ARG_WRONG, "Unknown orthogonalization type");
  KS_CHECK(refine >= 0 && refine <= 2, KS_ERR_ARG_WRONG, "Unknown refinement type");
  if (eta > 0.0) { KS_CHECK(eta <= 1.0, KS_ERR_ARG_OUTOFRANGE, "Invalid eta value"); if (eta != bv->orthog_eta) bv->spec.valid = false; bv->orthog_eta = eta; }
  if (type != bv->orthog_type || refine != bv->orthog_ref) bv->spec.valid = false;      // a chained Gram-Schmidt pass was predicted under the old policy
  bv->orthog_type = type; bv->orthog_ref = refine;
  return KS_SUCCESS;
}

extern "C" int ks_bv_get_array(ks_bv bv, double **dev) { KS_CHECK(bv && dev, KS_ERR_ARG_NULL, "NULL argument"); *dev = bv->array; bv->spec.valid = false; return KS_SUCCESS; }   // a writable view: whatever a Gram-Schmidt pass left for its successor no longer counts
extern "C" int ks_bv_get_buffer(ks_bv bv, double **dev) { KS_CHECK(bv && dev, KS_ERR_ARG_NULL, "NULL argument"); *dev = bv->buffer; return KS_SUCCESS; }
// BVSetBufferVec bvbasic.c:720 on a raw device array of (nc+m)*m doubles: the adapter hands over the array of the reference's
// bv->buffer Vec, so that BV_CleanCoefficients / BV_SetValue / BV_StoreCoefficients of the caller and the kernels of this library
// work on the same memory. NULL returns to the library's own allocation.
extern "C" int ks_bv_set_buffer(ks_bv bv, double *dev)
{
  KS_CHECK(bv, KS_ERR_ARG_NULL, "BV is NULL");
  if (dev != bv->buffer) bv->spec.valid = false;
  if (dev) { if (bv->own_buffer) { bv->buffer_own = bv->buffer; bv->own_buffer = false; } bv->buffer = dev; }
  else if (!bv->own_buffer) { bv->buffer = bv->buffer_own; bv->buffer_own = nullptr; bv->own_buffer = true; }
  return KS_SUCCESS;
}
// State-only mirror of the reference's BV fields nc / m (BVSetNumConstraints bvbasic.c:260-297 changes them, and shifts the columns,
// without going through an ops slot): no data moves here. nc + m must equal the number of allocated columns.
extern "C" int ks_bv_set_layout(ks_bv bv, int nc, int m)
{
  KS_CHECK(bv, KS_ERR_ARG_NULL, "BV is NULL");
  KS_CHECK(nc >= 0 && m > 0 && nc + m == bv->nc + bv->m, KS_ERR_ARG_OUTOFRANGE, "nc=%d, m=%d do not add up to the %d allocated columns", nc, m, bv->nc + bv->m);
  if (nc != bv->nc) bv->spec.valid = false;
  bv->nc = nc; bv->m = m;
  bv->l = std::min(bv->l, bv->m); bv->k = std::min(bv->k, bv->m);
  return KS_SUCCESS;
}
extern "C" int ks_bv_get_column(ks_bv bv, int j, double **dev)
{
  KS_CHECK(bv && dev, KS_ERR_ARG_NULL, "NULL argument");
  KS_CHECK(j < bv->m, KS_ERR_ARG_OUTOFRANGE, "You requested column %d but only columns 0 to %d are available", j, bv->m - 1);
  KS_CHECK(j >= -bv->nc, KS_ERR_ARG_OUTOFRANGE, "You requested constraint %d but only %d are available", -j, bv->nc);       // bvbasic.c BVGetColumn: negative = constraint
  *dev = ks_bv_col(bv, j);
  bv->spec.valid = false;
  return KS_SUCCESS;
}

extern "C" int ks_bv_set_column_host(ks_bv bv, int j, const double *host)
{
  KS_CHECK(bv && host, KS_ERR_ARG_NULL, "NULL argument");
  KS_CHECK(j >= 0 && j < bv->m, KS_ERR_ARG_OUTOFRANGE, "column %d out of range", j);
  bv->spec.valid = false;
  KS_HIP(hipSetDevice(bv->ctx->device));
  KS_HIP(hipMemcpyAsync(ks_bv_col(bv, j), host, sizeof(double) * bv->n, hipMemcpyHostToDevice, bv->ctx->stream));
  KS_HIP(ks_sync(bv->ctx));
  return KS_SUCCESS;
}

extern "C" int ks_bv_get_column_host(ks_bv bv, int j, double *host)
{
  KS_CHECK(bv && host, KS_ERR_ARG_NULL, "NULL argument");
  KS_CHECK(j >= -bv->nc && j < bv->m, KS_ERR_ARG_OUTOFRANGE, "column %d out of range", j);
  KS_HIP(hipSetDevice(bv->ctx->device));
  KS_HIP(hipMemcpyAsync(host, ks_bv_col(bv, j), sizeof(double) * bv->n, hipMemcpyDeviceToHost, bv->ctx->stream));
  KS_HIP(ks_sync(bv->ctx));
  return KS_SUCCESS;
}

extern "C" int ks_bv_get_buffer_host(ks_bv bv, double *host)
{
  KS_CHECK(bv && host, KS_ERR_ARG_NULL, "NULL argument");
  KS_HIP(hipSetDevice(bv->ctx->device));
  KS_HIP(hipMemcpyAsync(host, bv->buffer, sizeof(double) * (bv->nc + bv->m) * bv->m, hipMemcpyDeviceToHost, bv->ctx->stream));
  KS_HIP(ks_sync(bv->ctx));
  return KS_SUCCESS;
}

extern "C" int ks_bv_set_random_column(ks_bv bv, int j, uint64_t seed)   // BVSetRandomColumn bvops.c:482, reproducible variant :368-376
{
  KS_CHECK(bv, KS_ERR_ARG_NULL, "BV is NULL");
  KS_CHECK(j >= 0 && j < bv->m, KS_ERR_ARG_OUTOFRANGE, "Argument j has wrong value %d, the number of columns is %d", j, bv->m);
  KS_HIP(hipSetDevice(bv->ctx->device));
  if (bv->n) hipLaunchKernelGGL(k_random_column, dim3((bv->n + 255) / 256), dim3(256), 0, bv->ctx->stream, ks_bv_col(bv, j), bv->n, (unsigned long long)seed, (unsigned long long)j, (unsigned long long)bv->row_start);
  KS_HIP(hipGetLastError());
  return KS_SUCCESS;
}

// ---- ops->mult / multvec / multinplace -----------------------------------------------------------
static int panel_mult64(ks_ctx ctx, int kclass, const double *A, int lda, int n, int kin, const double *Qdev, int ldq, bool transq,
                        int nout, double alpha, double beta, double *C, int ldc);
// C(:,0:nout) = beta*C + alpha*A(:,0:kin)*Q for any kin, nout: 64 x 64 blocks of Q on the panel kernels, accumulated over the
// inner blocks. When C aliases columns of A (BVMultInPlace) and the product is wider than one block, the result goes to a
// temporary panel first (the reference's BVMultInPlace_BLAS_Private works through a workspace too, bvblas.c:74-106).
static int panel_mult(ks_ctx ctx, int kclass, const double *A, int lda, int n, int kin, const double *Qdev, int ldq, bool transq,
                      int nout, double alpha, double beta, double *C, int ldc)
{
  if (n == 0 || nout == 0) return KS_SUCCESS;
  KS_CHECK(kin >= 1, KS_ERR_PLIB, "panel product with %d inner columns", kin);
  if (kin <= KS_MAX_COLS && nout <= 64) return panel_mult64(ctx, kclass, A, lda, n, kin, Qdev, ldq, transq, nout, alpha, beta, C, ldc);
  const bool alias = (C < A + (size_t)kin * lda) && (A < C + (size_t)nout * ldc);
  double *T = C; int ldt = ldc;
  if (alias) {
    KS_CHECK(beta == 0.0, KS_ERR_PLIB, "in-place wide panel product with beta != 0");
    ldt = ((n + 31) / 32) * 32;
    KS_HIP(hipMalloc(&T, sizeof(double) * (size_t)ldt * nout));
  }
  int rc = KS_SUCCESS;
  for (int jo = 0; jo < nout && !rc; jo += 64) {
    const int nb = std::min(64, nout - jo);
    for (int ki = 0; ki < kin && !rc; ki += KS_MAX_COLS) {
      const int kb = std::min(KS_MAX_COLS, kin - ki);
      const double *qb = transq ? Qdev + jo + (size_t)ki * ldq : Qdev + ki + (size_t)jo * ldq;
      rc = panel_mult64(ctx, kclass, A + (size_t)ki * lda, lda, n, kb, qb, ldq, transq, nb, alpha, ki == 0 ? beta : 1.0, T + (size_t)jo * ldt, ldt);
    }
  }
  if (alias) {
    for (int j = 0; j < nout && !rc; j++) rc = ksk_copy(ctx, T + (size_t)j * ldt, C + (size_t)j * ldc, n);
    ks_sync(ctx);
    hipFree(T);
  }
  return rc;
}

static int panel_mult64(ks_ctx ctx, int kclass, const double *A, int lda, int n, int kin, const double *Qdev, int ldq, bool transq,
                        int nout, double alpha, double beta, double *C, int ldc)
{
  if (n == 0 || nout == 0) return KS_SUCCESS;
  KS_CHECK(kin >= 1 && kin <= KS_MAX_COLS, KS_ERR_SUP, "panel product with %d inner columns (max %d)", kin, KS_MAX_COLS);
  const bool use_mfma = !ctx->dbg.no_mfma;
  if (use_mfma && nout <= 64 && lda % 2 == 0 && aligned16(A))       // FP64 matrix cores; the VALU kernel below is the unaligned fallback
    return ksp_mult_mfma(ctx, kclass, A, lda, n, kin, Qdev, transq ? ldq : 1, transq ? 1 : ldq, nout, alpha, beta, C, ldc);
  const int grid = (int)std::min<long long>(((long long)n + SW_BLOCK - 1) / SW_BLOCK, (long long)ctx->num_cu * 8);
  KsProfScope ps(ctx, kclass, 8.0 * n * (kin + nout * (beta == 0.0 ? 1 : 2)));
#define LAUNCH_PM(KT)                                                                                                                      \
  do {                                                                                                                                     \
    const size_t sh = (size_t)nout * KT * sizeof(double);                                                                                  \
    if (transq) hipLaunchKernelGGL((k_panel_mult<KT, true>), dim3(grid), dim3(SW_BLOCK), sh, ctx->stream, A, (long long)lda, n, kin, Qdev, ldq, nout, alpha, beta, C, (long long)ldc); \
    else hipLaunchKernelGGL((k_panel_mult<KT, false>), dim3(grid), dim3(SW_BLOCK), sh, ctx->stream, A, (long long)lda, n, kin, Qdev, ldq, nout, alpha, beta, C, (long long)ldc);       \
  } while (0)
  KS_KT_DISPATCH(kin, LAUNCH_PM);
#undef LAUNCH_PM
  KS_HIP(hipGetLastError());
  return KS_SUCCESS;
}

extern "C" int ks_bv_mult(ks_bv Y, double alpha, double beta, ks_bv X, const double *Q, int ldq)   // bvops.c:49, svec.c:17-36
{
  KS_CHECK(Y && X, KS_ERR_ARG_NULL, "BV is NULL");
  KS_CHECK(X != Y, KS_ERR_ARG_WRONG, "X and Y arguments must be different");
  KS_CHECK(X->n == Y->n, KS_ERR_ARG_INCOMP, "Mismatching local dimension X %d, Y %d", X->n, Y->n);
  ks_ctx ctx = Y->ctx;
  KS_HIP(hipSetDevice(ctx->device));
  const int ny = Y->k - Y->l, kx = X->k - X->l;
  double *py = Y->array + (size_t)(Y->nc + Y->l) * Y->ld; const double *px = X->array + (size_t)(X->nc + X->l) * X->ld;
  if (ny <= 0) return KS_SUCCESS;
  if (!Q) {
    KS_CHECK(kx >= ny, KS_ERR_ARG_SIZ, "X has fewer active columns than Y");
    if (!Y->n) return KS_SUCCESS;
    KsProfScope ps(ctx, KS_K_MULT, 24.0 * Y->n * ny);
    const int grid = (int)std::min<long long>(((long long)Y->n + 255) / 256, (long long)ctx->num_cu * 8);
    hipLaunchKernelGGL(k_axpby, dim3(grid), dim3(256), 0, ctx->stream, px, (long long)X->ld, py, (long long)Y->ld, Y->n, ny, alpha, beta);
    KS_HIP(hipGetLastError());
    return KS_SUCCESS;
  }
  KS_CHECK(ldq >= X->k, KS_ERR_ARG_SIZ, "Mat argument has %d rows, should have at least %d", ldq, X->k);
  if (kx <= 0) { if (beta != 1.0) for (int j = 0; j < ny; j++) KS_CALL(ksk_scale(ctx, py + (size_t)j * Y->ld, Y->n, beta)); return KS_SUCCESS; }
  // stage the (X->k) x (Y->k) leading block of Q on the device (bvimpl.h:565-586 does the same per call)
  const size_t qlen = (size_t)ldq * Y->k;
  double *qdev = nullptr;
  KS_CALL(stage_coefs(Y, Q, qlen, &qdev));
  return panel_mult(ctx, KS_K_MULT, px, X->ld, Y->n, kx, qdev + (size_t)Y->l * ldq + X->l, ldq, false, ny, alpha, beta, py, Y->ld);
}

extern "C" int ks_bv_multvec(ks_bv X, double alpha, double beta, double *y_dev, const double *q)   // bvops.c:110, svec.c:38-52
{
  KS_CHECK(X && y_dev, KS_ERR_ARG_NULL, "NULL argument");
  ks_ctx ctx = X->ctx;
  KS_HIP(hipSetDevice(ctx->device));
  const int kx = X->k - X->l;
  const double *qdev = X->buffer;            // q==NULL: coefficients are in the buffer Vec (svec.c:46)
  if (q && kx > 0) { double *tmp; KS_CALL(stage_coefs(X, q, (size_t)kx, &tmp)); qdev = tmp; }
  if (kx <= 0) { if (beta != 1.0) return ksk_scale(ctx, y_dev, X->n, beta); return KS_SUCCESS; }
  const double *A = X->array + (size_t)(X->nc + X->l) * X->ld;
  for (int c0 = 0; c0 < kx; c0 += KS_MAX_COLS) {
    const int nc = std::min(KS_MAX_COLS, kx - c0);
    KS_CALL(ksk_multvec(X, A + (size_t)c0 * X->ld, X->ld, nc, alpha, c0 == 0 ? beta : 1.0, qdev + c0, y_dev));
  }
  return KS_SUCCESS;
}

extern "C" int ks_bv_multcolumn(ks_bv X, double alpha, double beta, int j, const double *q)   // bvops.c:165-198
{
  KS_CHECK(X, KS_ERR_ARG_NULL, "BV is NULL");
  KS_CHECK(j >= 0, KS_ERR_ARG_OUTOFRANGE, "Index j must be non-negative");
  KS_CHECK(j < X->m, KS_ERR_ARG_OUTOFRANGE, "Index j=%d but BV only has %d columns", j, X->m);
  const int ksave = X->k;
  X->k = j;
  int rc = ks_bv_multvec(X, alpha, beta, ks_bv_col(X, j), q);
  X->k = ksave;
  return rc;
}

static int multinplace(ks_bv V, const double *Q, int ldq, int s, int e, bool trans)   // bvops.c:220-250, svec.c:54-87
{
  KS_CHECK(V && Q, KS_ERR_ARG_NULL, "NULL argument");
  KS_CHECK(s >= V->l && s <= V->m, KS_ERR_ARG_OUTOFRANGE, "Argument s has wrong value %d, should be between %d and %d", s, V->l, V->m);
  KS_CHECK(e >= V->l && e <= V->m, KS_ERR_ARG_OUTOFRANGE, "Argument e has wrong value %d, should be between %d and %d", e, V->l, V->m);
  V->spec.valid = false;
  KS_CHECK(ldq >= (trans ? e : V->k), KS_ERR_ARG_SIZ, "Mat argument has %d rows, should have at least %d", ldq, trans ? e : V->k);
  if (s >= e || !V->n) return KS_SUCCESS;
  ks_ctx ctx = V->ctx;
  KS_HIP(hipSetDevice(ctx->device));
  const int kin = V->k - V->l;
  KS_CHECK(kin >= 1, KS_ERR_ARG_WRONGSTATE, "no active columns");
  // Q block needed: rows l..k-1, cols s..e-1 (or transposed); stage the leading max(k,e) x max(k,e) block
  const int ncolsq = trans ? V->k : e;
  double *qdev = nullptr;
  KS_CALL(stage_coefs(V, Q, (size_t)ldq * ncolsq, &qdev));
  double *A = V->array + (size_t)(V->nc + V->l) * V->ld;
  const double *B = qdev + (size_t)V->l * ldq + V->l;
  const int ss = s - V->l, ee = e - V->l;
  const double *pb = trans ? B + ss : B + (size_t)ss * ldq;
  return panel_mult(ctx, KS_K_MULTINPLACE, A, V->ld, V->n, kin, pb, ldq, trans, ee - ss, 1.0, 0.0, A + (size_t)ss * V->ld, V->ld);
}
extern "C" int ks_bv_multinplace(ks_bv V, const double *Q, int ldq, int s, int e) { return multinplace(V, Q, ldq, s, e, false); }
extern "C" int ks_bv_multinplace_trans(ks_bv V, const double *Q, int ldq, int s, int e) { return multinplace(V, Q, ldq, s, e, true); }

// ---- ops->dot / dotvec -----------------------------------------------------------------------------
// BVSetMatrix bvfunc.c:200-250 with indef = PETSC_FALSE: inner products become y^H B x
extern "C" int ks_bv_set_matrix(ks_bv bv, ks_mat B)
{
  KS_CHECK(bv, KS_ERR_ARG_NULL, "BV is NULL");
  if (B) {
    KS_CHECK(B->n == bv->n && B->n_global == bv->N, KS_ERR_ARG_INCOMP, "Mismatching dimensions of the inner-product matrix (%d) and the BV (%d)", B->n, bv->n);
    KS_HIP(hipSetDevice(bv->ctx->device));
    if (!bv->Bx) KS_HIP(hipMalloc(&bv->Bx, sizeof(double) * std::max(bv->n, 1)));
  }
  bv->matrix = B;
  return KS_SUCCESS;
}
extern "C" int ks_bv_get_matrix(ks_bv bv, ks_mat *B) { KS_CHECK(bv && B, KS_ERR_ARG_NULL, "NULL argument"); *B = bv->matrix; return KS_SUCCESS; }

int ksb_ipmatmult(ks_bv bv, const double *x, const double **z)      // BV_IPMatMult bvimpl.h:147-158 (recomputed on every use)
{
  if (!bv->matrix) { *z = x; return KS_SUCCESS; }
  KS_CALL(ks_mat_mult_internal(bv->matrix, x, bv->Bx));
  *z = bv->Bx;
  return KS_SUCCESS;
}

int ksb_norm_b(ks_bv bv, const double *x, double *val)              // BVNorm_Private bvglobal.c:444-453 + BV_SafeSqrt bvimpl.h:121-141
{
  ks_ctx ctx = bv->ctx;
  const double *z;
  KS_CALL(ksb_ipmatmult(bv, x, &z));
  double p = 0.0;
  if (bv->n > 0) { KS_CALL(ksk_dot(bv, z, bv->ld, 1, x, false)); KS_CALL(ksk_reduce_partials(bv, 1, bv->coef)); }
  else KS_HIP(hipMemsetAsync(bv->coef, 0, sizeof(double), ctx->stream));
  KS_CALL(ks_allreduce_sum(ctx, bv->coef, 1));
  KS_HIP(hipMemcpyAsync(&p, bv->coef, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  KS_HIP(ks_sync(ctx));
  KS_CHECK(p > -bv->deftol, KS_ERR_USER_INPUT, "The inner product is not well defined: indefinite matrix %g", p);
  *val = p < 0.0 ? 0.0 : sqrt(p);
  return KS_SUCCESS;
}

static int dotvec_impl(ks_bv X, const double *y_dev, double *m, bool reduce)
{
  KS_CHECK(X && y_dev, KS_ERR_ARG_NULL, "NULL argument");
  ks_ctx ctx = X->ctx;
  KS_HIP(hipSetDevice(ctx->device));
  const int kx = X->k - X->l;
  if (kx <= 0) return KS_SUCCESS;
  KS_CALL(ksb_ipmatmult(X, y_dev, &y_dev));          // svec.c:117-120: z = B*y when a matrix is set
  const double *A = X->array + (size_t)(X->nc + X->l) * X->ld;
  double *out = m ? X->coef : X->buffer;      // m==NULL: result goes to the buffer scratch (svec.c:123)
  KS_CHECK((size_t)kx <= X->coef_len, KS_ERR_ARG_SIZ, "too many columns");
  for (int c0 = 0; c0 < kx; c0 += KS_MAX_COLS) {
    const int nc = std::min(KS_MAX_COLS, kx - c0);
    if (X->n > 0) { KS_CALL(ksk_dot(X, A + (size_t)c0 * X->ld, X->ld, nc, y_dev, false)); KS_CALL(ksk_reduce_partials(X, nc, out + c0)); }
    else KS_HIP(hipMemsetAsync(out + c0, 0, sizeof(double) * nc, ctx->stream));
  }
  if (reduce) KS_CALL(ks_allreduce_sum(ctx, out, kx));
  if (m) { KS_HIP(hipMemcpyAsync(m, out, sizeof(double) * kx, hipMemcpyDeviceToHost, ctx->stream)); KS_HIP(ks_sync(ctx)); }
  return KS_SUCCESS;
}
extern "C" int ks_bv_dotvec(ks_bv X, const double *y_dev, double *m) { return dotvec_impl(X, y_dev, m, true); }          // bvglobal.c:151
extern "C" int ks_bv_dotvec_local(ks_bv X, const double *y_dev, double *m) { return dotvec_impl(X, y_dev, m, false); }   // ops->dotvec_local

extern "C" int ks_bv_dotcolumn(ks_bv X, int j, double *q)   // bvglobal.c:302-327
{
  KS_CHECK(X, KS_ERR_ARG_NULL, "BV is NULL");
  KS_CHECK(j >= 0, KS_ERR_ARG_OUTOFRANGE, "Index j must be non-negative");
  KS_CHECK(j < X->m, KS_ERR_ARG_OUTOFRANGE, "Index j=%d but BV only has %d columns", j, X->m);
  const int ksave = X->k;
  X->k = j;
  int rc = ks_bv_dotvec(X, ks_bv_col(X, j), q);
  X->k = ksave;
  return rc;
}

// M(ys:ye, xs:xe) = Y(:,ys:ye)^H X(:,xs:xe); X and Y may be the same BV
int ksb_dot_range(ks_bv X, int xs, int xe, ks_bv Y, int ys, int ye, double *M, int ldm)
{
  if (xs >= xe || ys >= ye) return KS_SUCCESS;
  ks_ctx ctx = X->ctx;
  KS_HIP(hipSetDevice(ctx->device));
  const int my = ye - ys, nx = xe - xs;
  if (my > KS_MAX_COLS || nx > 64) {                        // wide panels: 64 x 64 blocks of M, each one sweep on the matrix cores
    for (int yb = ys; yb < ye; yb += KS_MAX_COLS)
      for (int xb = xs; xb < xe; xb += 64)
        KS_CALL(ksb_dot_range(X, xb, std::min(xe, xb + 64), Y, yb, std::min(ye, yb + KS_MAX_COLS), M, ldm));
    return KS_SUCCESS;
  }
  KS_CHECK((size_t)my * nx <= X->coef_len, KS_ERR_ARG_SIZ, "result block too large");
  const double *py = Y->array + (size_t)(Y->nc + ys) * Y->ld;
  if (X->matrix) {
    // bvglobal.c:103-107: cached = B*X(:,xs:xe), then M = Y^H cached
    ks_bv W = nullptr;
    KS_CALL(ks_bv_create(ctx, X->n, X->N, nx, 0, &W));
    int rc = KS_SUCCESS;
    for (int j = 0; j < nx && !rc; j++) rc = ks_mat_mult_internal(X->matrix, X->array + (size_t)(X->nc + xs + j) * X->ld, ks_bv_col(W, j));
    if (!rc) {
      std::vector<double> T((size_t)ldm * nx, 0.0);
      rc = ksb_dot_range(W, 0, nx, Y, ys, ye, T.data(), ldm);
      if (!rc) for (int j = 0; j < nx; j++) for (int i = ys; i < ye; i++) M[(size_t)i + (size_t)(xs + j) * ldm] = T[(size_t)i + (size_t)j * ldm];
    }
    ks_bv_destroy(W);
    return rc;
  }
  const bool use_mfma = !ctx->dbg.no_mfma;
  const double *px0 = X->array + (size_t)(X->nc + xs) * X->ld;
  if (use_mfma && X->n > 0 && nx <= 64 && X->ld % 2 == 0 && Y->ld % 2 == 0 && aligned16(py) && aligned16(px0)) {
    KS_CALL(ksp_dot_mfma(X, py, Y->ld, my, px0, X->ld, nx, X->n, X->coef));      // one sweep over both panels on the matrix cores
  } else {
    KsProfScope ps(ctx, KS_K_BVDOT, 8.0 * X->n * (my + nx));
    const bool save = ctx->prof_on; ctx->prof_on = false;       // account the whole panel product as one class
    for (int jx = 0; jx < nx; jx++) {
      const double *xcol = X->array + (size_t)(X->nc + xs + jx) * X->ld;
      if (X->n > 0) { KS_CALL(ksk_dot(X, py, Y->ld, my, xcol, false)); KS_CALL(ksk_reduce_partials(X, my, X->coef + (size_t)jx * my)); }
      else KS_HIP(hipMemsetAsync(X->coef + (size_t)jx * my, 0, sizeof(double) * my, ctx->stream));
    }
    ctx->prof_on = save;
  }
  KS_CALL(ks_allreduce_sum(ctx, X->coef, my * nx));
  std::vector<double> tmp((size_t)my * nx);
  KS_HIP(hipMemcpyAsync(tmp.data(), X->coef, sizeof(double) * my * nx, hipMemcpyDeviceToHost, ctx->stream));
  KS_HIP(ks_sync(ctx));
  double *C = M + (size_t)xs * ldm + ys;
  for (int j = 0; j < nx; j++) memcpy(C + (size_t)j * ldm, tmp.data() + (size_t)j * my, sizeof(double) * my);
  return KS_SUCCESS;
}

extern "C" int ks_bv_dot(ks_bv X, ks_bv Y, double *M, int ldm)   // bvglobal.c:86-116, svec.c:89-107: M = Y^H X
{
  KS_CHECK(X && Y && M, KS_ERR_ARG_NULL, "NULL argument");
  KS_CHECK(ldm >= Y->k, KS_ERR_ARG_SIZ, "Mat argument has %d rows, should have at least %d", ldm, Y->k);
  KS_CHECK(X->n == Y->n, KS_ERR_ARG_INCOMP, "Mismatching local dimension X %d, Y %d", X->n, Y->n);
  return ksb_dot_range(X, X->l, X->k, Y, Y->l, Y->k, M, ldm);
}

// Y(:,ys:ye) = beta*Y(:,ys:ye) + alpha*X(:,xs:xe)*Q(xs:xe, ys:ye); X and Y may be the same BV when the ranges are disjoint
int ksb_mult_range(ks_bv Y, int ys, int ye, double alpha, double beta, ks_bv X, int xs, int xe, const double *Q, int ldq)
{
  const int ny = ye - ys, kx = xe - xs;
  if (ny <= 0 || kx <= 0) return KS_SUCCESS;
  ks_ctx ctx = Y->ctx;
  KS_HIP(hipSetDevice(ctx->device));
  double *qdev = nullptr;
  KS_CALL(stage_coefs(Y, Q, (size_t)ldq * ye, &qdev));
  return panel_mult(ctx, KS_K_MULT, X->array + (size_t)(X->nc + xs) * X->ld, X->ld, Y->n, kx, qdev + (size_t)ys * ldq + xs, ldq, false, ny, alpha, beta,
                    Y->array + (size_t)(Y->nc + ys) * Y->ld, Y->ld);
}

// ---- ops->scale / norm / copy ------------------------------------------------------------------------
extern "C" int ks_bv_scale(ks_bv bv, double alpha)   // bvops.c:311, svec.c:150-162 (j<0: (k-l)*ld contiguous scalars)
{
  KS_CHECK(bv, KS_ERR_ARG_NULL, "BV is NULL");
  if (alpha == 1.0 || !bv->n || bv->k <= bv->l) return KS_SUCCESS;
  bv->spec.valid = false;
  KS_HIP(hipSetDevice(bv->ctx->device));
  return ksk_scale(bv->ctx, bv->array + (size_t)(bv->nc + bv->l) * bv->ld, (size_t)(bv->k - bv->l) * bv->ld, alpha);
}

extern "C" int ks_bv_scalecolumn(ks_bv bv, int j, double alpha)   // bvops.c:341
{
  KS_CHECK(bv, KS_ERR_ARG_NULL, "BV is NULL");
  KS_CHECK(j >= 0 && j < bv->m, KS_ERR_ARG_OUTOFRANGE, "Argument j has wrong value %d, the number of columns is %d", j, bv->m);
  if (alpha == 1.0 || !bv->n) return KS_SUCCESS;
  bv->spec.valid = false;
  KS_HIP(hipSetDevice(bv->ctx->device));
  return ksk_scale(bv->ctx, ks_bv_col(bv, j), bv->n, alpha);
}

static int norm_core(ks_bv bv, const double *A, int ncols, int j, int type, double *val, bool reduce);
static int norm_impl(ks_bv bv, int j, int type, double *val, bool reduce)   // svec.c:164-190, bvlapack.c:37-83
{
  KS_CHECK(bv && val, KS_ERR_ARG_NULL, "NULL argument");
  KS_HIP(hipSetDevice(bv->ctx->device));
  const double *A; int ncols;
  if (j < 0) { A = bv->array + (size_t)(bv->nc + bv->l) * bv->ld; ncols = bv->k - bv->l; }
  else { KS_CHECK(j < bv->m, KS_ERR_ARG_OUTOFRANGE, "Argument j has wrong value %d, the number of columns is %d", j, bv->m); A = ks_bv_col(bv, j); ncols = 1; }
  return norm_core(bv, A, ncols, j, type, val, reduce);
}
// norm of the ncols columns starting at A (leading dimension of the BV); j >= 0 marks a single vector (2-norm allowed)
static int norm_core(ks_bv bv, const double *A, int ncols, int j, int type, double *val, bool reduce)
{
  ks_ctx ctx = bv->ctx;
  if (ncols <= 0) { *val = 0.0; return KS_SUCCESS; }
  KS_CHECK(ncols <= KS_PSTRIDE - 8, KS_ERR_SUP, "norm over more than %d columns", KS_PSTRIDE - 8);
  bv->spec.valid = false;                                       // the column sums go through the partials array
  const int grid = std::max(1, std::min((bv->n + SW_BLOCK - 1) / SW_BLOCK, std::min(ctx->num_cu * 4, KS_MAX_BLOCKS)));
  KsProfScope ps(ctx, KS_K_NORM, 8.0 * bv->n * ncols);
  std::vector<double> h;
  if (type == KS_NORM_FROBENIUS || type == KS_NORM_2 || type == KS_NORM_1) {
    if (type == KS_NORM_2) KS_CHECK(j >= 0, KS_ERR_SUP, "Requested norm not available");   // bvglobal.c:506
    if (type == KS_NORM_1) hipLaunchKernelGGL(k_colsum<1>, dim3(grid), dim3(SW_BLOCK), 0, ctx->stream, A, (long long)bv->ld, bv->n, ncols, bv->partials);
    else hipLaunchKernelGGL(k_colsum<0>, dim3(grid), dim3(SW_BLOCK), 0, ctx->stream, A, (long long)bv->ld, bv->n, ncols, bv->partials);
    KS_HIP(hipGetLastError());
    // column totals -> coef[0..ncols)
    for (int c0 = 0; c0 < ncols; c0 += KS_MAX_COLS) {
      const int nc = std::min(KS_MAX_COLS, ncols - c0);
      hipLaunchKernelGGL(k_reduce_only, dim3(1), dim3(1024), 0, ctx->stream, bv->partials + (size_t)c0 * grid, grid, nc, bv->coef + c0);
    }
    if (reduce) KS_CALL(ks_allreduce_sum(ctx, bv->coef, ncols));     // sum of squares / abs sums add across ranks
    h.resize(ncols);
    KS_HIP(hipMemcpyAsync(h.data(), bv->coef, sizeof(double) * ncols, hipMemcpyDeviceToHost, ctx->stream));
    KS_HIP(ks_sync(ctx));
    if (type == KS_NORM_1) { double mx = 0.0; for (double v : h) mx = std::max(mx, v); *val = mx; }
    else {
      double s = 0.0; for (double v : h) s += v;
      *val = sqrt(s);
      if (!(s < 1e300) || s < 1e-290) {
        // the plain sum of squares overflowed (or may have flushed to zero): redo it scaled by the largest entry, the
        // overflow-safe combination the reference gets from lange + MPIU_LAPY2 (bvlapack.c:20-32,60-75)
        hipLaunchKernelGGL(k_colsum<3>, dim3(grid), dim3(SW_BLOCK), 0, ctx->stream, A, (long long)bv->ld, bv->n, ncols, bv->partials, 1.0);
        std::vector<double> hm((size_t)grid * ncols);
        KS_HIP(hipMemcpyAsync(hm.data(), bv->partials, sizeof(double) * hm.size(), hipMemcpyDeviceToHost, ctx->stream));
        KS_HIP(ks_sync(ctx));
        double mx = 0.0; for (double v : hm) mx = std::max(mx, v);
        if (reduce && ctx->comm.size > 1) { std::vector<double> all(ctx->comm.size); KS_CALL(ks_comm_allgather_host(ctx, &mx, sizeof(double), all.data())); for (double v : all) mx = std::max(mx, v); }
        if (mx == 0.0 || !(mx < std::numeric_limits<double>::infinity())) *val = mx;
        else {
          hipLaunchKernelGGL(k_colsum<2>, dim3(grid), dim3(SW_BLOCK), 0, ctx->stream, A, (long long)bv->ld, bv->n, ncols, bv->partials, 1.0 / mx);
          for (int c0 = 0; c0 < ncols; c0 += KS_MAX_COLS) {
            const int nc = std::min(KS_MAX_COLS, ncols - c0);
            hipLaunchKernelGGL(k_reduce_only, dim3(1), dim3(1024), 0, ctx->stream, bv->partials + (size_t)c0 * grid, grid, nc, bv->coef + c0);
          }
          if (reduce) KS_CALL(ks_allreduce_sum(ctx, bv->coef, ncols));
          KS_HIP(hipMemcpyAsync(h.data(), bv->coef, sizeof(double) * ncols, hipMemcpyDeviceToHost, ctx->stream));
          KS_HIP(ks_sync(ctx));
          double s2 = 0.0; for (double v : h) s2 += v;
          *val = mx * sqrt(s2);
        }
      }
    }
  } else if (type == KS_NORM_INFINITY) {
    hipLaunchKernelGGL(k_rowsum_max, dim3(grid), dim3(SW_BLOCK), 0, ctx->stream, A, (long long)bv->ld, bv->n, ncols, bv->partials);
    KS_HIP(hipGetLastError());
    h.resize(grid);
    KS_HIP(hipMemcpyAsync(h.data(), bv->partials, sizeof(double) * grid, hipMemcpyDeviceToHost, ctx->stream));
    KS_HIP(ks_sync(ctx));
    double mx = 0.0; for (double v : h) mx = std::max(mx, v);
    if (reduce && ctx->comm.size > 1) {                        // MAX across ranks through the host allgather of the provider
      std::vector<double> all(ctx->comm.size);
      KS_CALL(ks_comm_allgather_host(ctx, &mx, sizeof(double), all.data()));
      for (double v : all) mx = std::max(mx, v);
    }
    *val = mx;
  } else KS_FAIL(KS_ERR_ARG_WRONG, "unknown norm type %d", type);
  return KS_SUCCESS;
}
extern "C" int ks_bv_norm(ks_bv bv, int type, double *val)                   // bvglobal.c:498
{
  KS_CHECK(type != KS_NORM_2, KS_ERR_SUP, "Requested norm not available");
  return norm_impl(bv, -1, type, val, true);
}
extern "C" int ks_bv_normcolumn(ks_bv bv, int j, int type, double *val)      // bvglobal.c:662
{
  KS_CHECK(j >= 0, KS_ERR_ARG_OUTOFRANGE, "Argument j has wrong value %d", j);
  if (bv && bv->matrix) {                                                   // bvglobal.c:683-687: sqrt(V[j]'*B*V[j]), type ignored
    KS_CHECK(val, KS_ERR_ARG_NULL, "NULL argument");
    KS_CHECK(j < bv->m, KS_ERR_ARG_OUTOFRANGE, "Argument j has wrong value %d, the number of columns is %d", j, bv->m);
    KS_HIP(hipSetDevice(bv->ctx->device));
    return ksb_norm_b(bv, ks_bv_col(bv, j), val);
  }
  return norm_impl(bv, j, type, val, true);
}
extern "C" int ks_bv_norm_local(ks_bv bv, int j, int type, double *val) { return norm_impl(bv, j, type, val, false); }
extern "C" int ks_bv_normvec(ks_bv bv, const double *v_dev, int type, double *val)   // BVNormVec bvglobal.c:530-571: B-norm when a matrix is set
{
  KS_CHECK(bv && v_dev && val, KS_ERR_ARG_NULL, "NULL argument");
  KS_HIP(hipSetDevice(bv->ctx->device));
  if (bv->matrix) return ksb_norm_b(bv, v_dev, val);
  return norm_core(bv, v_dev, 1, 0, type, val, true);
}

// ---- split reductions: BVDotVecBegin/End, BVDotColumnBegin/End, BVNormVecBegin/End, BVNormColumnBegin/End (bvglobal.c) ----
// Begin computes this rank's part into a queue on the device; nothing crosses ranks until the first End, which reduces the
// whole queue with one allreduce. Ends must come in the order of the Begins (as PetscSplitReduction requires).
static int split_begin(ks_ctx ctx, int cnt, int kind, double **slot)
{
  auto &sp = ctx->split;
  if (sp.reduced && sp.nread == sp.entries.size()) { sp.entries.clear(); sp.used = 0; sp.nread = 0; sp.reduced = false; }
  KS_CHECK(!sp.reduced, KS_ERR_ORDER, "Called a Begin operation after an End: finish the pending End calls first");
  if (!sp.dev) { KS_HIP(hipMalloc(&sp.dev, sizeof(double) * 16384)); sp.cap = 16384; }
  KS_CHECK(sp.used + cnt <= sp.cap, KS_ERR_ARG_SIZ, "too many values queued in split reductions (%d)", sp.used + cnt);
  *slot = sp.dev + sp.used;
  sp.entries.push_back({sp.used, cnt, kind});
  sp.used += cnt;
  return KS_SUCCESS;
}
static int split_end(ks_ctx ctx, int cnt, int kind, double *out, double deftol)
{
  auto &sp = ctx->split;
  KS_CHECK(sp.nread < sp.entries.size(), KS_ERR_ORDER, "End operation without a matching Begin");
  if (!sp.reduced) {
    KS_CALL(ks_allreduce_sum(ctx, sp.dev, sp.used));
    sp.host.resize(sp.used);
    KS_HIP(hipMemcpyAsync(sp.host.data(), sp.dev, sizeof(double) * sp.used, hipMemcpyDeviceToHost, ctx->stream));
    KS_HIP(ks_sync(ctx));
    sp.reduced = true;
  }
  const auto e = sp.entries[sp.nread];
  KS_CHECK(e.cnt == cnt && e.kind == kind, KS_ERR_ORDER, "End operation does not match the Begin at this position (they must come in the same order)");
  sp.nread++;
  if (kind == 0) { for (int i = 0; i < cnt; i++) out[i] = sp.host[e.off + i]; }
  else {
    const double p = sp.host[e.off];
    KS_CHECK(p > -deftol, KS_ERR_USER_INPUT, "The inner product is not well defined: indefinite matrix %g", p);
    out[0] = p < 0.0 ? 0.0 : sqrt(p);
  }
  return KS_SUCCESS;
}
static int dots_local(ks_bv X, const double *A, int kx, const double *y_dev, double *out)     // out[0..kx) = A(:,0:kx)' y, this rank only
{
  ks_ctx ctx = X->ctx;
  for (int c0 = 0; c0 < kx; c0 += KS_MAX_COLS) {
    const int nc = std::min(KS_MAX_COLS, kx - c0);
    if (X->n > 0) { KS_CALL(ksk_dot(X, A + (size_t)c0 * X->ld, X->ld, nc, y_dev, false)); KS_CALL(ksk_reduce_partials(X, nc, out + c0)); }
    else KS_HIP(hipMemsetAsync(out + c0, 0, sizeof(double) * nc, ctx->stream));
  }
  return KS_SUCCESS;
}
extern "C" int ks_bv_dotvec_begin(ks_bv X, const double *y_dev, double *m)   // BVDotVecBegin bvglobal.c:207
{
  KS_CHECK(X && y_dev && m, KS_ERR_ARG_NULL, "NULL argument");
  KS_HIP(hipSetDevice(X->ctx->device));
  const int kx = X->k - X->l;
  if (kx <= 0) return KS_SUCCESS;
  KS_CALL(ksb_ipmatmult(X, y_dev, &y_dev));
  double *slot = nullptr;
  KS_CALL(split_begin(X->ctx, kx, 0, &slot));
  return dots_local(X, X->array + (size_t)(X->nc + X->l) * X->ld, kx, y_dev, slot);
}
extern "C" int ks_bv_dotvec_end(ks_bv X, const double *y_dev, double *m)     // BVDotVecEnd bvglobal.c:256
{
  KS_CHECK(X && m, KS_ERR_ARG_NULL, "NULL argument");
  (void)y_dev;
  const int kx = X->k - X->l;
  if (kx <= 0) return KS_SUCCESS;
  return split_end(X->ctx, kx, 0, m, 0.0);
}
extern "C" int ks_bv_dotcolumn_begin(ks_bv X, int j, double *q)              // BVDotColumnBegin bvglobal.c:343
{
  KS_CHECK(X && q, KS_ERR_ARG_NULL, "NULL argument");
  KS_CHECK(j >= 0 && j < X->m, KS_ERR_ARG_OUTOFRANGE, "Index j=%d but BV only has %d columns", j, X->m);
  const int ksave = X->k; X->k = j;
  const int rc = ks_bv_dotvec_begin(X, ks_bv_col(X, j), q);
  X->k = ksave;
  return rc;
}
extern "C" int ks_bv_dotcolumn_end(ks_bv X, int j, double *q)                // BVDotColumnEnd bvglobal.c:395
{
  KS_CHECK(X && q, KS_ERR_ARG_NULL, "NULL argument");
  KS_CHECK(j >= 0 && j < X->m, KS_ERR_ARG_OUTOFRANGE, "Index j=%d but BV only has %d columns", j, X->m);
  const int ksave = X->k; X->k = j;
  const int rc = ks_bv_dotvec_end(X, nullptr, q);
  X->k = ksave;
  return rc;
}
static int norm2_begin(ks_bv bv, const double *v_dev)
{
  const double *z;
  KS_CALL(ksb_ipmatmult(bv, v_dev, &z));                                      // x' B x, or x' x
  double *slot = nullptr;
  KS_CALL(split_begin(bv->ctx, 1, 1, &slot));
  return dots_local(bv, z, 1, v_dev, slot);
}
extern "C" int ks_bv_normvec_begin(ks_bv bv, const double *v_dev, int type, double *val)   // BVNormVecBegin bvglobal.c:573
{
  KS_CHECK(bv && v_dev && val, KS_ERR_ARG_NULL, "NULL argument");
  KS_CHECK(type == KS_NORM_2 || type == KS_NORM_FROBENIUS || bv->matrix, KS_ERR_SUP, "the split form is built for the 2-norm (a SUM reduction)");
  KS_HIP(hipSetDevice(bv->ctx->device));
  return norm2_begin(bv, v_dev);
}
extern "C" int ks_bv_normvec_end(ks_bv bv, const double *v_dev, int type, double *val)     // BVNormVecEnd bvglobal.c:615
{
  KS_CHECK(bv && val, KS_ERR_ARG_NULL, "NULL argument");
  (void)v_dev; (void)type;
  return split_end(bv->ctx, 1, 1, val, bv->deftol);
}
extern "C" int ks_bv_normcolumn_begin(ks_bv bv, int j, int type, double *val)              // BVNormColumnBegin bvglobal.c:705
{
  KS_CHECK(bv && val, KS_ERR_ARG_NULL, "NULL argument");
  KS_CHECK(j >= 0 && j < bv->m, KS_ERR_ARG_OUTOFRANGE, "Argument j has wrong value %d, the number of columns is %d", j, bv->m);
  return ks_bv_normvec_begin(bv, ks_bv_col(bv, j), type, val);
}
extern "C" int ks_bv_normcolumn_end(ks_bv bv, int j, int type, double *val)                // BVNormColumnEnd bvglobal.c:751
{
  KS_CHECK(bv && val, KS_ERR_ARG_NULL, "NULL argument");
  KS_CHECK(j >= 0 && j < bv->m, KS_ERR_ARG_OUTOFRANGE, "Argument j has wrong value %d, the number of columns is %d", j, bv->m);
  return ks_bv_normvec_end(bv, nullptr, type, val);
}

extern "C" int ks_bv_copy(ks_bv V, ks_bv W)   // svec.c:232-247
{
  KS_CHECK(V && W, KS_ERR_ARG_NULL, "BV is NULL");
  KS_CHECK(V->n == W->n, KS_ERR_ARG_INCOMP, "Mismatching local dimension V %d, W %d", V->n, W->n);
  KS_CHECK(V->k - V->l == W->k - W->l, KS_ERR_ARG_SIZ, "W has %d active columns, should match %d active columns in V", W->k - W->l, V->k - V->l);
  if (V == W || !V->n) return KS_SUCCESS;
  KS_HIP(hipSetDevice(V->ctx->device));
  for (int j = 0; j < V->k - V->l; j++) KS_CALL(ksk_copy(V->ctx, ks_bv_col(V, V->l + j), ks_bv_col(W, W->l + j), V->n));
  return KS_SUCCESS;
}

extern "C" int ks_bv_copycolumn(ks_bv V, int j, int i)   // svec.c:249-259
{
  KS_CHECK(V, KS_ERR_ARG_NULL, "BV is NULL");
  KS_CHECK(j >= 0 && j < V->m && i >= 0 && i < V->m, KS_ERR_ARG_OUTOFRANGE, "column index out of range (%d -> %d, m=%d)", j, i, V->m);
  if (j == i) return KS_SUCCESS;
  V->spec.valid = false;
  KS_HIP(hipSetDevice(V->ctx->device));
  return ksk_copy(V->ctx, ks_bv_col(V, j), ks_bv_col(V, i), V->n);
}

// ---- ops->matmult -----------------------------------------------------------------------------------
extern "C" int ks_bv_matmultcolumn(ks_bv V, ks_mat A, int j)   // bvops.c:862-885
{
  KS_CHECK(V && A, KS_ERR_ARG_NULL, "NULL argument");
  KS_CHECK(j >= 0, KS_ERR_ARG_OUTOFRANGE, "Index j must be non-negative");
  KS_CHECK(j + 1 < V->m, KS_ERR_ARG_OUTOFRANGE, "Result should go in index j+1=%d but BV only has %d columns", j + 1, V->m);
  KS_CHECK(A->n == V->n, KS_ERR_ARG_INCOMP, "Mismatching local row dimension A %d, V %d", A->n, V->n);
  KS_HIP(hipSetDevice(V->ctx->device));
  return ks_mat_mult_internal(A, ks_bv_col(V, j), ks_bv_col(V, j + 1));
}

extern "C" int ks_bv_matmult(ks_bv V, ks_mat A, ks_bv W)   // bvops.c BVMatMult, svec.c:213-222 (column loop)
{
  KS_CHECK(V && A && W, KS_ERR_ARG_NULL, "NULL argument");
  KS_CHECK(A->n == W->n && A->n == V->n, KS_ERR_ARG_INCOMP, "Mismatching local row dimension");
  KS_CHECK(V->k - V->l == W->k - W->l, KS_ERR_ARG_SIZ, "Y has %d active columns, should match %d active columns in V", W->k - W->l, V->k - V->l);
  KS_HIP(hipSetDevice(V->ctx->device));
  for (int j = 0; j < V->k - V->l; j++) KS_CALL(ks_mat_mult_internal(A, ks_bv_col(V, V->l + j), ks_bv_col(W, W->l + j)));
  return KS_SUCCESS;
}
