// Block orthogonalisation and the panel operations built on the MFMA products.
//
// Restates BVOrthogonalize and its five block methods (src/sys/classes/bv/interface/bvorthog.c:492-767):
//   GS        column by column with BVOrthogonalizeColumn                              (:510-553)
//   CHOL      R = chol(V'V), Q = V inv(R)                                              (:601-617, bvlapack.c:136-204)
//   SVQB      S = D U Lambda^-1/2 from the eigendecomposition of D V'V D               (:674-690, bvlapack.c:259-341)
//   TSQR      tall-skinny QR: per-block Householder R factors, combined pairwise       (:622-641, bvlapack.c:346-451)
//   TSQRCHOL  R by TSQR only, Q = V inv(R)                                             (:646-669, bvlapack.c:483-565)
// and BVMatProject (bvglobal.c:1014-1160), BVNormalize (bvglobal.c:855-938).
// The Gram matrices, V*inv(R) and the block Gram-Schmidt against leading columns are the FP64 MFMA panel kernels of
// ks_panel.hip; the k x k factorisations run on the host (ks_dense.cpp). TSQR accumulates its Householder reflectors into
// an explicit Q as the reference does (geqrf / orgqr per block and per tree level, bvlapack.c:380-451): the factor kernel
// leaves the reflectors in place of the panel, the stack of the blocks' triangular factors is factored on the host with
// its orthogonal factor formed explicitly, and a second kernel applies every block's reflectors, last tile first, to its
// block of that factor. TSQRCHOL keeps Q = V inv(R) (bvlapack.c:483-565).
#include "ksgpu_internal.h"
#include "ks_dense.h"
#include <algorithm>

namespace {

constexpr int TS_T = 64;      // rows per tile = one row per lane
constexpr int TS_BLOCK = 256; // 4 wavefronts share the columns of a step

__device__ __forceinline__ double wsum(double v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Each block walks its contiguous row range in tiles of 64 rows and keeps a running R (nc x nc, upper triangular) in LDS. The block's FIRST
// tile is factored on its own (dgeqr2: reflector j has its 1 in tile row j and acts on rows j..63), every later tile as [R; tile] by reflectors
// that touch row j of R and the 64 tile rows only. (Starting the running R at zero instead - [0; tile] - is the same arithmetic for R but
// not for Q: the orthogonal factor of the augmented matrix has a top block that is zero only in exact arithmetic, eps ||A|| / sigma_min in
// practice, and the rows that belong to the panel lose that much of their norm.)
// REFL: leave the reflectors behind - v_j of every tile in place of the panel's entries (the leading 1 is implicit), tau_j in tau[tile][j] -
// for k_tsqr_formq.
template <bool REFL>
__global__ __launch_bounds__(TS_BLOCK) void k_tsqr_local(double *__restrict__ V, long long ld, long long n, int nc, long long rows_per_block, double *__restrict__ Rout, double *__restrict__ tau_out)
{
  __shared__ double Rr[64 * 64];     // column-major, pitch 64
  __shared__ double Tt[64 * 64];     // tile, column-major: Tt[c*64 + lane]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 64 * 64; i += TS_BLOCK) Rr[i] = 0.0;
  const long long r0 = (long long)blockIdx.x * rows_per_block, r1 = min(n, r0 + rows_per_block);
  for (long long t0 = r0; t0 < r1; t0 += TS_T) {
    __syncthreads();
    const long long row = t0 + lane;
    const bool first = t0 == r0;
    for (int c = wave; c < nc; c += TS_BLOCK / 64) Tt[c * 64 + lane] = (row < r1) ? V[(size_t)c * ld + row] : 0.0;
    __syncthreads();
    for (int j = 0; j < nc; j++) {
      // every wave derives the reflector of column j redundantly (same LDS inputs, same arithmetic)
      const double tj = Tt[j * 64 + lane];
      const double x = first ? (lane > j ? tj : 0.0) : tj;                 // the part of column j the reflector annihilates
      const double alpha = first ? Tt[j * 64 + j] : Rr[j * 64 + j];       // its pivot: tile row j, or row j of the running R
      const double xn2 = wsum(x * x);
      double tau = 0.0, v = 0.0, beta = alpha;
      if (xn2 != 0.0) {
        const double nrm = sqrt(alpha * alpha + xn2);
        beta = (alpha >= 0.0) ? -nrm : nrm;
        tau = (beta - alpha) / beta;
        v = x / (alpha - beta);
      }
      // apply H = I - tau [1; v][1; v]^T to the remaining columns, one column per wave at a time
      for (int c = j + 1 + wave; c < nc; c += TS_BLOCK / 64) {
        const double w = Tt[c * 64 + lane];
        const double pjc = first ? Tt[c * 64 + j] : Rr[c * 64 + j];        // the column's entry in the pivot row
        const double d = tau * (wsum(v * w) + pjc);
        if (first) Tt[c * 64 + lane] = (lane == j) ? pjc - d : w - d * v;   // (v = 0 in rows <= j: they keep their values)
        else { Tt[c * 64 + lane] = w - d * v; if (lane == 0) Rr[c * 64 + j] = pjc - d; }
      }
      __syncthreads();
      if (first) { if (wave == 0 && lane <= j) Rr[j * 64 + lane] = lane == j ? beta : Tt[j * 64 + lane]; }      // column j of R: rows 0..j-1 are final, the pivot becomes beta
      else if (threadIdx.x == 0) Rr[j * 64 + j] = beta;
      // column j of the tile is spent; the barrier above already separates this step from the next one's reads
      if (REFL && wave == 0) { Tt[j * 64 + lane] = v; if (lane == 0) tau_out[(size_t)(t0 / TS_T) * nc + j] = tau; }
    }
    if (REFL) {
      __syncthreads();
      for (int c = wave; c < nc; c += TS_BLOCK / 64) if (row < r1) V[(size_t)c * ld + row] = Tt[c * 64 + lane];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nc * nc; i += TS_BLOCK) { const int r = i % nc, c = i / nc; Rout[(size_t)blockIdx.x * nc * nc + i] = (r <= c) ? Rr[c * 64 + r] : 0.0; }
}

// Q of one block: its reflectors applied, last tile first and within a tile last reflector first, to [C; 0; ...; 0] with C the block's
// nc x nc piece of the combine step's orthogonal factor. A column of the result depends on the same column of C only: lane = row of the tile,
// a wave owns its columns from the last tile to the first (the column's part of C in LDS, its 64 tile entries in a register) - no barrier
// inside a tile. The first tile holds C itself in its first nc rows (its reflectors act inside the tile).
__global__ __launch_bounds__(TS_BLOCK) void k_tsqr_formq(double *__restrict__ V, long long ld, long long n, int nc, long long rows_per_block, const double *__restrict__ Cin, const double *__restrict__ tau_in)
{
  __shared__ double Cc[64 * 64];     // column-major, pitch 64: Cc[c*64 + j]
  __shared__ double Vt[64 * 64];     // the tile's reflectors
  __shared__ double ta[64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < nc * nc; i += TS_BLOCK) { const int r = i % nc, c = i / nc; Cc[c * 64 + r] = Cin[(size_t)blockIdx.x * nc * nc + i]; }
  const long long r0 = (long long)blockIdx.x * rows_per_block, r1 = min(n, r0 + rows_per_block);
  if (r0 >= r1) return;
  const long long ntile = (r1 - r0 + TS_T - 1) / TS_T;
  for (long long tt = ntile - 1; tt >= 0; tt--) {
    const long long t0 = r0 + tt * TS_T, row = t0 + lane;
    const bool first = tt == 0;
    __syncthreads();
    for (int c = wave; c < nc; c += TS_BLOCK / 64) Vt[c * 64 + lane] = (row < r1) ? V[(size_t)c * ld + row] : 0.0;
    if (threadIdx.x < nc) ta[threadIdx.x] = tau_in[(size_t)(t0 / TS_T) * nc + threadIdx.x];
    __syncthreads();
    for (int c = wave; c < nc; c += TS_BLOCK / 64) {
      double z = first ? (lane < nc ? Cc[c * 64 + lane] : 0.0) : 0.0;
      for (int j = nc - 1; j >= 0; j--) {
        const double tj = ta[j];
        if (tj == 0.0) continue;
        const double vj = Vt[j * 64 + lane];                               // first tile: zero in rows <= j
        if (first) {
          const double zj = __shfl(z, j, 64);
          const double w = tj * (zj + wsum(vj * z));
          z = (lane == j) ? z - w : z - w * vj;
        } else {
          const double w = tj * (Cc[c * 64 + j] + wsum(vj * z));
          z -= w * vj;
          if (lane == 0) Cc[c * 64 + j] -= w;
        }
      }
      // the tile's reflectors are all in LDS (every wave reads every one of them): the panel's entries can take the result
      if (row < r1) V[(size_t)c * ld + row] = z;
    }
  }
}

// R factor (host, nc x nc upper triangular, ldr) of V(:, s:s+nc): per-block factors, combined in block order, then in
// rank order across the communicator - deterministic and identical on every rank
int tsqr_r(ks_bv V, int s, int nc, double *R, int ldr)
{
  ks_ctx ctx = V->ctx;
  KS_CHECK(nc >= 1 && nc <= 64, KS_ERR_SUP, "TSQR with %d columns (max 64)", nc);
  const long long n = V->n;
  long long nb = std::min<long long>((n + TS_T - 1) / TS_T, (long long)ctx->num_cu * 2);
  if (nb < 1) nb = 1;
  long long rpb = (n + nb - 1) / nb; rpb = (rpb + TS_T - 1) / TS_T * TS_T; if (rpb < TS_T) rpb = TS_T;
  nb = std::max<long long>(1, (n + rpb - 1) / rpb);
  double *dR = nullptr;
  KS_HIP(hipMalloc(&dR, sizeof(double) * (size_t)nb * nc * nc));
  {
    KsProfScope ps(ctx, KS_K_OTHER, 8.0 * n * nc);
    hipLaunchKernelGGL(k_tsqr_local<false>, dim3((unsigned)nb), dim3(TS_BLOCK), 0, ctx->stream, V->array + (size_t)(V->nc + s) * V->ld, (long long)V->ld, n, nc, rpb, dR, (double *)nullptr);
  }
  int rc = hipGetLastError() == hipSuccess ? KS_SUCCESS : KS_ERR_LIB;
  std::vector<double> h((size_t)nb * nc * nc);
  if (!rc && hipMemcpyAsync(h.data(), dR, sizeof(double) * h.size(), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = KS_ERR_LIB;
  if (!rc && ks_sync(ctx) != hipSuccess) rc = KS_ERR_LIB;
  hipFree(dR);
  KS_CHECK(!rc, KS_ERR_LIB, "TSQR panel kernel failed");
  for (long long b = 1; b < nb; b++) ksd::tsqr_combine(nc, h.data(), nc, h.data() + (size_t)b * nc * nc, nc);
  if (ks_is_multi(ctx) && ctx->comm.size > 1) {
    std::vector<double> all((size_t)ctx->comm.size * nc * nc);
    KS_CALL(ks_comm_allgather_host(ctx, h.data(), (int)(sizeof(double) * nc * nc), all.data()));
    memcpy(h.data(), all.data(), sizeof(double) * nc * nc);
    for (int r = 1; r < ctx->comm.size; r++) ksd::tsqr_combine(nc, h.data(), nc, all.data() + (size_t)r * nc * nc, nc);
  }
  for (int c = 0; c < nc; c++) for (int r = 0; r < nc; r++) R[(size_t)r + (size_t)c * ldr] = (r <= c) ? h[(size_t)r + (size_t)c * nc] : 0.0;
  return KS_SUCCESS;
}

// V(:, s:s+nc) = Q R with Q formed from the accumulated reflectors, in place; R (host, nc x nc, ldr) identical on every rank.
// Two read-write passes over the panel: factor (reflectors left in place) and form-Q.
int tsqr_q(ks_bv V, int s, int nc, double *R, int ldr)
{
  ks_ctx ctx = V->ctx;
  KS_CHECK(nc >= 1 && nc <= 64, KS_ERR_SUP, "TSQR with %d columns (max 64)", nc);
  const long long n = V->n;
  long long nb = std::min<long long>((n + TS_T - 1) / TS_T, (long long)ctx->num_cu * 2);
  if (nb < 1) nb = 1;
  long long rpb = (n + nb - 1) / nb; rpb = (rpb + TS_T - 1) / TS_T * TS_T; if (rpb < TS_T) rpb = TS_T;
  nb = std::max<long long>(1, (n + rpb - 1) / rpb);
  const long long ntiles = (n + TS_T - 1) / TS_T + nb;      // (a block's last tile may be partial: tile indices are t0 / 64 with t0 a multiple of 64)
  double *dR = nullptr, *dTau = nullptr, *panel = V->array + (size_t)(V->nc + s) * V->ld;
  KS_HIP(hipMalloc(&dR, sizeof(double) * (size_t)nb * nc * nc));
  KS_HIP(hipMalloc(&dTau, sizeof(double) * (size_t)std::max<long long>(ntiles, 1) * nc));
  auto done = [&](int rc) { hipFree(dR); hipFree(dTau); (void)hipGetLastError(); return rc; };
  {
    KsProfScope ps(ctx, KS_K_OTHER, 16.0 * n * nc);
    hipLaunchKernelGGL(k_tsqr_local<true>, dim3((unsigned)nb), dim3(TS_BLOCK), 0, ctx->stream, panel, (long long)V->ld, n, nc, rpb, dR, dTau);
  }
  if (hipGetLastError() != hipSuccess) return done(KS_ERR_LIB);
  // the combine step on the host: the stack of the blocks' triangular factors, A = Qs Rs with Qs explicit (nb nc x nc)
  const int M = (int)(nb * nc);
  std::vector<double> h((size_t)nb * nc * nc), stack((size_t)M * nc), Qs((size_t)M * nc), Rs((size_t)nc * nc);
  if (hipMemcpyAsync(h.data(), dR, sizeof(double) * h.size(), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || ks_sync(ctx) != hipSuccess) return done(KS_ERR_LIB);
  for (long long b = 0; b < nb; b++) for (int c = 0; c < nc; c++) for (int r = 0; r < nc; r++) stack[(size_t)(b * nc + r) + (size_t)c * M] = h[(size_t)b * nc * nc + (size_t)c * nc + r];
  ksd::qr_explicit(M, nc, stack.data(), M, Rs.data(), nc, Qs.data(), M);
  std::vector<double> Cb((size_t)nb * nc * nc);             // block b: Qs(b nc : (b+1) nc, :), column-major nc x nc
  if (ks_is_multi(ctx) && ctx->comm.size > 1) {
    // one more level across the ranks: the ranks' factors stacked in rank order, the same arithmetic on every rank
    const int size = ctx->comm.size, Mr = size * nc;
    std::vector<double> all((size_t)size * nc * nc), st2((size_t)Mr * nc), Q2((size_t)Mr * nc), R2((size_t)nc * nc);
    int rc = ks_comm_allgather_host(ctx, Rs.data(), (int)(sizeof(double) * nc * nc), all.data());
    if (rc) return done(rc);
    for (int q = 0; q < size; q++) for (int c = 0; c < nc; c++) for (int r = 0; r < nc; r++) st2[(size_t)(q * nc + r) + (size_t)c * Mr] = all[(size_t)q * nc * nc + (size_t)c * nc + r];
    ksd::qr_explicit(Mr, nc, st2.data(), Mr, R2.data(), nc, Q2.data(), Mr);
    const double *Cr = Q2.data() + (size_t)ctx->comm.rank * nc;          // this rank's nc x nc block, leading dimension Mr
    for (long long b = 0; b < nb; b++)
      for (int c = 0; c < nc; c++) for (int r = 0; r < nc; r++) {
        double t = 0.0;
        for (int p = 0; p < nc; p++) t += Qs[(size_t)(b * nc + r) + (size_t)p * M] * Cr[(size_t)p + (size_t)c * Mr];
        Cb[(size_t)b * nc * nc + (size_t)c * nc + r] = t;
      }
    Rs = R2;
  } else {
    for (long long b = 0; b < nb; b++) for (int c = 0; c < nc; c++) for (int r = 0; r < nc; r++) Cb[(size_t)b * nc * nc + (size_t)c * nc + r] = Qs[(size_t)(b * nc + r) + (size_t)c * M];
  }
  if (hipMemcpyAsync(dR, Cb.data(), sizeof(double) * Cb.size(), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return done(KS_ERR_LIB);
  {
    KsProfScope ps(ctx, KS_K_OTHER, 16.0 * n * nc);
    hipLaunchKernelGGL(k_tsqr_formq, dim3((unsigned)nb), dim3(TS_BLOCK), 0, ctx->stream, panel, (long long)V->ld, n, nc, rpb, dR, dTau);
  }
  if (hipGetLastError() != hipSuccess || ks_sync(ctx) != hipSuccess) return done(KS_ERR_LIB);       // Cb is pageable: the upload must have left before it goes out of scope
  for (int c = 0; c < nc; c++) for (int r = 0; r < nc; r++) R[(size_t)r + (size_t)c * ldr] = (r <= c) ? Rs[(size_t)r + (size_t)c * nc] : 0.0;
  return done(KS_SUCCESS);
}

// BVOrthogonalize_BlockGS bvorthog.c:492-505: V2 -= V1 (V1' V2), coefficients into Rb(0:l, l:k)
int block_gs(ks_bv V, double *Rb, int ldb)
{
  const int l = V->l, k = V->k;
  KS_CALL(ksb_dot_range(V, l, k, V, 0, l, Rb, ldb));
  return ksb_mult_range(V, l, k, -1.0, 1.0, V, 0, l, Rb, ldb);
}

int orthogonalize_gs(ks_bv V, double *R, int ldr)                    // bvorthog.c:510-553
{
  std::vector<double> h(V->m + 1);
  const int l = V->l;
  for (int j = l; j < V->k; j++) {
    double norm = 0.0;
    V->l = 0;                                     // store the coefficients of the leading columns too (bvorthog.c:536-539)
    const int rc = ks_bv_orthogonalizecolumn(V, j, h.data(), &norm, nullptr);
    V->l = l;
    if (rc) return rc;
    if (R) { for (int i = 0; i < j; i++) R[(size_t)i + (size_t)j * ldr] = h[i]; R[(size_t)j + (size_t)j * ldr] = norm; }
    KS_CHECK(norm != 0.0, KS_ERR_CONV_FAILED, "Breakdown in BVOrthogonalize due to a linearly dependent column");
    KS_CALL(ks_bv_scalecolumn(V, j, 1.0 / norm));
  }
  return KS_SUCCESS;
}

// copy the block result into the caller's R: columns l..k-1, rows 0..j (tri) or 0..k-1 (BV_StoreCoeffsBlock_Default)
void store_block(const ks_bv V, const double *Rb, int ldb, double *R, int ldr, bool tri)
{
  if (!R) return;
  for (int j = V->l; j < V->k; j++) { const int rows = tri ? j + 1 : V->k; for (int i = 0; i < rows; i++) R[(size_t)i + (size_t)j * ldr] = Rb[(size_t)i + (size_t)j * ldb]; }
}

} // namespace

extern "C" int ks_bv_set_orthog_block(ks_bv bv, int block)            // BVSetOrthogonalization's fourth argument
{
  KS_CHECK(bv, KS_ERR_ARG_NULL, "BV is NULL");
  KS_CHECK(block >= KS_BV_ORTHOG_BLOCK_GS && block <= KS_BV_ORTHOG_BLOCK_SVQB, KS_ERR_ARG_WRONG, "Unknown block orthogonalization type");
  bv->orthog_block = block;
  return KS_SUCCESS;
}

extern "C" int ks_bv_orthogonalize(ks_bv V, double *R, int ldr)       // BVOrthogonalize bvorthog.c:729-767
{
  KS_CHECK(V, KS_ERR_ARG_NULL, "BV is NULL");
  if (R) KS_CHECK(ldr >= V->k, KS_ERR_ARG_SIZ, "Mat size %d is smaller than the number of BV active columns %d", ldr, V->k);
  KS_CHECK(!V->nc, KS_ERR_SUP, "Not implemented for BV with constraints, use BVOrthogonalizeColumn() instead");
  const int l = V->l, k = V->k, nact = k - l;
  if (nact <= 0) return KS_SUCCESS;
  KS_HIP(hipSetDevice(V->ctx->device));
  if (V->orthog_block == KS_BV_ORTHOG_BLOCK_GS) return orthogonalize_gs(V, R, ldr);
  if (nact > 64 && (V->orthog_block == KS_BV_ORTHOG_BLOCK_TSQR || V->orthog_block == KS_BV_ORTHOG_BLOCK_TSQRCHOL)) {
    // the running R factor of the TSQR kernel lives in LDS (64 columns): wider windows go panel by panel, every panel first
    // block-orthogonalised against everything before it (the leading-columns step below), then factored; R fills up column block by column block
    KS_CHECK(!V->matrix, KS_ERR_SUP, "Orthogonalization method not available for non-standard inner product");
    int rc = KS_SUCCESS;
    for (int s = l; s < k && !rc; s += 64) { V->l = s; V->k = std::min(s + 64, k); rc = ks_bv_orthogonalize(V, R, ldr); }
    V->l = l; V->k = k;
    return rc;
  }
  const int ldb = k;
  std::vector<double> Rb((size_t)ldb * k, 0.0), S((size_t)ldb * k, 0.0);     // Rb plays V->Abuffer, S the inverse
  double *R22 = Rb.data() + (size_t)l * ldb + l, *S22 = S.data() + (size_t)l * ldb + l;
  if (l) KS_CALL(block_gs(V, Rb.data(), ldb));
  const double eps = 2.220446049250313e-16;
  switch (V->orthog_block) {
    case KS_BV_ORTHOG_BLOCK_CHOL: {
      KS_CALL(ksb_dot_range(V, l, k, V, l, k, Rb.data(), ldb));
      std::vector<double> G((size_t)nact * nact);
      for (int j = 0; j < nact; j++) for (int i = 0; i < nact; i++) G[(size_t)i + (size_t)j * nact] = R22[(size_t)i + (size_t)j * ldb];
      int info = ksd::potrf_upper(nact, R22, ldb);
      if (info) {                                                      // retry on a diagonally perturbed matrix (bvlapack.c:177-185)
        for (int j = 0; j < nact; j++) { for (int i = 0; i < nact; i++) R22[(size_t)i + (size_t)j * ldb] = G[(size_t)i + (size_t)j * nact]; R22[(size_t)j + (size_t)j * ldb] += 50.0 * eps; }
        info = ksd::potrf_upper(nact, R22, ldb);
        KS_CHECK(!info, KS_ERR_LIB, "Error in LAPACK subroutine potrf: info=%d", info);
      }
      for (int j = 0; j < nact; j++) for (int i = 0; i < nact; i++) { if (i > j) R22[(size_t)i + (size_t)j * ldb] = 0.0; S22[(size_t)i + (size_t)j * ldb] = R22[(size_t)i + (size_t)j * ldb]; }
      info = ksd::trtri_upper(nact, S22, ldb);
      KS_CHECK(!info, KS_ERR_LIB, "Error in LAPACK subroutine trtri: info=%d", info);
      KS_CALL(ks_bv_multinplace(V, S.data(), ldb, l, k));
      store_block(V, Rb.data(), ldb, R, ldr, true);
      break;
    }
    case KS_BV_ORTHOG_BLOCK_SVQB: {
      KS_CALL(ksb_dot_range(V, l, k, V, l, k, Rb.data(), ldb));
      std::vector<double> D(nact), eig(nact), U((size_t)nact * nact);
      for (int i = 0; i < nact; i++) D[i] = 1.0 / sqrt(R22[(size_t)i + (size_t)i * ldb]);
      for (int j = 0; j < nact; j++) for (int i = 0; i < nact; i++) U[(size_t)i + (size_t)j * nact] = R22[(size_t)i + (size_t)j * ldb] * D[i] * D[j];
      const int info = ksd::sym_eig(nact, U.data(), nact, eig.data());
      KS_CHECK(!info, KS_ERR_LIB, "Error in LAPACK subroutine syev: info=%d", info);
      for (int j = 0; j < nact; j++) for (int i = 0; i < nact; i++) {
        const double u = U[(size_t)i + (size_t)j * nact];
        S22[(size_t)i + (size_t)j * ldb] = u * D[i] / sqrt(eig[j]);                          // S = D U Lambda^-1/2
        R22[(size_t)j + (size_t)i * ldb] = u * sqrt(eig[j]) / D[i];                          // R = inv(S) = Lambda^1/2 U' / D
      }
      KS_CALL(ks_bv_multinplace(V, S.data(), ldb, l, k));
      store_block(V, Rb.data(), ldb, R, ldr, false);
      break;
    }
    case KS_BV_ORTHOG_BLOCK_TSQR:
    case KS_BV_ORTHOG_BLOCK_TSQRCHOL: {
      KS_CHECK(!V->matrix, KS_ERR_SUP, "Orthogonalization method not available for non-standard inner product");   // bvorthog.c:750,754
      if (V->orthog_block == KS_BV_ORTHOG_BLOCK_TSQR) {
        KS_CALL(tsqr_q(V, l, nact, R22, ldb));                     // Q from the accumulated reflectors (bvlapack.c:380-451)
      } else {
        KS_CALL(tsqr_r(V, l, nact, R22, ldb));                     // TSQRCHOL: R by TSQR only, Q = V inv(R) (bvlapack.c:483-565)
        for (int j = 0; j < nact; j++) for (int i = 0; i < nact; i++) S22[(size_t)i + (size_t)j * ldb] = R22[(size_t)i + (size_t)j * ldb];
        const int info = ksd::trtri_upper(nact, S22, ldb);
        KS_CHECK(!info, KS_ERR_LIB, "Error in LAPACK subroutine trtri: info=%d", info);
        KS_CALL(ks_bv_multinplace(V, S.data(), ldb, l, k));
      }
      store_block(V, Rb.data(), ldb, R, ldr, true);
      break;
    }
  }
  return KS_SUCCESS;
}

// BVMatProject bvglobal.c:1014-1160 (no inner-product matrix): M(ly:ky, lx:kx) = Y(:,ly:ky)^H A X(:,lx:kx); A NULL = identity
extern "C" int ks_bv_matproject(ks_bv X, ks_mat A, ks_bv Y, double *M, int ldm)
{
  KS_CHECK(X && Y && M, KS_ERR_ARG_NULL, "NULL argument");
  KS_CHECK(X->n == Y->n, KS_ERR_ARG_INCOMP, "Mismatching local dimension X %d, Y %d", X->n, Y->n);
  KS_CHECK(ldm >= Y->k, KS_ERR_ARG_SIZ, "Matrix M has %d rows, should have at least %d", ldm, Y->k);
  if (!A) return ksb_dot_range(X, X->l, X->k, Y, Y->l, Y->k, M, ldm);
  KS_CHECK(A->n == X->n, KS_ERR_ARG_INCOMP, "Mismatching local row dimension A %d, X %d", A->n, X->n);
  ks_bv W = nullptr;
  const int nx = X->k - X->l;
  if (nx <= 0 || Y->k <= Y->l) return KS_SUCCESS;
  KS_CALL(ks_bv_create(X->ctx, X->n, X->N, nx, 0, &W));                  // BVDuplicateResize + BVMatMult (bvglobal.c:1127-1131)
  int rc = KS_SUCCESS;
  for (int j = 0; j < nx && !rc; j++) rc = ks_mat_mult_internal(A, ks_bv_col(X, X->l + j), ks_bv_col(W, j));
  if (!rc) {
    std::vector<double> T((size_t)ldm * nx);
    rc = ksb_dot_range(W, 0, nx, Y, Y->l, Y->k, T.data(), ldm);
    if (!rc) for (int j = 0; j < nx; j++) for (int i = Y->l; i < Y->k; i++) M[(size_t)i + (size_t)(X->l + j) * ldm] = T[(size_t)i + (size_t)j * ldm];
  }
  ks_bv_destroy(W);
  return rc;
}

// BVNormalize bvglobal.c:855-938 (real scalars): scale every active column to unit 2-norm; with eigi, the columns of
// a complex-conjugate pair (eigi[j] != 0) are scaled together by the norm of xr + i*xi
extern "C" int ks_bv_normalize(ks_bv V, const double *eigi)
{
  KS_CHECK(V, KS_ERR_ARG_NULL, "BV is NULL");
  for (int j = V->l; j < V->k; j++) {
    double nr = 0.0;
    KS_CALL(ks_bv_normcolumn(V, j, KS_NORM_2, &nr));
    if (eigi && eigi[j - V->l] != 0.0 && j + 1 < V->k) {
      double ni = 0.0;
      KS_CALL(ks_bv_normcolumn(V, j + 1, KS_NORM_2, &ni));
      const double nrm = hypot(nr, ni);
      KS_CALL(ks_bv_scalecolumn(V, j, 1.0 / nrm)); KS_CALL(ks_bv_scalecolumn(V, j + 1, 1.0 / nrm));
      j++;
    } else KS_CALL(ks_bv_scalecolumn(V, j, 1.0 / nr));
  }
  return KS_SUCCESS;
}
