#include "ks_csr.h"
#include <algorithm>
#include <thread>
#include <sched.h>
#include <system_error>

#pragma clang fp contract(off)      // p_ij = a_ij + (alpha * b_ij): the product is rounded before the sum, as two MatSetValues would

namespace ksc {
namespace {
struct RowB { const int *c; const double *v; int len; int idc; double idv; };      // a row of B, or the one diagonal entry of I
inline bool ascending(const int *c, int len) { for (int i = 1; i < len; i++) if (c[i] <= c[i - 1]) return false; return true; }

// one row; out == nullptr: count only. Returns the row's length.
inline int merge_row(const int *ca, const double *va, int la, const int *cb, const double *vb, int lb, double alpha, int *oc, double *ov)
{
  if (ascending(ca, la) && ascending(cb, lb)) {
    int i = 0, j = 0, k = 0;
    while (i < la || j < lb) {
      if (j >= lb || (i < la && ca[i] < cb[j])) { if (oc) { oc[k] = ca[i]; ov[k] = va[i]; } i++; }
      else if (i >= la || cb[j] < ca[i]) { if (oc) { oc[k] = cb[j]; const double t = alpha * vb[j]; ov[k] = t; } j++; }
      else { if (oc) { oc[k] = ca[i]; const double t = alpha * vb[j]; ov[k] = va[i] + t; } i++; j++; }
      k++;
    }
    return k;
  }
  int k = la;
  if (oc) for (int i = 0; i < la; i++) { oc[i] = ca[i]; ov[i] = va[i]; }
  for (int j = 0; j < lb; j++) {
    int hit = -1;
    for (int i = 0; i < la && hit < 0; i++) if (ca[i] == cb[j]) hit = i;
    if (!oc) { if (hit < 0) k++; continue; }
    const double t = alpha * vb[j];
    if (hit >= 0) ov[hit] = ov[hit] + t; else { oc[k] = cb[j]; ov[k] = t; k++; }
  }
  return k;
}
} // namespace

bool csr_axpy(int n, int row_start, const int *rpa, const int *ca, const double *va, double alpha, const int *rpb, const int *cb, const double *vb,
              std::vector<int> &rp, std::vector<int> &col, std::vector<double> &val)
{
  rp.assign((size_t)n + 1, 0);
  unsigned ncpu = std::thread::hardware_concurrency();
  { cpu_set_t cs; CPU_ZERO(&cs); if (sched_getaffinity(0, sizeof(cs), &cs) == 0 && CPU_COUNT(&cs) > 0) ncpu = (unsigned)CPU_COUNT(&cs); }
  const unsigned nthr = (n < 100000) ? 1u : std::max(1u, std::min(16u, ncpu));
  const double one = 1.0;
  auto rows = [&](bool fill, int r0, int r1) {
    for (int r = r0; r < r1; r++) {
      const int id = row_start + r;
      const int *bc = rpb ? cb + rpb[r] : &id; const double *bv = rpb ? vb + rpb[r] : &one; const int lb = rpb ? rpb[r + 1] - rpb[r] : 1;
      if (!fill) rp[r + 1] = merge_row(ca + rpa[r], va + rpa[r], rpa[r + 1] - rpa[r], bc, bv, lb, alpha, nullptr, nullptr);
      else merge_row(ca + rpa[r], va + rpa[r], rpa[r + 1] - rpa[r], bc, bv, lb, alpha, col.data() + rp[r], val.data() + rp[r]);
    }
  };
  auto parallel = [&](bool fill) {
    std::vector<std::thread> th;
    const int chunk = (n + (int)nthr - 1) / (int)nthr;
    unsigned started = 1;
    for (unsigned t = 1; t < nthr; t++) {
      try { th.emplace_back([&, t] { rows(fill, std::min(n, (int)t * chunk), std::min(n, (int)(t + 1) * chunk)); }); started = t + 1; }
      catch (const std::system_error &) { break; }
    }
    rows(fill, 0, std::min(n, chunk));
    for (unsigned t = started; t < nthr; t++) rows(fill, std::min(n, (int)t * chunk), std::min(n, (int)(t + 1) * chunk));     // chunks whose thread did not start
    for (auto &x : th) x.join();
  };
  parallel(false);
  long long total = 0;
  for (int r = 0; r < n; r++) total += rp[r + 1];
  if (total > 2147483647LL) return false;
  for (int r = 0; r < n; r++) rp[r + 1] += rp[r];
  col.resize((size_t)rp[n]); val.resize((size_t)rp[n]);
  parallel(true);
  return true;
}
void csr_transpose(int nrows, int ncols, const int *rp, const int *col, const double *val, std::vector<int> &rpt, std::vector<int> &colt, std::vector<double> &valt)
{
  const size_t nnz = (size_t)rp[nrows];
  rpt.assign((size_t)ncols + 1, 0); colt.resize(nnz); valt.resize(nnz);
  for (size_t p = 0; p < nnz; p++) rpt[(size_t)col[p] + 1]++;
  for (int c = 0; c < ncols; c++) rpt[c + 1] += rpt[c];
  std::vector<int> cur(rpt.begin(), rpt.end() - 1);
  for (int r = 0; r < nrows; r++)
    for (int p = rp[r]; p < rp[r + 1]; p++) { const int q = cur[col[p]]++; colt[q] = r; valt[q] = val[p]; }
}
} // namespace ksc

#ifdef KSD_TEST_HOOKS
extern "C" {
// test hook: B = A^T (square blocks of order n); rpt has n + 1 entries, colt / valt nnz
void ksc_csr_transpose(int n, const int *rp, const int *col, const double *val, int *rpt, int *colt, double *valt)
{
  std::vector<int> r, c; std::vector<double> v;
  ksc::csr_transpose(n, n, rp, col, val, r, c, v);
  std::copy(r.begin(), r.end(), rpt); std::copy(c.begin(), c.end(), colt); std::copy(v.begin(), v.end(), valt);
}
// test hook: P = A + alpha B (B arrays NULL: the identity); returns nnz(P), fills rp always and col/val when they are given (capacity cap entries)
long long ksc_csr_axpy(int n, int row_start, const int *rpa, const int *ca, const double *va, double alpha, const int *rpb, const int *cb, const double *vb,
                       int *rp, int *col, double *val, long long cap)
{
  std::vector<int> r, c; std::vector<double> v;
  if (!ksc::csr_axpy(n, row_start, rpa, ca, va, alpha, rpb, cb, vb, r, c, v)) return -1;
  std::copy(r.begin(), r.end(), rp);
  if (col && val && (long long)c.size() <= cap) { std::copy(c.begin(), c.end(), col); std::copy(v.begin(), v.end(), val); }
  return (long long)c.size();
}
}
#endif
