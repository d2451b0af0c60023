// Host-side CSR helpers of the assembly path (no device code).
#pragma once
#include <vector>

namespace ksc {
// P = A + alpha * B on CSR arrays of the same row block (global column indices): MatDuplicate + MatAXPY(P, alpha, B, DIFFERENT_NONZERO_PATTERN),
// the way STMatMAXPY_Private assembles A - sigma B in ST_MATMODE_COPY (src/sys/classes/st/interface/stsolve.c:611-626). rpb == nullptr: B = I
// (MatShift, the nmat = 1 branch, stsolve.c:625), the diagonal of row r being global column row_start + r.
// Entry by entry p_ij = a_ij + (alpha * b_ij) (two roundings: the scaled entry is added as a value of its own), alpha * b_ij where only B has the
// entry, a_ij where only A has it. Rows whose columns ascend strictly in both operands are merged and come out sorted (PETSc's AIJ rows always
// are); a row that is not (repeated or unordered columns, which ks_mat_create_csr accepts) keeps A's entries as they stand, an entry of B going
// to the first entry of A with its column, or to the end of the row. Returns false (nothing built) when the result has more than 2^31 - 1 entries.
bool csr_axpy(int n, int row_start, const int *rpa, const int *ca, const double *va, double alpha, const int *rpb, const int *cb, const double *vb,
              std::vector<int> &rp, std::vector<int> &col, std::vector<double> &val);
// B = A^T for a CSR block of nrows x ncols (MatTranspose, MAT_INITIAL_MATRIX): a counting sort by column; the rows of B list their entries in
// ascending row order of A (sorted columns out whatever the order inside A's rows; repeated entries of A stay separate entries).
void csr_transpose(int nrows, int ncols, const int *rp, const int *col, const double *val, std::vector<int> &rpt, std::vector<int> &colt, std::vector<double> &valt);
}
