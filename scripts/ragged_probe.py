"""Ragged long rows (lengths uniform in [10, 40], columns within +-2000 of the row): the entry-side CSR kernel (the automatic choice: SELL-64 would pad by more than
12.5 %) against SELL-64 forced on the same matrix (KSGPU_SPMV=sell: the padding is stored, not multiplied) - how much a padding-free sliced layout could be worth."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slepc_amd as ks

ctx = ks.Context(0)
n = 1_000_000
rng = np.random.default_rng(0)
lens = rng.integers(10, 41, n)
rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
nnz = int(rowptr[-1])
# PROBE_COLS=near: entry j of a row sits at column row + j - len / 2 (a variable-width band: the gathers of neighbouring rows share their lines);
# default: a random column within +-2000 of the row (every lane of a gather its own line)
if os.environ.get("PROBE_COLS") == "near":
    j = np.arange(nnz) - np.repeat(rowptr[:-1].astype(np.int64), lens)
    col = (np.repeat(np.arange(n), lens) + j - np.repeat(lens // 2, lens)).clip(0, n - 1).astype(np.int32)
else:
    col = (np.repeat(np.arange(n), lens) + rng.integers(-2000, 2001, nnz)).clip(0, n - 1).astype(np.int32)
val = rng.uniform(-1, 1, nnz)
x = rng.standard_normal(n)
ref = None
for fmt in ("auto", "csr", "sell", "csrvec"):
    os.environ.pop("KSGPU_SPMV", None) if fmt == "auto" else os.environ.__setitem__("KSGPU_SPMV", fmt)
    A = ks.Mat.from_csr(ctx, rowptr, col, val)
    y = A.mult(x)
    if ref is None:
        ref = y
    V = ks.BV(ctx, n, 2); V.SetRandomColumn(0)
    for _ in range(10):
        A.mult_dev(V.column_ptr(0), V.column_ptr(1))
    ctx.synchronize()
    ctx.prof_enable(True, classes=["spmv_csr"]); ctx.prof_reset()
    for _ in range(50):
        A.mult_dev(V.column_ptr(0), V.column_ptr(1))
    ctx.synchronize()
    p = ctx.prof_get(); ctx.prof_enable(False)
    us = 1e3 * p["spmv_csr"]["ms"] / p["spmv_csr"]["launches"]
    print("%-7s layout=%-5s %7.1f us  %.2f TB/s of the CSR bytes   max |y - y_first| %.1e" % (fmt, A.layout(), us, (12.0 * nnz + 20.0 * n) / us / 1e6, np.abs(y - ref).max()), flush=True)
    A.destroy(); del V
