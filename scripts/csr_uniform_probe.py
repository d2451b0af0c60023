"""The two forms of the short-row CSR product (KSGPU_SPMV=csr: LDS-DMA, lane-linear image; csrregs: register-staged, skewed image) on matrices whose rows ALL
have the same length L = 4 .. 12, columns in a band: where the lane-linear image meets LDS bank conflicts (L a multiple of 8), does the DMA form still win?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slepc_amd as ks

ctx = ks.Context(0)
n = 2_000_000
rng = np.random.default_rng(0)
for L in [int(v) for v in os.environ.get("PROBE_LENS", "4,6,7,8,9,12").split(",")]:
    rowptr = (np.arange(n + 1, dtype=np.int64) * L).astype(np.int32)
    col = (np.repeat(np.arange(n), L) + np.tile(np.arange(L) * 97 - 300, n)).clip(0, n - 1).astype(np.int32)
    val = rng.uniform(-1, 1, n * L)
    out = {}
    for fmt in ("csr", "csrregs"):
        os.environ["KSGPU_SPMV"] = fmt
        A = ks.Mat.from_csr(ctx, rowptr, col, val)
        V = ks.BV(ctx, n, 2); V.SetRandomColumn(0)
        x, y = V.column_ptr(0), V.column_ptr(1)
        for _ in range(10):
            A.mult_dev(x, y)
        ctx.synchronize()
        ctx.prof_enable(True, classes=["spmv_csr"]); ctx.prof_reset()
        for _ in range(50):
            A.mult_dev(x, y)
        ctx.synchronize()
        p = ctx.prof_get(); ctx.prof_enable(False)
        out[fmt] = 1e3 * p["spmv_csr"]["ms"] / p["spmv_csr"]["launches"]
        A.destroy(); del V
    b = 12.0 * n * L + 20.0 * n
    print("rows of %2d entries: LDS-DMA %6.1f us (%.2f TB/s)   register-staged %6.1f us (%.2f TB/s)" % (L, out["csr"], b / out["csr"] / 1e6, out["csrregs"], b / out["csrregs"] / 1e6), flush=True)
