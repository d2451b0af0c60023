"""Per-kernel-class time of the config-5-shaped solve (n rows, 32 nnz/row, sinvert at 0, step cap)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import slepc_amd as ks
import nhep_cases as nc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000000
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 120
ctx = ks.Context(0)
Ao, Bo = nc.config5_pencil_fast(n)
A = ks.Mat.from_csr(ctx, Ao.rowptr, Ao.col, Ao.val); B = ks.Mat.from_csr(ctx, Bo.rowptr, Bo.col, Bo.val)
print("nnz A", A.nnz, "nnz B", B.nnz, flush=True)
# raw SpMV timing
V = ks.BV(ctx, n, 2); V.set_column(0, np.random.default_rng(0).standard_normal(n))
for M, name in ((A, "A"), (B, "B")):
    for _ in range(3): M.mult_dev(V.column_ptr(0), V.column_ptr(1))
    ctx.L.ks_ctx_synchronize(ctx.h); t = time.time()
    for _ in range(20): M.mult_dev(V.column_ptr(0), V.column_ptr(1))
    ctx.L.ks_ctx_synchronize(ctx.h); dt = (time.time() - t) / 20
    print("SpMV %s: %.3f ms, %.1f GB/s algorithmic" % (name, dt * 1e3, M.spmv_bytes() / dt / 1e9), flush=True)
eps = ks.EPS(ctx)
eps.SetOperators(A, B); eps.SetProblemType(ks.EPS_GNHEP); eps.SetDimensions(20, 60); eps.SetTarget(0.0)
st = eps.GetST(); st.SetType("sinvert")
eps.SetMaxSteps(cap)
eps.Solve()                       # warm (allocations)
ctx.prof_enable(True); ctx.prof_reset()
t = time.time(); eps.Solve(); dt = time.time() - t
ctx.prof_enable(False)
steps = eps.GetStats()["arnoldi_steps"]; s = st.GetKSPStats()
print("steps %d in %.2f s -> %.1f steps/s ; %.2f ms/step" % (steps, dt, steps / dt, dt / steps * 1e3), flush=True)
tot = 0
for k, v in sorted(ctx.prof_get().items(), key=lambda kv: -kv[1]["ms"]):
    print("  %-22s launches %7d  ms %9.2f  ms/step %.3f" % (k, v["launches"], v["ms"], v["ms"] / steps)); tot += v["ms"]
print("  kernel total ms/step %.3f" % (tot / steps))
