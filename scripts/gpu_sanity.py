"""Quick end-to-end sanity run on a GPU box (developer tool; the real checks live in tests/)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slepc_amd as ks
from oracle import oracle as O

ctx = ks.Context(0)
print(ctx.device_info(), flush=True)

# SpMV
A_o = O.laplacian3d(12, 11, 10)
A = ks.Mat.from_csr(ctx, A_o.rowptr, A_o.col, A_o.val)
x = np.random.default_rng(0).standard_normal(A_o.n)
y = A.mult(x); y0 = A_o.mult(x)
print("spmv err", np.abs(y - y0).max(), flush=True)
A2 = ks.Mat.laplacian3d(ctx, 12, 11, 10)
print("spmv(gen) err", np.abs(A2.mult(x) - y0).max(), flush=True)

# BV test1
n, k, l = 10, 5, 3
X = ks.BV(ctx, n, k); Y = ks.BV(ctx, n, l)
Xh = np.zeros((n, k)); Yh = np.zeros((n, l))
for j in range(k):
    for i in range(4):
        if i + j < n: Xh[i + j, j] = 3 * i + j - 2
for j in range(l): Yh[:, j] = (j + 1) / 4.0
X.set_dense(Xh); Y.set_dense(Yh)
Q = np.array([[2.0 if i < j else -0.5 for j in range(l)] for i in range(k)], order='F')
Y.Mult(2.0, 1.0, X, Q)
print("Mult\n", Y.dense()[:4], flush=True)
z = np.array([2.0 * (-0.5) ** i for i in range(k)])
X.MultVec(-1.0, 1.0, Y.column_ptr(0), z)
M = np.zeros((l, k), order='F'); X.Dot(Y, M); print("Dot\n", M)
print("DotVec", X.DotVec(Y.column_ptr(0)))
X.MultInPlace(Q, 1, l); X.Scale(2.0)
print(X.dense()[:5]); print(X.NormColumn(0), X.Norm(), flush=True)

# Lanczos vs oracle
Ao = O.laplacian2d(30)
Ag = ks.Mat.from_csr(ctx, Ao.rowptr, Ao.col, Ao.val)
m = 12
Vo = O.BV(Ao.n, m + 1); Vg = ks.BV(ctx, Ao.n, m + 1)
Vo.SetRandomColumn(0); Vg.SetRandomColumn(0)
print("rand col diff", np.abs(Vo.column(0) - Vg.column(0)).max())
_, nrm, _ = Vo.OrthogonalizeColumn(0); Vo.ScaleColumn(0, 1 / nrm)
_, nrmg, _ = Vg.OrthogonalizeColumn(0); Vg.ScaleColumn(0, 1 / nrmg)
print("norm0", nrm, nrmg)
To = np.zeros((m + 1, 3), order='F'); Tg = np.zeros((m + 1, 3), order='F')
print(Vo.MatLanczos(Ao, To, 0, m), Vg.MatLanczos(Ag, Tg, 0, m))
print("T diff", np.abs(To - Tg).max(), "V diff", np.abs(Vo.dense() - Vg.dense()).max(), "passes", Vo.passes_total(), Vg.gs_passes(), flush=True)

# EPS ex2
Ao = O.laplacian2d(72)
Ag = ks.Mat.laplacian2d(ctx, 72)
t0 = time.time()
eps = ks.EPS(ctx); eps.SetOperators(Ag); eps.SetProblemType(ks.EPS_HEP); eps.SetDimensions(4, 20); eps.Solve()
print("EPS", eps.GetConverged(), eps.GetIterationNumber(), [eps.GetEigenvalue(i)[0] for i in range(4)], eps.GetStats(), time.time() - t0)
print([eps.ComputeError(i) for i in range(4)])
r = O.eps_krylovschur_hep(Ao, 4, ncv=20)
print("oracle", r.nconv, r.its, r.eigr[r.perm][:4], r.steps, r.passes, flush=True)

# timing, 3D
for N in (100, 216):
    Ag = ks.Mat.laplacian3d(ctx, N, N, N)
    eps = ks.EPS(ctx); eps.SetOperators(Ag); eps.SetProblemType(ks.EPS_HEP); eps.SetDimensions(10, 30); eps.SetMaxSteps(60); eps.Solve()  # warm
    ctx.prof_enable(True); ctx.prof_reset()
    eps.SetMaxSteps(150)
    t0 = time.time(); eps.Solve(); ctx.synchronize(); dt = time.time() - t0
    st = eps.GetStats()
    print(N, "steps", st, "time", dt, "steps/s", st["arnoldi_steps"] / dt)
    for kname, v in ctx.prof_get().items():
        print("   %-22s n=%6d  ms=%9.3f  GB/s=%8.1f" % (kname, v["launches"], v["ms"], v["alg_bytes"] / v["ms"] / 1e6 if v["ms"] > 0 else 0))
    ctx.prof_enable(False)
    del eps, Ag
