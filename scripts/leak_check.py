"""Device-memory leak check: free memory (hipMemGetInfo through torch) before and after many create/solve/destroy cycles
covering the solver variants, ST types, constraints, wide bases and block orthogonalisation."""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import slepc_amd as ks
import scenarios as sc
import nhep_cases as nc
from oracle import oracle as O

ctx = ks.Context(0)


def cycle(i):
    S = sc.graph_laplacian_2d(20, 17)
    A = ks.Mat.from_csr(ctx, S.indptr, S.indices, S.data)
    eps = ks.EPS(ctx); eps.SetOperators(A); eps.SetProblemType(ks.EPS_HEP); eps.SetDimensions(3, 16 if i % 2 else 80)
    eps.SetWhichEigenpairs("smallest_real"); eps.SetDeflationSpace(np.ones((S.shape[0], 1))); eps.Solve()
    Ao = nc.planted_pairs(600)
    An = ks.Mat.from_csr(ctx, Ao.rowptr, Ao.col, Ao.val)
    e2 = ks.EPS(ctx); e2.SetOperators(An); e2.SetProblemType(ks.EPS_NHEP); e2.SetDimensions(3, 20); e2.SetTarget(-2.0)
    st = e2.GetST(); st.SetType(["sinvert", "cayley"][i % 2]); st.SetKSP(rtol=1e-12)
    if i % 2:
        st.CayleySetAntishift(5.0)
    e2.Solve()
    Lo = O.laplacian2d(24, 17)
    L = ks.Mat.from_csr(ctx, Lo.rowptr, Lo.col, Lo.val)
    d = 1.0 + 0.5 * np.cos(np.arange(Lo.n)) ** 2
    B = ks.Mat.from_csr(ctx, np.arange(Lo.n + 1, dtype=np.int32), np.arange(Lo.n, dtype=np.int32), d)
    e3 = ks.EPS(ctx); e3.SetOperators(L, B); e3.SetProblemType(ks.EPS_GHEP); e3.SetDimensions(3, 16); e3.GetST().SetKSP(rtol=1e-13); e3.Solve()
    V = ks.BV(ctx, 3000, 90); V.set_dense(np.random.default_rng(i).standard_normal((3000, 90)))
    V.SetOrthogBlock(["chol", "svqb", "gs", "tsqr"][i % 4])
    if i % 4 == 3:
        V.SetActiveColumns(0, 40)
    V.Orthogonalize(None)
    for o in (eps, e2, e3):
        assert o.GetConverged() >= 1
    # the binned and the XCD-sliced SpMV layouts (forced on a small random matrix), and the communicator's one-shot mailboxes
    os.environ["KSGPU_SPMV"] = ["binned", "sliced"][i % 2]
    rng = np.random.default_rng(i)
    nb = 6000; rp = np.arange(0, 8 * nb + 1, 8, dtype=np.int32)
    M = ks.Mat.from_csr(ctx, rp, rng.integers(0, nb, 8 * nb).astype(np.int32), rng.uniform(-1, 1, 8 * nb))
    del os.environ["KSGPU_SPMV"]
    assert M.layout() == ["binned", "sliced"][i % 2]
    M.mult(np.ones(nb))
    M.destroy()
    del eps, e2, e3, V, A, An, L, B, st


cycle(0); cycle(1); cycle(2); cycle(3)
gc.collect(); ctx.synchronize()
free0, total = torch.cuda.mem_get_info(0)
N = 40
for i in range(N):
    cycle(i)
gc.collect(); ctx.synchronize()
free1, _ = torch.cuda.mem_get_info(0)
print("free before %.1f MiB, after %d cycles %.1f MiB, difference %.2f MiB" % (free0 / 2**20, N, free1 / 2**20, (free0 - free1) / 2**20))
assert free0 - free1 < 8 * 2**20, "device memory leak"
print("no leak")
