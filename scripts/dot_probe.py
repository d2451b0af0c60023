"""Time of the dot sweep (BVDotVec) and of the update (BVMultVec) at n = 10 077 696 for several column counts."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slepc_amd as ks
ctx = ks.Context(0)
n = 10077696
V = ks.BV(ctx, n, 34)
W = ks.BV(ctx, n, 1)
for j in range(34):
    V.SetRandomColumn(j)
W.SetRandomColumn(0)
for k in (20, 21, 24, 25, 26, 27, 28, 29, 32):
    V.SetActiveColumns(0, k)
    for _ in range(3): V.DotVec(W.column_ptr(0))
    ctx.L.ks_ctx_synchronize(ctx.h); t = time.perf_counter()
    for _ in range(20): V.DotVec(W.column_ptr(0))
    ctx.L.ks_ctx_synchronize(ctx.h); dt = (time.perf_counter() - t) / 20
    print("dotvec k=%2d: %.1f us  %.0f GB/s (k+1 columns of 8n)" % (k, dt * 1e6, 8.0 * n * (k + 1) / dt / 1e9), flush=True)
