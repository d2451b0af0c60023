"""Time of the MFMA panel kernels at n = 216^3: BVDot(X,Y) with 31+31 columns, the Gram matrix BVDot(X,X), and the
restart product BVMultInPlace (30 columns in, 20 out)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slepc_amd as ks

ctx = ks.Context(0)
n, m = 216 ** 3, 32
X = ks.BV(ctx, n, m); Y = ks.BV(ctx, n, m)
for j in range(m):
    X.SetRandomColumn(j); Y.SetRandomColumn(j, 7)
X.SetActiveColumns(0, 31); Y.SetActiveColumns(0, 31)
M = np.zeros((31, 31), order="F")

def timed(f, reps=20):
    """kernel time only: HIP events of the library's profiler around the panel kernel launches (the wall time of a call also
    holds the block reduction, the D2H of the result and a stream synchronisation)"""
    f(); ctx.synchronize()
    ctx.prof_enable(True, classes=["bv_dot_panel", "bv_multinplace", "bv_mult"]); ctx.prof_reset()
    for _ in range(reps):
        f()
    ctx.synchronize()
    p = ctx.prof_get(); ctx.prof_enable(False)
    return sum(v["ms"] for v in p.values()) / sum(v["launches"] for v in p.values()) * 1e-3

t = timed(lambda: X.Dot(Y, M)); print("BVDot(X,Y) 31x31: %.0f us  %.2f TB/s (62 columns)" % (t * 1e6, 62 * 8.0 * n / t / 1e12))
t = timed(lambda: X.Dot(X, M)); print("BVDot(X,X) 31x31: %.0f us  %.2f TB/s of the 31 columns it has to read, %.2f TB/s in BVDot's 62" % (t * 1e6, 31 * 8.0 * n / t / 1e12, 62 * 8.0 * n / t / 1e12))
G = M.copy(); X.Dot(Y, M); X.Dot(X, M)
Xh = None
Q = np.asfortranarray(np.random.default_rng(0).standard_normal((31, 31)))
X.SetActiveColumns(0, 30)
t = timed(lambda: X.MultInPlace(Q, 0, 20), reps=5); print("BVMultInPlace 30 -> 20: %.0f us  %.2f TB/s" % (t * 1e6, 50 * 8.0 * n / t / 1e12))
