"""Kernel time (HIP events) of the dot sweep (BVDotVec) at n = 10 077 696 for several column counts."""
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
for k in (20, 24, 28, 29, 32):
    V.SetActiveColumns(0, k)
    for _ in range(3): V.DotVec(W.column_ptr(0))
    ctx.prof_enable(True); ctx.prof_reset()
    for _ in range(20): V.DotVec(W.column_ptr(0))
    p = ctx.prof_get()["bv_dot_sweep"]; ctx.prof_enable(False)
    us = p["ms"] / p["launches"] * 1e3
    print("dotvec k=%2d: kernel %.1f us  %.0f GB/s (k+1 columns of 8n)" % (k, us, 8.0 * n * (k + 1) / us / 1e3), flush=True)
