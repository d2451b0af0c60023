"""Update and dot sweeps for 33..64 columns (216^3 Laplacian, nev 20, ncv 64): per compiled column-tile size, time and bandwidth by HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slepc_amd as ks
ctx = ks.Context(0)
A = ks.Mat.laplacian3d(ctx, 216, 216, 216)
eps = ks.EPS(ctx); eps.SetOperators(A); eps.SetProblemType(ks.EPS_HEP); eps.SetDimensions(20, 64); eps.SetTolerances(1e-300, 1 << 30); eps.SetMaxSteps(64 + 3 * 22)
eps.Solve()
ctx.prof_enable(True); ctx.prof_reset()
eps.Solve()
prof = ctx.prof_get(by_variant=True)
ctx.prof_enable(False)
for (name, var), v in sorted(prof.items(), key=lambda kv: (kv[0][0], kv[0][1])):
    if name in ("gs_update", "gs_update_fused_dot", "bv_dot_sweep") and v["launches"]:
        print("%-22s KT %2d  launches %3d  avg %8.1f us  %7.1f GB/s (compulsory bytes)" % (name, var, v["launches"], 1e3 * v["ms"] / v["launches"], v["hbm_bytes"] / v["ms"] / 1e6))
