"""Throughput of the other BASELINE configs on one GPU (not the bench line): C1 (2-D 100^2), C2 (2-D 1000^2), each
nev=4 m=20, and C3 for reference; prints steps/s and the per-class kernel time."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slepc_amd as ks

ctx = ks.Context(0)
for name, mk, nev, ncv, steps in [("C1 2-D 100^2", lambda: ks.Mat.laplacian2d(ctx, 100), 4, 20, 2000),
                                  ("C2 2-D 1000^2", lambda: ks.Mat.laplacian2d(ctx, 1000), 4, 20, 2000),
                                  ("C3 3-D 216^3", lambda: ks.Mat.laplacian3d(ctx, 216, 216, 216), 10, 30, 300)]:
    A = mk()
    eps = ks.EPS(ctx); eps.SetOperators(A); eps.SetProblemType(ks.EPS_HEP); eps.SetDimensions(nev, ncv); eps.SetTolerances(1e-8, 1 << 30)
    def run(k):
        done = 0; s = 0
        while done < k:
            eps.SetRandomSeed(100 + s); eps.SetMaxSteps(k - done); eps.Solve(); done += eps.GetStats()["arnoldi_steps"]; s += 1
        return done
    run(ncv + 40)
    ctx.L.ks_ctx_synchronize(ctx.h); t = time.time(); d = run(steps); ctx.L.ks_ctx_synchronize(ctx.h); dt = time.time() - t
    ctx.prof_enable(True); ctx.prof_reset(); d2 = run(min(steps, 400)); ctx.prof_enable(False)
    prof = ctx.prof_get()
    ktot = sum(v["ms"] for v in prof.values())
    print("%s: n=%d  %.1f steps/s (%.1f us/step); kernel time %.1f us/step: %s" % (
        name, A.n, d / dt, dt / d * 1e6, ktot / d2 * 1e3,
        ", ".join("%s %.1f" % (k, v["ms"] / d2 * 1e3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:6])), flush=True)
    del eps, A
