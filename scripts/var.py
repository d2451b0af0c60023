import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import slepc_amd as ks
import bench
ctx = ks.Context(0)
A = ks.Mat.laplacian3d(ctx, 216, 216, 216)
eps = bench.make_eps(ks, ctx, A)
bench.run_steps(ks, ctx, A, 30, 1, eps)
torch.cuda.synchronize()
for mode in ("off", "upd", "upd", "all", "off", "off", "upd", "upd"):
    if mode == "off": ctx.prof_enable(False)
    elif mode == "upd": ctx.prof_enable(True, classes=["gs_update_fused_dot", "gs_update"]); ctx.prof_reset()
    else: ctx.prof_enable(True); ctx.prof_reset()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bench.run_steps(ks, ctx, A, 150, 1, eps)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(mode, "%.1f steps/s  %.4f ms/step" % (150 / dt, 1e3 * dt / 150), flush=True)
