#!/usr/bin/env python3
"""bench.py -- Arnoldi steps/s of the Krylov-Schur expansion on MI355X (BASELINE.json metric).

A "step" is one Arnoldi/Lanczos step of EPSSolve_KrylovSchur_Default: BVMatMultColumn (CSR SpMV) +
BVOrthonormalizeColumn (CGS with refinement, all passes, scaling).  The K timed steps are the first K
steps of a real Krylov-Schur solve (nev=10, ncv=m=30, tol 1e-8, keep 0.5, largest magnitude), INCLUDING
its restarts (host DS solve + BVMultInPlace + BVCopyColumn): whole-job throughput, inputs resident in HBM.

Workload (config.workload):
  N=1 : BASELINE config 3 - 3-D 7-point Laplacian 216^3 (n = 10 077 696, nnz = 70 263 936)
  N>1 : weak scaling towards config 4 - 432 x 432 x (54 N) grid, each rank owns a 54-plane z-slab
        (10 077 696 rows per GPU; N=8 is the 432^3 = 80.6 M row problem); allreduce of the CGS
        coefficients and the SpMV halo go through RCCL over xGMI.
Launch: `python bench.py --gpus 1 ...` or
        `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s is the measured copy ceiling
NEV, NCV = 10, 30


def make_eps(ks, ctx, A):
    """EPSCreate + EPSSetOperators + EPSSetDimensions; the basis V (ncv+1 columns) is allocated by the first solve
    and reused by later ones, so the timed region holds no allocation (inputs and workspace resident in HBM)."""
    eps = ks.EPS(ctx)
    eps.SetOperators(A)
    eps.SetProblemType(ks.EPS_HEP)
    eps.SetDimensions(NEV, NCV)
    eps.SetTolerances(1e-8, 1 << 30)
    return eps


def run_steps(ks, ctx, A, steps, seed, eps=None):
    """Perform exactly `steps` Arnoldi steps of Krylov-Schur solves on A (fresh solve; restarts on convergence)."""
    done, passes, restarts, solves = 0, 0, 0, 0
    if eps is None:
        eps = make_eps(ks, ctx, A)
    while done < steps:
        eps.SetRandomSeed(seed + solves)
        eps.SetMaxSteps(steps - done)
        eps.Solve()
        st = eps.GetStats()
        if st["arnoldi_steps"] == 0:
            raise RuntimeError("solver made no progress")
        done += st["arnoldi_steps"]; passes += st["gs_passes"]; restarts += st["restarts"]; solves += 1
    return eps, {"steps": done, "gs_passes": passes, "restarts": restarts, "solves": solves}


def cpu_baseline(n_side, max_seconds=30.0):
    """The CPU oracle (port of the reference's CPU path, OpenMP row split) on a bounded sample of the SAME
    workload: the first Lanczos run (m=30 steps, k=1..30) on the 216^3 Laplacian, host cores of this box."""
    from oracle import oracle as O
    want = int(os.environ.get("BENCH_CPU_THREADS", "0")) or O.usable_cores()      # the cgroup quota of the box, not its CPU count
    O.lib(omp=True).orc_set_num_threads(want)
    threads = O.lib(omp=True).orc_num_threads()
    t0 = time.time()
    A = O.laplacian3d(n_side, n_side, n_side, omp=True)
    V = O.BV(A.n, NCV + 1, omp=True)
    V.SetRandomColumn(0)
    _, nrm, _ = V.OrthogonalizeColumn(0)
    V.ScaleColumn(0, 1.0 / nrm)
    import numpy as np
    T = np.zeros((NCV + 1, 3), order="F")
    # time 6 steps first to bound the sample, then the rest of the run if it fits
    t1 = time.time()
    m1 = 6
    V.MatLanczos(A, T, 0, m1)
    dt6 = time.time() - t1
    # steps get more expensive with k (8n(2k+3) per pass): estimate the full run ~ (sum_{k<=30}(168+32k))/(sum_{k<=6}) * dt6
    est = dt6 * sum(168 + 32 * k for k in range(1, NCV + 1)) / sum(168 + 32 * k for k in range(1, m1 + 1))
    if est <= max_seconds:
        t2 = time.time()
        V.MatLanczos(A, T, m1, NCV)
        dt = dt6 + (time.time() - t2)
        steps = NCV
        # then what a Krylov-Schur cycle costs after a restart: steps k = 16..30 against the kept half of the basis,
        # repeated (same vectors every time) until about 12 s of CPU work have been sampled
        k0, cycles = NCV // 2, 0
        while dt < min(12.0, max_seconds):
            t2 = time.time()
            V.MatLanczos(A, T, k0, NCV)
            dt += time.time() - t2
            steps += NCV - k0; cycles += 1
        sample = ("first Lanczos run of the %d^3 workload (%d steps, k=1..%d) + %d restart-cycle expansions (k=%d..%d), CGS2; "
                  "no restart GEMM or projected solve in the CPU sample") % (n_side, NCV, NCV, cycles, k0 + 1, NCV)
    else:
        dt, steps = dt6, m1
        sample = "first %d Lanczos steps (k=1..%d) of the %d^3 workload (full run estimated %.0f s > budget)" % (m1, m1, n_side, est)
    out = {"value": steps / dt, "unit": "steps/s", "cores": threads, "kind": "port", "sample": sample,
           "seconds": round(dt, 3), "setup_seconds": round(t1 - t0, 2), "gs_passes": V.passes_total(),
           "same_6_steps_all_cores": {"value": m1 / dt6, "unit": "steps/s", "cores": threads}}
    # one core (the serial build of the oracle), on the first 6 steps only so that it stays within a few seconds
    try:
        del V
        A1 = O.laplacian3d(n_side, n_side, n_side)
        V1 = O.BV(A1.n, m1 + 2)
        V1.SetRandomColumn(0)
        _, nrm, _ = V1.OrthogonalizeColumn(0)
        V1.ScaleColumn(0, 1.0 / nrm)
        T1 = np.zeros((m1 + 2, 3), order="F")
        t3 = time.time()
        V1.MatLanczos(A1, T1, 0, m1)
        d1 = time.time() - t3
        out["single_core"] = {"value": m1 / d1, "unit": "steps/s", "cores": 1, "seconds": round(d1, 3),
                              "sample": "first %d Lanczos steps (k=1..%d) of the %d^3 workload; compare same_6_steps_all_cores" % (m1, m1, n_side)}
    except Exception as e:      # noqa: BLE001
        out["single_core"] = {"value": None, "sample": "failed: %r" % (e,)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=60)
    ap.add_argument("--side", type=int, default=216, help="grid side per GPU slab (216 -> 10 077 696 rows per GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="disable the per-kernel HIP-event timing")
    args = ap.parse_args()

    # RCCL / the HIP runtime may print banners on stdout: park stdout on stderr until the one JSON line is due
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import slepc_amd as ks

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 bench.py --gpus %d" % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dist = None
    force_dist = os.environ.get("BENCH_FORCE_DIST") == "1"     # rehearse the N>1 code path (RCCL comm, slab grid) on one GPU
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    ctx = ks.Context(local_rank)
    if world > 1 or force_dist:
        idt = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(ks.Context.get_unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, 0)
        ctx.init_rccl(rank, world, bytes(idt.cpu().numpy().tobytes()))

    side = args.side
    if world == 1 and not force_dist:
        nx = ny = nz = side
        A = ks.Mat.laplacian3d(ctx, nx, ny, nz)
        workload = "3-D 7-pt Laplacian %d^3 (BASELINE config 3), Krylov-Schur nev=%d m=%d" % (side, NEV, NCV)
    else:
        nx = ny = 2 * side
        planes = side // 4                      # 54 planes of 432^2 = 216^3 rows per GPU
        nz = planes * world
        A = ks.Mat.laplacian3d(ctx, nx, ny, nz, rank * planes, planes)
        workload = "3-D 7-pt Laplacian %dx%dx%d in %d z-slabs of %d planes (BASELINE config 4 at N=8), Krylov-Schur nev=%d m=%d" % (nx, ny, nz, world, planes, NEV, NCV)

    def barrier():
        ctx.synchronize()                 # the library's own stream
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed: allocate the solver workspace and touch every kernel once (one full cycle + a restart),
    # then the W warmup steps proper: the first W steps of the same solve
    eps = make_eps(ks, ctx, A)
    run_steps(ks, ctx, A, NCV + 15, 0x12345678, eps)
    if args.warmup > 0:
        run_steps(ks, ctx, A, args.warmup, 0x12345678, eps)
    barrier()
    # Timed region. HIP events are recorded only around the Gram-Schmidt update kernel (the dominant kernel
    # symbol, k_gs_update<KT,2>): timing every launch costs ~6 % of the step rate, this subset ~2 %.
    upd_classes = ["gs_update_fused_dot", "gs_update", "gated_noop"]
    if not args.no_prof:
        ctx.prof_enable(True, classes=upd_classes[:2])
        ctx.prof_reset()
    barrier()
    t0 = time.perf_counter()
    eps, st = run_steps(ks, ctx, A, args.steps, 0x12345678, eps)
    barrier()
    t1 = time.perf_counter()
    dt = t1 - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert st["steps"] == args.steps
    prof_timed = {} if args.no_prof else ctx.prof_get(by_variant=True)

    # Untimed second pass of the same K steps with events on every kernel class: the per-kernel table.
    prof = {}
    if not args.no_prof:
        ctx.prof_enable(True)
        ctx.prof_reset()
        _, st2 = run_steps(ks, ctx, A, args.steps, 0x12345678, eps)
        barrier()
        prof = ctx.prof_get(by_variant=True)
        ctx.prof_enable(False)

    if rank == 0:
        n_local = A.n
        out = {
            "metric": "Arnoldi steps/sec (and GB/s vs HBM roofline), 3D Laplacian n=10M, m=30, 1/2/4/8 GPU",
            # weak scaling: the unit is one Arnoldi step on one GPU's 10 077 696-row shard (the N=1 workload), so the
            # whole job processes world*steps of them; the global solver itself advances steps/dt steps per second
            "value": world * args.steps / dt, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "rows_per_gpu": n_local, "nnz_per_gpu": A.nnz, "n_global": A.N, "nev": NEV, "ncv": NCV,
                       "orthog": "CGS, refine ifneeded eta=0.7071", "gs_passes_per_step": st["gs_passes"] / st["steps"],
                       "restarts": st["restarts"], "parallelism": "row-slab x%d" % world,
                       "spmv_layout": A.layout() + (" (2-byte entries: offset code + value code, lossless; y bit-identical to SELL-64; KSGPU_SPMV=sell disables)" if A.layout() == "dict" else ""),
                       "unit_of_value": "Arnoldi steps on a 10 077 696-row shard, summed over the %d shard(s): value = n_gpus * global_steps_per_s" % world,
                       "global_steps_per_s": args.steps / dt},
        }
        # step-level algorithmic traffic by the SURVEY 8d formulas (reference-equivalent work) vs time
        if prof:
            kernels = []
            for (name, var), v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
                sym = ks.kernel_symbol(name, var)
                kernels.append({"class": name, "variant": var, "kernel": sym, "launches": v["launches"], "ms_total": round(v["ms"], 3),
                                "avg_us": round(1e3 * v["ms"] / v["launches"], 2),
                                "alg_GBps": round(v["alg_bytes"] / v["ms"] / 1e6, 1) if v["ms"] > 0 else 0.0,
                                "hbm_GBps": round(v["hbm_bytes"] / v["ms"] / 1e6, 1) if v["ms"] > 0 else 0.0})
            # roofline: the k_gs_update<KT,2> symbol with the largest total time IN THE TIMED REGION.
            # One symbol = fused launches + final launches + launches that exited at their device-side gate.
            sym = {}
            for (name, var), v in prof_timed.items():
                if name in upd_classes and var > 0:
                    e = sym.setdefault(var, {"ms_exec": 0.0, "n_exec": 0, "alg": 0.0, "hbm": 0.0, "ms_noop": 0.0, "n_noop": 0})
                    if name == "gated_noop":
                        e["ms_noop"] += v["ms"]; e["n_noop"] += v["launches"]
                    else:
                        e["ms_exec"] += v["ms"]; e["n_exec"] += v["launches"]; e["alg"] += v["alg_bytes"]; e["hbm"] += v["hbm_bytes"]
            kt, d = max(sym.items(), key=lambda kv: kv[1]["ms_exec"])
            achieved = d["hbm"] / d["ms_exec"] / 1e6          # bytes this kernel must move per launch / its time
            kname = "k_gs_update<%d, 2>" % kt
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic_r01.json")
            if os.path.exists(tpath):
                try:
                    traffic = json.load(open(tpath)).get(kname, {}).get("hbm_bytes_per_executed_launch")
                except Exception:
                    traffic = None
            out["roofline"] = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                               "kernel": kname, "launches_executed": d["n_exec"], "launches_gated_off": d["n_noop"],
                               "avg_launch_us_executed": round(1e3 * d["ms_exec"] / d["n_exec"], 2),
                               "avg_launch_us_all_launches": round(1e3 * (d["ms_exec"] + d["ms_noop"]) / (d["n_exec"] + d["n_noop"]), 2),
                               "bytes_per_executed_launch": d["hbm"] / d["n_exec"],
                               "survey8d_bytes_per_executed_launch": d["alg"] / d["n_exec"],
                               "survey8d_equivalent_GBps": round(d["alg"] / d["ms_exec"] / 1e6, 1),
                               "note": "HIP events on the library stream over the timed region. One symbol is launched in two forms (DESIGN.md "
                                       "section 4): the first CGS pass reads k basis columns and the vector, 8n(k+1) bytes, keeps its result in "
                                       "registers and produces the k+1 dot products of the next pass (work the reference pays another 8n(k+1) "
                                       "bytes for, SURVEY 8d: survey8d_*); the final pass reads the same and writes the vector, 8n(k+2). "
                                       "bytes_per_executed_launch is the mean over both forms, as is the PMC figure in traffic. rocprofv3's "
                                       "per-symbol average covers all launches incl. the ones that exit at their device-side gate: compare "
                                       "avg_launch_us_all_launches."}
            tot_alg = sum(v["alg_bytes"] for v in prof.values()); tot_hbm = sum(v["hbm_bytes"] for v in prof.values())
            tot_ms = sum(v["ms"] for v in prof.values())
            out["step_traffic"] = {"alg_GB_per_step": round(tot_alg / args.steps / 1e9, 3), "alg_GBps_vs_wall": round(tot_alg / dt / 1e9, 1),
                                   "frac_of_hbm_peak_alg": round(tot_alg / dt / 1e9 / HBM_PEAK_GBS, 4),
                                   "compulsory_GB_per_step": round(tot_hbm / args.steps / 1e9, 3), "compulsory_GBps_vs_wall": round(tot_hbm / dt / 1e9, 1),
                                   "kernel_ms_per_step": round(tot_ms / args.steps, 4)}
            # every column count has its own compiled kernel, so the per-symbol table is long: all classes, top symbols
            classes = {}
            for (name, var), v in prof.items():
                c = classes.setdefault(name, {"launches": 0, "ms_total": 0.0, "alg": 0.0, "hbm": 0.0})
                c["launches"] += v["launches"]; c["ms_total"] += v["ms"]; c["alg"] += v["alg_bytes"]; c["hbm"] += v["hbm_bytes"]
            out["kernel_classes_untimed_instrumented_pass"] = [
                {"class": name, "launches": c["launches"], "ms_total": round(c["ms_total"], 3), "ms_per_step": round(c["ms_total"] / args.steps, 4),
                 "alg_GBps": round(c["alg"] / c["ms_total"] / 1e6, 1) if c["ms_total"] > 0 else 0.0,
                 "hbm_GBps": round(c["hbm"] / c["ms_total"] / 1e6, 1) if c["ms_total"] > 0 else 0.0}
                for name, c in sorted(classes.items(), key=lambda kv: -kv[1]["ms_total"])]
            out["kernels_untimed_instrumented_pass"] = kernels[:12]
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(side)
            except Exception as e:       # noqa: BLE001 - the baseline must not take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "steps/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
