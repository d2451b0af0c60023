#!/usr/bin/env python3
"""bench.py -- Arnoldi steps/s of the Krylov-Schur expansion on MI355X (BASELINE.json metric).

A "step" is one Arnoldi/Lanczos step of EPSSolve_KrylovSchur_Default: BVMatMultColumn (CSR SpMV) +
BVOrthonormalizeColumn (CGS with refinement, all passes, scaling).

What is timed.  ONE Krylov-Schur solve (nev=10, ncv=m=30, tol 1e-8, keep 0.5, largest magnitude) runs through
warm-up, timed region and an instrumented tail without being restarted in between; the phase boundaries sit at restart
boundaries of that solve (the solver's stopping-test callback, krylovschur.c:289), so the timed region is a whole number
of restart cycles in steady state - expansion (k = 16..30 here), host projected solve, restart BVMultInPlace +
BVCopyColumn - and its rate does not depend on --steps:
  warm-up : the first cycle (k = 1..30) and whole cycles until at least --warmup steps (>= 45) have run;
  timed   : whole cycles until at least max(--steps, --min-steps) steps have run (--min-steps 200 keeps the region
            above 200 ms on one GPU); "steps" in the JSON line is the number actually timed;
  tail    : the same number of steps again with HIP events on every kernel class (untimed): the per-kernel table.
Inputs and workspace are resident in HBM when the timed region starts.

Workload (config.workload):
  N=1 : BASELINE config 3 - 3-D 7-point Laplacian 216^3 (n = 10 077 696, nnz = 70 263 936)
  N>1 : weak scaling towards config 4 - 432 x 432 x (54 N) grid, each rank owns a 54-plane z-slab
        (10 077 696 rows per GPU; N=8 is the 432^3 = 80.6 M row problem); allreduce of the CGS
        coefficients and the SpMV halo go through RCCL over xGMI. That is `value`. Beside it, measured by ONE child process of
        every rank after the headline (one JSON line per finished leg, stopped after 240 s): `strong_scaling` - the SAME 216^3
        problem cut into N z-slabs (the metric's literal "n=10M ... 1/2/4/8 GPU") - and `oneshot_allreduce` - both legs again with
        the one-shot allreduce and the peer-mapped halo. Every leg carries `multi_gpu_breakdown` (allreduce / halo / restart
        broadcast per step and per rank, rank skew).
Launch: `python bench.py --gpus N ...` starts its N ranks itself (one fresh child process per GPU through
        torch.distributed.run, before this process touches the GPU); under an existing launcher
        (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`) it is one of the ranks.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s is the measured copy ceiling
NEV, NCV = 10, 30
UPD_CLASSES = ["gs_update_fused_dot", "gs_update", "gated_noop"]
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic_r04.json")      # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE summary of this command


class Phases:
    """Stopping-test callback of one solver: moves through warm-up -> timed -> instrumented tail at restart boundaries.

    Every rank runs the same replicated control flow (identical step counts), so the boundaries - which hold a barrier -
    fall on the same restart everywhere without any extra collective."""

    def __init__(self, ks, ctx, eps, barrier, warmup, steps, tail_steps, timed_classes):
        self.ks, self.ctx, self.eps, self.barrier = ks, ctx, eps, barrier
        self.warmup, self.steps, self.tail_steps, self.timed_classes = warmup, steps, tail_steps, timed_classes
        self.phase = 0
        self.base = 0                 # steps of earlier solves (a solve that converges inside the region is followed by a fresh one)
        self.marks = {}               # name -> (steps, passes, time)
        self.cycles = []              # (phase, steps at this restart)
        self.prof_timed, self.prof_tail = {}, {}
        self.bcast = (0, 0.0)         # (calls, host seconds) of the projected-problem broadcast inside the timed region
        # the callback runs once per restart with the GPU idle behind it (the restart's product waits for the stopping test): keep it to one
        # library call and plain integer tests
        import ctypes as C
        self._s, self._p, self._r = C.c_longlong(), C.c_longlong(), C.c_int()
        self._get = (ctx.L.ks_eps_get_stats, eps.h, C.byref(self._s), C.byref(self._p), C.byref(self._r)) if hasattr(ctx, "L") else None

    def _steps(self):
        if self._get is None:
            return self.base + self.eps.GetStats()["arnoldi_steps"]
        f, h, a, b, c = self._get
        f(h, a, b, c)
        return self.base + self._s.value

    def _passes(self):
        return self.eps.GetBV().gs_passes()[0]

    def _mark(self, name, steps):
        t = self.barrier()
        self.marks[name] = (steps, self._passes(), t)
        # host time inside the per-restart broadcast of the projected problem (N > 1), over the timed region
        if hasattr(self.ctx, "bcast_stats"):
            if name == "t0":
                self.ctx.bcast_stats(reset=True)
            elif name == "t1":
                self.bcast = self.ctx.bcast_stats()

    def __call__(self, its, max_it, nconv, nev):
        steps = self._steps()
        self.cycles.append((self.phase, steps))
        if self.phase == 0 and steps >= self.warmup:
            if self.timed_classes is not None:
                self.ctx.prof_enable(True, classes=self.timed_classes)
                self.ctx.prof_reset()
            self.phase = 1
            self._mark("t0", steps)
        elif self.phase == 1 and steps - self.marks["t0"][0] >= self.steps:
            self._mark("t1", steps)
            if self.tail_steps > 0:
                self.prof_timed = self.ctx.prof_get(by_variant=True) if self.timed_classes is not None else {}
                self.ctx.prof_enable(True)
                self.ctx.prof_reset()
                self.phase = 2
            else:
                if self.timed_classes is not None:
                    self.prof_timed = self.ctx.prof_get(by_variant=True)
                self.phase = 3
                return self.ks.EPS_CONVERGED_USER
        elif self.phase == 2 and steps - self.marks["t1"][0] >= self.tail_steps:
            self._mark("t2", steps)
            self.prof_tail = self.ctx.prof_get(by_variant=True)
            self.ctx.prof_enable(False)
            self.phase = 3
            return self.ks.EPS_CONVERGED_USER
        # EPSStoppingBasic (epsdefault.c:290-303; EPS.StoppingBasic is the library's own)
        if nconv >= nev:
            return self.ks.EPS_CONVERGED_TOL
        return self.ks.EPS_DIVERGED_ITS if its >= max_it else 0

    def run(self, seed):
        solves = 0
        while self.phase < 3:
            self.eps.SetRandomSeed(seed + solves)
            self.eps.Solve()
            self.base += self.eps.GetStats()["arnoldi_steps"]
            solves += 1
            if solves > 1000:
                raise RuntimeError("solver made no progress")
        return solves

    def timed(self, ncv):
        (s0, p0, t0), (s1, p1, t1) = self.marks["t0"], self.marks["t1"]
        ends = [s for (ph, s) in self.cycles if ph == 1]                     # restarts inside the timed region (the last one closes it)
        starts = [s0] + ends[:-1]
        lens = [e - b for b, e in zip(starts, ends)]
        # column orthogonalised by a step = number of previous columns k; a cycle of L steps ends at column ncv: k = ncv-L+1 .. ncv
        ksum = sum(L * (2 * ncv - L + 1) / 2.0 for L in lens)
        return {"steps": s1 - s0, "seconds": t1 - t0, "gs_passes": p1 - p0, "cycles": len(lens), "cycle_steps": lens,
                "mean_k": ksum / max(1, s1 - s0)}


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
        dt_first = dt6 + (time.time() - t2)
        # The steady state the GPU line times: restart cycles = the restart itself (BVMultInPlace of the 30 active columns by an
        # orthogonal 30 x 15 factor, the host's Q of a projected solve stands in for by a fixed orthogonal matrix; BVCopyColumn of the
        # residual column) + the expansion, steps k = 16..30 against the kept half of the basis. SURVEY 8d: at least 3 repetitions,
        # median. A repetition is `per_rep` cycles, sized from one probe cycle so that the three of them take about 12 s.
        k0 = NCV // 2
        Qr, _ = np.linalg.qr(np.random.default_rng(7).standard_normal((NCV, NCV)))
        Qf = np.asfortranarray(Qr)

        def cycle():
            V.SetActiveColumns(0, NCV)
            V.MultInPlace(Qf, 0, k0)              # krylovschur.c:326 BVMultInPlace(eps->V,U,eps->nconv,k+l): V(:,0:k0) = V(:,0:ncv) Q(:,0:k0)
            V.CopyColumn(NCV, k0)                  # krylovschur.c:329
            V.SetActiveColumns(0, NCV + 1)
            V.MatLanczos(A, T, k0, NCV)
        t2 = time.time(); cycle(); probe = time.time() - t2
        per_rep = max(1, min(40, int(min(12.0, max_seconds) / 3.0 / max(probe, 1e-3))))
        reps = []
        for _ in range(3):
            t2 = time.time()
            for _ in range(per_rep):
                cycle()
            reps.append(time.time() - t2)
        dt = sorted(reps)[1]; steps = per_rep * (NCV - k0); cycles = per_rep
        sample = ("median of 3 repetitions of %d restart cycles each (BVMultInPlace 30 -> 15 columns + BVCopyColumn + expansion k=%d..%d, CGS2) of the %d^3 "
                  "workload after its first Lanczos run; repetition times %s s; the projected solve (O(m^3) on the host) is not in the CPU sample; "
                  "%s"
                  % (cycles, k0 + 1, NCV, n_side, ", ".join("%.2f" % r for r in reps),
                     "the restart product is the reference's 64-row-block algorithm (bvblas.c:74-106) on plain loops with OpenMP over the blocks, not the BLAS gemm a SLEPc build "
                     "calls (the same product as numpy / OpenBLAS dgemm over row blocks of the basis' storage took 2.7x as long on 8 cores when tried: the loops are the stronger baseline here)"))
        first = {"value": NCV / dt_first, "unit": "steps/s", "sample": "first Lanczos run, k=1..%d" % NCV}
    else:
        dt, steps, first = dt6, m1, None
        sample = "first %d Lanczos steps (k=1..%d) of the %d^3 workload (full run estimated %.0f s > budget)" % (m1, m1, n_side, est)
    out = {"value": steps / dt, "unit": "steps/s", "cores": threads, "kind": "port", "sample": sample, "method": "median of 3 repetitions (SURVEY 8d)",
           "seconds": round(dt, 3), "setup_seconds": round(t1 - t0, 2), "gs_passes": V.passes_total(), "first_cycle": first,
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


def class_table(prof, steps):
    classes = {}
    for (name, var), v in prof.items():
        c = classes.setdefault(name, {"launches": 0, "ms_total": 0.0, "alg": 0.0, "hbm": 0.0})
        c["launches"] += v["launches"]; c["ms_total"] += v["ms"]; c["alg"] += v["alg_bytes"]; c["hbm"] += v["hbm_bytes"]
    import slepc_amd as ks
    return [{"class": name, "event": ks.event_name(name), "launches": c["launches"], "ms_total": round(c["ms_total"], 3), "ms_per_step": round(c["ms_total"] / steps, 4),
             "alg_GBps": round(c["alg"] / c["ms_total"] / 1e6, 1) if c["ms_total"] > 0 else 0.0,
             "hbm_GBps": round(c["hbm"] / c["ms_total"] / 1e6, 1) if c["ms_total"] > 0 else 0.0}
            for name, c in sorted(classes.items(), key=lambda kv: -kv[1]["ms_total"])]


def update_kernel_roofline(ks, prof_timed):
    """The k_gs_update<KT,2> symbol with the largest total time IN THE TIMED REGION (HIP events on the library's stream).
    One symbol = fused launches + final launches + launches that exited at their device-side gate."""
    sym = {}
    for (name, var), v in prof_timed.items():
        if name in UPD_CLASSES and var > 0:
            e = sym.setdefault(var, {"ms_exec": 0.0, "n_exec": 0, "alg": 0.0, "hbm": 0.0, "ms_noop": 0.0, "n_noop": 0})
            if name == "gated_noop":
                e["ms_noop"] += v["ms"]; e["n_noop"] += v["launches"]
            else:
                e["ms_exec"] += v["ms"]; e["n_exec"] += v["launches"]; e["alg"] += v["alg_bytes"]; e["hbm"] += v["hbm_bytes"]
    sym = {k: d for k, d in sym.items() if d["n_exec"]}
    if not sym:
        return None
    kt, d = max(sym.items(), key=lambda kv: kv[1]["ms_exec"])
    achieved = d["hbm"] / d["ms_exec"] / 1e6          # bytes this kernel must move per launch / its time
    kname = "k_gs_update<%d, 2, false>" % kt
    traffic, tsrc = None, None
    if os.path.exists(TRAFFIC_FILE):
        try:
            tj = json.load(open(TRAFFIC_FILE))
            traffic = (tj.get(kname) or tj.get(kname.replace(", false>", ">")) or {}).get("hbm_bytes_per_executed_launch")
            tsrc = tj.get("_source")
        except Exception:      # noqa: BLE001
            traffic = None
    return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "traffic_kind": "lookup" if traffic is not None else None,      # not measured by this run: read from the committed PMC summary of the same command
            "traffic_source": tsrc if traffic is not None else None,
            "kernel": kname, "launches_executed": d["n_exec"], "launches_gated_off": d["n_noop"],
            "avg_launch_us_executed": round(1e3 * d["ms_exec"] / d["n_exec"], 2),
            "avg_launch_us_all_launches": round(1e3 * (d["ms_exec"] + d["ms_noop"]) / (d["n_exec"] + d["n_noop"]), 2),
            "bytes_per_executed_launch": d["hbm"] / d["n_exec"],
            "survey8d_bytes_per_executed_launch": d["alg"] / d["n_exec"],
            "survey8d_equivalent_GBps": round(d["alg"] / d["ms_exec"] / 1e6, 1),
            "note": "HIP events on the library stream over the timed region. achieved = bytes the kernel as designed must move "
                    "(bytes_per_executed_launch) / its average executed launch. One symbol is launched in two forms (DESIGN.md "
                    "section 4): the first CGS pass reads k basis columns and the vector, 8n(k+1) bytes, keeps its result in "
                    "registers and produces the k+1 dot products of the next pass (work the reference pays another 8n(k+1) "
                    "bytes for, SURVEY 8d: survey8d_*); the final pass reads the same and writes the vector, 8n(k+2). "
                    "bytes_per_executed_launch is the mean over both forms, as is the PMC figure in traffic (a separate rocprofv3 "
                    "--pmc run of this command, profiles/). rocprofv3's per-symbol average covers all launches incl. the ones "
                    "that exit at their device-side gate: compare avg_launch_us_all_launches."}


def live_traffic(kname, limit=90.0):
    """HBM bytes per executed launch of kernel symbol `kname`, measured NOW: two child runs of this same command under rocprofv3
    (--kernel-trace --pmc FETCH_SIZE, then WRITE_SIZE: separate passes, no other trace domain), the counters reduced as
    scripts/pmc_traffic.py does - KiB units, FETCH_SIZE x 2 on gfx950 (it counts half of a 128-byte request; calibration in
    profiles/r01b_pmc_calibration_and_mfma_summary.txt), launches that exit at their device-side gate (fetch < 1 MB) left out.
    Returns (bytes, note) or (None, reason); never raises."""
    import csv, glob, re, shutil, tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 is not on PATH"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this run is itself under a profiler"
    vals, child = {}, {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix="ks_pmc_", dir="/tmp")
            try:
                env = dict(os.environ); env["TMPDIR"] = "/tmp"
                # the interpreter itself directly behind "--" (no env / shell hop: the profiler's preloaded library has initialised the GPU by then)
                cmd = ["rocprofv3", "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
                       "--steps", "60", "--warmup", "20", "--min-steps", "60", "--no-cpu-baseline", "--no-configs", "--no-pmc"]
                try:
                    cp = subprocess.run(cmd, env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=limit)
                    child[ctr] = (cp.returncode, " | ".join(cp.stderr.decode(errors="replace").strip().splitlines()[-3:])[-400:])
                except subprocess.TimeoutExpired:
                    return None, "the %s pass was stopped after %.0f s" % (ctr, limit)
                got = []
                for path in glob.glob(d + "/*/*counter_collection.csv"):
                    for r in csv.DictReader(open(path)):
                        if r["Counter_Name"] != ctr:
                            continue
                        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "").replace("ksk::", "").split("(")[0]
                        if name == kname:
                            got.append(float(r["Counter_Value"]))
                vals[ctr] = got
            finally:
                shutil.rmtree(d, ignore_errors=True)
        f, w = vals.get("FETCH_SIZE", []), vals.get("WRITE_SIZE", [])
        ex = [v for v in f if v > 1024.0]
        if not ex:
            return None, "no executed launch of %s in the counter pass (child exit codes and last stderr lines: %r)" % (kname, child)
        exw = sorted(w)[len(w) - len(ex):] if len(w) >= len(ex) else w          # the executed launches are the ones that write
        fb = 2.0 * 1024.0 * sum(ex) / len(ex); wb = 1024.0 * (sum(exw) / len(exw) if exw else 0.0)
        return fb + wb, ("measured by this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate child passes of this command with --steps 60 "
                         "--warmup 20 --min-steps 60), (2*FETCH_SIZE + WRITE_SIZE)*1024 per executed launch, %d launches: fetched %.1f MB + written %.1f MB"
                         % (len(ex), fb / 1e6, wb / 1e6))
    except Exception as e:       # noqa: BLE001 - the counters must not take the headline down with them
        return None, "counter pass failed: %r" % (e,)


def measured_copy_ceiling(torch, gib=1.0, reps=10):
    """the spec peak confirmed on the box - a device-to-device copy of `gib` GiB (hipMemcpyDtoD through torch, timed
    with events on torch's stream, where the copy runs), read + write bytes over the time. Not a library kernel: the yardstick."""
    n = int(gib * (1 << 30)) // 8
    x = torch.ones(n, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
    for _ in range(3):
        y.copy_(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        y.copy_(x)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    del x, y
    return 2.0 * n * 8 / ms / 1e6


def measure(ks, ctx, A, B, barrier, warmup, steps, min_steps, nev, ncv, ptype, prof=True, tail=True, setup=None, seed=0x12345678, timed_events=True):
    """Run the phased solve on (A, B); returns (Phases, timed dict). timed_events=False: no HIP events inside the timed region (a pair per
    update launch costs a step of 100 us about 5 %); the per-kernel figures then come from the instrumented tail alone."""
    eps = ks.EPS(ctx)
    eps.SetOperators(A, B)
    eps.SetProblemType(ptype)
    eps.SetDimensions(nev, ncv)
    eps.SetTolerances(1e-8, 1 << 30)
    if setup:
        setup(eps)
    k_eff = max(steps, min_steps)
    ph = Phases(ks, ctx, eps, barrier, max(warmup, ncv + ncv // 2), k_eff, k_eff if (prof and tail) else 0, UPD_CLASSES[:2] if (prof and timed_events) else None)
    eps.SetStoppingTestFunction(ph)
    ph.run(seed)
    eps.SetStoppingTestFunction(None)
    return eps, ph, ph.timed(ncv)


def spmv_leg(ks, ctx, make_mat, reps=40):
    """Stand-alone MatMult timing of one device layout (untimed side leg): HIP events around `reps` launches."""
    import numpy as np
    A = make_mat()
    x = ks.BV(ctx, A.n, 2)
    x.SetRandomColumn(0)
    for _ in range(5):
        A.mult_dev(x.column_ptr(0), x.column_ptr(1))
    ctx.synchronize()
    ctx.prof_enable(True, classes=["spmv_csr"]); ctx.prof_reset()
    for _ in range(reps):
        A.mult_dev(x.column_ptr(0), x.column_ptr(1))
    ctx.synchronize()
    p = ctx.prof_get()
    ctx.prof_enable(False)
    v = p.get("spmv_csr", {"launches": 0, "ms": 0.0, "alg_bytes": 0.0, "hbm_bytes": 0.0})
    us = 1e3 * v["ms"] / max(1, v["launches"])
    out = {"layout": A.layout(), "avg_us": round(us, 2), "survey8d_GBps": round(v["alg_bytes"] / v["ms"] / 1e6, 1) if v["ms"] else None,
           "own_bytes_GBps": round(v["hbm_bytes"] / v["ms"] / 1e6, 1) if v["ms"] else None,
           "own_bytes_per_launch": v["hbm_bytes"] / max(1, v["launches"])}
    del x
    A.destroy()
    return out


def dropin_slot_leg(ks, ctx, A, ncv, cycles=3):
    """The SAME expansion driven the way SLEPc's own BVMatLanczos drives a BV type (the adapter's path, adapters/slepc/hipks.c):
    per step MatMult, then BVOrthonormalizeColumn = BV_CleanCoefficients, one ops->gramschmidt call PER PASS (host scalars back
    each time, the refinement test on the host), BV_SetValue, BVScaleColumn. Restart-cycle expansions k = ncv/2+1 .. ncv."""
    import numpy as np
    V = ks.BV(ctx, A.n, ncv + 1, N=A.N)
    V.SetRandomColumn(0)
    _, nrm, _ = V.OrthogonalizeColumn(0); V.ScaleColumn(0, 1.0 / nrm)
    T = np.zeros((ncv + 1, 3), order="F")
    V.MatLanczos(A, T, 0, ncv)                                  # a valid orthonormal basis to expand against (native path)
    ld = ncv + 1
    buf = V.buffer_ptr()
    k0 = ncv // 2
    ctx.synchronize()
    steps, passes = 0, 0
    state = 1                                                   # the BV's PetscObjectState, bumped where the reference bumps it
    s0 = ctx.sync_count()
    t0 = time.perf_counter()
    for _ in range(cycles):
        for j in range(k0, ncv):
            A.mult_dev(V.column_ptr(j), V.column_ptr(j + 1))    # BVMatMultColumn (MatMult into the column Vec)
            state += 1                                          #   BVRestoreColumn of a written Vec (bvbasic.c:1176)
            col = j + 1
            ctx.memset(buf + 8 * col * ld, 0, 8 * col)          # BV_CleanCoefficients_HIP: hipMemset on the stream (bvhip.hip.cpp:345-360)
            V.SetState(state)                                   # HipksSync at the head of the slot
            onrm, nrm = V.GramSchmidtPass(col); passes += 1
            l = 1
            while l < 3 and nrm != 0.0 and abs(nrm) < 0.7071 * abs(onrm):
                l += 1
                V.SetState(state)
                onrm, nrm = V.GramSchmidtPass(col); passes += 1
            ctx.memcpy_h2d(buf + 8 * (col * ld + col), np.array([nrm]))   # BV_SetValue_HIP: synchronous hipMemcpy of one scalar (bvhip.hip.cpp:397-413)
            state += 1                                          #   end of BVOrthogonalizeColumn (bvorthog.c:338)
            V.ScaleColumn(col, 1.0 / nrm)                       # BVScaleColumn
            state += 1                                          #   (bvops.c:356)
            steps += 1
    ctx.synchronize()
    dt = time.perf_counter() - t0
    ch = V.GsChainStats()
    return {"value": steps / dt, "unit": "steps/s", "steps": steps, "gs_passes_per_step": passes / steps,
            "passes_chained": ch["chained"], "passes_with_own_dot_sweep": ch["fresh"], "host_waits_per_step": (ctx.sync_count() - s0) / steps,
            "note": "one ops->gramschmidt call per pass through the C ABI, driven from Python here as the reference's BVMatLanczos / "
                    "BVOrthogonalizeGS would from C (clean coefficients, pass, refinement test on the host, pass, set value, scale), the "
                    "object state announced as adapters/slepc/hipks.c does. A pass is one dot sweep (first pass only) and one update "
                    "launch whose prologue hands onrm / nrm to the host; the second pass is chained to the dots the first one left: 3 "
                    "reads of V per CGS2 step, one extra 8n write and a separate scale against the library's own enqueued run (the headline)"}


def side_configs(ks, ctx, barrier, args):
    """The other single-GPU BASELINE configurations, measured with the same phased harness (N=1 only, after the headline)."""
    import numpy as np
    out = {}
    # C2: 2-D 5-pt Laplacian 1000^2, nev 4, m 20
    try:
        A = ks.Mat.laplacian2d(ctx, 1000)
        eps, ph, t = measure(ks, ctx, A, None, barrier, 60, 2000, 0, 4, 20, ks.EPS_HEP, timed_events=False)
        n = A.n
        rl = update_kernel_roofline(ks, ph.prof_tail)       # the instrumented pass behind the timed region: same cycles, events on every launch
        comp = sum(v["hbm_bytes"] for v in ph.prof_tail.values()); ms = sum(v["ms"] for v in ph.prof_tail.values())
        out["C2"] = {"workload": "2-D 5-pt Laplacian 1000^2 (n=%d), Krylov-Schur nev=4 m=20" % n, "value": t["steps"] / t["seconds"], "unit": "steps/s",
                     "steps": t["steps"], "us_per_step": 1e6 * t["seconds"] / t["steps"], "mean_k": round(t["mean_k"], 2), "cycles": t["cycles"],
                     "gs_passes_per_step": t["gs_passes"] / t["steps"], "spmv_layout": A.layout(),
                     "roofline": rl and {k: rl[k] for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "avg_launch_us_executed", "bytes_per_executed_launch")},
                     "basis_MB": round(21 * n * 8 / 1e6, 1),
                     "note": "the 168 MB basis fits the 256 MB Infinity Cache: rates above the HBM figure are possible and the step is partly latency-bound; "
                             "value: timed region without HIP events; roofline and kernel_classes: the instrumented pass of the same cycles behind it",
                     "kernel_classes": class_table(ph.prof_tail, max(1, ph.marks["t2"][0] - ph.marks["t1"][0])) if ph.prof_tail else None}
        del eps
        A.destroy()
    except Exception as e:      # noqa: BLE001
        out["C2"] = {"value": None, "error": repr(e)}
    # C5: random nonsymmetric CSR n = 5e6, ~33 nnz/row, generalized, shift-and-invert at target 0, nev 20, m 60
    if not args.no_c5:
        try:
            from slepc_amd.workloads import config5_pencil_arrays
            n5 = args.c5_n
            t0 = time.time()
            (ar, ac, av), (br, bc, bv) = config5_pencil_arrays(n5)
            A = ks.Mat.from_csr(ctx, ar, ac, av); B = ks.Mat.from_csr(ctx, br, bc, bv)
            nnz = int(ar[-1]); nnzb = int(br[-1])
            del ar, ac, av, br, bc, bv
            tgen = time.time() - t0
            stref = {}

            def setup(eps):
                eps.SetTarget(0.0)
                st = eps.GetST(); st.SetType("sinvert")
                stref["st"] = st
            eps, ph, t = measure(ks, ctx, A, B, barrier, 60, 150, 0, 20, 60, ks.EPS_GNHEP, setup=setup)
            kst = stref["st"].GetKSPStats()
            tail_steps = max(1, ph.marks["t2"][0] - ph.marks["t1"][0])
            tab = class_table(ph.prof_tail, tail_steps)
            sp = ph.prof_tail
            spmv_ms = sum(v["ms"] for (nm, _), v in sp.items() if nm == "spmv_csr"); spmv_n = sum(v["launches"] for (nm, _), v in sp.items() if nm == "spmv_csr")
            spmv_alg = sum(v["alg_bytes"] for (nm, _), v in sp.items() if nm == "spmv_csr")
            spmv_hbm = sum(v["hbm_bytes"] for (nm, _), v in sp.items() if nm == "spmv_csr")
            its_per_solve = kst["iterations"] / max(1, kst["solves"])
            out["C5"] = {"workload": "random nonsymmetric CSR n=%d nnz=%d (+ tridiagonal B, nnz=%d), GNHEP shift-and-invert target 0, nev=20 m=60, GMRES(30)+Jacobi inner solves (classical Gram-Schmidt without refinement, the KSP default)"
                                     % (n5, nnz, nnzb),
                         "generator_deviations_from_survey_8d": "slepc_amd/workloads.py: column indices drawn WITH replacement (duplicates in a row stay separate entries; SURVEY 8d says "
                                                                "without); 'diagonal += 40' is an extra leading entry (i, i) = 40 in every row, which MatMult and MatGetDiagonal sum with any random "
                                                                "entry on the diagonal; row lengths Poisson(32) clipped to [1,64] (+ that entry), values uniform(-1,1), seed 42 as in 8d",
                         "value": t["steps"] / t["seconds"], "unit": "steps/s", "steps": t["steps"], "ms_per_step": 1e3 * t["seconds"] / t["steps"],
                         "mean_k": round(t["mean_k"], 2), "cycles": t["cycles"], "inner_iterations_per_step": round(its_per_solve, 2),
                         "spmv_layout": A.layout(), "setup_seconds": round(tgen, 1),
                         "roofline": {"bound": "hbm", "kernel": ("k_binned_gather + k_binned_reduce" if A.layout() == "binned" else "k_spmv_sliced (+ k_sum_parts)")
                                      + " and the small SpMV of B, all MatMult launches of a step",
                                      "achieved": round(spmv_alg / spmv_ms / 1e6, 1) if spmv_ms else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": round(spmv_alg / spmv_ms / 1e6 / HBM_PEAK_GBS, 4) if spmv_ms else None,
                                      "own_bytes_GBps": round(spmv_hbm / spmv_ms / 1e6, 1) if spmv_ms else None,
                                      "own_bytes_frac": round(spmv_hbm / spmv_ms / 1e6 / HBM_PEAK_GBS, 4) if spmv_ms else None,
                                      "avg_launch_us": round(1e3 * spmv_ms / max(1, spmv_n), 1), "launches_per_step": round(spmv_n / tail_steps, 2),
                                      "pmc_bytes_per_product_of_A": {"fetched_GB": 0.40 + 3.25, "written_GB": 1.37 + 0.04, "kind": "lookup",
                                                                     "source": "profiles/r03_pmc_binned_spmv.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                                                                               "k_binned_gather + k_binned_reduce; FETCH_SIZE x2 on gfx950)",
                                                                     "ratio_to_survey8d_bytes": round((0.40 + 3.25 + 1.37 + 0.04) / 2.08, 2)},
                                      "structural_note": "2.4-2.5x the SURVEY 8d bytes by design: a (column slice, wave-bin) tile of a uniformly random matrix holds about 157 "
                                                         "entries, so no one-pass kernel can keep both its piece of x and its rows of y in LDS; the layout pays an 8 B per nonzero "
                                                         "round trip of the gathered x between two streaming phases (28 B per nonzero) for having no random access leave the CU "
                                                         "(row-ordered layouts pull a 128-byte line of x per nonzero: 1.3 ms instead of 0.89)",
                                      "bytes": "achieved / frac: SURVEY 8d's CSR bytes, 12 nnz + 4(n+1) + 16 n per MatMult. own_bytes_*: what the layout itself streams - "
                                               "the binned layout moves 28 bytes per nonzero in two passes (16-bit column -> gathered x out; gathered x, value and 16-bit "
                                               "row -> y) so that no random access leaves the CU (DESIGN.md section 6)"},
                         "kernel_classes": tab}
            del eps
            A.destroy(); B.destroy()
        except Exception as e:      # noqa: BLE001
            out["C5"] = {"value": None, "error": repr(e)}
    return out


BREAKDOWN_CLASSES = ["allreduce", "halo_exchange", "spmv_csr", "bv_dot_sweep", "gs_update_fused_dot", "gs_update", "gs_bookkeeping", "bv_multinplace", "bv_copy"]


def slab_problem(ks, ctx, side, world, rank, scaling, force_dist=False):
    """This rank's matrix and the workload string of one leg. weak: `world` z-slabs of side^3 rows each (432 x 432 x 54 N for side 216: config 4 at
    N = 8). strong: the side^3 grid of config 3 itself cut into `world` z-slabs the way PetscSplitOwnership cuts it (bvbasic.c:129-134) - the literal
    reading of the metric, "3D Laplacian n=10M ... 1/2/4/8 GPU"."""
    from slepc_amd import partition as P
    if scaling == "strong":
        z0, z1 = P.split_ownership(side, world)[rank]
        A = ks.Mat.laplacian3d(ctx, side, side, side, z0, z1 - z0) if (world > 1 or force_dist) else ks.Mat.laplacian3d(ctx, side, side, side)
        return A, ("3-D 7-pt Laplacian %d^3 (BASELINE config 3, n = %d) in %d z-slab(s) of %s planes, Krylov-Schur nev=%d m=%d"
                   % (side, side ** 3, world, "/".join(str(b - a) for a, b in P.split_ownership(side, world)), NEV, NCV))
    if world == 1 and not force_dist:
        return ks.Mat.laplacian3d(ctx, side, side, side), "3-D 7-pt Laplacian %d^3 (BASELINE config 3), Krylov-Schur nev=%d m=%d" % (side, NEV, NCV)
    nx = ny = 2 * side
    planes = side // 4                      # 54 planes of 432^2 = 216^3 rows per GPU
    nz = planes * world
    A = ks.Mat.laplacian3d(ctx, nx, ny, nz, rank * planes, planes)
    return A, "3-D 7-pt Laplacian %dx%dx%d in %d z-slabs of %d planes (BASELINE config 4 at N=8), Krylov-Schur nev=%d m=%d" % (nx, ny, nz, world, planes, NEV, NCV)


def rank_record(ph, t, tail_steps):
    """What every rank contributes to the N > 1 breakdown: its own timed seconds, the host time of its restart broadcasts inside the timed region, and its
    per-class kernel time over the instrumented tail (HIP events: an allreduce's time includes waiting for the slowest rank)."""
    ms = {c: 0.0 for c in BREAKDOWN_CLASSES}; n = {c: 0 for c in BREAKDOWN_CLASSES}
    for (name, _var), v in (ph.prof_tail or {}).items():
        if name in ms:
            ms[name] += v["ms"]; n[name] += v["launches"]
    return [t["seconds"], float(ph.bcast[0]), ph.bcast[1], float(tail_steps)] + [ms[c] for c in BREAKDOWN_CLASSES] + [float(n[c]) for c in BREAKDOWN_CLASSES]


def comm_breakdown(records, steps, leg):
    """The N > 1 line's per-phase table from one rank_record per rank (pure: tests/test_bench_host.py feeds it faked ranks). Times in microseconds per
    Arnoldi step; per-rank lists in rank order, so that a slow first number can be read: allreduce large on all ranks but one = that rank is late
    (skew), large on all = the transport; halo_exchange = pack -> exchange -> unpack on the halo stream, hidden under the diagonal-block product unless
    it exceeds it; restart_bcast = host wall time of the projected-problem broadcast, once per restart."""
    nc = len(BREAKDOWN_CLASSES)
    secs = [r[0] for r in records]
    per_rank = {}
    for i, c in enumerate(BREAKDOWN_CLASSES):
        per_rank[c] = [round(1e3 * r[4 + i] / max(1.0, r[3]), 2) for r in records]
    calls = {c: [r[4 + nc + i] / max(1.0, r[3]) for r in records] for i, c in enumerate(BREAKDOWN_CLASSES)}
    bc = [round(1e6 * r[2] / max(1, steps), 2) for r in records]
    mean = lambda xs: sum(xs) / max(1, len(xs))        # noqa: E731
    out = {"leg": leg, "ranks": len(records), "rank_timed_seconds": [round(s, 5) for s in secs], "rank_seconds_min": round(min(secs), 5),
           "rank_seconds_max": round(max(secs), 5), "rank_skew_pct": round(100.0 * (max(secs) - min(secs)) / max(secs), 3) if max(secs) > 0 else 0.0,
           "per_step_us": {"allreduce": round(mean(per_rank["allreduce"]), 2), "halo_exchange": round(mean(per_rank["halo_exchange"]), 2),
                           "restart_bcast": round(mean(bc), 2), "spmv": round(mean(per_rank["spmv_csr"]), 2), "dot_sweep": round(mean(per_rank["bv_dot_sweep"]), 2),
                           "update": round(mean(per_rank["gs_update_fused_dot"]) + mean(per_rank["gs_update"]), 2),
                           "bookkeeping": round(mean(per_rank["gs_bookkeeping"]), 2), "restart_gemm_copy": round(mean(per_rank["bv_multinplace"]) + mean(per_rank["bv_copy"]), 2)},
           "per_rank_us_per_step": {"allreduce": per_rank["allreduce"], "halo_exchange": per_rank["halo_exchange"], "restart_bcast": bc,
                                    "spmv": per_rank["spmv_csr"], "dot_sweep": per_rank["bv_dot_sweep"]},
           "allreduce_calls_per_step": round(mean(calls["allreduce"]), 3), "halo_exchanges_per_step": round(mean(calls["halo_exchange"]), 3),
           "restart_bcasts": int(records[0][1]), "restart_bcast_us_each": round(1e6 * records[0][2] / records[0][1], 1) if records[0][1] else None,
           "source": "per-class HIP events of the instrumented pass behind the timed region (same cycles), every rank's own; restart_bcast: host clock inside the "
                     "timed region; rank_timed_seconds: each rank's wall time between the two barriers of the timed region"}
    return out


def strong_entry(world, steps, dt, t, workload, mat, breakdown):
    """The strong-scaling leg of the line: the global solver's rate on the FIXED n = side^3 problem (total work fixed as N grows)."""
    return {"scaling": "strong", "value": steps / dt, "unit": "steps/s", "n_gpus": world, "steps": steps, "ms_per_step": 1e3 * dt / steps, "workload": workload,
            "rows_per_gpu": mat["n"], "n_global": mat["N"], "gs_passes_per_step": t["gs_passes"] / steps, "cycles": t["cycles"], "mean_k": round(t["mean_k"], 2),
            "multi_gpu_breakdown": breakdown,
            "note": "value = Arnoldi steps per second of the one global solver on the fixed-size problem; at N = 1 it is the headline's workload, so "
                    "value(N) / value(1) is the strong-scaling speed-up"}


def headline(world, steps, dt, warm_steps, t, args, workload, mat):
    """The contract part of the JSON line from the measured numbers (rank 0). `dt` is already the maximum over the ranks; weak scaling:
    the unit is one Arnoldi step on one GPU's 10 077 696-row shard (the N=1 workload), so the whole job processes world*steps of them while
    the global solver itself advances steps/dt steps per second. mat: n, nnz, N, layout of this rank's matrix."""
    return {
        "metric": "Arnoldi steps/sec (and GB/s vs HBM roofline), 3D Laplacian n=10M, m=30, 1/2/4/8 GPU",
        "value": world * steps / dt, "unit": "steps/s", "n_gpus": world, "steps": steps, "warmup": warm_steps,
        "ms_per_step": 1e3 * dt / steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload, "rows_per_gpu": mat["n"], "nnz_per_gpu": mat["nnz"], "n_global": mat["N"], "nev": NEV, "ncv": NCV,
                   "orthog": "CGS, refine ifneeded eta=0.7071", "gs_passes_per_step": t["gs_passes"] / steps,
                   "timed_region": "whole restart cycles of one continuing solve in steady state (after the first cycle and %d warm-up steps)" % warm_steps,
                   "steps_requested": args.steps, "min_steps": args.min_steps, "cycles": t["cycles"], "restarts": t["cycles"],
                   "steps_per_cycle": round(steps / t["cycles"], 2), "mean_k": round(t["mean_k"], 2), "timed_seconds": round(dt, 4),
                   "parallelism": "row-slab x%d" % world,
                   "spmv_layout": mat["layout"] + (" (2-byte entries: offset code + value code, lossless; y bit-identical to SELL-64; KSGPU_SPMV=sell disables; "
                                                   "general-matrix layouts: spmv_layouts)" if mat["layout"] == "dict" else ""),
                   "unit_of_value": "Arnoldi steps on a 10 077 696-row shard, summed over the %d shard(s): value = n_gpus * global_steps_per_s" % world,
                   "global_steps_per_s": steps / dt},
    }


def attach_side_legs(out, legs):
    """The N>1 side legs (measured by ONE child process of every rank, after the headline has been taken) into the line; never fatal.
    legs: {"strong_rccl": {...}, "oneshot_weak": {...}, "oneshot_strong": {...}, "child": {...}} - whatever the child got to before it ended."""
    if legs is None:
        return out
    child = legs.get("child") or {}
    st = legs.get("strong_rccl")
    out["strong_scaling"] = st if st is not None else {"scaling": "strong", "value": None, "reason": child.get("reason", "the side-leg child did not report this leg")}
    ow, os_ = legs.get("oneshot_weak"), legs.get("oneshot_strong")
    leg = dict(ow) if ow is not None else {"active": child.get("oneshot_active", "unknown"), "reason": child.get("oneshot_reason") or child.get("reason", "the side-leg child did not report this leg")}
    if leg.get("value"):
        leg["vs_provider_allreduce"] = round(leg["value"] / out["value"], 4)
    if os_ is not None:
        leg["strong_scaling"] = os_
        if os_.get("value") and st is not None and st.get("value"):
            leg["strong_vs_provider_allreduce"] = round(os_["value"] / st["value"], 4)
    leg["note"] = ("the same measurements in a child process of every rank with ks_comm_set_allreduce(ONESHOT) and ks_mat_set_halo(PEER): the "
                   "Gram-Schmidt sums go through peer-mapped mailboxes, one kernel per rank, instead of ncclAllReduce, and the boundary entries "
                   "of x go straight into the neighbours' ghost mailboxes instead of grouped ncclSend / ncclRecv; `value` above is the RCCL path")
    out["oneshot_allreduce"] = leg
    if child:
        out["side_legs_child"] = child
    return out


def spawn_ranks(args):
    """`python bench.py --gpus N` outside a launcher: start the N ranks as fresh child processes (this process has not
    touched the GPU) and hand their exit code back. Rank 0's JSON line goes straight to our stdout."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def parse_side_legs(stdout_bytes):
    """{"leg name": report} from what the side-leg child printed: one JSON object per finished leg, each with a "leg" key."""
    legs = {}
    for ln in (stdout_bytes or b"").decode(errors="replace").splitlines():
        if ln.startswith("{"):
            try:
                d = json.loads(ln)
            except ValueError:
                continue
            if isinstance(d, dict) and "leg" in d:
                legs[d.pop("leg")] = d
    return legs


def free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def side_legs_child(args, limit=240.0, port=None):
    """Everything of the N>1 line that is not the headline, in ONE child process per rank (same launcher environment, another rendezvous port):
    the strong-scaling leg with the RCCL provider, then the weak and the strong leg once more with the one-shot allreduce and the peer-mapped
    halo. The child prints one JSON line per finished leg; whatever happens to a later leg - mailboxes that cannot be mapped, a check that fails,
    a hang - stays in the children, which are stopped at `limit` seconds, and the legs finished before that are still reported. The headline
    has been taken before and is printed regardless."""
    env = dict(os.environ)
    # the children's rendezvous port: one the OS reported free on rank 0 and the ranks agreed on (port=), else derived from the parents' port
    env["MASTER_PORT"] = str(port) if port else str(20000 + (int(env.get("MASTER_PORT", "29511")) + 1789) % 20000)
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)      # under torchrun the ranks would look for the agent's store on the old port: rank 0's child opens its own
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--min-steps", str(args.min_steps), "--side", str(args.side), "--no-configs", "--no-cpu-baseline", "--side-legs"]
    if getattr(args, "no_oneshot", False):
        cmd.append("--no-oneshot")
    t0 = time.time()
    try:
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=None if env.get("RANK", "0") == "0" else subprocess.DEVNULL, timeout=limit)
        legs = parse_side_legs(p.stdout)
        legs.setdefault("child", {}).update({"exit_code": p.returncode, "seconds": round(time.time() - t0, 1)})
        if not any(k != "child" for k in legs):
            legs["child"]["reason"] = "the child left no report (exit code %d)" % p.returncode
        return legs
    except subprocess.TimeoutExpired as e:
        legs = parse_side_legs(e.stdout)
        legs.setdefault("child", {}).update({"reason": "the child was stopped after %.0f s" % limit, "seconds": round(time.time() - t0, 1)})
        return legs
    except Exception as e:       # noqa: BLE001 - the side legs must not take the headline down with them
        return {"child": {"reason": "%r" % (e,)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=60)
    ap.add_argument("--min-steps", type=int, default=200, help="lower bound of the timed region in steps (>= 200 ms on one GPU)")
    ap.add_argument("--side", type=int, default=216, help="grid side per GPU slab (216 -> 10 077 696 rows per GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="disable the per-kernel HIP-event timing")
    ap.add_argument("--no-configs", action="store_true", help="skip the side legs (configs C2 / C5, SpMV layouts)")
    ap.add_argument("--no-c5", action="store_true")
    ap.add_argument("--c5-n", type=int, default=5000000)
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc child passes that measure roofline.traffic (the committed summary is used instead)")
    ap.add_argument("--no-oneshot", action="store_true", help="N>1: skip the side legs that repeat the measurements with the one-shot allreduce")
    ap.add_argument("--no-side-legs", action="store_true", help="N>1: the weak headline only (no strong-scaling leg, no one-shot legs)")
    ap.add_argument("--side-legs", action="store_true", help=argparse.SUPPRESS)       # the side legs themselves (a child process of every rank)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args))

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
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    # BENCH_PROVIDER=gloo: REHEARSAL of the N>1 line with several ranks on ONE GPU (RCCL refuses two ranks on a device): torch.distributed over gloo and the
    # host-staged provider of slepc_amd/gloo_provider.py instead of RCCL. Everything above the transport is the real thing - slabs, halo plans, the split
    # bookkeeping, the records of every rank, the side-leg child with its one-shot allreduce and peer-mapped halo (hipIpc between the processes). Not a measurement.
    # Keep it to --gpus 2 on a one-GPU box: every rank and every rank's child hold the GPU open (a GPU box admits six processes on its card).
    rehearsal = os.environ.get("BENCH_PROVIDER") == "gloo"
    device = local_rank % max(1, torch.cuda.device_count()) if rehearsal else local_rank
    torch.cuda.set_device(device)
    dist = None
    force_dist = os.environ.get("BENCH_FORCE_DIST") == "1"     # rehearse the N>1 code path (RCCL comm, slab grid) on one GPU
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    ctx = ks.Context(device)
    if force_dist:
        ctx.set_debug("force_multi")        # ... with the collectives really issued (allreduce per pass, broadcast per restart)
    if (world > 1 or force_dist) and rehearsal:
        from slepc_amd import gloo_provider
        gloo_provider.install(ctx, dist, torch, rank, world)
        ctx.comm_check()
    elif world > 1 or force_dist:
        idt = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(ks.Context.get_unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, 0)
        ctx.init_rccl(rank, world, bytes(idt.cpu().numpy().tobytes()))
        ctx.comm_check()                  # allreduce / allgather / neighbour exchange against known answers before anything is timed
    if args.side_legs and dist is None:
        raise SystemExit("--side-legs is the N>1 child")
    side = args.side

    def barrier():
        ctx.synchronize()                 # the library's own stream
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter()

    def gather_records(rec):
        if dist is None:
            return [rec]
        tt = torch.tensor(rec, dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        parts = [torch.empty_like(tt) for _ in range(world)]
        dist.all_gather(parts, tt)
        return [p.cpu().tolist() for p in parts]

    def run_leg(scaling, prof, peer_halo=False):
        """One phased measurement on this rank's slab of the weak or the strong problem; every rank leaves with all ranks' records, so dt (the
        maximum over the ranks, as the contract asks) and the breakdown are the same everywhere."""
        A, workload = slab_problem(ks, ctx, side, world, rank, scaling, force_dist)
        halo = A.set_halo("peer") if peer_halo else None          # collective: boundary entries straight into the neighbours' ghost mailboxes
        eps, ph, t = measure(ks, ctx, A, None, barrier, args.warmup, args.steps, args.min_steps, NEV, NCV, ks.EPS_HEP, prof=prof)
        tail_steps = (ph.marks["t2"][0] - ph.marks["t1"][0]) if "t2" in ph.marks else 0
        recs = gather_records(rank_record(ph, t, tail_steps))
        return {"A": A, "workload": workload, "eps": eps, "ph": ph, "t": t, "dt": max(r[0] for r in recs), "recs": recs, "tail_steps": tail_steps, "halo": halo,
                "mat": {"n": A.n, "nnz": A.nnz, "N": A.N, "layout": A.layout()}}

    def close_leg(L):
        if L["halo"] == "peer":
            L["A"].set_halo("provider")          # collective: every rank finishes, unmaps the neighbours' mailboxes, then frees its own
        L["eps"] = None; L["ph"] = None          # the solver (and the callback that holds it) go before the matrix they refer to
        import gc
        gc.collect()
        L["A"].destroy()

    def report(name, d):
        if rank == 0:
            d = dict(d); d["leg"] = name
            os.dup2(real_stdout, 1); print(json.dumps(d), flush=True); os.dup2(2, 1)

    def weak_entry(L):
        st = L["t"]
        return {"value": world * st["steps"] / L["dt"], "unit": "steps/s", "ms_per_step": 1e3 * L["dt"] / st["steps"], "steps": st["steps"],
                "gs_passes_per_step": st["gs_passes"] / st["steps"], "multi_gpu_breakdown": comm_breakdown(L["recs"], st["steps"], "weak")}

    def strong_of(L):
        st = L["t"]
        return strong_entry(world, st["steps"], L["dt"], st, L["workload"], L["mat"], comm_breakdown(L["recs"], st["steps"], "strong"))

    if args.side_legs:
        # (1) the metric's literal reading beside the weak headline: the SAME 216^3 problem cut into N slabs (total work fixed), RCCL provider
        L = run_leg("strong", prof=True)
        report("strong_rccl", strong_of(L))
        close_leg(L)
        if not args.no_oneshot:
            # (2), (3) both legs again with the one-shot allreduce and the peer-mapped halo; every rank's verdict is the same at each exit
            # (set_allreduce and comm_check agree among the ranks)
            info = {"oneshot_active": ctx.set_allreduce("oneshot")}
            if info["oneshot_active"] == "oneshot":
                try:
                    ctx.comm_check()
                except RuntimeError as e:
                    info = {"oneshot_active": "provider", "oneshot_reason": "known-answer check failed with the one-shot path: %s" % e}
                    ctx.set_allreduce("provider")
            else:
                info["oneshot_reason"] = "some rank could not map the other ranks' mailboxes"
            report("child", info)
            if info["oneshot_active"] == "oneshot":
                for scaling in ("weak", "strong"):
                    L = run_leg(scaling, prof=True, peer_halo=True)
                    d = weak_entry(L) if scaling == "weak" else strong_of(L)
                    d.update({"active": "oneshot", "halo_active": L["halo"]})
                    report("oneshot_" + scaling, d)
                    close_leg(L)
        dist.barrier(); dist.destroy_process_group()
        return

    weak = run_leg("weak", prof=not args.no_prof)
    A, eps, ph, t, dt, workload = weak["A"], weak["eps"], weak["ph"], weak["t"], weak["dt"], weak["workload"]
    steps = t["steps"]
    prof_timed, prof = ph.prof_timed, ph.prof_tail
    tail_steps = weak["tail_steps"]

    legs = None
    if dist is not None and not args.no_side_legs:
        pt = torch.zeros(1, dtype=torch.int64, device="cpu" if rehearsal else "cuda")
        if rank == 0:
            pt[0] = free_port()
        dist.broadcast(pt, 0)             # rank 0 asks the OS for a free port, every rank's child meets there
        legs = side_legs_child(args, port=int(pt.item()))      # every rank starts its own child; rank 0's child reports, one line per finished leg
    if rank == 0:
        out = headline(world, steps, dt, ph.marks["t0"][0], t, args, workload, weak["mat"])
        if rehearsal:
            out["rehearsal"] = ("BENCH_PROVIDER=gloo: %d rank(s) sharing device %d through a host-staged gloo provider - the N>1 code paths exercised, "
                                "NOT a measurement of anything" % (world, device))
        try:
            from slepc_amd import _lib as _kslib
            ri = _kslib.runtime_info()
            out["hip_runtime"] = {"path": ri["hip_runtime_path"], "version": ri["hip_runtime_version"], "runtimes_mapped": len(ri["hip_runtimes_mapped"]),
                                  "occupancy_query_failures": ri["occupancy_query_failures"],
                                  "note": "the HIP runtime libksgpu.so is bound to in this process (torch's bundled copy when torch is installed: slepc_amd/_lib.py bind_hip_runtime)"}
        except Exception as e:       # noqa: BLE001
            out["hip_runtime"] = {"error": repr(e)}
        if dist is not None:
            out["multi_gpu_breakdown"] = comm_breakdown(weak["recs"], steps, "weak")
        if prof_timed:
            rl = update_kernel_roofline(ks, prof_timed)
            if rl:
                try:
                    cp = measured_copy_ceiling(torch)
                    rl["measured_copy_GBps"] = round(cp, 1)
                    rl["frac_of_measured_copy"] = round(rl["achieved"] / cp, 4)
                    rl["measured_note"] = ("device-to-device copy of 1 GiB on this box (read + write bytes / time); a pure 31-column read by a "
                                           "stand-alone kernel reaches 6.8-7.0 TB/s (profiles/r02_micro_update_write*.txt)")
                except Exception as e:       # noqa: BLE001
                    rl["measured_copy_GBps"] = None; rl["measured_note"] = "copy probe failed: %r" % (e,)
                if world == 1 and not force_dist and not args.no_pmc:
                    tb, tnote = live_traffic(rl["kernel"])
                    if tb is not None:
                        rl["traffic_lookup"] = rl["traffic"]
                        rl["traffic"] = tb; rl["traffic_kind"] = "measured"; rl["traffic_source"] = tnote
                        rl["traffic_over_bytes_per_executed_launch"] = round(tb / rl["bytes_per_executed_launch"], 4)
                    else:
                        rl["traffic_live_note"] = tnote
                out["roofline"] = rl
        if prof:
            kernels = []
            for (name, var), v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
                kernels.append({"class": name, "event": ks.event_name(name), "variant": var, "kernel": ks.kernel_symbol(name, var), "launches": v["launches"], "ms_total": round(v["ms"], 3),
                                "avg_us": round(1e3 * v["ms"] / v["launches"], 2),
                                "alg_GBps": round(v["alg_bytes"] / v["ms"] / 1e6, 1) if v["ms"] > 0 else 0.0,
                                "hbm_GBps": round(v["hbm_bytes"] / v["ms"] / 1e6, 1) if v["ms"] > 0 else 0.0})
            tot_alg = sum(v["alg_bytes"] for v in prof.values()); tot_hbm = sum(v["hbm_bytes"] for v in prof.values())
            tot_ms = sum(v["ms"] for v in prof.values())
            mip_ms = sum(v["ms"] for (nm, _), v in prof.items() if nm in ("bv_multinplace", "bv_copy"))
            # the tail repeats the timed region's work (same cycle structure) with events on every launch: scale its byte
            # counts to the timed region's step count and divide by the timed wall time
            sc = steps / max(1, tail_steps)
            out["step_traffic"] = {"alg_GB_per_step": round(tot_alg / tail_steps / 1e9, 3), "alg_GBps_vs_wall": round(tot_alg * sc / dt / 1e9, 1),
                                   "frac_of_hbm_peak_alg": round(tot_alg * sc / dt / 1e9 / HBM_PEAK_GBS, 4),
                                   "compulsory_GB_per_step": round(tot_hbm / tail_steps / 1e9, 3), "compulsory_GBps_vs_wall": round(tot_hbm * sc / dt / 1e9, 1),
                                   "frac_of_hbm_peak_compulsory": round(tot_hbm * sc / dt / 1e9 / HBM_PEAK_GBS, 4),
                                   "kernel_ms_per_step": round(tot_ms / tail_steps, 4)}
            out["config"]["restart_share_of_kernel_time"] = round(mip_ms / tot_ms, 4) if tot_ms else None
            out["kernel_classes_untimed_instrumented_pass"] = class_table(prof, tail_steps)
            out["kernels_untimed_instrumented_pass"] = kernels[:12]
        del eps
        if world == 1 and not force_dist and not args.no_configs:
            try:
                legs = {}
                for fmt in ("dict", "odict", "sell", "csr", "csrregs", "csrvec"):
                    os.environ["KSGPU_SPMV"] = fmt
                    try:
                        legs[fmt] = spmv_leg(ks, ctx, lambda: ks.Mat.laplacian3d(ctx, side, side, side))
                    finally:
                        os.environ.pop("KSGPU_SPMV", None)
                out["spmv_layouts"] = {"workload": "MatMult of the %d^3 7-pt Laplacian alone, each device layout (KSGPU_SPMV=...)" % side, "legs": legs,
                                       "note": "dict needs <= 255 distinct values and <= 256 distinct column offsets, odict only the offsets; "
                                               "sell = SELL-64 for any stencil-like matrix; csr = CSR row blocks streamed through wave-private LDS (a wave per 64 rows), for ragged ones - for short rows by LDS-DMA "
                                               "(global_load_lds_dwordx4, round 4), csrregs = its register-staged form; csrvec = the CSR-vector kernel it replaced"}
            except Exception as e:      # noqa: BLE001
                out["spmv_layouts"] = {"error": repr(e)}
            # the same measurement with the general-matrix SpMV (CSR row blocks; nothing Laplacian-specific in the layout): untimed for the headline
            try:
                os.environ["KSGPU_SPMV"] = "csr"
                try:
                    Ag = ks.Mat.laplacian3d(ctx, side, side, side)
                finally:
                    os.environ.pop("KSGPU_SPMV", None)
                eg, phg, tg = measure(ks, ctx, Ag, None, barrier, args.warmup, args.steps, args.min_steps, NEV, NCV, ks.EPS_HEP, prof=False)
                out["value_general_layout"] = {"value": tg["steps"] / tg["seconds"], "unit": "steps/s", "ms_per_step": 1e3 * tg["seconds"] / tg["steps"], "steps": tg["steps"],
                                               "spmv_layout": Ag.layout(), "spmv_avg_us": out.get("spmv_layouts", {}).get("legs", {}).get("csr", {}).get("avg_us"),
                                               "note": "the headline's solve with KSGPU_SPMV=csr: MatMult through the CSR row-block kernel (12 B per nonzero, any "
                                                       "matrix) instead of the 2-byte dictionary layout the assembly picks for this stencil; everything else identical"}
                del eg
                Ag.destroy()
            except Exception as e:      # noqa: BLE001
                out["value_general_layout"] = {"value": None, "error": repr(e)}
            try:
                Ad = ks.Mat.laplacian3d(ctx, side, side, side)
                out["dropin_slot_path"] = dropin_slot_leg(ks, ctx, Ad, NCV)
                Ad.destroy()
            except Exception as e:      # noqa: BLE001
                out["dropin_slot_path"] = {"value": None, "error": repr(e)}
            out["configs"] = side_configs(ks, ctx, barrier, args)
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(side)
            except Exception as e:       # noqa: BLE001 - the baseline must not take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "steps/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}
        attach_side_legs(out, legs)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
