"""Host logic of bench.py without a GPU: the phase machine that places warm-up / timed / tail boundaries on restart
boundaries of one continuing solve, its cycle accounting (mean k, steps per cycle), and the N > 1 self-launch command."""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench      # noqa: E402


class FakeBV:
    def __init__(self, eps): self.eps = eps
    def gs_passes(self): return (2 * self.eps.total_steps, 2)


class FakeCtx:
    def __init__(self): self.enabled = []; self.resets = 0
    def prof_enable(self, on, classes=None): self.enabled.append((on, tuple(classes) if classes else None))
    def prof_reset(self): self.resets += 1
    def prof_get(self, by_variant=False): return {("gs_update", 30): {"launches": 1, "ms": 1.0, "alg_bytes": 1.0, "hbm_bytes": 1.0}}


class FakeEPS:
    """A solver whose first cycle has ncv steps and every later one ncv/2, calling the stopping test after each."""
    def __init__(self, ncv): self.ncv = ncv; self.steps = 0; self.total_steps = 0; self.cb = None; self.solves = 0
    def SetRandomSeed(self, s): pass
    def GetStats(self): return {"arnoldi_steps": self.steps}
    def GetBV(self): return FakeBV(self)
    def StoppingBasic(self, its, max_it, nconv, nev): return 0
    def Solve(self):
        self.solves += 1; self.steps = 0; its = 0
        while True:
            its += 1
            L = self.ncv if its == 1 else self.ncv // 2
            self.steps += L; self.total_steps += L
            if self.cb(its, 10 ** 9, 0, 10):
                return


def test_phases_cover_whole_cycles_and_do_not_depend_on_the_requested_steps():
    ks = types.SimpleNamespace(EPS_CONVERGED_USER=2)
    clock = [0.0]

    def barrier():
        clock[0] += 1.0
        return clock[0]
    for want in (20, 300):
        eps = FakeEPS(30); ctx = FakeCtx()
        k_eff = max(want, 200)
        ph = bench.Phases(ks, ctx, eps, barrier, max(5, 45), k_eff, k_eff, bench.UPD_CLASSES[:2])
        eps.cb = ph
        ph.run(1)
        t = ph.timed(30)
        # warm-up = first cycle (30) + one restart cycle (15) = 45 steps, then whole 15-step cycles
        assert ph.marks["t0"][0] == 45
        assert t["steps"] % 15 == 0 and t["steps"] >= k_eff and t["steps"] - k_eff < 15
        assert t["cycles"] == t["steps"] // 15 and all(L == 15 for L in t["cycle_steps"])
        assert abs(t["mean_k"] - 23.0) < 1e-12                  # k = 16..30 in every timed cycle
        assert t["gs_passes"] == 2 * t["steps"]
        assert ph.marks["t2"][0] - ph.marks["t1"][0] >= k_eff    # the instrumented tail repeats the region
        assert ctx.enabled[0] == (True, ("gs_update_fused_dot", "gs_update")) and ctx.enabled[-1] == (False, None)
        assert eps.solves == 1                                   # one continuing solve


def test_self_launch_command(monkeypatch):
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"] = cmd; seen["env"] = env
        return 0
    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20"])
    args = types.SimpleNamespace(gpus=4)
    assert bench.spawn_ranks(args) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "20"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def _fake_timed(steps=210, cycles=14):
    return {"steps": steps, "seconds": 0.25, "gs_passes": 2 * steps, "cycles": cycles, "cycle_steps": [15] * cycles, "mean_k": 23.0}


def test_n_gt_1_line_assembles_from_a_faked_two_rank_result():
    """What rank 0 prints at N = 2, built from numbers as the two ranks would have produced them: the contract keys, whole-job value =
    n_gpus x global steps per second, weak scaling - and it survives json.dumps."""
    import json
    args = types.SimpleNamespace(steps=20, min_steps=200, warmup=5)
    t = _fake_timed()
    dt = max(0.2500, 0.2531)                         # the maximum over the ranks
    mat = {"n": 10077696, "nnz": 70263936 - 2 * 186624 // 2, "N": 2 * 10077696, "layout": "dict"}
    out = bench.headline(2, t["steps"], dt, 45, t, args, "3-D 7-pt Laplacian 432x432x108 in 2 z-slabs of 54 planes", mat)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in out, key
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["vs_baseline"] is None and out["dtype"] == "f64"
    assert abs(out["value"] - 2 * 210 / dt) < 1e-9 and abs(out["ms_per_step"] - 1e3 * dt / 210) < 1e-12
    assert out["config"]["global_steps_per_s"] == 210 / dt and out["config"]["parallelism"] == "row-slab x2"
    assert out["config"]["steps_requested"] == 20 and out["config"]["min_steps"] == 200
    assert out["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    # the side legs: a child that reported every leg, one that was stopped after the first, none at all
    strong = {"scaling": "strong", "value": 3000.0, "unit": "steps/s", "n_gpus": 2}
    ow = {"active": "oneshot", "halo_active": "peer", "value": 1.01 * out["value"], "unit": "steps/s", "ms_per_step": 0.9, "steps": 210, "gs_passes_per_step": 2.0}
    os_ = {"scaling": "strong", "value": 3300.0, "active": "oneshot", "halo_active": "peer"}
    legs = {"strong_rccl": strong, "oneshot_weak": ow, "oneshot_strong": os_, "child": {"oneshot_active": "oneshot", "exit_code": 0, "seconds": 61.2}}
    line = json.loads(json.dumps(bench.attach_side_legs(dict(out), legs)))
    assert line["strong_scaling"]["value"] == 3000.0 and line["scaling"] == "weak"
    assert line["oneshot_allreduce"]["vs_provider_allreduce"] == 1.01 and "note" in line["oneshot_allreduce"]
    assert line["oneshot_allreduce"]["strong_scaling"]["value"] == 3300.0 and line["oneshot_allreduce"]["strong_vs_provider_allreduce"] == 1.1
    assert line["side_legs_child"]["exit_code"] == 0
    stopped = {"strong_rccl": strong, "child": {"oneshot_active": "oneshot", "reason": "the child was stopped after 240 s", "seconds": 240.0}}
    line = json.loads(json.dumps(bench.attach_side_legs(dict(out), stopped)))
    assert line["strong_scaling"]["value"] == 3000.0                        # the leg finished before the stop is kept
    assert "vs_provider_allreduce" not in line["oneshot_allreduce"] and "stopped after" in line["oneshot_allreduce"]["reason"]
    line = json.loads(json.dumps(bench.attach_side_legs(dict(out), {"child": {"reason": "the child left no report (exit code 3)"}})))
    assert line["strong_scaling"]["value"] is None and "no report" in line["strong_scaling"]["reason"]
    assert "oneshot_allreduce" not in bench.attach_side_legs(dict(out), None) and "strong_scaling" not in bench.attach_side_legs(dict(out), None)


def test_side_legs_child_command_reports_and_time_limit(monkeypatch):
    """The child of the N>1 run: its command line (same sizes, no single-GPU side legs, --side-legs), another rendezvous port, one JSON line per
    finished leg parsed from its stdout; a child that prints nothing, or is stopped at the limit after some legs, becomes a reason string next to
    the legs it did finish - never an exception."""
    import json
    import subprocess
    seen = {}
    l1 = json.dumps({"leg": "strong_rccl", "scaling": "strong", "value": 3000.0}).encode()
    l2 = json.dumps({"leg": "child", "oneshot_active": "oneshot"}).encode()
    l3 = json.dumps({"leg": "oneshot_weak", "active": "oneshot", "halo_active": "peer", "value": 7000.0}).encode()

    def fake_run(cmd, env=None, stdout=None, stderr=None, timeout=None):
        seen.update(cmd=cmd, env=env, timeout=timeout)
        return types.SimpleNamespace(returncode=0, stdout=b"RCCL banner\n" + l1 + b"\n" + l2 + b"\n{not json\n" + l3 + b"\n")
    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setenv("MASTER_PORT", "29511"); monkeypatch.setenv("RANK", "0")
    args = types.SimpleNamespace(gpus=8, steps=20, warmup=5, min_steps=200, side=216, no_oneshot=False)
    legs = bench.side_legs_child(args)
    assert legs["strong_rccl"] == {"scaling": "strong", "value": 3000.0} and legs["oneshot_weak"]["value"] == 7000.0 and "oneshot_strong" not in legs
    assert legs["child"]["oneshot_active"] == "oneshot" and legs["child"]["exit_code"] == 0
    cmd = seen["cmd"]
    assert cmd[cmd.index("--gpus") + 1] == "8" and "--side-legs" in cmd and "--no-configs" in cmd and "--no-cpu-baseline" in cmd and "--no-oneshot" not in cmd
    assert seen["env"]["MASTER_PORT"] != "29511" and seen["timeout"] == 240.0
    bench.side_legs_child(args, port=31234)                     # the port rank 0 asked the OS for and broadcast to the others
    assert seen["env"]["MASTER_PORT"] == "31234"

    def silent(cmd, env=None, stdout=None, stderr=None, timeout=None):
        return types.SimpleNamespace(returncode=3, stdout=b"")
    monkeypatch.setattr(bench.subprocess, "run", silent)
    assert "no report" in bench.side_legs_child(args)["child"]["reason"]

    def hung(cmd, env=None, stdout=None, stderr=None, timeout=None):
        raise subprocess.TimeoutExpired(cmd, timeout, output=l1 + b"\n")
    monkeypatch.setattr(bench.subprocess, "run", hung)
    legs = bench.side_legs_child(args)
    assert "stopped after" in legs["child"]["reason"] and legs["strong_rccl"]["value"] == 3000.0


def test_breakdown_and_strong_leg_of_the_n_gt_1_line_from_faked_ranks():
    """The per-phase table of the N > 1 line from one record per rank (what rank_record produces: timed seconds, restart broadcasts, per-class
    HIP-event time of the instrumented pass), and the strong-scaling entry beside the weak headline. Rank 1 is the slow one here: it shows as the
    maximum of the timed seconds and as the SMALLEST allreduce time (the others wait for it)."""
    import json
    nc = len(bench.BREAKDOWN_CLASSES)

    def rec(seconds, bcasts, bcast_s, tail_steps, ms, calls):
        return [seconds, float(bcasts), bcast_s, float(tail_steps)] + [ms.get(c, 0.0) for c in bench.BREAKDOWN_CLASSES] + [float(calls.get(c, 0)) for c in bench.BREAKDOWN_CLASSES]
    calls = {"allreduce": 420, "halo_exchange": 210, "spmv_csr": 210, "bv_dot_sweep": 210}
    r0 = rec(0.2500, 14, 14 * 60e-6, 210, {"allreduce": 8.4, "halo_exchange": 6.3, "spmv_csr": 14.7, "bv_dot_sweep": 66.0, "gs_update": 72.0, "gs_update_fused_dot": 62.0}, calls)
    r1 = rec(0.2531, 14, 14 * 55e-6, 210, {"allreduce": 2.1, "halo_exchange": 6.1, "spmv_csr": 14.9, "bv_dot_sweep": 67.0, "gs_update": 73.0, "gs_update_fused_dot": 63.0}, calls)
    assert len(r0) == 4 + 2 * nc
    b = bench.comm_breakdown([r0, r1], 210, "weak")
    assert b["ranks"] == 2 and b["leg"] == "weak" and b["rank_seconds_max"] == 0.2531 and b["rank_seconds_min"] == 0.25
    assert abs(b["rank_skew_pct"] - 100 * 0.0031 / 0.2531) < 1e-3
    assert b["per_rank_us_per_step"]["allreduce"] == [40.0, 10.0] and b["per_step_us"]["allreduce"] == 25.0
    assert b["per_rank_us_per_step"]["halo_exchange"] == [30.0, 29.05] and b["allreduce_calls_per_step"] == 2.0 and b["halo_exchanges_per_step"] == 1.0
    assert b["restart_bcasts"] == 14 and b["restart_bcast_us_each"] == 60.0
    assert b["per_rank_us_per_step"]["restart_bcast"] == [4.0, 3.67] and abs(b["per_step_us"]["restart_bcast"] - 3.835) < 0.006
    assert abs(b["per_step_us"]["update"] - (1e3 * (72.0 + 73.0) / 2 / 210 + 1e3 * (62.0 + 63.0) / 2 / 210)) < 0.02
    t = _fake_timed()
    s = bench.strong_entry(2, 210, 0.14, t, "3-D 7-pt Laplacian 216^3 in 2 z-slab(s) of 108/108 planes", {"n": 5038848, "nnz": 0, "N": 10077696, "layout": "dict"}, b)
    assert s["scaling"] == "strong" and abs(s["value"] - 1500.0) < 1e-9 and s["n_gpus"] == 2 and s["n_global"] == 10077696 and s["multi_gpu_breakdown"]["ranks"] == 2
    json.loads(json.dumps({"multi_gpu_breakdown": b, "strong_scaling": s}))


def test_slab_problems_of_the_weak_and_the_strong_leg():
    """Which slab of which grid a rank assembles: weak = N slabs of 216^3 rows each (432 x 432 x 54 N), strong = the 216^3 grid itself cut as
    PetscSplitOwnership cuts it (uneven when N does not divide the planes)."""
    made = []

    class FakeMat:
        @staticmethod
        def laplacian3d(ctx, nx, ny, nz, z0=0, nz_local=None):
            made.append((nx, ny, nz, z0, nz_local)); return "A"
    ks = types.SimpleNamespace(Mat=FakeMat)
    _, w = bench.slab_problem(ks, None, 216, 8, 3, "weak")
    assert made[-1] == (432, 432, 432, 3 * 54, 54) and "8 z-slabs of 54 planes" in w
    _, w = bench.slab_problem(ks, None, 216, 8, 7, "strong")
    assert made[-1] == (216, 216, 216, 7 * 27, 27) and "n = 10077696" in w
    _, w = bench.slab_problem(ks, None, 216, 5, 0, "strong")                   # 216 = 5 * 43 + 1: the first rank takes the extra plane
    assert made[-1] == (216, 216, 216, 0, 44) and "44/43/43/43/43" in w
    _, w = bench.slab_problem(ks, None, 216, 1, 0, "weak")
    assert made[-1] == (216, 216, 216, 0, None)
    _, w = bench.slab_problem(ks, None, 216, 1, 0, "weak", force_dist=True)    # the N > 1 code path rehearsed on one rank
    assert made[-1] == (432, 432, 54, 0, 54)
