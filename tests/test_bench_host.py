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
