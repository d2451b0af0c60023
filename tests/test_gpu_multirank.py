"""The multi-rank code path of libksgpu on ONE GPU.

* two processes share cuda:0; the communicator operations (ks_comm_set_ops: allreduce, allgather, neighbour
  exchange) are served by torch.distributed/gloo through host staging -> exercises the row-slab Mat (diag/off-diag
  split, halo plan, ghost indexing), the split reduce | allreduce | bookkeeping Gram-Schmidt kernels and the
  replicated host control flow of the Krylov-Schur driver, against the single-rank CPU oracle;
* one process with the NATIVE RCCL provider at size 1 and the force_multi test hook -> exercises ncclCommInitRank /
  ncclAllReduce on the library stream in the same split-kernel path.
(8-GPU runs over xGMI are the driver's job; this is what one GPU can prove.)"""
import ctypes
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))




def _collect(q, procs, n, timeout=400):
    """n results from the workers' queue; whatever happens, no worker outlives the test (a rank left waiting for a peer that failed would keep the
    interpreter from exiting, and the GPU run behind it from starting)."""
    import queue
    import time
    t0, out = time.time(), []
    try:
        while len(out) < n:
            try:
                out.append(q.get(timeout=2))
            except queue.Empty:
                dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
                if dead:             # a rank that died before reporting: fail now, not after the full wait (the peers are reaped below)
                    raise RuntimeError("a worker exited with code %s before reporting (%d of %d results in)" % (dead, len(out), n))
                if time.time() - t0 > timeout:
                    raise
        return out
    finally:
        for p in procs:
            p.join(20 if p.is_alive() else 0)
        for p in procs:
            if p.is_alive():
                p.terminate(); p.join(10)
            if p.is_alive():
                p.kill(); p.join(5)


def _free_port():
    """A rendezvous port the OS says is free right now (ports computed from the pid collided with the ephemeral ports of earlier tests' gloo pairs:
    EADDRINUSE on the box, round 4)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _install_gloo_ops(ks, ctx, dist, torch, rank, size, perturb=False):
    from slepc_amd import gloo_provider
    gloo_provider.install(ctx, dist, torch, rank, size, perturb=perturb)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import slepc_amd as ks
        from slepc_amd import partition as P
        from oracle import oracle as O
        ctx = ks.Context(0)
        _install_gloo_ops(ks, ctx, dist, torch, rank, world)
        ctx.comm_check()
        nx, ny, nz = 12, 10, 9
        plane = nx * ny
        z0, z1 = P.split_ownership(nz, world)[rank]
        r0, r1 = z0 * plane, z1 * plane
        Aglob = O.laplacian3d(nx, ny, nz)
        x = np.random.default_rng(1).standard_normal(Aglob.n)
        yref = Aglob.mult(x)
        res = {}
        # (1) both constructors: device generator and host CSR block with global column indices
        rp, col, val = P.local_block(Aglob.rowptr, Aglob.col, Aglob.val, r0, r1)
        mats = {"gen": ks.Mat.laplacian3d(ctx, nx, ny, nz, z0, z1 - z0),
                "csr": ks.Mat.from_csr(ctx, rp, col, val, row_start=r0, n_global=Aglob.n)}
        X = ks.BV(ctx, r1 - r0, 2, N=Aglob.n)
        X.set_column(0, x[r0:r1])
        for name, A in mats.items():
            A.mult_dev(X.column_ptr(0), X.column_ptr(1))
            res["spmv_" + name] = float(np.abs(X.column(1) - yref[r0:r1]).max())
        # (2) sharded Lanczos
        m = 12
        A = mats["gen"]
        V = ks.BV(ctx, r1 - r0, m + 1, N=Aglob.n, row_start=r0)
        V.SetRandomColumn(0)
        _, nrm, _ = V.OrthogonalizeColumn(0); V.ScaleColumn(0, 1.0 / nrm)
        T = np.zeros((m + 1, 3), order="F")
        mm, beta, brk = V.MatLanczos(A, T, 0, m)
        res["T"] = T[:m, :2].copy(); res["beta"] = beta; res["mm"] = mm; res["passes"] = V.gs_passes()[0]
        M = np.zeros((m + 1, m + 1), order="F"); V.SetActiveColumns(0, m + 1); V.Dot(V, M)
        res["orth"] = float(np.abs(M - np.eye(m + 1)).max())
        res["normF"] = V.Norm(ks.NORM_FROBENIUS)
        res["normInf"] = V.Norm(ks.NORM_INFINITY); res["norm1"] = V.Norm(ks.NORM_1)
        V.ScaleColumn(3, 1e200); res["big"] = V.NormColumn(3); V.ScaleColumn(3, 1e-200)      # overflow-safe 2-norm across ranks
        # split reductions: two Begins, ONE allreduce at the first End, results as the plain calls
        V.SetActiveColumns(0, m)
        a = V.DotVecBegin(V.column_ptr(m)); V.NormColumnBegin(1)
        d = V.DotVecEnd(V.column_ptr(m), a); nrm1 = V.NormColumnEnd(1)
        res["split"] = float(max(np.abs(d - V.DotVec(V.column_ptr(m))).max(), abs(nrm1 - V.NormColumn(1))))
        V.SetActiveColumns(0, m + 1)
        # (2b) the ops->gramschmidt slot across ranks, with pass chaining (ks_bv_set_state): BVOrthogonalizeGS's loop on the caller's side, every pass
        # one dot sweep (first pass only) + reduce / allreduce + update launch; the chained pass reduces the partial sums the update kernel left
        Xg = np.random.default_rng(9).standard_normal((Aglob.n, 6))
        Xg[:, 3] = Xg[:, 0] + 2.0 ** -10 * Xg[:, 3]; Xg[:, 5] = Xg[:, 1] - 2.0 * Xg[:, 2] + 2.0 ** -12 * Xg[:, 5]
        W = ks.BV(ctx, r1 - r0, 6, N=Aglob.n, row_start=r0)
        W.set_dense(Xg[r0:r1])
        wbuf = W.buffer_ptr(); state = 1; slot = []
        for j in range(6):
            if j > 0:
                ctx.memset(wbuf + 8 * j * 6, 0, 8 * j)
            W.SetState(state)
            onrm, nrm = W.GramSchmidtPass(j); passes = 1
            while passes < 3 and nrm != 0.0 and abs(nrm) < 0.7071 * abs(onrm):
                passes += 1
                W.SetState(state)
                onrm, nrm = W.GramSchmidtPass(j)
            ctx.memcpy_h2d(wbuf + 8 * (j * 6 + j), np.array([nrm])); state += 1
            W.ScaleColumn(j, 1.0 / nrm); state += 1
            slot.append((nrm, passes))
        res["slot"] = slot; res["slot_H"] = W.buffer(); res["slot_chain"] = W.GsChainStats()
        M6 = np.zeros((6, 6), order="F"); W.SetActiveColumns(0, 6); W.Dot(W, M6)
        res["slot_orth"] = float(np.abs(M6 - np.eye(6)).max())
        # (3) full Krylov-Schur solve, replicated control flow
        eps = ks.EPS(ctx); eps.SetOperators(A); eps.SetProblemType(ks.EPS_HEP); eps.SetDimensions(3, 12); eps.Solve()
        res["eig"] = [eps.GetEigenvalue(i)[0] for i in range(3)]
        res["its"] = eps.GetIterationNumber(); res["nconv"] = eps.GetConverged()
        res["err"] = [eps.ComputeError(i) for i in range(3)]
        # (4) a deflation space (row-sharded constraint vectors) and a basis wider than the fused kernels (host-driven passes)
        Cg = np.random.default_rng(5).standard_normal((Aglob.n, 2))
        e2 = ks.EPS(ctx); e2.SetOperators(A); e2.SetProblemType(ks.EPS_HEP); e2.SetDimensions(3, 12); e2.SetDeflationSpace(Cg[r0:r1]); e2.Solve()
        res["defl_eig"] = [e2.GetEigenvalue(i)[0] for i in range(3)]; res["defl_its"] = e2.GetIterationNumber()
        res["defl_cross"] = float(max(abs(np.dot(Cg[r0:r1, 0], e2.GetEigenvector(i))) for i in range(3)))     # local part only: summed by the parent
        res["defl_cross_vec"] = [[float(np.dot(Cg[r0:r1, c], e2.GetEigenvector(i))) for c in range(2)] for i in range(3)]
        e3 = ks.EPS(ctx); e3.SetOperators(A); e3.SetProblemType(ks.EPS_HEP); e3.SetDimensions(20, 70); e3.Solve()
        res["wide_eig"] = [e3.GetEigenvalue(i)[0] for i in range(20)]; res["wide_its"] = e3.GetIterationNumber(); res["wide_nconv"] = e3.GetConverged()
        # (5) generalized non-symmetric shift-and-invert (config-5 shape) across ranks: both matrices sharded by rows, the inner
        # GMRES + Jacobi solves and the halo exchanges of A and B on the same communicator
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import nhep_cases as nc
        Ag, Bg = nc.config5_pencil(900)
        q0, q1 = P.split_ownership(Ag.n, world)[rank]
        Al = ks.Mat.from_csr(ctx, *P.local_block(Ag.rowptr, Ag.col, Ag.val, q0, q1), row_start=q0, n_global=Ag.n)
        Bl = ks.Mat.from_csr(ctx, *P.local_block(Bg.rowptr, Bg.col, Bg.val, q0, q1), row_start=q0, n_global=Bg.n)
        e4 = ks.EPS(ctx); e4.SetOperators(Al, Bl); e4.SetProblemType(ks.EPS_GNHEP); e4.SetDimensions(4, 20); e4.SetTarget(36.0)
        s4 = e4.GetST(); s4.SetType("sinvert"); s4.SetKSP(rtol=1e-13)
        e4.Solve()
        res["c5_eig"] = [list(e4.GetEigenvalue(i)) for i in range(4)]; res["c5_its"] = e4.GetIterationNumber()
        res["c5_err"] = [e4.ComputeError(i) for i in range(4)]
        # (5b) the same solve with the matrix of the inner solves assembled (ST_MATMODE_COPY): MatAXPY on every rank's row block (global
        # columns), the halo plan of the sum built like any other matrix's
        Ak = ks.Mat.from_csr(ctx, *P.local_block(Ag.rowptr, Ag.col, Ag.val, q0, q1), row_start=q0, n_global=Ag.n, keep_csr=True)
        Bk = ks.Mat.from_csr(ctx, *P.local_block(Bg.rowptr, Bg.col, Bg.val, q0, q1), row_start=q0, n_global=Bg.n, keep_csr=True)
        Pk = Ak.axpy_new(-36.0, Bk)
        xg = np.random.default_rng(12).standard_normal(Ag.n)
        Xp = ks.BV(ctx, q1 - q0, 2, N=Ag.n); Xp.set_column(0, xg[q0:q1])
        Pk.mult_dev(Xp.column_ptr(0), Xp.column_ptr(1))
        yp = (Ag.to_scipy() - 36.0 * Bg.to_scipy()) @ xg
        res["copy_axpy"] = float(np.abs(Xp.column(1) - yp[q0:q1]).max() / np.abs(yp).max())
        e4c = ks.EPS(ctx); e4c.SetOperators(Ak, Bk); e4c.SetProblemType(ks.EPS_GNHEP); e4c.SetDimensions(4, 20); e4c.SetTarget(36.0)
        s4c = e4c.GetST(); s4c.SetType("sinvert"); s4c.SetMatMode("copy"); s4c.SetKSP(rtol=1e-13)
        e4c.Solve()
        res["c5copy_eig"] = [list(e4c.GetEigenvalue(i)) for i in range(4)]; res["c5copy_its"] = e4c.GetIterationNumber()
        # (5c) block Jacobi on the row-sharded P: every rank inverts the diagonal blocks of ITS rows (global columns minus its row offset)
        s5 = ks.ST(ctx); s5.SetType("sinvert"); s5.SetShift(36.0); s5.SetMatrices(Ak, Bk); s5.SetKSP(rtol=1e-13); s5.SetPC("bjacobi", 5)
        Xq = ks.BV(ctx, q1 - q0, 2, N=Ag.n); Xq.set_column(0, xg[q0:q1])
        import ctypes
        ks._lib.check(ctx.L.ks_st_apply(s5.h, ctypes.c_void_p(Xq.column_ptr(0)), ctypes.c_void_p(Xq.column_ptr(1))))
        yq = O.ST(Ag, Bg, "sinvert", 36.0).apply(xg)
        res["bj_err"] = float(np.abs(Xq.column(1) - yq[q0:q1]).max() / np.abs(yq).max()); res["bj_its"] = s5.GetKSPStats()["iterations"]
        # (6) test39.c: one solver, two solves with matrices whose LOCAL sizes differ (the 10x11 2-D Laplacian with one row moved
        # from rank 1 to rank 0, then the other way); EPSSetOperators drops what was sized by the first matrix
        L2 = O.laplacian2d(10, 11)
        e5 = ks.EPS(ctx); e5.SetProblemType(ks.EPS_HEP); e5.SetWhichEigenpairs("smallest_real"); e5.SetDimensions(3)
        res["t39"] = []
        for shift in (1, -1):
            cnt = [b - a for a, b in P.split_ownership(L2.n, world)]
            cnt[0] += shift; cnt[1] -= shift
            s0 = sum(cnt[:rank]); s1 = s0 + cnt[rank]
            Ml = ks.Mat.from_csr(ctx, *P.local_block(L2.rowptr, L2.col, L2.val, s0, s1), row_start=s0, n_global=L2.n)
            e5.SetOperators(Ml); e5.Solve()
            res["t39"].append(([e5.GetEigenvalue(i)[0] for i in range(3)], e5.GetIterationNumber(), e5.GetConverged(),
                               max(e5.ComputeError(i) for i in range(3)), len(e5.GetEigenvector(0)), s1 - s0))
        # (7) an allreduce that differs by one ulp on rank 1: with the projected solve synchronised from rank 0 (the default,
        # DSSynchronize krylovschur.c:281) every rank still takes the same decisions and reports the same bits
        ctx2 = ks.Context(0)
        _install_gloo_ops(ks, ctx2, dist, torch, rank, world, perturb=True)
        Ap = ks.Mat.laplacian3d(ctx2, nx, ny, nz, z0, z1 - z0)
        e6 = ks.EPS(ctx2); e6.SetOperators(Ap); e6.SetProblemType(ks.EPS_HEP); e6.SetDimensions(3, 12); e6.Solve()
        res["pert_eig"] = [e6.GetEigenvalue(i)[0] for i in range(3)]; res["pert_its"] = e6.GetIterationNumber(); res["pert_nconv"] = e6.GetConverged()
        res["pert_err"] = [e6.ComputeError(i) for i in range(3)]
        e7 = ks.EPS(ctx2); e7.SetOperators(Ap); e7.SetProblemType(ks.EPS_NHEP); e7.SetDimensions(3, 12); e7.Solve()
        res["pert_nhep"] = ([list(e7.GetEigenvalue(i)) for i in range(3)], e7.GetIterationNumber(), e7.GetConverged())
        del e6, e7, Ap
        dist.barrier()
        q.put((rank, res))
    except Exception as e:      # noqa: BLE001
        import traceback
        q.put((rank, {"error": traceback.format_exc()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4])
def test_ranks_sharing_one_gpu_against_oracle(world):
    """world = 2: every rank has one slab neighbour; world = 4: the inner ranks exchange with two (9 planes split 3,2,2,2)."""
    import torch.multiprocessing as mp
    from oracle import oracle as O
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    out = dict(_collect(q, procs, world))
    for r in range(world):
        assert "error" not in out[r], out[r].get("error")
    # single-rank oracle reference
    nx, ny, nz = 12, 10, 9
    A = O.laplacian3d(nx, ny, nz); m = 12
    V = O.BV(A.n, m + 1); V.SetRandomColumn(0)
    _, nrm, _ = V.OrthogonalizeColumn(0); V.ScaleColumn(0, 1 / nrm)
    p0 = V.passes_total()
    T = np.zeros((m + 1, 3), order="F")
    mm, beta, brk = V.MatLanczos(A, T, 0, m)
    r = O.eps_krylovschur_hep(A, 3, ncv=12)
    for rk in range(world):
        o = out[rk]
        assert o["spmv_gen"] < 1e-13 and o["spmv_csr"] < 1e-13
        assert o["mm"] == mm and abs(o["beta"] - beta) < 1e-12
        assert np.abs(o["T"] - T[:m, :2]).max() < 1e-12
        assert o["passes"] - 1 == V.passes_total() - p0      # the start vector cost one pass on each side
        assert o["orth"] < 1e-13
        assert abs(o["normF"] - np.sqrt(m + 1)) < 1e-12       # global Frobenius norm of an orthonormal basis
        assert o["split"] < 1e-15
        assert abs(o["normInf"] - V.Norm(O.NORM_INFINITY)) < 1e-11 and abs(o["norm1"] - V.Norm(O.NORM_1)) < 1e-11
        assert abs(o["big"] / 1e200 - 1.0) < 1e-13
        assert o["nconv"] == r.nconv and o["its"] == r.its
        assert np.allclose(o["eig"], r.eigr[r.perm][:3], rtol=1e-10)
        assert max(o["err"]) < 1e-8
    # the slot across ranks: identical pass counts and (replicated) scalars on every rank, coefficients as the single-rank oracle's, every
    # pass after a column's first chained to the dots its predecessor left
    Xg = np.random.default_rng(9).standard_normal((A.n, 6))
    Xg[:, 3] = Xg[:, 0] + 2.0 ** -10 * Xg[:, 3]; Xg[:, 5] = Xg[:, 1] - 2.0 * Xg[:, 2] + 2.0 ** -12 * Xg[:, 5]
    Wo = O.BV(A.n, 6)
    for j in range(6):
        Wo.set_column(j, Xg[:, j])
    oref = []
    for j in range(6):
        _, nrm, lin = Wo.OrthogonalizeColumn(j); oref.append((nrm, Wo.passes_last())); Wo.ScaleColumn(j, 1.0 / nrm)
    Bo = np.array(Wo.buffer)
    for rk in range(world):
        o = out[rk]
        assert [p for _, p in o["slot"]] == [p for _, p in oref], (o["slot"], oref)
        assert np.allclose([x for x, _ in o["slot"]], [x for x, _ in oref], rtol=1e-7)
        for j in range(1, 6):
            assert np.allclose(o["slot_H"][:j, j], Bo[:j, j], rtol=1e-7, atol=1e-9), j
        assert o["slot_orth"] < 1e-12
        total = sum(p for _, p in o["slot"][1:])
        assert o["slot_chain"] == {"chained": total - 5, "fresh": 5} and total - 5 >= 2
        assert o["slot"] == out[0]["slot"]
    Cg = np.random.default_rng(5).standard_normal((A.n, 2))
    rd = O.eps_krylovschur_hep(A, 3, ncv=12, deflation=Cg)
    rw = O.eps_krylovschur_hep(A, 20, ncv=70)
    for rk in range(world):
        o = out[rk]
        assert o["defl_its"] == rd.its and np.allclose(o["defl_eig"], rd.eigr[rd.perm][:3], rtol=1e-10)
        assert o["wide_its"] == rw.its and o["wide_nconv"] == rw.nconv and np.allclose(o["wide_eig"], rw.eigr[rw.perm][:20], rtol=1e-9)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import nhep_cases as nc
    Ag, Bg = nc.config5_pencil(900)
    r5 = O.eps_krylovschur_nhep(Ag, 4, ncv=20, which=O.which_target_magnitude(36.0), st=O.ST(Ag, Bg, "sinvert", 36.0))
    ref5 = np.array([[r5.eigr[j], r5.eigi[j]] for j in r5.perm[:4]])
    for rk in range(world):
        assert out[rk]["c5_its"] == r5.its and np.allclose(np.array(out[rk]["c5_eig"]), ref5, rtol=1e-8, atol=1e-9)
        assert max(out[rk]["c5_err"]) < 1e-6
        assert out[rk]["bj_err"] < 1e-10 and 0 < out[rk]["bj_its"] < 100                        # (5c)
        assert out[rk]["copy_axpy"] < 1e-14                                                     # (5b) P = A - 36 B assembled per row block
        assert out[rk]["c5copy_its"] == r5.its and np.allclose(np.array(out[rk]["c5copy_eig"]), ref5, rtol=1e-8, atol=1e-9)
    import golden_inputs as gi
    r39 = O.eps_krylovschur_hep(O.laplacian2d(10, 11), 3, which="smallest_real")
    g39 = gi.eigenvalue_lines(gi.read("eps/eps_test39_1.out"))                  # first and second solve
    assert len(g39) == 2
    for rk in range(world):
        first, second = out[rk]["t39"]
        for (lam, its, nconv, err, nvec, nloc), gold in zip((first, second), g39):
            assert its == r39.its and nconv == r39.nconv and err < 1e-8 and nvec == nloc
            assert np.allclose(lam, r39.eigr[r39.perm][:3], rtol=1e-10)
            assert np.allclose(np.round(lam, 5), gold, atol=1.5e-5)
        assert first[5] - second[5] == (2 if rk == 0 else -2 if rk == 1 else 0)
    cross = np.sum([np.array(out[rk]["defl_cross_vec"]) for rk in range(world)], axis=0)     # global C' x from the ranks' parts
    assert np.abs(cross).max() < 1e-10
    for rk in range(1, world):
        assert out[0]["eig"] == out[rk]["eig"]               # replicated scalars are bitwise identical on all ranks
        # the perturbed provider: same decisions and the same bits everywhere, because rank 0's projected solve is broadcast
        assert out[0]["pert_eig"] == out[rk]["pert_eig"] and out[0]["pert_its"] == out[rk]["pert_its"] and out[0]["pert_nconv"] == out[rk]["pert_nconv"]
        assert out[0]["pert_nhep"] == out[rk]["pert_nhep"]
    assert out[0]["pert_nconv"] >= 3 and np.allclose(out[0]["pert_eig"], r.eigr[r.perm][:3], rtol=1e-10) and max(out[0]["pert_err"]) < 1e-8


def _rccl_worker(q):
    sys.path.insert(0, ROOT)
    try:
        import slepc_amd as ks
        from oracle import oracle as O
        ctx = ks.Context(0)
        ctx.set_debug("force_multi")           # test hook: the collectives really issued on a communicator of one rank
        ctx.init_rccl(0, 1, ks.Context.get_unique_id())
        ctx.comm_check()                  # ncclAllReduce, ncclAllGather (staging buffer reused by a second call), ncclSend/ncclRecv with itself
        ctx.comm_check()
        Ao = O.laplacian2d(40)
        A = ks.Mat.from_csr(ctx, Ao.rowptr, Ao.col, Ao.val)
        eps = ks.EPS(ctx); eps.SetOperators(A); eps.SetProblemType(ks.EPS_HEP); eps.SetDimensions(4, 20); eps.Solve()
        r = O.eps_krylovschur_hep(Ao, 4, ncv=20)
        ok = (eps.GetConverged() == r.nconv and eps.GetIterationNumber() == r.its and
              np.allclose([eps.GetEigenvalue(i)[0] for i in range(4)], r.eigr[r.perm][:4], rtol=1e-10))
        ctx.prof_enable(True); ctx.prof_reset()
        ctx.bcast_stats(reset=True)
        eps.Solve()
        n_allreduce = ctx.prof_get().get("allreduce", {}).get("launches", 0)
        nb, sb = ctx.bcast_stats()           # one ncclBroadcast of the projected problem per restart (DSSynchronize), counted with its host time
        q.put({"ok": bool(ok), "allreduces": n_allreduce, "steps": eps.GetStats()["arnoldi_steps"], "bcasts": nb, "bcast_seconds": sb, "its": eps.GetIterationNumber()})
    except Exception:      # noqa: BLE001
        import traceback
        q.put({"error": traceback.format_exc()})


@pytest.mark.timeout(600)
def test_native_rccl_provider_single_rank_forced_collectives():
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    p = mpc.Process(target=_rccl_worker, args=(q,))
    p.start()
    out = _collect(q, [p], 1)[0]
    assert "error" not in out, out.get("error")
    assert out["ok"]
    assert out["allreduces"] >= 2 * out["steps"]          # one ncclAllReduce per executed Gram-Schmidt pass went through RCCL
    assert out["bcasts"] == out["its"] and out["bcast_seconds"] > 0.0      # and one broadcast of the projected problem per restart (ks_comm_bcast_stats)


def _oneshot_worker(rank, world, port, q, absent_rank):
    """absent_rank >= 0: that rank skips one allreduce - the others must come back with an error, not hang."""
    sys.path.insert(0, ROOT)
    if absent_rank >= 0:
        os.environ["KSGPU_ONESHOT_TIMEOUT_MS"] = "300"
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import slepc_amd as ks
        from slepc_amd import partition as P
        ctx = ks.Context(0)
        _install_gloo_ops(ks, ctx, dist, torch, rank, world)
        res = {"active": ctx.set_allreduce("oneshot")}
        if absent_rank >= 0:
            import time
            X = ks.BV(ctx, 64, 12, N=64 * world)
            d = X.buffer_ptr()
            ctx.memcpy_h2d(d, np.arange(1.0, 9.0) * (rank + 1))
            dist.barrier()
            if rank == absent_rank:
                time.sleep(1.5)
            t0 = time.perf_counter()
            try:
                for _ in range(64):                   # 64 enqueued calls must not wait 64 times
                    ctx.allreduce_sum_dev(d, 8)
                ctx.synchronize()
                res["raised"] = False
            except RuntimeError as e:
                res["raised"] = str(e)
            res["seconds"] = time.perf_counter() - t0
            out = np.empty(8); ctx.memcpy_d2h(out, d)
            res["nan"] = bool(np.isnan(out).all())
            res["back"] = ctx.set_allreduce("provider")
            ctx.comm_check()                      # the provider is unharmed
            dist.barrier()
            q.put((rank, res))
            return
        ctx.comm_check()
        # the values every rank ends up with are the same bits: sums of rank-dependent irrational numbers, 1..128 long (128 = the
        # longest one-shot message; longer ones go through the provider)
        X = ks.BV(ctx, 64, 12, N=64 * world)             # its coefficient buffer (12 x 12 doubles) is the device scratch
        bits = []
        for count in (1, 2, 31, 61, 128, 129):
            h = np.sqrt(np.arange(1, count + 1) * (rank + 2.0)) * (-1.0) ** np.arange(count)
            d = X.buffer_ptr()
            ctx.memcpy_h2d(d, h)
            ctx.allreduce_sum_dev(d, count)
            out = np.empty(count); ctx.memcpy_d2h(out, d)
            want = sum(np.sqrt(np.arange(1, count + 1) * (r + 2.0)) * (-1.0) ** np.arange(count) for r in range(world))
            assert np.allclose(out, want, rtol=1e-14, atol=0), (count, out, want)
            bits.append(out.tobytes())
        res["bits"] = bits
        nx, ny, nz = 12, 10, 9
        z0, z1 = P.split_ownership(nz, world)[rank]
        A = ks.Mat.laplacian3d(ctx, nx, ny, nz, z0, z1 - z0)
        m = 12
        V = ks.BV(ctx, A.n, m + 1, N=A.N, row_start=z0 * nx * ny)
        V.SetRandomColumn(0)
        _, nrm, _ = V.OrthogonalizeColumn(0); V.ScaleColumn(0, 1.0 / nrm)
        T = np.zeros((m + 1, 3), order="F")
        mm, beta, brk = V.MatLanczos(A, T, 0, m)
        res["T"] = T[:m, :2].copy(); res["beta"] = beta; res["mm"] = mm; res["passes"] = V.gs_passes()[0]
        M = np.zeros((m + 1, m + 1), order="F"); V.SetActiveColumns(0, m + 1); V.Dot(V, M)
        res["orth"] = float(np.abs(M - np.eye(m + 1)).max())
        eps = ks.EPS(ctx); eps.SetOperators(A); eps.SetProblemType(ks.EPS_HEP); eps.SetDimensions(3, 12); eps.Solve()
        res["eig"] = [eps.GetEigenvalue(i)[0] for i in range(3)]
        res["its"] = eps.GetIterationNumber(); res["nconv"] = eps.GetConverged()
        res["err"] = [eps.ComputeError(i) for i in range(3)]
        # a basis wider than the fused kernels: the chunked dots reduce 65..71 coefficients per pass, still one-shot (<= 128)
        e3 = ks.EPS(ctx); e3.SetOperators(A); e3.SetProblemType(ks.EPS_HEP); e3.SetDimensions(20, 70); e3.Solve()
        res["wide_eig"] = [e3.GetEigenvalue(i)[0] for i in range(20)]; res["wide_its"] = e3.GetIterationNumber(); res["wide_nconv"] = e3.GetConverged()
        del e3
        res["back"] = ctx.set_allreduce("provider")
        ctx.comm_check()
        dist.barrier()
        q.put((rank, res))
    except Exception:      # noqa: BLE001
        import traceback
        q.put((rank, {"error": traceback.format_exc()}))
    finally:
        dist.destroy_process_group()


def _run_oneshot(world, absent_rank=-1):
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_oneshot_worker, args=(r, world, port, q, absent_rank)) for r in range(world)]
    for p in procs: p.start()
    out = dict(_collect(q, procs, world))
    for r in range(world):
        assert "error" not in out[r], out[r].get("error")
    return out


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4])
def test_oneshot_allreduce_between_processes_sharing_one_gpu(world):
    """SURVEY 8e's one-shot allreduce: every process maps the others' mailboxes through hipIpc (the route the ranks of an 8-GPU node
    take over xGMI; here the mailboxes sit on one card) and one kernel per rank does the sum. Same Lanczos coefficients, pass
    counts and eigenvalues as the single-rank oracle, identical bits on every rank."""
    from oracle import oracle as O
    out = _run_oneshot(world)
    A = O.laplacian3d(12, 10, 9); m = 12
    V = O.BV(A.n, m + 1); V.SetRandomColumn(0)
    _, nrm, _ = V.OrthogonalizeColumn(0); V.ScaleColumn(0, 1 / nrm)
    p0 = V.passes_total()
    T = np.zeros((m + 1, 3), order="F")
    mm, beta, brk = V.MatLanczos(A, T, 0, m)
    r = O.eps_krylovschur_hep(A, 3, ncv=12)
    for rk in range(world):
        o = out[rk]
        assert o["active"] == "oneshot" and o["back"] == "provider"
        assert o["mm"] == mm and abs(o["beta"] - beta) < 1e-12 and np.abs(o["T"] - T[:m, :2]).max() < 1e-12
        assert o["passes"] - 1 == V.passes_total() - p0
        assert o["orth"] < 1e-13
        assert o["its"] == r.its and o["nconv"] == r.nconv and max(o["err"]) < 1e-8
        assert np.allclose(o["eig"], r.eigr[r.perm][:3], rtol=1e-10)
        assert o["bits"] == out[0]["bits"] and o["eig"] == out[0]["eig"] and o["beta"] == out[0]["beta"]
        assert o["wide_eig"] == out[0]["wide_eig"] and o["wide_its"] == out[0]["wide_its"] and o["wide_nconv"] >= 20
    rw = O.eps_krylovschur_hep(A, 20, ncv=70)
    assert out[0]["wide_its"] == rw.its and np.allclose(out[0]["wide_eig"], rw.eigr[rw.perm][:20], rtol=1e-10)


@pytest.mark.timeout(600)
def test_oneshot_allreduce_gives_up_when_a_rank_never_arrives():
    """Rank 1 arrives 1.5 s late at 64 back-to-back allreduces, the time limit is 0.3 s: rank 0 gives up (its 64 enqueued calls do not
    wait 64 times), rank 1 then finds rank 0's packets overwritten and gives up as well; both get NaN and KS_ERR_LIB at their next host
    wait, and go back to the provider's allreduce, which still works."""
    out = _run_oneshot(2, absent_rank=1)
    for rk in (0, 1):
        assert out[rk]["active"] == "oneshot" and out[rk]["back"] == "provider"
        assert out[rk]["raised"] and "one-shot" in out[rk]["raised"], out[rk]
        assert out[rk]["nan"]
    assert out[0]["seconds"] < 6.0


# ---- peer-mapped halo exchange (ks_mat_set_halo) and the one-shot stamps across their 32-bit wrap ---------------------------------------
def _peer_halo_worker(rank, world, port, q, mode):
    """mode "halo": products through the neighbours' ghost mailboxes against the provider's exchange and the oracle;
    mode "absent": rank 1 arrives late at a product - the others give up within the time limit, nobody hangs;
    mode "wrap": the one-shot allreduce with its stamps started just below 2^32."""
    sys.path.insert(0, ROOT)
    if mode == "absent":
        os.environ["KSGPU_ONESHOT_TIMEOUT_MS"] = "300"
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import slepc_amd as ks
        from slepc_amd import partition as P
        from oracle import oracle as O
        ctx = ks.Context(0)
        _install_gloo_ops(ks, ctx, dist, torch, rank, world)
        res = {}
        if mode == "wrap":
            ctx.set_debug("oneshot_seq0", 0xFFFFFFE0)           # test hook: first stamp of the one-shot allreduce
            res["active"] = ctx.set_allreduce("oneshot")
            ctx.comm_check()                                   # 64 back-to-back calls: the stamps run through 0xFFFFFFFF -> 1
            X = ks.BV(ctx, 64, 12, N=64 * world)
            d = X.buffer_ptr()
            for it in range(40):
                h = np.sqrt(np.arange(1, 32) * (rank + 2.0 + it))
                ctx.memcpy_h2d(d, h); ctx.allreduce_sum_dev(d, 31)
                out = np.empty(31); ctx.memcpy_d2h(out, d)
                want = sum(np.sqrt(np.arange(1, 32) * (r + 2.0 + it)) for r in range(world))
                assert np.allclose(out, want, rtol=1e-14, atol=0), (it, out, want)
            ctx.comm_check()
            res["back"] = ctx.set_allreduce("provider")
            dist.barrier(); q.put((rank, res)); return
        nx, ny, nz = 12, 10, 9
        plane = nx * ny
        z0, z1 = P.split_ownership(nz, world)[rank]
        r0, r1 = z0 * plane, z1 * plane
        Aglob = O.laplacian3d(nx, ny, nz)
        rng = np.random.default_rng(1)
        A = ks.Mat.laplacian3d(ctx, nx, ny, nz, z0, z1 - z0)
        X = ks.BV(ctx, r1 - r0, 3, N=Aglob.n)
        xs = [rng.standard_normal(Aglob.n) for _ in range(6)]
        # the provider's exchange first (reference bits), then the same products through the mailboxes
        prov = []
        for x in xs:
            X.set_column(0, x[r0:r1]); A.mult_dev(X.column_ptr(0), X.column_ptr(1)); prov.append(X.column(1))
        res["active"] = A.set_halo("peer")
        if mode == "absent":
            import time
            if rank == 1:
                time.sleep(1.5)
            t0 = time.perf_counter()
            X.set_column(0, xs[0][r0:r1])
            try:
                A.mult_dev(X.column_ptr(0), X.column_ptr(1)); ctx.synchronize(); res["raised"] = False
            except RuntimeError as e:
                res["raised"] = str(e)
            res["seconds"] = time.perf_counter() - t0
            dist.barrier(); q.put((rank, res)); return
        peer = []
        for x in xs:
            X.set_column(0, x[r0:r1]); A.mult_dev(X.column_ptr(0), X.column_ptr(1)); peer.append(X.column(1))
        res["bits_equal"] = all(np.array_equal(a, b) for a, b in zip(prov, peer))
        res["spmv_err"] = float(max(np.abs(p - Aglob.mult(x)[r0:r1]).max() for p, x in zip(peer, xs)))
        # many products back to back, no reduction in between: y_{k+1} = A y_k / 8 alternating between two columns (the slots' parities
        # and acknowledgements are all that keeps a fast rank from overwriting what a slow one has not read yet)
        X.set_column(0, xs[0][r0:r1]); X.ScaleColumn(0, 1.0)
        yo = xs[0].copy()
        for k in range(60):
            a, b = (0, 1) if k % 2 == 0 else (1, 0)
            A.mult_dev(X.column_ptr(a), X.column_ptr(b)); X.ScaleColumn(b, 0.125)
            yo = Aglob.mult(yo) * 0.125
        res["chain_err"] = float(np.abs(X.column(0) - yo[r0:r1]).max() / np.abs(yo).max())
        # a general CSR matrix with a ONE-WAY pattern: rank r needs entries of rank r+1 only (upper bidiagonal blocks), so a sender gets no
        # data back from its destination - only acknowledgements
        n = 64 * world
        rows = np.arange(n)
        colsU = np.minimum(rows + 40, n - 1)
        rp = np.arange(0, 2 * n + 1, 2, dtype=np.int32)
        col = np.stack([rows, colsU], axis=1).astype(np.int32).ravel()         # rows <= colsU: already in column order
        val = np.stack([2.0 + rows * 0.01, -1.0 + rows * 0.001], axis=1).ravel()
        U = O.CSR(n, rp, col, val)
        q0, q1 = rank * 64, (rank + 1) * 64
        Ul = ks.Mat.from_csr(ctx, *P.local_block(U.rowptr, U.col, U.val, q0, q1), row_start=q0, n_global=n)
        res["oneway_active"] = Ul.set_halo("peer")
        W = ks.BV(ctx, 64, 2, N=n)
        w = rng.standard_normal(n); wo = w.copy()
        W.set_column(0, w[q0:q1])
        for k in range(30):
            a, b = (0, 1) if k % 2 == 0 else (1, 0)
            Ul.mult_dev(W.column_ptr(a), W.column_ptr(b)); W.ScaleColumn(b, 0.25)
            wo = U.mult(wo) * 0.25
        res["oneway_err"] = float(np.abs(W.column(0) - wo[q0:q1]).max() / max(np.abs(wo).max(), 1e-300))
        # Lanczos and the full solver on top of the mailbox halo
        m = 12
        V = ks.BV(ctx, r1 - r0, m + 1, N=Aglob.n, row_start=r0)
        V.SetRandomColumn(0)
        _, nrm, _ = V.OrthogonalizeColumn(0); V.ScaleColumn(0, 1.0 / nrm)
        T = np.zeros((m + 1, 3), order="F")
        mm, beta, brk = V.MatLanczos(A, T, 0, m)
        res["T"] = T[:m, :2].copy(); res["beta"] = beta; res["mm"] = mm
        eps = ks.EPS(ctx); eps.SetOperators(A); eps.SetProblemType(ks.EPS_HEP); eps.SetDimensions(3, 12); eps.Solve()
        res["eig"] = [eps.GetEigenvalue(i)[0] for i in range(3)]; res["its"] = eps.GetIterationNumber(); res["nconv"] = eps.GetConverged()
        res["err"] = [eps.ComputeError(i) for i in range(3)]
        res["back"] = A.set_halo("provider")
        X.set_column(0, xs[1][r0:r1]); A.mult_dev(X.column_ptr(0), X.column_ptr(1))
        res["back_bits"] = bool(np.array_equal(X.column(1), prov[1]))
        dist.barrier()
        q.put((rank, res))
    except Exception:      # noqa: BLE001
        import traceback
        q.put((rank, {"error": traceback.format_exc()}))
    finally:
        dist.destroy_process_group()


def _run_peer(world, mode):
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_peer_halo_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs: p.start()
    out = dict(_collect(q, procs, world))
    for r in range(world):
        assert "error" not in out[r], out[r].get("error")
    return out


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4])
def test_peer_mapped_halo_between_processes_sharing_one_gpu(world):
    """SURVEY 8e "neighbour P2P of boundary x entries": every process maps its neighbours' ghost mailboxes through hipIpc (the route the
    ranks of an 8-GPU node take over xGMI; here on one card), the pack kernel stores straight into them. Same bits as the provider's
    exchange, the oracle's products, Lanczos coefficients and eigenvalues; 60 products back to back and a one-way communication
    pattern (acknowledgements only flowing back) stay correct."""
    from oracle import oracle as O
    out = _run_peer(world, "halo")
    A = O.laplacian3d(12, 10, 9); m = 12
    V = O.BV(A.n, m + 1); V.SetRandomColumn(0)
    _, nrm, _ = V.OrthogonalizeColumn(0); V.ScaleColumn(0, 1 / nrm)
    T = np.zeros((m + 1, 3), order="F")
    mm, beta, brk = V.MatLanczos(A, T, 0, m)
    r = O.eps_krylovschur_hep(A, 3, ncv=12)
    for rk in range(world):
        o = out[rk]
        assert o["active"] == "peer" and o["oneway_active"] == "peer" and o["back"] == "provider"
        assert o["bits_equal"] and o["back_bits"] and o["spmv_err"] < 1e-13
        assert o["chain_err"] < 1e-13 and o["oneway_err"] < 1e-13
        assert o["mm"] == mm and abs(o["beta"] - beta) < 1e-12 and np.abs(o["T"] - T[:m, :2]).max() < 1e-12
        assert o["its"] == r.its and o["nconv"] == r.nconv and max(o["err"]) < 1e-8
        assert np.allclose(o["eig"], r.eigr[r.perm][:3], rtol=1e-10)


@pytest.mark.timeout(600)
def test_peer_mapped_halo_gives_up_when_a_neighbour_never_arrives():
    """Rank 1 reaches the product 1.5 s late, the time limit is 0.3 s: rank 0 fills its ghosts with NaN, raises the error word and its next
    host wait fails - it does not hang."""
    out = _run_peer(2, "absent")
    assert out[0]["active"] == "peer"
    assert out[0]["raised"] and "halo" in out[0]["raised"], out[0]
    assert out[0]["seconds"] < 6.0


@pytest.mark.timeout(600)
def test_oneshot_allreduce_across_the_wrap_of_its_stamps():
    """The slot parity flips on every call, also where the 32-bit stamp skips 0: more than 100 calls started at 0xFFFFFFE0."""
    out = _run_peer(2, "wrap")
    for rk in (0, 1):
        assert out[rk]["active"] == "oneshot" and out[rk]["back"] == "provider"


# ---- the N > 1 bench line, end to end, with two ranks sharing the GPU -------------------------------------------------------------------------
@pytest.mark.gpu
def test_bench_n_gt_1_line_with_two_ranks_on_one_gpu():
    """`bench.py --gpus 2` as the driver will start it on a multi-GPU node, rehearsed here with both ranks on the one GPU and a gloo provider in
    place of RCCL (BENCH_PROVIDER=gloo; small slabs): the weak headline with its per-rank breakdown, and from the side-leg child the strong-scaling
    leg and both legs again with the one-shot allreduce and the peer-mapped halo between the two processes. The numbers mean nothing; the line does."""
    import json
    import subprocess
    env = dict(os.environ); env["BENCH_PROVIDER"] = "gloo"; env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--min-steps", "60", "--side", "96"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and "rehearsal" in d
    assert d["config"]["rows_per_gpu"] == 96 ** 3 and d["config"]["n_global"] == 2 * 96 ** 3 and d["config"]["gs_passes_per_step"] == 2.0
    b = d["multi_gpu_breakdown"]
    assert b["ranks"] == 2 and len(b["rank_timed_seconds"]) == 2 and b["allreduce_calls_per_step"] == 2.0 and b["halo_exchanges_per_step"] == 1.0
    assert b["restart_bcasts"] == d["config"]["cycles"] and all(x > 0 for x in b["per_rank_us_per_step"]["allreduce"])
    s = d["strong_scaling"]
    assert s["scaling"] == "strong" and s["value"] > 0 and s["n_global"] == 96 ** 3 and s["rows_per_gpu"] == 96 ** 3 // 2 and s["multi_gpu_breakdown"]["ranks"] == 2
    o = d["oneshot_allreduce"]
    assert o["active"] == "oneshot" and o["halo_active"] == "peer" and o["value"] > 0 and o["strong_scaling"]["value"] > 0, o
    assert d["side_legs_child"]["exit_code"] == 0
