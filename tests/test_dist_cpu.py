"""N>1 path on CPU: world_size-2 gloo run of the row-sharded Lanczos expansion.

The GPU path cannot run here, so this test drives the SAME distributed formulation with the oracle's local
kernels: per-rank row slab, halo exchange of the SpMV boundary entries, local partial dots, allreduce(SUM) of
the k+1 coefficients between the reduce and the bookkeeping halves of each CGS pass (bvblas.c:255), replicated
scalar control flow.  It checks the sharding helpers used by bench.py (slepc_amd.partition) and that the sharded
run reproduces the single-rank T, beta, pass counts and Ritz values."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)



def _free_port():
    """A rendezvous port the OS says is free right now (ports computed from the pid collided with the ephemeral ports of earlier tests' gloo pairs:
    EADDRINUSE on the box, round 4)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, shape, m, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from slepc_amd import partition as P
    nx, ny, nz = shape
    plane = nx * ny
    own = P.split_ownership(nz, world)                         # z-planes per rank
    z0, z1 = own[rank]
    A = O.laplacian3d(nx, ny, nz, z0, z1 - z0)                 # my rows, GLOBAL column indices
    r0, r1 = z0 * plane, z1 * plane
    n = r1 - r0
    ghosts = P.ghost_columns(A.col, r0, r1)
    assert len(ghosts) == plane * ((z0 > 0) + (z1 < nz))       # one plane per neighbour
    eta, deftol = 0.7071, 10 * np.finfo(float).eps

    def allreduce(a):
        t = torch.from_numpy(np.ascontiguousarray(a)); dist.all_reduce(t); return t.numpy()

    def spmv(xl):
        # halo: every rank publishes its first and last plane (all_gather keeps the test simple)
        parts = [None] * world
        dist.all_gather_object(parts, (xl[:plane].copy(), xl[-plane:].copy()))
        xg = np.zeros(nx * ny * nz)
        xg[r0:r1] = xl
        if rank > 0: xg[r0 - plane:r0] = parts[rank - 1][1]
        if rank < world - 1: xg[r1:r1 + plane] = parts[rank + 1][0]
        return A.mult(xg)

    V = np.zeros((n, m + 1)); H = np.zeros((m + 1, m + 1)); passes = 0
    lib = O.lib()
    v0 = np.array([lib.orc_random_value(0x12345678, 0, r0 + i) for i in range(n)])
    V[:, 0] = v0 / np.sqrt(allreduce(np.array([v0 @ v0]))[0])
    beta = 0.0
    for j in range(m):
        w = spmv(V[:, j]); k = j + 1
        l, onrm, nrm = 0, 0.0, 0.0
        while True:
            c = allreduce(np.array([V[:, i] @ w for i in range(k)] + [w @ w]))      # k+1 dots, ONE allreduce per pass
            assert c[k] > -deftol
            bta = np.sqrt(max(c[k], 0.0))
            w = w - V[:, :k] @ c[:k]
            s = float(np.sum(c[:k] ** 2)); n2 = bta * bta - s
            nrm = np.sqrt(n2) if n2 > 0 else np.sqrt(allreduce(np.array([w @ w]))[0])
            onrm = bta; H[:k, k] += c[:k]; l += 1; passes += 1
            if not (l < 3 and nrm != 0 and abs(nrm) < eta * abs(onrm)):
                break
        H[k, k] = nrm; beta = nrm
        V[:, k] = w / nrm
    if rank == 0:
        out.put((np.diag(H, 0)[1:m + 1].copy(), np.array([H[j + 1, j + 1] for j in range(m)]), np.array([H[j, j + 1] for j in range(m)]), beta, passes))
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_lanczos_matches_single_rank():
    import torch.multiprocessing as mp
    from oracle import oracle as O
    shape, m = (6, 5, 8), 10
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = _free_port()
    procs = [ctxm.Process(target=_worker, args=(r, 2, port, shape, m, q)) for r in range(2)]
    for p in procs: p.start()
    try:
        nrm_diag, betas, alphas, beta, passes = q.get(timeout=240)
        for p in procs: p.join(60); assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate(); p.join(10)
    # single-rank oracle
    A = O.laplacian3d(*shape)
    V = O.BV(A.n, m + 1); V.SetRandomColumn(0)
    _, nrm, _ = V.OrthogonalizeColumn(0); V.ScaleColumn(0, 1 / nrm)
    T = np.zeros((m + 1, 3), order="F")
    p0 = V.passes_total()
    mm, b1, brk = V.MatLanczos(A, T, 0, m)
    assert mm == m and not brk
    assert np.allclose(alphas, T[:m, 0], rtol=1e-12, atol=1e-13)
    assert np.allclose(betas, T[:m, 1], rtol=1e-12, atol=1e-13)
    assert abs(beta - b1) < 1e-12
    assert passes == V.passes_total() - p0


def test_partition_helpers():
    from slepc_amd import partition as P
    assert P.split_ownership(10, 3) == [(0, 4), (4, 7), (7, 10)]
    assert P.split_ownership(8, 8) == [(i, i + 1) for i in range(8)]
    (nx, ny, nz), slabs = P.slab_grid(216, 8)
    assert (nx, ny, nz) == (432, 432, 432) and slabs[3] == (162, 54)
    assert nx * ny * slabs[0][1] == 216 ** 3 == 10077696          # rows per GPU identical to config 3
    assert nx * ny * nz == 80621568                                # config 4
    (a, b, c), s1 = P.slab_grid(216, 1)
    assert (a, b, c) == (216, 216, 216) and s1 == [(0, 216)]
    for w in (2, 4):
        (nx, ny, nz), slabs = P.slab_grid(216, w)
        assert nz == 54 * w and all(nzl == 54 for _, nzl in slabs) and [z for z, _ in slabs] == [54 * r for r in range(w)]
    rowptr = [0, 2, 3, 5, 6]; col = np.array([0, 3, 1, 0, 2, 3]); val = np.arange(6.0)
    rp, c, v = P.local_block(rowptr, col, val, 1, 3)
    assert rp == [0, 1, 3] and list(c) == [1, 0, 2] and list(v) == [2.0, 3.0, 4.0]
    assert P.ghost_columns(c, 1, 3) == [0]
