"""One-shot allreduce between W processes sharing one GPU: time per call (HIP events around 2000 back-to-back calls of 31 doubles)
and the Lanczos step rate of a small sharded problem with the one-shot path against the host-staged gloo provider.
usage: python scripts/oneshot_probe.py [W]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import slepc_amd as ks
    from test_gpu_multirank import _install_gloo_ops
    ctx = ks.Context(0)
    _install_gloo_ops(ks, ctx, dist, torch, rank, world)
    res = {"active": ctx.set_allreduce("oneshot")}
    ctx.comm_check()
    X = ks.BV(ctx, 64, 12, N=64 * world)
    d = X.buffer_ptr()
    for count in (1, 31, 61, 128):
        ctx.memcpy_h2d(d, np.ones(count))
        for _ in range(20): ctx.allreduce_sum_dev(d, count)
        ctx.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        for _ in range(2000):
            ctx.memcpy_h2d(d, np.ones(count)) if False else None
            ctx.allreduce_sum_dev(d, count)
        ctx.synchronize()
        res["us_per_call_%d" % count] = 1e6 * (time.perf_counter() - t0) / 2000
    dist.barrier()
    q.put((rank, res))
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=worker, args=(r, world, 30500 + os.getpid() % 1000, q)) for r in range(world)]
    for p in procs: p.start()
    out = dict(q.get(timeout=300) for _ in range(world))
    for p in procs: p.join(60)
    for r in range(world):
        print(r, out[r])
