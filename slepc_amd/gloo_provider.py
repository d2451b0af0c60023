"""A caller-supplied communicator for libksgpu (ks_comm_set_ops) backed by torch.distributed's gloo backend - the shape an MPI-backed provider has:
three operations (allreduce of a device buffer, allgather of host bytes, neighbour exchange of device segments), each ordering itself on the stream the
library hands it. Host-staged: for tests and for rehearsing the multi-rank code paths with several ranks on ONE GPU (where RCCL refuses to run);
the product's communicator is the native RCCL one (ks_comm_init_rccl)."""
import ctypes

import numpy as np


def install(ctx, dist, torch, rank, size, perturb=False):
    """Host-staged provider. Every operation orders itself on the stream the library hands it (the halo exchange comes on
    the halo stream, the reductions on the main one). perturb: rank 1 returns every reduced value one ulp up - an
    allreduce that is not bitwise identical across ranks, which a caller-supplied provider is allowed to be."""
    def allreduce_sum(ptr, count, stream):
        h = np.empty(count)
        ctx.memcpy_d2h(h, ptr, stream)
        t = torch.from_numpy(h)
        dist.all_reduce(t)
        if perturb and rank == 1:
            h[:] = np.nextafter(h, np.inf)
        ctx.memcpy_h2d(ptr, h, stream)
        return 0

    def allgather_host(send, nbytes, recv):
        buf = torch.frombuffer(bytearray(ctypes.string_at(send, nbytes)), dtype=torch.uint8)
        outs = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(size)]
        dist.all_gather(outs, buf)
        ctypes.memmove(recv, b"".join(o.numpy().tobytes() for o in outs), nbytes * size)
        return 0

    def exchange(peers, dsend, soff, scnt, drecv, roff, rcnt, eb, stream):
        ops, recvs = [], []
        for i, p in enumerate(peers):
            if scnt[i]:
                h = np.empty(scnt[i] * eb, dtype=np.uint8)
                ctx.memcpy_d2h(h, dsend + soff[i] * eb, stream)
                ops.append(dist.P2POp(dist.isend, torch.from_numpy(h), p))
            if rcnt[i]:
                r = torch.empty(rcnt[i] * eb, dtype=torch.uint8)
                recvs.append((i, r))
                ops.append(dist.P2POp(dist.irecv, r, p))
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        for i, r in recvs:
            ctx.memcpy_h2d(drecv + roff[i] * eb, r.numpy(), stream)
        return 0

    ctx.set_comm_ops(rank, size, allreduce_sum, allgather_host, exchange)
