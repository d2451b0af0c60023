"""Row-wise distribution helpers (host logic of the multi-rank tests).

The path shards exactly like the reference: contiguous row blocks per rank (PetscLayout,
src/sys/classes/bv/interface/bvbasic.c:129-134), small dense matrices replicated."""


def split_ownership(N, size):
    """PetscSplitOwnership: n_r = N/size + (r < N % size). Returns the list of (start, end) per rank."""
    out, start = [], 0
    for r in range(size):
        n = N // size + (1 if r < N % size else 0)
        out.append((start, start + n))
        start += n
    return out


def slab_grid(side, world):
    """Weak-scaling grid of bench.py: `world` z-slabs of side^3 rows each.
    world == 1: side x side x side.  world > 1: (2 side) x (2 side) x (side/4 * world), side/4 planes per rank."""
    if world == 1:
        return (side, side, side), [(0, side)]
    assert side % 4 == 0, "side must be a multiple of 4 so that a slab of (2 side)^2 planes holds side^3 rows"
    planes = side // 4
    return (2 * side, 2 * side, planes * world), [(r * planes, planes) for r in range(world)]


def local_block(rowptr, col, val, start, end):
    """Rows [start,end) of a global CSR, keeping GLOBAL column indices (what ks_mat_create_csr takes)."""
    p0, p1 = int(rowptr[start]), int(rowptr[end])
    return [int(x) - p0 for x in rowptr[start:end + 1]], col[p0:p1], val[p0:p1]


def ghost_columns(col, start, end):
    """Sorted global column indices referenced by a row block but owned by other ranks (PETSc garray)."""
    return sorted({int(c) for c in col if c < start or c >= end})
