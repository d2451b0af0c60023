"""Known-byte-count launches for calibrating rocprofv3 FETCH_SIZE/WRITE_SIZE on gfx950 and for the MFMA counters
of the panel kernels. Run under: rocprofv3 --pmc <COUNTERS> --kernel-trace --output-format csv -d <dir> -- python3 scripts/pmc_calib.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slepc_amd as ks
ctx = ks.Context(0)
n, m = 216 ** 3, 32
for ld in (0, n + 1):                       # ld=0: 16-byte loads (VEC=2); odd ld: 8-byte loads (VEC=1)
    X = ks.BV(ctx, n, m, ld)
    for j in range(m):
        X.SetRandomColumn(j)
    X.SetActiveColumns(0, 31)
    for _ in range(3):
        X.DotVec(X.column_ptr(31))          # k_dot_sweep<32,VEC>: reads 31 columns + y = 32 * 8n bytes
    if ld == 0:
        Y = ks.BV(ctx, n, m)
        for j in range(m):
            Y.SetRandomColumn(j, 7)
        Y.SetActiveColumns(0, 31)
        M = np.zeros((31, 31), order="F")
        for _ in range(3):
            X.Dot(Y, M)                     # k_panel_dot_mfma<2,2>: reads 62 columns; 2*n*31*31 flop (padded 32x32 on the MFMA)
        Q = np.asfortranarray(np.random.default_rng(0).standard_normal((31, 31)))
        X.SetActiveColumns(0, 30)
        for _ in range(3):
            X.MultInPlace(Q, 0, 20)         # k_panel_mult_mfma<8,2>: reads 30 columns, writes 20
        del Y
    del X
ctx.synchronize()
print("done")
