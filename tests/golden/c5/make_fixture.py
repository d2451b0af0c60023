"""Generates tests/golden/c5/c5_n900_target36.json: an INDEPENDENT known answer for the config-5-shaped problem ("SciPy ... used
only to generate small committed fixtures ... random-nonsymmetric GNHEP cases for C5"). The reference holds no fixture for this configuration
(its shift-and-invert tests use PETSc's direct LU, absent here), so the oracle's generalized shift-and-invert Krylov-Schur is pinned against
LAPACK's dense generalized eigensolver (scipy.linalg.eig -> dggev) on the same pencil at n = 900: a different algorithm in a different library.
Also records checksums of the generated arrays, so that the generator itself (tests/nhep_cases.py config5_pencil, seed 42) is pinned.
Run from the repository root:  python tests/golden/c5/make_fixture.py"""
import hashlib
import json
import os
import sys

import numpy as np
import scipy.linalg as sl

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nhep_cases as nc      # noqa: E402

n, target, keep = 900, 36.0, 12
A, B = nc.config5_pencil(n)
Ad, Bd = A.to_scipy().toarray(), B.to_scipy().toarray()
lam = sl.eig(Ad, Bd, right=False)
order = np.argsort(np.abs(lam - target), kind="stable")
sel = lam[order[:keep]]
out = {
    "what": "generalized eigenvalues of the config-5-shaped pencil (A, B) at n = 900 closest to the target, by scipy.linalg.eig (LAPACK dggev)",
    "n": n, "target": target, "generator": "tests/nhep_cases.py config5_pencil(900), seed 42",
    "sha256": {k: hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest() for k, v in
               (("A.rowptr", A.rowptr), ("A.col", A.col), ("A.val", A.val), ("B.rowptr", B.rowptr), ("B.col", B.col), ("B.val", B.val))},
    "nnz": {"A": int(A.rowptr[-1]), "B": int(B.rowptr[-1])},
    "eigenvalues_by_distance_to_target": [[float(np.real(z)), float(np.imag(z))] for z in sel],
    "versions": {"numpy": np.__version__, "scipy": __import__("scipy").__version__},
}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c5_n900_target36.json")
json.dump(out, open(path, "w"), indent=1)
print("wrote", path)
for z in sel[:6]:
    print("  %.15g %+.15gi   |lambda - target| = %.6f" % (np.real(z), np.imag(z), abs(z - target)))
