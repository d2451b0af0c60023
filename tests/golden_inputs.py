"""Deterministic inputs of the reference's BV tests (formulas from SURVEY.md section 4) and a parser for its
golden .out files (tests/golden/, data copied from the reference's test-suite)."""
import os
import re

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_NUM = r"[-+]?(?:\d+\.?\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?)"


def read(rel):
    return open(os.path.join(GOLDEN, rel)).read()


def numeric_blocks(text):
    """Consecutive lines that consist only of numbers -> list of 2-D arrays (ragged lines kept per line)."""
    blocks, cur = [], []
    for line in text.splitlines():
        toks = line.split()
        if toks and all(re.fullmatch(_NUM, t) for t in toks):
            cur.append([float(t) for t in toks])
        else:
            if cur:
                blocks.append(cur)
                cur = []
    if cur:
        blocks.append(cur)
    out = []
    for b in blocks:
        w = {len(r) for r in b}
        out.append(np.array(b) if len(w) == 1 else [np.array(r) for r in b])
    return out


def section_after(text, label):
    """Numeric blocks that follow `label` up to the next 'After' label."""
    i = text.index(label) + len(label)
    j = text.find("After ", i)
    return numeric_blocks(text[i:j if j >= 0 else len(text)])


def value_after(text, label):
    m = re.search(re.escape(label) + r"\s*(" + _NUM + ")", text)
    return float(m.group(1))


def eigenvalues_line(text):
    """The '%.5f, %.5f, ...' line printed by EPSErrorView in -terse mode."""
    for line in text.splitlines():
        toks = [t for t in line.replace(",", " ").split()]
        if len(toks) >= 2 and all(re.fullmatch(_NUM, t) for t in toks) and "." in toks[0]:
            return np.array([float(t) for t in toks])
    raise ValueError("no eigenvalue line")


def complex_eigenvalues_line(text):
    """The -terse eigenvalue line when it holds conjugate pairs: '-243874.97870+6999.66927i, ..., -212991.49278'."""
    pat = re.compile(r"^\s*(" + _NUM + r")(?:([-+]\d+\.?\d*)i)?\s*$")
    for line in text.splitlines():
        toks = [t.strip() for t in line.split(",")]
        if len(toks) >= 2 and all(pat.match(t) for t in toks) and "." in toks[0]:
            out = []
            for t in toks:
                m = pat.match(t)
                out.append(complex(float(m.group(1)), float(m.group(2)) if m.group(2) else 0.0))
            return np.array(out)
    raise ValueError("no eigenvalue line")


def matrix_path(name):
    return os.path.join(GOLDEN, "matrices", name)


def table_first_column(text):
    """First numeric column of an EPSErrorView-style table (test29_1.out)."""
    vals = []
    for line in text.splitlines():
        toks = line.split()
        if toks and re.fullmatch(_NUM, toks[0]) and len(toks) >= 2 and "." in toks[0]:
            vals.append(float(toks[0]))
    return np.array(vals)


# ---- inputs (src/sys/classes/bv/tests/test1.c:60-99, test2.c:58-69, test4.c, test13.c) -----------------
def test1_X(n=10, k=5):
    X = np.zeros((n, k))
    for j in range(k):
        for i in range(4):
            if i + j < n:
                X[i + j, j] = 3 * i + j - 2
    return X


def test1_Y(n=10, l=3):
    Y = np.zeros((n, l))
    for j in range(l):
        Y[:, j] = (j + 1) / 4.0
    return Y


def test1_Q(k=5, l=3):
    return np.array([[2.0 if i < j else -0.5 for j in range(l)] for i in range(k)], order="F")


def test2_X(n=20, k=8):
    X = np.zeros((n, k))
    for j in range(k):
        for i in range(n // 2 + 1):
            if i + j < n:
                X[i + j, j] = (3.0 * i + j - 2) / (2 * (i + j + 1))
    return X


def eigenvalue_lines(text):
    """Every line that follows 'All requested eigenvalues computed ...' (one per solve of the program), as arrays."""
    out = []
    lines = text.splitlines()
    for i, line in enumerate(lines):
        if "All requested eigenvalues computed" in line:
            out.append(np.array([float(t) for t in lines[i + 1].replace(",", " ").split()]))
    return out


def complex_eigenvalue_lines(text):
    """As eigenvalue_lines, for values printed as 'a+bi' / 'a-bi' / plain reals (EPSErrorView of a non-symmetric problem)."""
    out = []
    lines = text.splitlines()
    for i, line in enumerate(lines):
        if "All requested eigenvalues computed" in line:
            out.append(np.array([complex(t.replace("i", "j")) for t in lines[i + 1].replace(",", " ").split()]))
    return out


def eigenvalues_block(text):
    """All numbers after 'All requested eigenvalues computed ...' up to the next blank line (EPSErrorView wraps long lists)."""
    lines = text.splitlines()
    i = next(k for k, l in enumerate(lines) if "All requested eigenvalues computed" in l) + 1
    out = []
    while i < len(lines) and lines[i].strip():
        out += [float(t) for t in lines[i].replace(",", " ").split()]
        i += 1
    return np.array(out)


def eigenvalues_after(text, marker):
    """The numbers that follow `marker` on its own line (programs that print their values after a sentence)."""
    line = next(l for l in text.splitlines() if marker in l)
    return np.array([float(t) for t in line.split(marker, 1)[1].replace(",", " ").split()])
