import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Fixtures are plain numeric arrays written by tests/golden/make_golden.py."""
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_err(a, b):
    """max |a-b| / max(|b|_inf, tiny): the 'relative' of BASELINE.json's 1e-5."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def assert_elementwise(a, b, rtol=1e-5, atol=1e-12, what=""):
    """BASELINE.json's "within 1e-5 relative" read ELEMENT BY ELEMENT: every entry of ``a`` within
    ``rtol * |b| + atol`` of its reference value (``rel_err`` is the norm-wise reading; small
    entries of V are individually held to the tolerance here)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    bad = ~(np.abs(a - b) <= rtol * np.abs(b) + atol)  # (NaN anywhere counts as bad)
    if bad.any():
        i = np.unravel_index(int(np.argmax(np.where(bad, np.abs(a - b), -1.0))), a.shape) if a.ndim else ()
        raise AssertionError(f"{what}: {int(bad.sum())} of {a.size} elements outside {rtol}*|b|+{atol}; "
                             f"worst at {i}: got {a[i]!r}, want {b[i]!r}")


def check_matrix_summary(g, prefix, M, tight, what=""):
    """A parameter matrix against its ``published_*.npz`` summary (make_golden_published.py):
    every stride-th row norm-wise at ``tight`` and element-wise at the contract tolerance, plus
    the row sums, column sums and sum of squares of the WHOLE matrix (norm-wise)."""
    M = np.asarray(M, dtype=np.float64)
    assert tuple(M.shape) == tuple(int(v) for v in g[f"{prefix}_shape"]), (what, prefix, M.shape)
    rows = M[:: int(g[f"{prefix}_stride"])]
    assert rel_err(rows, g[f"{prefix}_rows"]) < tight, (what, prefix, rel_err(rows, g[f"{prefix}_rows"]))
    assert_elementwise(rows, g[f"{prefix}_rows"], what=f"{what} {prefix} rows")
    assert rel_err(M.sum(axis=1), g[f"{prefix}_rowsum"]) < tight, (what, prefix, "row sums")
    assert rel_err(M.sum(axis=0), g[f"{prefix}_colsum"]) < tight, (what, prefix, "column sums")
    assert abs(float((M * M).sum()) - float(g[f"{prefix}_sqsum"])) <= tight * float(g[f"{prefix}_sqsum"]), (what, prefix)
