import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    # a fresh checkout has no built artefacts (they are git-ignored): build the HIP library (hipcc
    # cross-compiles without a GPU) and the C oracle once, as __graft_entry__.build() does
    from pyrad_amd import _native
    if not os.path.isfile(_native.LIB_PATH):
        _native.build()
    from oracle import c_oracle
    c_oracle.build()


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def unpack_lines(z, prefix):
    from pyrad_amd import synthetic
    return {f: np.ascontiguousarray(z["%s.%s" % (prefix, f)]) for f in synthetic.FIELDS}


def rel_err(a, b, floor=0.0):
    """max |a-b| / max(|b|, floor) over elements where the denominator is non-zero;
    exact zeros in b must be matched by exact zeros in a."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    both_nan = np.isnan(a) & np.isnan(b)
    den = np.maximum(np.abs(b), floor)
    zero = den == 0
    if zero.any():
        assert np.all(a[zero] == 0), "non-zero where the reference is exactly zero"
    ok = ~zero & ~both_nan
    if not ok.any():
        return 0.0
    return float(np.max(np.abs(a[ok] - b[ok]) / den[ok]))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def write_xsc_tree(z, root):
    """Materialise the xsc files stored in the G8 golden (bytes written by the reference's own
    writer) under ``root`` in PyRad's data/xsc layout."""
    import json, os
    for i, key in enumerate(json.loads(str(z["tree_json"]))):
        mol, fn = key.split("/")
        os.makedirs(os.path.join(root, mol), exist_ok=True)
        with open(os.path.join(root, mol, fn), "wb") as f:
            f.write(z["tree.%d" % i].tobytes())
