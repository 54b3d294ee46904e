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


C2_INTENSITY = 299792458.0 * 6.62607004e-34 * 100 / 1.38064852E-23      # pyradIntensity.py:13
RTOL_BASE = 2e-12      # summation association + libm-vs-ocml last bits away from nu -> 0.  Measured: every point of
#                        C2 / C3 / three C5 layers <= 1.2e-14 (tests/test_gpu_whole_spectrum.py prints the worst);
#                        random cells away from 0 cm^-1 <= 1e-12 (scripts/stress_random_cells.py)
ULPS_STIM = 16.0       # last-bit differences of the four exp() in the stimulated-emission ratio, both sides


def point_tolerance(nu_axis, T, dfc, rtol_base=RTOL_BASE):
    """Per-point relative tolerance of a device spectrum against the oracle / the goldens.

    Both sides are fp64.  Away from nu = 0 they differ by summation order and the last bits of exp/log/pow
    (RTOL_BASE).  The reference's stimulated-emission factor 1 - exp(-c2 nu / T) (pyradIntensity.py:23-27)
    cancels for nu -> 0: a last-bit difference between NumPy's and the device's exp is amplified by
    T / (c2 nu) in the line intensity, on BOTH sides, so the bound for a grid point is stated for the
    lowest line that can reach it (nu - dfc): ULPS_STIM * 2^-53 * T / (c2 * max(nu - dfc, tiny)).
    At 100 cm^-1 that term is 4e-15; it passes 1e-11 only below 0.04 cm^-1."""
    nu = np.maximum(np.asarray(nu_axis, dtype=np.float64) - dfc, 1e-6)
    return rtol_base + ULPS_STIM * 2.0 ** -53 * float(T) / (C2_INTENSITY * nu)


def rel_err_points(a, b, floor=0.0):
    """|a-b| / max(|b|, floor) per element (0 where both are exactly 0; inf where only b is)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.maximum(np.abs(b), floor)
    with np.errstate(divide="ignore", invalid="ignore"):
        e = np.abs(a - b) / den
    e[(den == 0) & (a == 0)] = 0.0
    e[(den == 0) & (a != 0)] = np.inf
    return e


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
