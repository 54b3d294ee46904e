"""ctypes loader of the plain-C oracle (oracle/lbl_oracle.c).  TEST INFRASTRUCTURE ONLY:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "_build", "liblbl_oracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
    return LIB


def load():
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB):
            build()
        _lib = C.CDLL(LIB)
        _lib.lbl_oracle_xsec.restype = C.c_int64
        _lib.lbl_oracle_xsec.argtypes = [C.c_void_p] * 7 + [C.c_int64] + [C.c_double] * 8 + [C.c_int64, C.c_int64,
                                                                                           C.c_void_p, C.c_void_p]
    return _lib


def create_cross_section_work(lines, T, P, conc, molmass, q_T, q296, grid):
    """Work-grid cross section (before the regrid of cls:401-405), regime counts, exact evals."""
    lib = load()
    order = ("nu", "sw", "elower", "gamma_air", "gamma_self", "n_air", "delta_air")
    arrs = [np.ascontiguousarray(lines[k], dtype=np.float64) for k in order]
    out = np.zeros(int(grid["n_work"]), dtype=np.float64)
    counts = (C.c_int64 * 3)()
    evals = lib.lbl_oracle_xsec(*[a.ctypes.data_as(C.c_void_p) for a in arrs], len(arrs[0]), float(T), float(P),
                                float(conc), float(molmass), float(q_T), float(q296), float(grid["range_min"]),
                                float(grid["resolution"]), int(grid["W"]), int(grid["n_work"]),
                                out.ctypes.data_as(C.c_void_p), counts)
    if evals < 0:
        raise MemoryError("lbl_oracle_xsec")
    return out, tuple(int(c) for c in counts), int(evals)
