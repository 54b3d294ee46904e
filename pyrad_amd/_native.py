"""ctypes binding of ``libpyrad_hip.so`` (the C ABI in ``include/pyrad_hip.h``).

No PyTorch, no Triton: the Python host talks to the HIP kernels through plain
pointers and sizes.  If the shared library is missing or cannot be loaded this module
raises — there is deliberately no CPU fallback in the product path.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PYRAD_HIP_LIB") or os.path.join(_HERE, "lib", "libpyrad_hip.so")
CSRC = os.path.join(_HERE, "csrc")

LBL_OK = 0
ERR_NAMES = {0: "LBL_OK", -1: "LBL_ERR_BAD_ARG", -2: "LBL_ERR_NO_DEVICE", -3: "LBL_ERR_HIP",
             -4: "LBL_ERR_RCCL", -5: "LBL_ERR_OOM", -6: "LBL_ERR_STATE"}
UNIQUE_ID_BYTES = 128


class LblError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("%s (%d): %s" % (ERR_NAMES.get(code, "LBL_ERR_?"), code, message))
        self.code = code


class NoDeviceError(LblError):
    pass


class IsoParams(C.Structure):
    """lbl_iso_params"""
    _fields_ = [("T", C.c_double), ("P", C.c_double), ("q_frac", C.c_double),
                ("molmass", C.c_double), ("Q_T", C.c_double), ("Q_296", C.c_double)]


class Grid(C.Structure):
    """lbl_grid"""
    _fields_ = [("range_min", C.c_double), ("range_max", C.c_double), ("resolution", C.c_double),
                ("base_resolution", C.c_double), ("n_work", C.c_int64), ("n_base", C.c_int64),
                ("window", C.c_int64), ("shard_first", C.c_int64), ("shard_count", C.c_int64)]


_P = C.c_void_p
_D = C.POINTER(C.c_double)

# name -> (restype, argtypes): every symbol include/pyrad_hip.h declares
SIGNATURES = {
    "lbl_abi_version": (C.c_int, []),
    "lbl_limit": (C.c_int, [C.c_char_p, C.POINTER(C.c_int64)]),
    "lbl_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "lbl_ctx_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "lbl_ctx_destroy": (C.c_int, [_P]),
    "lbl_last_error": (C.c_char_p, [_P]),
    "lbl_sync": (C.c_int, [_P]),
    "lbl_ctx_stream": (C.c_int, [_P, C.POINTER(_P)]),
    "lbl_ctx_chain_accumulate": (C.c_int, [_P, _P]),
    "lbl_lines_view": (C.c_int, [_P, C.c_int64, C.c_int64, C.POINTER(_P)]),
    "lbl_device_info": (C.c_int, [_P, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "lbl_set_option": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "lbl_profile_enable": (C.c_int, [_P, C.c_int]),
    "lbl_profile_read": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int64), _D]),
    "lbl_profile_reset": (C.c_int, [_P]),
    "lbl_profile_reserve": (C.c_int, [_P, C.c_int]),
    "lbl_buffer_create": (C.c_int, [_P, C.c_int64, C.POINTER(_P)]),
    "lbl_buffer_destroy": (C.c_int, [_P]),
    "lbl_buffer_size": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "lbl_buffer_upload": (C.c_int, [_P, _P, C.c_int64, C.c_int64]),
    "lbl_buffer_download": (C.c_int, [_P, _P, C.c_int64, C.c_int64]),
    "lbl_buffer_fill": (C.c_int, [_P, C.c_double]),
    "lbl_buffer_download_async": (C.c_int, [_P, _P, C.c_int64, C.c_int64]),
    "lbl_download_wait": (C.c_int, [_P]),
    "lbl_buffer_devptr": (C.c_int, [_P, C.POINTER(_P)]),
    "lbl_host_alloc": (C.c_int, [_P, C.c_int64, C.POINTER(_P)]),
    "lbl_host_free": (C.c_int, [_P, _P]),
    "lbl_lines_create": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, C.c_int64, C.POINTER(_P)]),
    "lbl_lines_destroy": (C.c_int, [_P]),
    "lbl_lines_count": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "lbl_xsec_accumulate": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, C.c_int64, C.POINTER(IsoParams),
                                      C.POINTER(Grid), _P, C.POINTER(C.c_int64)]),
    "lbl_xsec_accumulate_dev": (C.c_int, [_P, C.c_int, C.POINTER(_P), C.POINTER(IsoParams), C.POINTER(Grid),
                                          C.POINTER(_P)]),
    "lbl_schedule_export": (C.c_int, [_P, C.c_int, _P, C.c_int64, _P, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                      C.POINTER(C.c_int32)]),
    "lbl_last_regime_counts": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int64)]),
    "lbl_line_quantities": (C.c_int, [_P, _P, C.POINTER(IsoParams), C.POINTER(Grid), _P, _P, _P, _P, _P]),
    "lbl_layer_sweep_dev": (C.c_int, [_P, C.c_int, C.POINTER(_P), C.POINTER(C.c_int32), C.c_int, _D,
                                      C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int64,
                                      C.c_int64, C.c_int64, _P, C.c_double, _P, _P, _P]),
    "lbl_layer_step_dev": (C.c_int, [_P, C.c_int, C.POINTER(_P), C.POINTER(IsoParams), C.POINTER(Grid), C.POINTER(_P),
                                     C.POINTER(C.c_int32), C.c_int, _D, C.c_double, _P, C.c_double, _P, _P, _P]),
    "lbl_layer_merged_step_dev": (C.c_int, [_P, C.c_int, C.POINTER(_P), C.POINTER(IsoParams), C.POINTER(Grid),
                                            C.POINTER(C.c_int32), C.c_int, _D, C.c_double, _P, C.c_double, _P, _P, _P]),
    "lbl_layers_merged_accumulate_dev": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int32), C.POINTER(_P), C.POINTER(IsoParams),
                                                   C.POINTER(Grid), C.POINTER(C.c_int32), C.POINTER(C.c_int32), _D,
                                                   C.POINTER(_P)]),
    "lbl_column_fold_dev": (C.c_int, [_P, C.c_int, C.POINTER(_P), _D, _D, C.c_double, C.c_double, C.c_int64, C.c_int64,
                                      C.c_int64, _P, C.c_double, C.POINTER(_P), _P]),
    "lbl_column_create": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int32), C.POINTER(_P), C.POINTER(IsoParams), C.POINTER(Grid),
                                    C.POINTER(C.c_int32), C.POINTER(C.c_int32), _D, _D, C.POINTER(_P), C.POINTER(_P)]),
    "lbl_column_destroy": (C.c_int, [_P]),
    "lbl_column_set_layer": (C.c_int, [_P, C.c_int, C.POINTER(_P), C.POINTER(IsoParams), C.POINTER(Grid), _D, C.c_double, _P]),
    "lbl_column_transmission": (C.c_int, [_P, C.POINTER(C.c_uint8), _P, C.c_double, _P, _P, C.c_int]),
    "lbl_column_sweep_dev": (C.c_int, [_P, C.c_int, C.POINTER(_P), _D, C.c_double, C.c_double, C.c_int64,
                                       C.c_int64, C.c_int64, _P, C.c_double, _P]),
    "lbl_column_step_dev": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int32), C.POINTER(_P), C.POINTER(C.c_int32),
                                      C.POINTER(C.c_int32), _D, _D, _D, _D, C.c_double, C.c_double, C.c_int64, C.c_int64,
                                      C.c_int64, _P, C.c_double, C.POINTER(_P), C.POINTER(_P), _P]),
    "lbl_optical_dev": (C.c_int, [_P, _P, C.c_int64, C.c_int, _P]),
    "lbl_sum_dev": (C.c_int, [_P, C.c_int, C.POINTER(_P), C.c_int64, _P]),
    "lbl_planck_dev": (C.c_int, [_P, C.c_double, C.c_double, C.c_int64, C.c_double, _P]),
    "lbl_band_integral": (C.c_int, [_P, _P, C.c_int64, C.c_double, C.c_double, _D]),
    "lbl_line_survey_dev": (C.c_int, [_P, _P, C.POINTER(Grid), _P]),
    "lbl_comm_unique_id": (C.c_int, [C.c_char_p]),
    "lbl_comm_create": (C.c_int, [_P, C.c_char_p, C.c_int, C.c_int, C.POINTER(_P)]),
    "lbl_comm_destroy": (C.c_int, [_P]),
    "lbl_allgather_dev": (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P]),
    "lbl_allgather_overlap_dev": (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P, C.c_int]),
    "lbl_comm_fence_dev": (C.c_int, [_P, C.c_int]),
    "lbl_gather_stage_dev": (C.c_int, [_P, C.c_int64, _P, C.c_int64, C.c_int64]),
    "lbl_capture_begin": (C.c_int, [_P]),
    "lbl_capture_end": (C.c_int, [_P, C.POINTER(_P)]),
    "lbl_graph_launch": (C.c_int, [_P]),
    "lbl_graph_destroy": (C.c_int, [_P]),
    "lbl_gather_compact_dev": (C.c_int, [_P, _P, C.c_int, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _P]),
}

_lib = None


def build(force: bool = False) -> str:
    """Compile the HIP sources for gfx950 with hipcc (``make`` in pyrad_amd/csrc)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", CSRC, "-j4"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def source_hash() -> str:
    """sha256 over what decides the kernels and their launch shapes (csrc/lbl_kernels.hip, lbl_api.hip,
    lbl_device.h, Makefile): profiles/pmc_traffic.json records it so that bench.py can tell when
    committed PMC numbers were measured on other kernels than the ones it is timing."""
    import hashlib
    h = hashlib.sha256()
    for f in ("Makefile", "lbl_api.hip", "lbl_device.h", "lbl_kernels.hip", "lbl_launch_shapes.h"):
        h.update(f.encode())
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def load():
    """Load the shared library (once) and declare every signature."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C pyrad_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


_limits = {}


def limit(name: str) -> int:
    """A fixed size of the library (lbl_limit): "merged_lists_per_job", "arrays_per_layer", "arrays_per_sum",
    "arrays_per_column", "layers_per_column", "jobs_per_batch"."""
    if name not in _limits:
        v = C.c_int64()
        rc = load().lbl_limit(name.encode(), C.byref(v))
        if rc != LBL_OK:
            raise LblError(rc, "unknown limit %r" % name)
        _limits[name] = int(v.value)
    return _limits[name]


def _as_f64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
    """One HIP device + one stream (lbl_ctx).  Not thread-safe."""

    def __init__(self, device: int = 0):
        self.lib = load()
        h = _P()
        rc = self.lib.lbl_ctx_create(int(device), C.byref(h))
        if rc != LBL_OK:
            msg = (self.lib.lbl_last_error(None) or b"").decode()
            raise (NoDeviceError if rc == -2 else LblError)(rc, msg)
        self.h = h
        self.device = int(device)
        self._children = []

    # -- plumbing ------------------------------------------------------------------------
    def check(self, rc):
        if rc != LBL_OK:
            raise LblError(rc, (self.lib.lbl_last_error(self.h) or b"").decode())

    def close(self):
        if getattr(self, "h", None):
            # views before the line lists they window (lbl_lines_destroy refuses a list with live views), then
            # everything else newest first
            for child in reversed(list(self._children)):
                if isinstance(child, Lines) and child._parent is not None:
                    child.free()
            for child in reversed(list(self._children)):
                child.free()
            for blocks in self.__dict__.pop("_pinned_pool", {}).values():      # idle blocks; a block a caller still
                for ptr in blocks:                                              # holds an array in is freed with that array
                    self.lib.lbl_host_free(self.h, ptr)
            self.check(self.lib.lbl_ctx_destroy(self.h))
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self.check(self.lib.lbl_sync(self.h))

    def download_wait(self):
        self.check(self.lib.lbl_download_wait(self.h))

    def stream(self) -> int:
        s = _P()
        self.check(self.lib.lbl_ctx_stream(self.h, C.byref(s)))
        return s.value or 0

    def device_info(self):
        name = C.create_string_buffer(256)
        ncu = C.c_int()
        hbm = C.c_int64()
        self.check(self.lib.lbl_device_info(self.h, name, 256, C.byref(ncu), C.byref(hbm)))
        return dict(name=name.value.decode(), n_cu=ncu.value, hbm_bytes=hbm.value)

    def chain_accumulate(self, predecessor):
        """This context's accumulate kernels start after those ``predecessor`` has enqueued (lbl_ctx_chain_accumulate);
        None ends the chaining."""
        self.check(self.lib.lbl_ctx_chain_accumulate(self.h, predecessor.h if predecessor is not None else None))

    def set_option(self, key: str, value: int):
        self.check(self.lib.lbl_set_option(self.h, key.encode(), int(value)))
        self.__dict__.setdefault("options", {})[key] = int(value)      # what was set through this object (hosts choose routes by it)

    def option(self, key: str, default: int = 0) -> int:
        return self.__dict__.get("options", {}).get(key, default)

    PROFILE_KINDS = {"line_prep": 0, "xsec_accumulate": 1, "regrid": 2, "layer_sweep": 3, "column_sweep": 4,
                     "allgather": 5}

    def profile_enable(self, kinds=True):
        """kinds: True (all), False (off) or an iterable of PROFILE_KINDS names."""
        if kinds is True:
            mask = 0x3F
        elif not kinds:
            mask = 0
        else:
            mask = 0
            for k in kinds:
                mask |= 1 << self.PROFILE_KINDS[k]
        self.check(self.lib.lbl_profile_enable(self.h, mask))

    def profile_reserve(self, n_events: int):
        self.check(self.lib.lbl_profile_reserve(self.h, int(n_events)))

    def profile_reset(self):
        self.check(self.lib.lbl_profile_reset(self.h))

    def profile_read(self) -> dict:
        """{kernel class: (launches, total_ms)} measured with HIP events on the context stream."""
        out = {}
        for name, kind in self.PROFILE_KINDS.items():
            n = C.c_int64()
            ms = C.c_double()
            self.check(self.lib.lbl_profile_read(self.h, kind, C.byref(n), C.byref(ms)))
            out[name] = (n.value, ms.value)
        return out

    # -- graph capture -------------------------------------------------------------------
    def capture(self, enqueue_fn) -> "Graph":
        """Capture what ``enqueue_fn()`` enqueues into a graph (run it once yourself first)."""
        self.check(self.lib.lbl_capture_begin(self.h))
        err = None
        try:
            enqueue_fn()
        except Exception as e:                 # the capture has to be ended whatever happened inside
            err = e
        h = _P()
        rc = self.lib.lbl_capture_end(self.h, C.byref(h))
        if err is not None:
            if rc == LBL_OK and h:
                self.lib.lbl_graph_destroy(h)
            raise err
        self.check(rc)
        return Graph(self, h)

    # -- page-locked host arrays ---------------------------------------------------------
    def host_array(self, n: int) -> np.ndarray:
        """float64 array of n elements in page-locked host memory (lbl_host_alloc): downloads into it
        and uploads from it run at the link's DMA rate.  The block returns to a per-context pool when
        the array (and every view of it) is gone, and is released with the context."""
        import weakref
        nbytes = max(int(n), 1) * 8
        cls = 1 << max(nbytes - 1, 1).bit_length()             # pool by power-of-two size class
        pool = self.__dict__.setdefault("_pinned_pool", {})
        # the smallest idle block that is large enough, up to four times the need (page-locking a new one costs ~0.2 ms per
        # MB); else a new one of this class
        fit = min((c for c, blocks in pool.items() if cls <= c <= 4 * cls and blocks), default=None)
        if fit is not None:
            cls = fit
            ptr = pool[fit].pop()
        else:
            p = _P()
            self.check(self.lib.lbl_host_alloc(self.h, cls, C.byref(p)))
            ptr = p.value
        raw = (C.c_char * nbytes).from_address(ptr)
        arr = np.frombuffer(raw, dtype=np.float64, count=int(n))
        weakref.finalize(raw, Context._recycle, weakref.ref(self), self.lib, cls, ptr)
        return arr

    @staticmethod
    def _recycle(ctx_ref, lib, cls, ptr):
        """the last array over a block is gone: back to the pool, or freed if its context has closed"""
        ctx = ctx_ref()
        if ctx is not None and getattr(ctx, "h", None):
            ctx.__dict__.setdefault("_pinned_pool", {}).setdefault(cls, []).append(ptr)
        else:
            lib.lbl_host_free(None, ptr)

    # -- objects -------------------------------------------------------------------------
    def buffer(self, n: int, data=None) -> "Buffer":
        b = Buffer(self, n)
        if data is not None:
            b.upload(data)
        return b

    def lines(self, lines: dict) -> "Lines":
        return Lines(self, lines)

    # -- hot path ------------------------------------------------------------------------
    def xsec_accumulate(self, lines: dict, iso: IsoParams, grid: Grid):
        """One-shot host in / host out (lbl_xsec_accumulate). Returns (xsec, regime_counts).
        The C entry point wants nu non-decreasing; an unsorted list is sorted here (stable)."""
        nu = _as_f64(lines["nu"])
        if nu.size > 1 and np.any(np.diff(nu) < 0):
            order = np.argsort(nu, kind="stable")
            lines = {k: np.asarray(lines[k])[order] for k in Lines.ORDER}
        arrs = [_as_f64(lines[k]) for k in Lines.ORDER]
        out = np.empty(int(grid.n_base), dtype=np.float64)
        counts = (C.c_int64 * 3)()
        self.check(self.lib.lbl_xsec_accumulate(self.h, *[_ptr(a) for a in arrs], len(arrs[0]),
                                                C.byref(iso), C.byref(grid), _ptr(out), counts))
        return out, tuple(int(c) for c in counts)

    def xsec_accumulate_dev(self, jobs):
        """jobs: list of (Lines, IsoParams, Grid, Buffer). Asynchronous."""
        n = len(jobs)
        if n == 0:
            return
        L = (_P * n)(*[j[0].h for j in jobs])
        I = (IsoParams * n)(*[j[1] for j in jobs])
        G = (Grid * n)(*[j[2] for j in jobs])
        O = (_P * n)(*[j[3].h for j in jobs])
        self.check(self.lib.lbl_xsec_accumulate_dev(self.h, n, L, I, G, O))

    def schedule_export(self, k: int = 0):
        """(list[n, 2] of (job, tile), tabs[spans, 8], built_on_device) of the k-th most recently used schedule."""
        n = C.c_int64(); nt = C.c_int64(); dev = C.c_int32()
        self.check(self.lib.lbl_schedule_export(self.h, int(k), None, 0, None, 0, C.byref(n), C.byref(nt), C.byref(dev)))
        lst = np.empty((n.value, 2), np.int32); tabs = np.empty((max(nt.value, 0) // 8, 8), np.int32)
        self.check(self.lib.lbl_schedule_export(self.h, int(k), _ptr(lst), lst.size, _ptr(tabs), tabs.size, C.byref(n),
                                                C.byref(nt), C.byref(dev)))
        return lst, tabs, bool(dev.value)

    def last_regime_counts(self, n_jobs: int):
        counts = (C.c_int64 * (3 * n_jobs))()
        self.check(self.lib.lbl_last_regime_counts(self.h, n_jobs, counts))
        return np.array(counts, dtype=np.int64).reshape(n_jobs, 3)

    def line_quantities(self, lines: "Lines", iso: IsoParams, grid: Grid):
        n = lines.n
        index = np.empty(n, np.int64); lhw = np.empty(n); ghw = np.empty(n); inten = np.empty(n)
        regime = np.empty(n, np.int32)
        self.check(self.lib.lbl_line_quantities(self.h, lines.h, C.byref(iso), C.byref(grid), _ptr(index), _ptr(lhw),
                                                _ptr(ghw), _ptr(inten), _ptr(regime)))
        return dict(index=index, lhw=lhw, ghw=ghw, intensity=inten, regime=regime)

    def layer_sweep_dev(self, xsec, iso_mol, conc, P, T, depth, range_min, range_max, n,
                        I_in=None, surface_T=0.0, abs_coef=None, trans=None, I_out=None, first=0, count=0):
        n_iso = len(xsec)
        X = (_P * max(n_iso, 1))(*[b.h for b in xsec])
        M = (C.c_int32 * max(n_iso, 1))(*[int(m) for m in iso_mol])
        cc = (C.c_double * max(len(conc), 1))(*[float(c) for c in conc])
        self.check(self.lib.lbl_layer_sweep_dev(
            self.h, n_iso, X, M, len(conc), cc, float(P), float(T), float(depth), float(range_min),
            float(range_max), int(n), int(first), int(count), I_in.h if I_in is not None else None,
            float(surface_T), abs_coef.h if abs_coef is not None else None, trans.h if trans is not None else None,
            I_out.h if I_out is not None else None))

    def layer_step_dev(self, lines, iso, grid: Grid, xsec, iso_mol, conc, depth, I_in=None, surface_T=0.0,
                       abs_coef=None, trans=None, I_out=None):
        """Accumulate + sweep of one layer in one launch sequence (lbl_layer_step_dev): ``lines`` /
        ``iso`` / ``xsec`` are per line list (a single Lines / IsoParams / Buffer is taken as one),
        ``iso_mol`` maps line list -> molecule, ``conc`` is per molecule."""
        if isinstance(lines, Lines):
            lines, iso, xsec = [lines], [iso], [xsec]
        if np.isscalar(conc):
            conc = [conc]
        if iso_mol is None:
            iso_mol = list(range(len(lines)))
        n = len(lines)
        L = (_P * n)(*[l.h for l in lines])
        I = (IsoParams * n)(*iso)
        X = (_P * n)(*[b.h for b in xsec])
        M = (C.c_int32 * n)(*[int(m) for m in iso_mol])
        cc = (C.c_double * max(len(conc), 1))(*[float(c) for c in conc])
        h = lambda b: b.h if b is not None else None
        self.check(self.lib.lbl_layer_step_dev(self.h, n, L, I, C.byref(grid), X, M, len(conc), cc, float(depth), h(I_in),
                                               float(surface_T), h(abs_coef), h(trans), h(I_out)))

    def layer_merged_step_dev(self, lines, iso, grid: Grid, iso_mol, conc, depth, I_in=None, surface_T=0.0,
                              abs_coef=None, trans=None, I_out=None):
        """One layer through ONE accumulate job over its merged, factor-weighted line lists with the sweep in the
        output stage (lbl_layer_merged_step_dev): the absorption coefficient is accumulated directly, no
        per-line-list cross section is written.  Arguments as layer_step_dev without ``xsec``."""
        if isinstance(lines, Lines):
            lines, iso = [lines], [iso]
        if np.isscalar(conc):
            conc = [conc]
        if iso_mol is None:
            iso_mol = list(range(len(lines)))
        n = len(lines)
        L = (_P * n)(*[l.h for l in lines])
        I = (IsoParams * n)(*iso)
        M = (C.c_int32 * n)(*[int(m) for m in iso_mol])
        cc = (C.c_double * max(len(conc), 1))(*[float(c) for c in conc])
        h = lambda b: b.h if b is not None else None
        self.check(self.lib.lbl_layer_merged_step_dev(self.h, n, L, I, C.byref(grid), M, len(conc), cc, float(depth),
                                                      h(I_in), float(surface_T), h(abs_coef), h(trans), h(I_out)))

    def layers_merged_accumulate_dev(self, layers):
        """layers: list of dict(lines=[Lines], iso=[IsoParams], grid=Grid, iso_mol=[int], conc=[float], abs_coef=Buffer):
        every layer's absorption coefficient through one merged accumulate job per layer, all in one launch sequence
        (lbl_layers_merged_accumulate_dev)."""
        nl = len(layers)
        if nl == 0:
            return
        arr = lambda typ, vals: (typ * max(len(vals), 1))(*vals)
        ls = [l for L in layers for l in L["lines"]]
        self.check(self.lib.lbl_layers_merged_accumulate_dev(
            self.h, nl, arr(C.c_int32, [len(L["lines"]) for L in layers]), arr(_P, [l.h for l in ls]),
            arr(IsoParams, [i for L in layers for i in L["iso"]]), arr(Grid, [L["grid"] for L in layers]),
            arr(C.c_int32, [int(m) for L in layers for m in L["iso_mol"]]), arr(C.c_int32, [len(L["conc"]) for L in layers]),
            arr(C.c_double, [float(c) for L in layers for c in L["conc"]]), arr(_P, [L["abs_coef"].h for L in layers])))

    def column(self, layers):
        """Resident column (lbl_column_create): ``layers`` as for layers_merged_accumulate_dev, each with ``depth``."""
        return Column(self, layers)

    def column_fold_dev(self, abs_coef, layer_T, depth, range_min, range_max, n, I_out, I_in=None, surface_T=0.0,
                        trans=None, first=0, count=0):
        """Column step from the layers' absorption coefficients, bottom to top (lbl_column_fold_dev); ``trans``: None or
        a list (entries may be None) of buffers that receive the layers' transmittances."""
        nl = len(abs_coef)
        arr = lambda typ, vals: (typ * max(len(vals), 1))(*vals)
        hb = lambda b: b.h if b is not None else None
        self.check(self.lib.lbl_column_fold_dev(
            self.h, nl, arr(_P, [b.h for b in abs_coef]), arr(C.c_double, [float(t) for t in layer_T]),
            arr(C.c_double, [float(d) for d in depth]), float(range_min), float(range_max), int(n), int(first), int(count),
            hb(I_in), float(surface_T), arr(_P, [hb(b) for b in trans]) if trans is not None else None, I_out.h))

    def gather_compact_dev(self, gathered, slot, bounds, out):
        """padded all-gather result (slot r = rank r's shard) -> grid order (lbl_gather_compact_dev)."""
        w = len(bounds)
        F = (C.c_int64 * w)(*[int(f) for f, _ in bounds])
        K = (C.c_int64 * w)(*[int(c) for _, c in bounds])
        self.check(self.lib.lbl_gather_compact_dev(self.h, gathered.h, w, int(slot), F, K, out.h))

    def column_step_dev(self, layers, range_min, range_max, n, I_out, I_in=None, surface_T=0.0, first=0, count=0):
        """layers: bottom to top, each dict(xsec=[Buffer], iso_mol=[int], conc=[float], P, T, depth,
        trans=Buffer|None, abs_coef=Buffer|None)."""
        nl = len(layers)
        xs = [b for L in layers for b in L["xsec"]]
        im = [int(m) for L in layers for m in L["iso_mol"]]
        cc = [float(c) for L in layers for c in L["conc"]]
        arr = lambda typ, vals: (typ * max(len(vals), 1))(*vals)
        n_iso = arr(C.c_int32, [len(L["xsec"]) for L in layers])
        n_mol = arr(C.c_int32, [len(L["conc"]) for L in layers])
        hb = lambda b: b.h if b is not None else None
        self.check(self.lib.lbl_column_step_dev(
            self.h, nl, n_iso, arr(_P, [b.h for b in xs]), arr(C.c_int32, im), n_mol, arr(C.c_double, cc),
            arr(C.c_double, [float(L["P"]) for L in layers]), arr(C.c_double, [float(L["T"]) for L in layers]),
            arr(C.c_double, [float(L["depth"]) for L in layers]), float(range_min), float(range_max), int(n),
            int(first), int(count), hb(I_in), float(surface_T),
            arr(_P, [hb(L.get("abs_coef")) for L in layers]), arr(_P, [hb(L.get("trans")) for L in layers]), I_out.h))

    def column_sweep_dev(self, trans, layer_T, range_min, range_max, n, I_out, I_in=None, surface_T=0.0,
                         first=0, count=0):
        nl = len(trans)
        Tb = (_P * max(nl, 1))(*[b.h for b in trans])
        TT = (C.c_double * max(nl, 1))(*[float(t) for t in layer_T])
        self.check(self.lib.lbl_column_sweep_dev(self.h, nl, Tb, TT, float(range_min), float(range_max), int(n),
                                                 int(first), int(count),
                                                 I_in.h if I_in is not None else None, float(surface_T), I_out.h))

    def sum_dev(self, bufs, n, out):
        """out = zeros + bufs[0] + bufs[1] + ... in list order (pyradClasses.py:566-571, 684-689).  A list longer than the
        library takes at once is chained with the partial sum first: 0 + partial is exact, the order of additions the same."""
        cap = limit("arrays_per_sum")
        bufs = list(bufs)
        first = True
        while first or bufs:
            take = bufs[:cap] if first else [out] + bufs[:cap - 1]
            bufs = bufs[cap:] if first else bufs[cap - 1:]
            B = (_P * max(len(take), 1))(*[b.h for b in take])
            self.check(self.lib.lbl_sum_dev(self.h, len(take), B, int(n), out.h))
            first = False

    def optical_dev(self, trans, n, kind, out):
        self.check(self.lib.lbl_optical_dev(self.h, trans.h, int(n), int(kind), out.h))

    def planck_dev(self, range_min, range_max, n, T, out):
        self.check(self.lib.lbl_planck_dev(self.h, float(range_min), float(range_max), int(n), float(T), out.h))

    def band_integral(self, spectrum, n, unit_angle, res) -> float:
        r = C.c_double()
        self.check(self.lib.lbl_band_integral(self.h, spectrum.h, int(n), float(unit_angle), float(res), C.byref(r)))
        return r.value

    def line_survey_dev(self, lines, grid, out):
        self.check(self.lib.lbl_line_survey_dev(self.h, lines.h, C.byref(grid), out.h))


class Column:
    """lbl_column: the argument blocks of a column's merged accumulate jobs and of its fold, kept on the C side between
    calls.  ``layers``: list of dict(lines=[Lines], iso=[IsoParams], grid=Grid, iso_mol=[int], conc=[float], depth=float,
    abs_coef=Buffer).  The handle owns nothing on the device; the Python object keeps the line lists and buffers it was
    given alive."""

    def __init__(self, ctx: "Context", layers):
        self.ctx = ctx
        self.n_layers = len(layers)
        arr = lambda typ, vals: (typ * max(len(vals), 1))(*vals)
        ls = [l for L in layers for l in L["lines"]]
        self._keep = [(list(L["lines"]), L["abs_coef"]) for L in layers]
        h = _P()
        ctx.check(ctx.lib.lbl_column_create(
            ctx.h, self.n_layers, arr(C.c_int32, [len(L["lines"]) for L in layers]), arr(_P, [l.h for l in ls]),
            arr(IsoParams, [i for L in layers for i in L["iso"]]), arr(Grid, [L["grid"] for L in layers]),
            arr(C.c_int32, [int(m) for L in layers for m in L["iso_mol"]]), arr(C.c_int32, [len(L["conc"]) for L in layers]),
            arr(C.c_double, [float(c) for L in layers for c in L["conc"]]), arr(C.c_double, [float(L["depth"]) for L in layers]),
            arr(_P, [L["abs_coef"].h for L in layers]), C.byref(h)))
        self.h = h
        self._due = (C.c_uint8 * self.n_layers)()
        ctx._children.append(self)

    def set_layer(self, l, lines, iso, grid, conc, depth, abs_coef):
        arr = lambda typ, vals: (typ * max(len(vals), 1))(*vals)
        self.ctx.check(self.ctx.lib.lbl_column_set_layer(self.h, int(l), arr(_P, [x.h for x in lines]), arr(IsoParams, list(iso)),
                                                         C.byref(grid), arr(C.c_double, [float(c) for c in conc]), float(depth),
                                                         abs_coef.h))
        self._keep[l] = (list(lines), abs_coef)

    def transmission(self, due, I_out, host=None, I_in=None, surface_T=0.0, pieces=1):
        """Enqueue: the merged accumulate jobs of the layers with due[l] true (None: all), the fold in ``pieces`` pieces, each
        piece's part of I_out on its way into ``host`` (Context.host_array); Context.download_wait() before reading it."""
        flags = None
        if due is not None:
            flags = self._due
            for l, d in enumerate(due):
                flags[l] = 1 if d else 0
        hp = None
        if host is not None:
            if not (host.dtype == np.float64 and host.flags["C_CONTIGUOUS"]):
                raise ValueError("the host array must be contiguous float64")
            hp = _ptr(host)
        self.ctx.check(self.ctx.lib.lbl_column_transmission(self.h, flags, I_in.h if I_in is not None else None, float(surface_T),
                                                            I_out.h, hp, int(pieces)))

    def free(self):
        if self.h:
            self.ctx.check(self.ctx.lib.lbl_column_destroy(self.h))
            self.h = None
            self._keep = []
            if self in self.ctx._children:
                self.ctx._children.remove(self)


class Buffer:
    """Device float64 array (lbl_buffer)."""

    def __init__(self, ctx: Context, n: int):
        self.ctx = ctx
        self.n = int(n)
        h = _P()
        ctx.check(ctx.lib.lbl_buffer_create(ctx.h, self.n, C.byref(h)))
        self.h = h
        ctx._children.append(self)

    def stage_from_dev(self, src: "Buffer", src_offset: int, n: int, dst_offset: int = 0):
        """self[dst_offset : dst_offset + n] = src[src_offset : src_offset + n], device to device, async on this
        buffer's context stream (lbl_gather_stage_dev: staging shards into an all-gather batch buffer)."""
        self.ctx.check(self.ctx.lib.lbl_gather_stage_dev(self.h, int(dst_offset), src.h, int(src_offset), int(n)))
        return self

    def upload(self, data, offset: int = 0):
        a = _as_f64(data)
        self.ctx.check(self.ctx.lib.lbl_buffer_upload(self.h, _ptr(a), a.size, int(offset)))
        return self

    def download(self, n: int | None = None, offset: int = 0, pinned: bool = False) -> np.ndarray:
        """Host copy of n elements.  ``pinned``: into page-locked memory of the context's pool (DMA
        rate; the array must not outlive the context)."""
        n = self.n - offset if n is None else int(n)
        out = self.ctx.host_array(n) if pinned else np.empty(n, dtype=np.float64)
        self.ctx.check(self.ctx.lib.lbl_buffer_download(self.h, _ptr(out), n, int(offset)))
        return out

    def download_async(self, out: np.ndarray, n: int, offset: int = 0, out_offset: int = 0):
        """out[out_offset : out_offset + n] = self[offset : offset + n] without waiting (lbl_buffer_download_async: behind
        the work enqueued so far, beside what is enqueued next); ``out`` from Context.host_array; Context.download_wait()
        before reading it."""
        if not (out.dtype == np.float64 and out.flags["C_CONTIGUOUS"] and 0 <= out_offset and out_offset + int(n) <= out.size):
            raise ValueError("download_async needs a contiguous float64 array that holds the range")
        self.ctx.check(self.ctx.lib.lbl_buffer_download_async(self.h, _ptr(out[out_offset:]), int(n), int(offset)))
        return out

    def fill(self, value: float):
        self.ctx.check(self.ctx.lib.lbl_buffer_fill(self.h, float(value)))
        return self

    def devptr(self) -> int:
        p = _P()
        self.ctx.check(self.ctx.lib.lbl_buffer_devptr(self.h, C.byref(p)))
        return p.value or 0

    def free(self):
        if self.h:
            self.ctx.check(self.ctx.lib.lbl_buffer_destroy(self.h))
            self.h = None
            if self in self.ctx._children:
                self.ctx._children.remove(self)

    def __del__(self):
        try:
            if self.ctx.h:
                self.free()
        except Exception:
            pass


class Graph:
    """A captured launch sequence (lbl_graph)."""

    def __init__(self, ctx: Context, h):
        self.ctx, self.h = ctx, h
        ctx._children.insert(0, self)

    def launch(self):
        self.ctx.check(self.ctx.lib.lbl_graph_launch(self.h))

    def free(self):
        if self.h:
            self.ctx.check(self.ctx.lib.lbl_graph_destroy(self.h))
            self.h = None
            if self in self.ctx._children:
                self.ctx._children.remove(self)


class Lines:
    """Device-resident HITRAN line list (lbl_lines).  Sorts by nu on the host if needed."""
    ORDER = ("nu", "sw", "elower", "gamma_air", "gamma_self", "n_air", "delta_air")

    def __init__(self, ctx: Context, lines: dict):
        self.ctx = ctx
        nu = _as_f64(lines["nu"])
        if nu.size > 1 and np.any(np.diff(nu) < 0):
            order = np.argsort(nu, kind="stable")
            lines = {k: np.asarray(lines[k])[order] for k in self.ORDER}
        arrs = [_as_f64(lines[k]) for k in self.ORDER]
        self.n = int(arrs[0].size)
        self._parent = None
        h = _P()
        ctx.check(ctx.lib.lbl_lines_create(ctx.h, *[_ptr(a) for a in arrs], self.n, C.byref(h)))
        self.h = h
        ctx._children.append(self)

    def has_views(self) -> bool:
        return any(isinstance(c, Lines) and c._parent is self and c.h for c in self.ctx._children)

    def view(self, first: int, count: int) -> "Lines":
        """``count`` consecutive lines from line ``first`` on (sorted order), sharing this list's device arrays
        (lbl_lines_view).  Free the views before the list."""
        v = Lines.__new__(Lines)
        v.ctx, v.n = self.ctx, int(count)
        v._parent = self                       # the view keeps its list alive (a collected list could not free itself)
        h = _P()
        self.ctx.check(self.ctx.lib.lbl_lines_view(self.h, int(first), int(count), C.byref(h)))
        v.h = h
        self.ctx._children.append(v)
        return v

    def free(self):
        if self.h:
            self.ctx.check(self.ctx.lib.lbl_lines_destroy(self.h))
            self.h = None
            self._parent = None
            if self in self.ctx._children:
                self.ctx._children.remove(self)

    def __del__(self):
        try:
            if self.ctx.h:
                self.free()
        except Exception:
            pass


class Comm:
    """RCCL communicator over the context's device (lbl_comm), one rank per process."""

    def __init__(self, ctx: Context, unique_id: bytes, world_size: int, rank: int):
        self.ctx = ctx
        self.world_size, self.rank = int(world_size), int(rank)
        h = _P()
        buf = C.create_string_buffer(bytes(unique_id), UNIQUE_ID_BYTES)
        ctx.check(ctx.lib.lbl_comm_create(ctx.h, buf, self.world_size, self.rank, C.byref(h)))
        self.h = h
        ctx._children.insert(0, self)

    @staticmethod
    def unique_id() -> bytes:
        lib = load()
        buf = C.create_string_buffer(UNIQUE_ID_BYTES)
        rc = lib.lbl_comm_unique_id(buf)
        if rc != LBL_OK:
            raise LblError(rc, (lib.lbl_last_error(None) or b"").decode())
        return buf.raw

    def allgather_dev(self, send: Buffer, send_offset: int, count: int, recv: Buffer, overlap_slot=None):
        """In-stream all-gather, or (overlap_slot 0..6) one the context stream does not wait for.  "The
        context" is the one that owns send and recv (any context of the communicator's device): errors are
        reported there."""
        owner = send.ctx
        if overlap_slot is None:
            owner.check(owner.lib.lbl_allgather_dev(self.h, send.h, int(send_offset), int(count), recv.h))
        else:
            owner.check(owner.lib.lbl_allgather_overlap_dev(self.h, send.h, int(send_offset), int(count), recv.h,
                                                            int(overlap_slot)))

    def fence_dev(self, slot: int = -1):
        """Context stream waits (no host sync) for the collective issued with ``slot`` (-1: all)."""
        self.ctx.check(self.ctx.lib.lbl_comm_fence_dev(self.h, int(slot)))

    def free(self):
        if self.h:
            self.ctx.check(self.ctx.lib.lbl_comm_destroy(self.h))
            self.h = None
            if self in self.ctx._children:
                self.ctx._children.remove(self)


def device_count() -> int:
    lib = load()
    n = C.c_int()
    rc = lib.lbl_device_count(C.byref(n))
    return n.value if rc == LBL_OK else 0
