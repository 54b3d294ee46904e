"""Line-list sources: where ``Isotope.getData`` gets its HITRAN lines, partition sums and
molecule parameters from (the role of pyradUtilities.gatherData / getQData / readMolParams,
ut:173-197, 421-477).  No network code: the reference's HTTP download is out of scope.

Two sources:
  * ``MemorySource``   — in-memory line lists (synthetic or user supplied);
  * ``PyradDataDir``   — reader of PyRad's own on-disk cache ``data/<globalIso>/<seg>.pyr``,
                         ``q<iso>.txt`` and ``params.pyr``, so an
                         existing PyRad ``data/`` tree is usable as is.
Lines are handed over as a structure of arrays sorted by wavenumber.

Measured cross sections ("xsc" molecules) come from ``XscDir``, the
reader of PyRad's ``data/xsc/<molecule>/<file>.txt`` cache (ut:611-715); ``set_xsc_source``
installs one and fills ``EXOTIC_IDS`` (the table cls:1024 builds at import).
"""
from __future__ import annotations

import os
import re

import numpy as np

FIELDS = ("nu", "sw", "a", "elower", "gamma_air", "gamma_self", "delta_air", "n_air")
NULL_TAG = '#/null/#'          # ut:842 sentinel of a failed download


def _empty():
    return {f: np.zeros(0, dtype=np.float64) for f in FIELDS}


class MemorySource:
    """{global_iso: lines / q table / params} held in memory."""

    def __init__(self):
        self._lines, self._q, self._params = {}, {}, {}

    def register(self, global_iso: int, lines: dict, q: dict, params: list):
        order = np.argsort(np.asarray(lines["nu"]), kind="stable")
        full = {f: np.ascontiguousarray(np.asarray(lines[f], dtype=np.float64)[order]) for f in FIELDS if f in lines}
        if "a" not in full:
            full["a"] = np.zeros_like(full["nu"])
        # duplicated wavenumbers collapse once, here (rows with equal nu are in or out of any window together, so
        # collapsing the whole list first gives what collapsing each selection gave): a window is then a SLICE
        full = _Master({f: np.ascontiguousarray(v) for f, v in _dedupe_last_wins(full).items()})
        self._lines[global_iso] = full
        _MASTERS[id(full["nu"])] = full                   # lets a consumer recognise a window as a slice of this list
        self._q[global_iso] = q
        self._params[global_iso] = list(params)

    def readMolParams(self, global_iso):
        return list(self._params[global_iso])

    def getQData(self, global_iso):
        return self._q[global_iso]

    def gatherData(self, global_iso, range_min, range_max):
        lines = self._lines[global_iso]
        nu = lines["nu"]                                  # sorted, duplicates collapsed (register)
        first = int(np.searchsorted(nu, range_min, "right"))          # strict range_min < nu < range_max, ut:437-438
        end = max(int(np.searchsorted(nu, range_max, "left")), first)
        out = {f: v[first:end] for f, v in lines.items()}              # views: no copy
        _note_window(out, lines, first, end - first)
        return out


# line lists registered with a MemorySource, by id of their wavenumber array: a selection handed out by gatherData
# is a slice of one of them, which the device side turns into a VIEW of the list's one resident copy (lbl_lines_view)
class _Master(dict):
    """a registered line list (a dict that can be weakly referenced: the registry must not keep dropped sources alive)"""
    __slots__ = ("__weakref__",)


import weakref as _weakref
_MASTERS = _weakref.WeakValueDictionary()


# windows handed out by gatherData, by id of their wavenumber view: (weak reference to that view, master, first, count).  The
# general test below reads 14 __array_interface__ dicts per window (12 us; 1.1 ms for the 90 windows of a re-windowed 30-layer
# column, all of it before the first kernel is enqueued); a window that came from gatherData is recognised in under a microsecond.
_WINDOWS = {}


def _note_window(views, master, first, count):
    nu_view = views["nu"]
    key = id(nu_view)
    # (the sibling views themselves, held while the wavenumber view lives: the test below is object identity - an id can be
    #  reused by another view of the same list once the original has been dropped)
    _WINDOWS[key] = (_weakref.ref(nu_view, lambda _r, k=key: _WINDOWS.pop(k, None)), _weakref.ref(master), int(first), int(count),
                     {f: v for f, v in views.items() if f != "nu"})


def master_slice(lines: dict, fields):
    """(master dict, first, count) if every array of ``lines`` named in ``fields`` is the same contiguous slice of one
    registered line list; else None."""
    nu = lines.get("nu")
    hit = _WINDOWS.get(id(nu))
    if hit is not None and hit[0]() is nu:
        master, first, n = hit[1](), hit[2], hit[3]
        if master is not None and _MASTERS.get(id(master["nu"])) is master:
            sib = hit[4]
            for f in fields:
                # (the very view objects gatherData made together; an array somebody swapped in fails here)
                if f != "nu" and lines.get(f) is not sib.get(f, master):
                    break
            else:
                return master, first, n
    base = getattr(nu, "base", None)
    if base is None or nu.ndim != 1 or nu.dtype != np.float64 or not nu.flags.c_contiguous:
        return None
    master = _MASTERS.get(id(base))
    if master is None or master["nu"] is not base:
        return None
    first = (nu.__array_interface__["data"][0] - base.__array_interface__["data"][0]) // 8
    n = nu.size
    for f in fields:
        a, m = lines.get(f), master.get(f)
        if a is None or m is None or a.base is not m or a.size != n or a.dtype != np.float64 or not a.flags.c_contiguous:
            return None
        if (a.__array_interface__["data"][0] - m.__array_interface__["data"][0]) // 8 != first:
            return None
    return master, int(first), int(n)


def _dedupe_last_wins(lines):
    """The reference keys its line dict by wavenumber, so a duplicated nu keeps the LAST row's
    values at the FIRST row's position (ut:447).  Input is sorted by nu (stable)."""
    nu = lines["nu"]
    if nu.size < 2 or not np.any(np.diff(nu) == 0):
        return lines
    first = np.ones(nu.size, dtype=bool)
    first[1:] = np.diff(nu) != 0
    last = np.ones(nu.size, dtype=bool)
    last[:-1] = np.diff(nu) != 0
    return {f: v[last] if f != "nu" else v[first] for f, v in lines.items()}


class PyradDataDir:
    """Reader of PyRad's on-disk inputs under ``<root>`` (the reference's ``./data``).

    * ``<root>/<iso>/<seg>.pyr``: CSV rows ``molec_id,local_iso_id,nu,sw,a,elower,gamma_air,
      gamma_self,delta_air,n_air`` (request_params of ut:369-374; column indices ut:422-430),
      one file per 100 cm^-1 segment named by its lower edge (ut:175-184); leading ``#`` lines
      are skipped and a file whose first line carries NULL_TAG is empty (ut:96-101).
    * ``<root>/<iso>/q<iso>.txt``: whitespace ``T Q`` rows (ut:451-461).
    * ``<root>/<iso>/params.pyr``: comment lines then one CSV row
      ``globalIso,shortName,molNum,isoN,abundance,Q296,gj,molMass`` (ut:464-477).
    A missing segment file is an error here (the reference would try to download it).
    """

    def __init__(self, root: str, cache: bool = True):
        self.root = root
        self.cache = cache               # keep parsed segments (False: read and parse on every call, as the reference does)
        self._segments = {}              # (global_iso, segment) -> parsed file
        self._masters = {}               # global_iso -> the sorted, duplicate-free list of its kept segments

    @staticmethod
    def _rows(path):
        """openReturnLines (ut:90-101)."""
        if not os.path.isfile(path):
            return None
        with open(path) as f:
            rows = f.readlines()
        if not rows or NULL_TAG in rows[0]:
            return []
        while rows[0][0] == '#' and len(rows) > 1:
            rows.pop(0)
        return rows

    def readMolParams(self, global_iso):
        rows = self._rows('%s/%s/params.pyr' % (self.root, global_iso))
        if not rows:
            raise FileNotFoundError('%s/%s/params.pyr' % (self.root, global_iso))
        c = rows[0].split(',')
        return [int(c[0]), c[1], int(c[2]), int(c[3]), float(c[4]), float(c[5]), int(c[6]), float(c[7])]

    def getQData(self, global_iso):
        """readQFile (ut:451-461): every row must hold ``T Q``; a blank row raises IndexError there too."""
        q = {}
        with open('%s/%s/q%s.txt' % (self.root, global_iso, global_iso)) as f:
            for row in f.readlines():
                cell = row.split()
                q[int(cell[0])] = float(cell[1])
        return q

    @staticmethod
    def segments(range_min, range_max):
        """Segment lower edges visited by gatherData (ut:175-180)."""
        out = []
        segment = int(range_min / 100) * 100
        while segment < range_max:
            out.append(segment)
            segment += 100
        return out

    def gatherData(self, global_iso, range_min, range_max):
        """gatherData (ut:173-189) over readHitranOnlineFile (ut:421-448).  Parsed segments are kept (the reference re-reads
        and re-parses the files on every range or pressure change: a 30-layer column asks for the same ten files ninety
        times); while every kept segment of the isotopologue is well formed the window is a SLICE of one sorted,
        duplicate-free list per isotopologue - what lets the device keep one resident copy and hand out views - and the
        reference's row-by-row semantics are reproduced by construction (see _window_of_master); anything unusual (a row
        filed in the wrong segment, an unparsable column) takes the row-by-row path below, which is the reference's."""
        segs = self.segments(range_min, range_max)
        parsed = [self._segment(global_iso, seg) for seg in segs]
        if self.cache and all(p["clean"] for p in parsed):
            return self._window_of_master(global_iso, range_min, range_max)
        info = {}            # nu -> row, insertion ordered, later rows override (ut:447, dict.update ut:187)
        for seg, p in zip(segs, parsed):
            for row in p["rows"]:
                cell = row.split(',')                          # a malformed row raises IndexError / ValueError, as ut:434-446
                nu = float(cell[2])
                if range_min < nu and nu < range_max:          # ut:437-438
                    int(cell[1])                               # the reference parses the isotopologue column too (ut:439)
                    info[nu] = (float(cell[3]), float(cell[4]), float(cell[5]), float(cell[6]), float(cell[7]),
                                float(cell[8]), float(cell[9]))
        if not info:
            return _empty()
        nu = np.fromiter(info.keys(), dtype=np.float64, count=len(info))
        vals = np.array(list(info.values()), dtype=np.float64).reshape(len(info), 7)
        out = {"nu": nu, "sw": vals[:, 0], "a": vals[:, 1], "elower": vals[:, 2], "gamma_air": vals[:, 3],
               "gamma_self": vals[:, 4], "delta_air": vals[:, 5], "n_air": vals[:, 6]}
        order = np.argsort(nu, kind="stable")
        return {k: np.ascontiguousarray(v[order]) for k, v in out.items()}

    def _segment(self, global_iso, segment):
        """One segment file, read once (again if its size or modification time changes): its rows as text, and - when every
        row parses and lies inside the segment's own 100 cm^-1 - as arrays ("clean")."""
        path = '%s/%s/%s.pyr' % (self.root, global_iso, segment)
        try:
            st = os.stat(path)
            stamp = (st.st_mtime_ns, st.st_size)
        except OSError:
            stamp = None
        key = (global_iso, segment)
        hit = self._segments.get(key) if self.cache else None
        if hit is not None and hit["stamp"] == stamp:
            return hit
        rows = self._rows(path)
        if rows is None:
            raise FileNotFoundError("%s (PyRad would download it; this build has no network code)" % path)
        p = {"stamp": stamp, "rows": rows, "clean": False}
        try:
            cells = [row.split(',') for row in rows]
            nu = np.array([float(c[2]) for c in cells], dtype=np.float64)
            for c in cells:
                int(c[1])
            vals = np.array([[float(c[k]) for k in range(3, 10)] for c in cells], dtype=np.float64).reshape(len(cells), 7)
            p.update(nu=nu, vals=vals, clean=bool(np.all((nu >= segment) & (nu < segment + 100))) if nu.size else True)
        except (IndexError, ValueError):
            pass                                               # the row-by-row path raises where (and only if) the reference would
        if self.cache:
            self._segments[key] = p
            self._masters.pop(global_iso, None)                # the isotopologue's list is rebuilt from its kept segments
        return p

    def _window_of_master(self, global_iso, range_min, range_max):
        """All kept, well-formed segments of the isotopologue as one list in the reference's visiting order (segments
        ascending, rows in file order), duplicated wavenumbers collapsed last-wins (ut:447), sorted; a window is a slice.
        Equal to the row-by-row result: a well-formed row with range_min < nu < range_max lies in a segment the window
        visits, rows of equal nu are in or out of a window together, and among them the last in visiting order wins
        either way."""
        m = self._masters.get(global_iso)
        if m is None:
            keys = sorted(k for k in self._segments if k[0] == global_iso and self._segments[k]["clean"])
            nu = np.concatenate([self._segments[k]["nu"] for k in keys]) if keys else np.zeros(0)
            vals = np.concatenate([self._segments[k]["vals"] for k in keys]) if keys else np.zeros((0, 7))
            order = np.argsort(nu, kind="stable")
            full = {"nu": nu[order], "sw": vals[order, 0], "a": vals[order, 1], "elower": vals[order, 2],
                    "gamma_air": vals[order, 3], "gamma_self": vals[order, 4], "delta_air": vals[order, 5],
                    "n_air": vals[order, 6]}
            m = _Master({f: np.ascontiguousarray(v) for f, v in _dedupe_last_wins(full).items()})
            self._masters[global_iso] = m
            _MASTERS[id(m["nu"])] = m
        nu = m["nu"]
        first = int(np.searchsorted(nu, range_min, "right"))
        end = max(int(np.searchsorted(nu, range_max, "left")), first)
        return {f: v[first:end] for f, v in m.items()}

    # -- writer, so tests and users can materialise a tree in PyRad's format ----------------
    @staticmethod
    def write_tree(root, global_iso, lines, q, params, mol_id=0, local_iso=1):
        d = '%s/%s' % (root, global_iso)
        os.makedirs(d, exist_ok=True)
        nu = np.asarray(lines["nu"])
        if nu.size:
            for seg in range(int(nu.min() / 100) * 100, int(nu.max() / 100) * 100 + 100, 100):
                m = (nu >= seg) & (nu < seg + 100)
                with open('%s/%s.pyr' % (d, seg), 'w') as f:
                    for i in np.nonzero(m)[0]:
                        f.write('%d,%d,%r,%r,%r,%r,%r,%r,%r,%r\n' % (
                            mol_id, local_iso, float(nu[i]), float(lines["sw"][i]), float(lines["a"][i]),
                            float(lines["elower"][i]), float(lines["gamma_air"][i]), float(lines["gamma_self"][i]),
                            float(lines["delta_air"][i]), float(lines["n_air"][i])))
        with open('%s/q%s.txt' % (d, global_iso), 'w') as f:
            for T in sorted(q):
                f.write('%d %r\n' % (T, float(q[T])))
        with open('%s/params.pyr' % d, 'w') as f:
            f.write("#\t#\t#\n# Molecule params for pyrad\n#\t#\t#\n")
            f.write(','.join(str(x) for x in params) + '\n')


_source = None


def set_source(source):
    global _source
    _source = source
    return source


def get_source():
    if _source is None:
        raise RuntimeError("no line-list source: call pyrad_amd.data.set_source(PyradDataDir('data')) "
                           "or set_source(MemorySource()) first (there is no HITRAN download in this build)")
    return _source


def synthetic_source(species_lines: dict):
    """MemorySource filled from {species name: lines SoA} using pyrad_amd.synthetic tables."""
    from . import synthetic
    src = MemorySource()
    for species, lines in species_lines.items():
        sp = synthetic.SPECIES[species]
        src.register(sp["global_iso"], lines, synthetic.q_table(species), synthetic.mol_params(species))
    return src


# ------------------------------------------------------------------------------------------
# measured cross-section ("xsc") files: data/xsc/<molecule>/<name>.txt  (ut:611-715)
# ------------------------------------------------------------------------------------------
# Field -> pattern searched in the file name without its extension (ut:619-625).  The name is
# "<mol>_<T>K-<P>Torr_<min>-<max>_<res>_<broadener>_<id>_<id>" (writeXscFile, ut:538).
_XSC_NAME_FIELDS = {
    'RANGE': re.compile(r'(?<=_)[0-9.]*-[0-9.]*(?=_)'),
    'MOLECULE_SHORT_NAME': re.compile(r'^[A-Za-z0-9]*'),
    'TEMP': re.compile(r'[0-9.]*(?=K)'),
    'PRESSURE': re.compile(r'[0-9.]*(?=Torr)'),
    'RES': re.compile(r'(?<=_)[0-9]{1,}.[0-9]{1,}(?=_)'),
    'ID': re.compile(r'(?<=_)[0-9]*_[0-9]*$'),
    'BROADENER': re.compile(r'(?<=_)[A-Za-z0-9]*(?=_[0-9]*_[0-9]*$)'),
}
TORR_PER_MBAR = 0.75006          # cls:481


def parseXscFileName(file):
    """ut:611-641: the properties PyRad encodes in an xsc file name.  A field whose pattern is
    absent is False (BROADENER: ''); a name without the trailing ``_<n>_<n>`` id raises
    AttributeError, as the reference's ``False.replace`` does."""
    stem = re.sub('.txt', '', file)          # sic: '.' is a wildcard there too (ut:612)
    found = {}
    for field, pattern in _XSC_NAME_FIELDS.items():
        m = pattern.search(stem)
        found[field] = m.group(0) if m else False
    found['ID'] = found['ID'].replace('_', '-')
    if not found['BROADENER']:
        found['BROADENER'] = ''
    found['SHORT_FILENAME'] = stem
    found['LONG_FILENAME'] = stem + '.txt'
    return found


def returnXscFileContents(filepath):
    """ut:680-696: two space-separated columns (wavenumber, cross section); leading '#' rows are
    dropped by openReturnLines (ut:90-101).  PyRad's own writer puts no newline after its
    header comment (ut:541), so the first sample of such a file shares the comment's row and
    is lost — here as there.  A row that does not split into two numbers raises ValueError
    (the reference's handler for it dies on an undefined name, ut:692)."""
    rows = PyradDataDir._rows(filepath)
    if not rows:                              # missing, empty or NULL_TAG file (ut:91-98)
        print('Could not open:', filepath)
        return False
    out = {'wavenumber': [], 'intensity': []}
    for row in rows:
        cells = re.split('[ ]+', row.strip())
        if len(cells) != 2:
            raise ValueError("%s: cannot split %r into wavenumber and cross section" % (filepath, row))
        out['wavenumber'].append(float(cells[0]))
        out['intensity'].append(float(cells[1]))
    return out


class XscDir:
    """PyRad's measured cross-section cache ``<root>/<molecule>/<file>.txt`` (``data/xsc``, ut:19)."""

    def __init__(self, root: str, cache: bool = True):
        self.root = root
        self.cache = cache               # keep parsed segments (False: read and parse on every call, as the reference does)
        self._segments = {}              # (global_iso, segment) -> parsed file
        self._masters = {}               # global_iso -> the sorted, duplicate-free list of its kept segments

    parseXscFileName = staticmethod(parseXscFileName)

    def returnXscTemperaturePressureValues(self):
        """ut:644-677: {molecule dir: {file stem: {TEMP, PRESSURE, RANGEMIN, RANGEMAX, RES, filename}}}
        for every file whose name parses completely.  Keys keep the reference's
        ``file.strip('.txt')`` (a character strip, not a suffix strip) and its doubled
        ``filename`` extension."""
        table = {}
        for directory in os.listdir(self.root):
            target = '%s/%s' % (self.root, directory)
            if not os.path.isdir(target):
                continue
            entries = {}
            for file in os.listdir(target):
                props = parseXscFileName(file)
                if False in props.values():
                    print('error parsing values')
                    continue
                lo, hi = props['RANGE'].split('-')[:2]
                entries[file.strip('.txt')] = {
                    'TEMP': float(props['TEMP']), 'PRESSURE': float(props['PRESSURE']),
                    'RANGEMIN': float(lo), 'RANGEMAX': float(hi), 'RES': float(props['RES']),
                    'filename': file + '.txt'}
                table[directory] = entries
        return table

    def processXscFile(self, directory, filename):
        """ut:699-715.  A missing or empty file raises FileNotFoundError (the reference prints
        'Could not open' and then fails with a TypeError on the False it got back)."""
        path = '%s/%s/%s' % (self.root, directory, filename)
        contents = returnXscFileContents(path)
        if contents is False:
            raise FileNotFoundError(path)
        return {'wavenumber': contents['wavenumber'], 'intensity': contents['intensity'],
                'res': float(parseXscFileName(filename)['RES'])}


EXOTIC_IDS = {}        # cls:1024; filled by set_xsc_source
_xsc_source = None


def set_xsc_source(source):
    """Install the measured cross-section reader and rebuild EXOTIC_IDS from it."""
    global _xsc_source
    _xsc_source = source
    EXOTIC_IDS.clear()
    if source is not None:
        EXOTIC_IDS.update(source.returnXscTemperaturePressureValues() or {})
    return source


def get_xsc_source():
    if _xsc_source is None:
        raise RuntimeError("no measured cross-section source: call pyrad_amd.data.set_xsc_source(XscDir('data/xsc'))")
    return _xsc_source
