"""The Layer / Molecule / Isotope / Line / Atmosphere object model of pyradClasses, kept so
that PyRad-style drivers (pyradInteractive, main.py) run unchanged on top of the MI355X
engine.  Names, argument meaning, units, the lazy ``progressCrossSection`` protocol and the
error behaviour follow the reference; every number comes from the HIP kernels behind
include/pyrad_hip.h.  There is no CPU fallback: without libpyrad_hip.so or without a GPU the
first computation raises.

Reference map (cls = pyradClasses.py):
    Line cls:237-263 · Isotope cls:266-442 · Molecule cls:445-642 · Layer cls:645-787 ·
    Atmosphere cls:790-821 · getters cls:32-88 · reset protocol cls:38-58 ·
    converters cls:121-156 · integrateSpectrum cls:26-29 · returnPlot cls:824-839.
Measured cross-section ("xsc") molecules (cls:466-505, mergeArray cls:165-233) are host-side
table handling in the reference and here; their array then feeds the same device sweep.
Not carried over (out of scope): the HITRAN download, the curve cache, the
interactive menu.
"""
from __future__ import annotations

import math

import numpy as np

from . import _native as nat
from . import data as _data
from . import engine as _engine
from . import settings

utils = settings          # the reference spells it utils.BASE_RESOLUTION

c = 299792458.0
k = 1.38064852E-23
h = 6.62607004e-34
pi = 3.141592653589793
t0 = 296
p0 = 1013.25
avo = 6.022140857E23

VERSION = settings.VERSION
VERBOSE = False           # the reference prints progress bars; set True to see the regime line


def _say(*a, **kw):
    if VERBOSE:
        print(*a, **kw)


# ----------------------------------------------------------------------------------------
# tables (content of cls:951-1022 stored compactly)
# ----------------------------------------------------------------------------------------
_MOLECULES = ("h2o co2 o3 n2o co ch4 o2 no so2 no2 nh3 hno3 oh hf hcl hbr hi clo ocs h2co hocl n2 hcn ch3cl "
              "h2o2 c2h2 c2h6 ph3 cof2 sf6 h2s hcooh ho2 o clono2 no+ hobr c2h4 ch3oh ch3br ch3cn cf4 c4h2 hc3n "
              "h2 cs so3 c2n2 cocl2").split()
MOLECULE_ID = {name: i + 1 for i, name in enumerate(_MOLECULES)}

_GLOBAL_ISO_ROWS = (
    "1 2 3 4 5 6 129|7 8 9 10 11 12 13 14 121 15 120 122|16 17 18 19 20|21 22 23 24 25|26 27 28 29 30 31|"
    "32 33 34 35|36 37 38|39 40 41|42 43|44|45 46|47 117|48 49 50|51 110|52 53 107 108|19 11 111 112|56 113|"
    "57 58|59 60 61 62 63|64 65 66|67 68|69 118|70 71 72|73 74|75|76 77 105|78 106|79|80 119|126|81 82 83|84|85|"
    "86|127 128|87|88 89|90 91|92|93 94|95|96|116|109|103 115|97 98 99 100|114|123|124 125")
HITRAN_GLOBAL_ISO = {m + 1: {i + 1: int(g) for i, g in enumerate(row.split())}
                     for m, row in enumerate(_GLOBAL_ISO_ROWS.split("|"))}

COLOR_LIST = ['xkcd:white', 'xkcd:bright orange', 'xkcd:seafoam green', 'xkcd:bright blue', 'xkcd:salmon',
              'xkcd:light violet', 'xkcd:green yellow']
EXOTIC_IDS = _data.EXOTIC_IDS          # cls:1024; filled by data.set_xsc_source


# ----------------------------------------------------------------------------------------
# module-level helpers (cls:26-162)
# ----------------------------------------------------------------------------------------
def _ctx() -> nat.Context:
    return _engine.get_engine().ctx


def integrateSpectrum(spectrum, unitAngle=pi, res=settings.BASE_RESOLUTION):
    """cls:26-29 on the device (K6): sum(nan_to_num(spectrum)) * unitAngle * res.  Like the
    reference, the default ``res`` is frozen at import time."""
    spectrum = np.ascontiguousarray(spectrum, dtype=np.float64)
    ctx = _ctx()
    buf = ctx.buffer(max(spectrum.size, 1))
    try:
        buf.upload(spectrum)
        return ctx.band_integral(buf, spectrum.size, unitAngle, res)
    finally:
        buf.free()


def getCrossSection(obj):
    if not obj.progressCrossSection:
        obj.createCrossSection()
    return obj.crossSection


def resetCrossSection(obj):
    """cls:38-45: marks every Isotope/Molecule below ``obj`` dirty (a Layer itself is skipped)."""
    if not isinstance(obj, Layer):
        if not obj.exotic:
            n = int((obj.rangeMax - obj.rangeMin) / utils.BASE_RESOLUTION)
            type(obj).crossSection.defer(obj, lambda n=n: np.zeros(n))      # the zeros of cls:41, made when read
            if isinstance(obj, Isotope):
                obj._host_array_assigned("_crossSection_host", installed=False)      # (the zeros above are the reference's reset, not somebody's array)
                obj._inputs_version += 1             # (a merged layer step computed from this isotopologue is stale too)
            obj.progressCrossSection = False
    if isinstance(obj, Isotope):
        return                      # its children are Lines (the reference walks them and skips each one)
    for child in obj:
        if not isinstance(child, Line):
            resetCrossSection(child)


def resetData(obj):
    """cls:48-58: drop and reload the line data below ``obj`` (range or pressure changed)."""
    for child in obj:
        if isinstance(child, Isotope):
            child.clear_lines()
            child.getData()
        else:
            resetData(child)
    resetCrossSection(obj)


def getAbsCoef(obj):
    if not obj.progressCrossSection:
        obj.createCrossSection()
    return obj.absCoef


def getTransmittance(obj):
    if not obj.progressCrossSection:
        obj.createCrossSection()
    return obj.transmittance


def getOpticalDepth(obj):
    if not obj.progressCrossSection:
        obj.createCrossSection()
    return obj.opticalDepth                         # -log(transmittance), cls:76


def getAbsorbance(obj):
    if not obj.progressCrossSection:
        obj.createCrossSection()
    return obj.absorbance


def getEmissivity(obj):
    if not obj.progressCrossSection:
        obj.createCrossSection()
    return obj.emissivity


def getGlobalIsotope(ID, isotopeDepth):
    return [HITRAN_GLOBAL_ISO[ID][i] for i in range(1, isotopeDepth + 1)]


def totalConcentration(layer):
    total = 0
    for molecule in layer:
        total += molecule.concentration
    return total


def totalLineList(obj):
    if isinstance(obj, Isotope):
        return obj.linelist()
    fullList = []
    for item in obj:
        fullList += totalLineList(item)
    return fullList


def convertLength(value, units):
    if units == 'cm':
        return value
    if units in ['m', 'meter']:
        return value * 100
    if units in ['ft', 'feet']:
        return value * 30.48
    if units in ['in', 'inch']:
        return value * 2.54


def convertPressure(value, units):
    if units == 'mbar':
        return value
    if units in ['atm', 'atmospheres', 'atmosphere']:
        return value * 1013.25
    if units in ['b', 'bar']:
        return value * 1000
    if units in ['pa', 'pascal', 'pascals']:
        return value / 100


def convertRange(value, units):
    if units == 'cm-1':
        return value
    if units in ['um', 'micrometers', 'micrometer']:
        return 10000 / value


def convertTemperature(value, units):
    u = units[0].upper()
    if u == 'K':
        return value
    if u == 'C':
        return value + 273            # 273, not 273.15 (cls:154)
    if u == 'F':
        return (value - 32) * 5 / 9 + 273


def interpolateArray(hiResXAxis, loResXAxis, loResYValues):
    """cls:159-162 (used by the reference only inside createCrossSection, where the device
    regrid kernel replaces it; kept for callers)."""
    return np.interp(hiResXAxis, loResXAxis, loResYValues)


def isBetween(test, minValue, maxValue):
    return minValue <= test <= maxValue


def mergeArray(newX, oldX, oldY):
    """cls:165-233: lay a measured cross section (oldX, oldY) onto the layer axis newX by matching
    abscissae rounded to 0.01 cm^-1; zeros outside the overlap.  Kept as the reference has it:
    the last overlapping sample is dropped, positions are matched only at the first overlapping
    point (the copy is then index-for-index), a start value missing from the rounded axis raises
    ValueError, and a partial overlap returns an array longer than newX."""
    as_list = lambda v: v if isinstance(v, list) else v.tolist()
    oldY = as_list(oldY)
    nx = [round(x, 2) for x in as_list(newX)]
    ox = [round(x, 2) for x in as_list(oldX)]
    n_lo, n_hi, o_lo, o_hi = min(nx), max(nx), min(ox), max(ox)
    if n_hi < o_lo or n_lo > o_hi:
        return np.zeros(len(nx))
    if n_lo <= o_lo:
        lead, src = nx.index(o_lo), 0
    else:
        lead, src = 0, ox.index(n_lo)
    if n_hi >= o_hi:
        last_new, src_end = lead + len(ox) - 1, len(ox) - 1
    else:
        last_new, src_end = len(nx) - 1, src + len(nx) - 1
    if src < src_end and src_end > len(oldY):
        raise IndexError("list index out of range")
    return np.asarray([0] * lead + oldY[src:max(src_end, src)] + [0] * (len(nx) - last_new))


def concentration_from_kwargs(**abundance):
    """The volume fraction Molecule.__init__ derives from its keyword (cls:453-463, 543-560),
    including ppb -> x1e-8 (cls:554)."""
    conc = 0
    for key, v in abundance.items():
        if key == 'ppm':
            conc = v * 10**-6
        elif key == 'ppb':
            conc = v * 10**-8
        elif key in ('percentage', 'perc', '%'):
            conc = v / 100
        elif key == 'concentration':
            conc = (v * 1E6) * 10**-6
        else:
            print('Invalid concentration type. Use ppm, ppb, percentage, or concentration.')
    return conc


# ----------------------------------------------------------------------------------------
# device residency: cross sections and swept spectra stay in HBM between getters
# ----------------------------------------------------------------------------------------
class _LazyArray:
    """Attribute whose host copy is fetched from the device on first read.  The reference keeps
    plain NumPy arrays in ``crossSection`` / ``lineSurvey``; here the device owns the data and a
    host array is made only when somebody looks at it (a getter chain that ends in absCoef never
    does).  Assigning a host array drops the loader."""

    def __init__(self, name):
        self.host, self.loader = "_%s_host" % name, "_%s_loader" % name

    def __get__(self, obj, owner=None):
        if obj is None:
            return self
        load = obj.__dict__.get(self.loader)
        if load is not None:
            obj.__dict__[self.host] = load()
            obj.__dict__[self.loader] = None
        return obj.__dict__.get(self.host)

    def __set__(self, obj, value):
        obj.__dict__[self.host] = value
        obj.__dict__[self.loader] = None
        hook = getattr(obj, "_host_array_assigned", None)
        if hook is not None:
            hook(self.host)

    def defer(self, obj, loader):
        obj.__dict__[self.host] = None
        obj.__dict__[self.loader] = loader


def _zeros_later(descriptor, obj, n):
    """``obj.<array> = np.zeros(n)`` without making the array until somebody reads it (a 2.4e6-point layer with three
    molecules would otherwise write 150 MB of zeros at construction, most of which nobody ever looks at)"""
    loader = lambda n=int(n): np.zeros(n)
    descriptor.defer(obj, loader)
    obj.__dict__[descriptor.host + "_zeros"] = loader


def _copy_of_layer_cross_section(descriptor, obj, layer):
    """``obj.crossSection = np.copy(layer.crossSection)`` (cls:290, 512): while the layer's array is still its initial
    zeros the copy is zeros made on first read; a computed array is copied now, as the reference does"""
    d = Layer.crossSection
    pending = layer.__dict__.get(d.loader)
    if pending is not None and pending is layer.__dict__.get(d.host + "_zeros"):
        _zeros_later(descriptor, obj, int((layer.rangeMax - layer.rangeMin) / utils.BASE_RESOLUTION))
    else:
        setattr(obj, "crossSection", np.copy(layer.crossSection))


def _free_buffers(bufs):
    for b in bufs.values():
        try:
            if b.ctx.h:
                b.free()
        except Exception:
            pass
    bufs.clear()


class _SweepState:
    """Device buffers of one object's property chain (absorption coefficient, transmittance and
    scratch) and the key of what they were computed from; a getter re-sweeps only when the key
    (member cross-section versions, concentrations, P, T, depth, grid) has changed."""

    def __init__(self, owner):
        self.bufs = {}
        self.n = -1
        self.key = None
        import weakref
        weakref.finalize(owner, _free_buffers, self.bufs)

    def reserve(self, ctx, n):
        """buffers for n grid points: kept when they are large enough (a range change that shrinks the grid re-uses them:
        five hipFree + hipMalloc of 19 MB are 1.2 ms), what they held is stale either way"""
        if any(b.h is None or b.ctx is not ctx or b.n < n for b in self.bufs.values()):
            _free_buffers(self.bufs)
            self.key = None
        if self.n != n:
            self.n, self.key = n, None
        return self

    def buf(self, ctx, name):
        b = self.bufs.get(name)
        if b is None:
            b = self.bufs[name] = ctx.buffer(max(self.n, 1))
        return b


def _iso_params(iso):
    layer = iso.layer
    q_T = iso.q[layer.T]                       # KeyError for a non-integer temperature, as cls:389
    return nat.IsoParams(float(layer.T), float(layer.P), float(iso.molecule.concentration), float(iso.molmass),
                         float(q_T), float(iso.q296))


def _check_window(g):
    if g["W"] < 1:
        raise IndexError("index 0 is out of bounds for axis 0 with size 0")     # rightCurve[0], cls:393


def _mark_computed(ctx, isotopes, n):
    for iso in isotopes:
        iso._xs_version += 1
        iso._dev_xsec_valid = True
        iso._xs_deferred = False
        iso._xs_installed = False
        Isotope.crossSection.defer(iso, (lambda b=iso._dev_xsec, n=n: b.download(n, pinned=True)))
        iso._regime_counts = None
        iso.progressCrossSection = True
    if VERBOSE:
        for iso in isotopes:
            _say('\ngaussian only: %s\t lorentz only: %s\t voigt: %s\n' % iso.regimeCounts, end='\r')


def _compute_cross_sections(isotopes):
    """Batched Isotope.createCrossSection (cls:361-407) for every dirty isotopologue in the
    list: one prep launch + one accumulate launch for all of them.  The cross sections stay on
    the device; ``iso.crossSection`` downloads on first read."""
    dirty = [i for i in isotopes if (not i.progressCrossSection or i._xs_deferred) and not i.exotic]
    if not dirty:
        return
    ctx = _ctx()
    jobs = []
    for iso in dirty:
        g = iso.layer._grid()
        _check_window(g)
        jobs.append((iso._device_lines(ctx), _iso_params(iso), _engine.native_grid(g), iso._device_xsec(ctx, g["n_base"])))
    ctx.xsec_accumulate_dev(jobs)
    _mark_computed(ctx, dirty, dirty[0].layer._grid()["n_base"])


class _OpticalMixin:
    """The property chain shared by Isotope, Molecule and Layer (cls:322-340, 581-606, 707-732,
    784-787).  ``_sweep_members`` says which isotopologues and concentrations take part; the
    swept arrays live in a per-object _SweepState on the device and a getter downloads only the
    array it returns."""

    def _sweep_members(self):
        raise NotImplementedError

    def _ensure_swept(self):
        """(state, n) with absorption coefficient and transmittance of the CURRENT members resident.
        When every line-by-line member is dirty (first use, or after a temperature / pressure / range
        change) the whole layer step runs as one fused launch sequence (lbl_layer_step_dev: line
        prep, accumulate, sweep in the accumulate kernel's output stage); otherwise only the dirty
        isotopologues are accumulated and the sweep kernel runs if anything it reads has changed."""
        ctx = _ctx()
        layer = self._layer()
        g = layer._grid()
        n = g["n_base"]
        members, conc = self._sweep_members()
        flat = [iso for isos in members for iso in isos]
        st = self.__dict__.get("_sweep_state")
        if st is None:
            st = self.__dict__["_sweep_state"] = _SweepState(self)
        st.reserve(ctx, n)
        lbl = [i for i in flat if not i.exotic]
        dirty = [i for i in lbl if not i.progressCrossSection]
        fusable = (dirty and len(dirty) == len(flat) and g["resolution"] == g["base_resolution"]
                   and g["n_work"] == n and len(flat) <= nat.limit("arrays_per_layer"))
        if self._merged_step_applies(flat, lbl):
            # A LAYER whose line lists are due (any of them dirty): ONE accumulate job over the merged, factor-weighted
            # line lists - the absorption coefficient sum_m f_m sum_iso xs_iso (cls:707-712, 581-583, 566-571) accumulated
            # directly, the transmittance in the kernel's output stage.  No isotopologue cross section is written, and none
            # is marked computed: getCrossSection(isotope | molecule) produces it when asked (the reference's lazy
            # protocol, cls:32-88), and the layer's arrays are found again through the key below until an input changes.
            key = self._merged_key(flat, conc, layer, g)
            if st.key != key and isinstance(st.key, tuple) and st.key[:-1] == key[:-1]:
                # only the depth changed (changeDepth resets nothing, cls:754-755): the absorption coefficient stands,
                # the transmittance exp(-k depth) (cls:714-716) is redone from it
                ctx.column_fold_dev([st.bufs["abs_coef"]], [layer.T], [layer.depth], layer.rangeMin, layer.rangeMax, n,
                                    st.buf(ctx, "tmp"), surface_T=float(layer.T), trans=[st.buf(ctx, "trans")])
                st.key = key
            elif st.key != key:
                _check_window(g)
                iso_mol = [m for m, isos in enumerate(members) for _ in isos]
                ctx.layer_merged_step_dev([i._device_lines(ctx) for i in flat], [_iso_params(i) for i in flat],
                                          _engine.native_grid(g), iso_mol, conc, layer.depth,
                                          abs_coef=st.buf(ctx, "abs_coef"), trans=st.buf(ctx, "trans"))
                st.key = key
            for iso in flat:
                iso._defer_cross_section()
            self._members_ready()
            return st, n
        if fusable:
            _check_window(g)
            iso_mol = [m for m, isos in enumerate(members) for _ in isos]
            ctx.layer_step_dev([i._device_lines(ctx) for i in flat], [_iso_params(i) for i in flat],
                               _engine.native_grid(g), [i._device_xsec(ctx, n) for i in flat], iso_mol, conc,
                               layer.depth, abs_coef=st.buf(ctx, "abs_coef"), trans=st.buf(ctx, "trans"))
            _mark_computed(ctx, flat, n)
            st.key = self._sweep_key(flat, conc, layer, g)
        else:
            _compute_cross_sections(dirty)
            key = self._sweep_key(flat, conc, layer, g)
            if st.key != key:
                xs = [iso._device_xsec_current(ctx, n) for iso in flat]
                iso_mol = [m for m, isos in enumerate(members) for _ in isos]
                ctx.layer_sweep_dev(xs, iso_mol, conc, layer.P, layer.T, layer.depth, layer.rangeMin, layer.rangeMax, n,
                                    abs_coef=st.buf(ctx, "abs_coef"), trans=st.buf(ctx, "trans"))
                st.key = self._sweep_key(flat, conc, layer, g)
        self._members_ready()
        return st, n

    def _merged_step_applies(self, flat, lbl):
        return False                    # (Layer overrides: isotopologues and molecules keep the per-line-list path)

    @staticmethod
    def _merged_key(flat, conc, layer, g):
        return ("merged", tuple((id(i), i._inputs_version) for i in flat), tuple(float(c) for c in conc), layer.P, layer.T,
                layer.rangeMin, layer.rangeMax, g["n_base"], g["resolution"], settings.ACCURACY, layer.depth)     # (depth last)

    @staticmethod
    def _sweep_key(flat, conc, layer, g):
        return (tuple((id(i), i._xs_version) for i in flat), tuple(float(c) for c in conc), layer.P, layer.T, layer.depth,
                layer.rangeMin, layer.rangeMax, g["n_base"])

    def _members_ready(self):
        """hook: a Layer / Molecule marks the molecule sums it stands for as computed (cls:566-571)"""

    @property
    def absCoef(self):
        st, n = self._ensure_swept()
        return st.bufs["abs_coef"].download(n, pinned=True)

    @property
    def transmittance(self):
        st, n = self._ensure_swept()
        return st.bufs["trans"].download(n, pinned=True)

    def _optical(self, kind):
        st, n = self._ensure_swept()
        ctx = _ctx()
        out = st.buf(ctx, "tmp")
        ctx.optical_dev(st.bufs["trans"], n, kind, out)
        return out.download(n, pinned=True)

    @property
    def emissivity(self):
        return self._optical(0)                 # 1 - transmittance (cls:330-332)

    @property
    def emittance(self):
        return self.emissivity

    @property
    def absorbance(self):
        return self._optical(1)                 # log10(1 / transmittance) (cls:338-340)

    @property
    def opticalDepth(self):
        return self._optical(2)                 # -log(transmittance) (cls:73-76)

    def planck(self, temperature):
        return self._layer().planck(temperature)

    def transmission(self, surfaceSpectrum):
        """transmittance * surfaceSpectrum + emittance * planck(T)  (cls:784-787): the resident
        transmittance folded with the uploaded spectrum by the column kernel (one layer)."""
        st, n = self._ensure_swept()
        ctx = _ctx()
        layer = self._layer()
        I_in = np.ascontiguousarray(surfaceSpectrum, dtype=np.float64)
        if I_in.shape != (n,):
            raise ValueError("operands could not be broadcast together with shapes (%d,) %s" % (n, I_in.shape))
        src = st.buf(ctx, "I_in").upload(I_in)
        out = st.buf(ctx, "tmp")
        ctx.column_sweep_dev([st.bufs["trans"]], [layer.T], layer.rangeMin, layer.rangeMax, n, out, I_in=src)
        return out.download(n, pinned=True)


# ----------------------------------------------------------------------------------------
# Line (cls:237-263)
# ----------------------------------------------------------------------------------------
class Line:
    def __init__(self, wavenumber, intensity, einsteinA, airHalfWidth,
                 selfHalfWidth, lowerEnergy, tempExponent, pressureShift, parent):
        self.isotope = parent
        self.molecule = self.isotope.molecule
        self.layer = self.molecule.layer
        self.wavenumber = wavenumber
        self.intensity = intensity
        self.einsteinA = einsteinA
        self.airHalfWidth = airHalfWidth
        self.selfHalfWidth = selfHalfWidth
        self.lowerEnergy = lowerEnergy
        self.tempExponent = tempExponent
        self.pressureShift = pressureShift

    # introspection only: the device computes these per line in K1 (lbl_line_quantities)
    @property
    def broadenedLine(self):
        return self.wavenumber + self.pressureShift * self.layer.P / p0

    @property
    def lorentzHW(self):
        return (float((1 - self.molecule.concentration) * self.airHalfWidth + self.molecule.concentration
                      * self.selfHalfWidth) * (self.layer.P / p0) * (t0 / self.layer.T) ** self.tempExponent)

    @property
    def gaussianHW(self):
        return self.broadenedLine * math.sqrt(2 * k * self.layer.T / self.isotope.molMass / c ** 2)


# ----------------------------------------------------------------------------------------
# Isotope (cls:266-442): a list of Line, stored as a structure of arrays
# ----------------------------------------------------------------------------------------
class Isotope(_OpticalMixin, list):
    _FIELDS = ("nu", "sw", "a", "gamma_air", "gamma_self", "elower", "n_air", "delta_air")
    crossSection = _LazyArray("crossSection")      # device-resident after createCrossSection; downloaded on read
    lineSurvey = _LazyArray("lineSurvey")          # computed (K7) and downloaded on first read

    def __init__(self, number, molecule):
        super().__init__()
        self.molecule = molecule
        self.layer = self.molecule.layer
        self._dev_lines = None
        self._dev_xsec = None
        self._dev_xsec_valid = False
        self._xs_version = 0
        self._xs_deferred = False        # marked computed by a merged layer step: the array itself is made when somebody reads it
        self._xs_installed = False       # somebody assigned ``crossSection`` an array of their own (and nothing has recomputed it since):
                                         # the layer's sums must use THAT array, as the reference's getters do (cls:32-35, 566-571)
        self._inputs_version = 0         # bumped by everything that marks the cross section dirty (resetCrossSection, new lines)
        self._struct_version = 0         # bumped when the line list itself changes or somebody installs a cross section: a resident
                                         # column (Atmosphere._column_fast) then re-reads this layer's blocks
        self._regime_counts = (0, 0, 0)
        _copy_of_layer_cross_section(Isotope.crossSection, self, self.layer)
        self.exotic = molecule.exotic
        self._lines = {f: np.zeros(0) for f in self._FIELDS}
        if number not in EXOTIC_IDS:
            params = _data.get_source().readMolParams(number)
            self.globalIsoNumber = params[0]
            self.shortName = params[1]
            self.name = 'Isotope %s' % self.globalIsoNumber
            self.molNum = params[2]
            self.isoN = params[3]
            self.abundance = params[4]
            self.q296 = params[5]
            self.gj = params[6]
            self.molmass = params[7]
            self.q = {}
            _zeros_later(Isotope.lineSurvey, self, int((self.layer.rangeMax - self.layer.rangeMin) / utils.BASE_RESOLUTION))
            self.progressCrossSection = False

    # -- list protocol over the SoA ------------------------------------------------------
    def __len__(self):
        return int(self._lines["nu"].size)

    def __bool__(self):
        return True

    def _line(self, i):
        L = self._lines
        return Line(float(L["nu"][i]), float(L["sw"][i]), float(L["a"][i]), float(L["gamma_air"][i]),
                    float(L["gamma_self"][i]), float(L["elower"][i]), float(L["n_air"][i]),
                    float(L["delta_air"][i]), self)

    def __iter__(self):
        for i in range(len(self)):
            yield self._line(i)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._line(j) for j in range(*i.indices(len(self)))]
        n = len(self)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError("list index out of range")
        return self._line(i)

    def append(self, line):
        vals = (line.wavenumber, line.intensity, line.einsteinA, line.airHalfWidth, line.selfHalfWidth,
                line.lowerEnergy, line.tempExponent, line.pressureShift)
        for f, v in zip(self._FIELDS, vals):
            self._lines[f] = np.append(self._lines[f], float(v))
        self._invalidate_lines()

    def pop(self, index=-1):
        line = self[index]
        n = len(self)
        keep = np.ones(n, dtype=bool)
        keep[index] = False
        self._lines = {f: v[keep] for f, v in self._lines.items()}
        self._invalidate_lines()
        return line

    def clear_lines(self):
        self._lines = {f: np.zeros(0) for f in self._FIELDS}
        self._invalidate_lines()

    def set_lines(self, lines: dict):
        """Install a structure-of-arrays line list (nu, sw, a, elower, gamma_air, gamma_self, n_air, delta_air)."""
        self._lines = {f: np.ascontiguousarray(lines[f], dtype=np.float64) if f in lines
                       else np.zeros(len(lines["nu"])) for f in self._FIELDS}
        self._invalidate_lines()

    def _invalidate_lines(self):
        if self._dev_lines is not None:
            self._dev_lines.free()
            self._dev_lines = None
        self.progressCrossSection = False
        self._inputs_version += 1
        self._struct_version += 1

    def _host_array_assigned(self, which, installed=True):
        if which == "_crossSection_host":           # somebody installed a host array: the device copy is stale
            self._dev_xsec_valid = False
            self._xs_deferred = False
            self._xs_version += 1
            self._xs_installed = bool(installed)
            if installed:
                self._struct_version += 1

    def _defer_cross_section(self):
        """A merged layer step (one accumulate job over all the layer's line lists, settings.LAYER_STEP) has just produced
        the layer's absorption coefficient WITHOUT this isotopologue's cross-section array.  The reference's protocol
        (cls:32-88) says the cross section is now computed: progressCrossSection is set as createCrossSection would, and
        the array is made by the per-line-list path when somebody reads ``crossSection`` (or a device-side consumer asks
        for it): getCrossSection(isotope), the molecule sums and the per-molecule getters all go through here."""
        if self.exotic or (self.progressCrossSection and not self._xs_deferred):
            return                                  # a computed, current array exists
        self._xs_deferred = True
        self._xs_installed = False                  # (what stands now is computed from the lines)
        self._dev_xsec_valid = False
        self._xs_version += 1
        self._regime_counts = None
        self.progressCrossSection = True
        Isotope.crossSection.defer(self, self._materialise_cross_section)

    def _materialise_cross_section(self):
        _compute_cross_sections([self])             # K1 + K2 for this line list; leaves the download deferred
        return self.crossSection

    @property
    def regimeCounts(self):
        """(gaussian, lorentz, voigt) line counts as printed at cls:406, from the device's regime
        select (lbl_line_quantities) - fetched when asked for, not on every createCrossSection."""
        if self._regime_counts is None:
            ctx = _ctx()
            g = self.layer._grid()
            q = ctx.line_quantities(self._device_lines(ctx), _iso_params(self), _engine.native_grid(dict(g, W=max(g["W"], 1))))
            c = np.bincount(q["regime"], minlength=3)
            self._regime_counts = (int(c[0]), int(c[1]), int(c[2]))
        return self._regime_counts

    # -- device residency ----------------------------------------------------------------
    def _device_lines(self, ctx):
        if self._dev_lines is None or self._dev_lines.h is None:
            # a window handed out by a MemorySource is a slice of a registered list: a VIEW of that list's one resident
            # copy (lbl_lines_view) instead of another upload - the 30 layers of a column, or a layer whose range or
            # pressure keeps changing, re-window the same three lists
            pooled = _engine.get_engine().pooled_lines(self._lines)
            self._dev_lines = pooled if pooled is not None else ctx.lines(self._lines)
        return self._dev_lines

    def _device_xsec(self, ctx, n):
        if self._dev_xsec is None or self._dev_xsec.h is None or self._dev_xsec.n < n or self._dev_xsec.ctx is not ctx:
            if self._dev_xsec is not None and self._dev_xsec.h is not None and self._dev_xsec.ctx.h:
                self._dev_xsec.free()
            self._dev_xsec = ctx.buffer(max(n, 1))
            import weakref
            weakref.finalize(self, _free_buffers, {"xsec": self._dev_xsec})
        return self._dev_xsec

    def _device_xsec_current(self, ctx, n):
        """Device copy of self.crossSection (uploaded if a host array was installed since)."""
        if self._xs_deferred:
            _compute_cross_sections([self])         # promised by a merged layer step: made now, on the device
        if (self._dev_xsec_valid and self._dev_xsec is not None and self._dev_xsec.h is not None
                and self._dev_xsec.ctx is ctx):
            return self._dev_xsec
        xs = np.ascontiguousarray(self.crossSection, dtype=np.float64)
        if xs.shape != (n,):
            raise ValueError("cross section has %s points, layer grid has %d" % (xs.shape, n))
        buf = self._device_xsec(ctx, n)
        buf.upload(xs)
        self._dev_xsec_valid = True
        return buf

    # -- reference surface ---------------------------------------------------------------
    def _layer(self):
        return self.layer

    def _sweep_members(self):
        return [[self]], [self.molecule.concentration]          # cls:324 uses the molecule's concentration

    P = property(lambda self: self.layer.P)
    T = property(lambda self: self.layer.T)
    depth = property(lambda self: self.layer.depth)
    rangeMin = property(lambda self: self.layer.rangeMin)
    rangeMax = property(lambda self: self.layer.rangeMax)
    resolution = property(lambda self: self.layer.resolution)
    distanceFromCenter = property(lambda self: self.layer.distanceFromCenter)
    yAxis = property(lambda self: np.copy(self.layer.yAxis))
    xAxis = property(lambda self: np.copy(self.layer.xAxis))

    @property
    def molMass(self):
        return self.molmass / 1000 / avo

    def getData(self):
        _say('Getting data for %s, isotope %s' % (self.molecule.name, self.globalIsoNumber))
        src = _data.get_source()
        lines = src.gatherData(self.globalIsoNumber, self.layer.effectiveRangeMin, self.layer.effectiveRangeMax)
        self.q = src.getQData(self.globalIsoNumber)
        self.set_lines(lines)
        _ctx()                                            # the reference computes the survey here (cls:359): fail now without a GPU
        Isotope.lineSurvey.defer(self, self.createLineSurvey)      # ... the histogram itself on first read

    def createCrossSection(self):
        """cls:361-407 on the device: K1 line prep, K2 owner-computes accumulate, K3 regrid."""
        self.progressCrossSection = False
        _compute_cross_sections([self])

    def createLineSurvey(self):
        """cls:409-428 on the device (K7)."""
        ctx = _ctx()
        g = self.layer._grid()
        n = int((self.rangeMax - self.rangeMin) / utils.BASE_RESOLUTION)
        out = ctx.buffer(max(n, 1))
        try:
            grid = _engine.native_grid(dict(g, n_base=n, W=max(g["W"], 1)))
            ctx.line_survey_dev(self._device_lines(ctx), grid, out)
            survey = out.download(n)
        finally:
            out.free()
        self.lineSurvey = survey
        return survey

    def linelist(self):
        return list(self)


# ----------------------------------------------------------------------------------------
# Molecule (cls:445-642)
# ----------------------------------------------------------------------------------------
class Molecule(_OpticalMixin, list):
    crossSection = _LazyArray("crossSection")      # sum of the isotopologue cross sections, summed and downloaded on read

    def __init__(self, shortNameOrMolNum, layer, isotopeDepth=1, **abundance):
        super().__init__()
        self.layer = layer
        self.concText = ''
        self.concentration = 0
        self.exotic = False
        for key in abundance:
            if key == 'ppm':
                self.setPPM(abundance[key])
            elif key == 'ppb':
                self.setPPB(abundance[key])
            elif key == 'percentage' or key == 'perc' or key == '%':
                self.setPercentage(abundance[key])
            elif key == 'concentration':
                self.setConcentration(abundance[key])
            else:
                print('Invalid concentration type. Use ppm, ppb, percentage, or concentration.')
        self._xsc_member = None
        if type(shortNameOrMolNum) is dict:
            self._init_from_xsc(shortNameOrMolNum)
            return
        self.isotopeDepth = isotopeDepth
        _copy_of_layer_cross_section(Molecule.crossSection, self, layer)
        try:
            int(shortNameOrMolNum)
            self.ID = int(shortNameOrMolNum)
            self.name = False
            self._by_number = True
        except ValueError:
            self.name = shortNameOrMolNum
            self.ID = MOLECULE_ID[self.name]
            self._by_number = False
        for isotope in getGlobalIsotope(self.ID, isotopeDepth):
            isoClass = Isotope(isotope, self)
            self.append(isoClass)
            if not self.name:
                self.name = isoClass.shortName
        self.progressCrossSection = False
        self.exotic = False

    def _init_from_xsc(self, spec):
        """cls:466-505: a molecule given as {name: xsc file name or index}.  The file fixes the
        layer's temperature and pressure (Torr / 0.75006 -> mbar), its table is brought to
        0.01 cm^-1 if coarser and merged onto the layer axis; the molecule holds no isotopologues
        and its cross section is never invalidated (cls:40)."""
        name = list(spec.keys())[0]
        filename = list(spec.values())[0]
        if type(filename) == int:
            filename = list(EXOTIC_IDS[name].keys())[filename] + '.txt'
        source = _data.get_xsc_source()
        table = source.processXscFile(name, filename)
        props = source.parseXscFileName(filename)
        rangeMin, rangeMax = (float(v) for v in props['RANGE'].split('-')[:2])
        temp = int(float(props['TEMP']))
        pressure = float(props['PRESSURE']) / _data.TORR_PER_MBAR
        self.name = name
        self._xsc_spec = dict(spec)
        self.exotic = True
        self._xsc_member = Isotope(name, self)      # the reference's dummyIso: holds the device copy here
        if temp != self.layer.T:
            self.layer.changeTemperature(temp)
        if pressure != self.layer.P:
            self.layer.changePressure(pressure)
        xAxis = np.arange(rangeMin, rangeMax, .01)
        if float(props['RES']) > .01:
            measured = interpolateArray(xAxis, table['wavenumber'], table['intensity'])
        else:
            measured = table['intensity']
        self.crossSection = mergeArray(self.layer.xAxis, xAxis, measured)
        self.progressCrossSection = True
        self._xsc_member.crossSection = self.crossSection
        self._xsc_member.progressCrossSection = True

    def __str__(self):
        return '%s: %s' % (self.name, self.concText)

    def __bool__(self):
        return True

    def _layer(self):
        return self.layer

    def _members(self):
        """What the device sweep reads for this molecule: its isotopologues, or the holder of a
        measured cross section (kept in step with self.crossSection)."""
        if not self.exotic:
            return list(self)
        if self._xsc_member.crossSection is not self.crossSection:
            self._xsc_member.crossSection = self.crossSection
            self._xsc_member._dev_xsec_valid = False
        return [self._xsc_member]

    def _sweep_members(self):
        return [self._members()], [self.concentration]

    def returnCopy(self, layer=None):
        valueUnit = self.concText.split()
        tempDict = {valueUnit[1]: float(valueUnit[0])}
        if self.exotic:         # the reference's copy fails on these (no isotopeDepth, cls:539): re-read the file
            return Molecule(dict(self._xsc_spec), layer if layer is not None else self.layer, **tempDict)
        # a molecule made from its HITRAN number carries the upper-case short name, which is not
        # a MOLECULE_ID key (the reference's copy raises KeyError there): copy by number instead
        newMolecule = Molecule(self.ID if self._by_number else self.name, layer if layer is not None else self.layer,
                               isotopeDepth=int(self.isotopeDepth), **tempDict)
        newMolecule.getData()
        return newMolecule

    def setPercentage(self, percentage):
        self.concentration = percentage / 100
        self.concText = '%s %%' % percentage
        resetCrossSection(self)

    def setPPM(self, ppm):
        self.concentration = ppm * 10**-6
        self.concText = '%s ppm' % ppm
        resetCrossSection(self)

    def setPPB(self, ppb):
        self.concentration = ppb * 10**-8          # sic (cls:554)
        self.concText = '%s ppb' % ppb
        resetCrossSection(self)

    def setConcentration(self, concentration):
        self.setPPM(concentration * 1E6)
        resetCrossSection(self)

    def getData(self):
        for isotope in self:
            isotope.getData()

    def createCrossSection(self):
        """cls:566-571: sum of the isotopologue cross sections (no abundance weighting).  The
        isotopologues are accumulated (with this molecule's sweep folded in when all of them are
        due); the sum itself is formed on the device when ``crossSection`` is read."""
        if self.exotic:
            return
        self._ensure_swept()
        self._mark_sum_ready(force=True)          # the reference re-sums on every call (cls:566-571)

    def _mark_sum_ready(self, force=False):
        """Defer ``crossSection`` = zeros + sum of the isotopologue cross sections to the first read.  A sum that
        is already deferred or loaded is kept only while the isotopologue cross sections it was made from are
        still the current ones (``_xs_version``: bumped by every recomputation and by an assigned host array);
        the loader always adds the device copies that are current when it runs."""
        if self.exotic:
            return
        isos = list(self)
        versions = [i._xs_version for i in isos]
        if self.progressCrossSection and not force and self.__dict__.get("_sum_versions") == versions:
            return
        n = int((self.rangeMax - self.rangeMin) / utils.BASE_RESOLUTION)

        def load():
            ctx = _ctx()
            return _sum_on_device(ctx, [i._device_xsec_current(ctx, n) for i in isos], n)
        Molecule.crossSection.defer(self, load)
        self.__dict__["_sum_versions"] = versions
        self.progressCrossSection = True

    def _members_ready(self):
        self._mark_sum_ready()

    @property
    def lineSurvey(self):
        """cls:589-594: sum of the isotopologue surveys (device sum, list order)."""
        return _sum_host_arrays([isotope.lineSurvey for isotope in self],
                                int((self.rangeMax - self.rangeMin) / utils.BASE_RESOLUTION))

    P = property(lambda self: self.layer.P)
    T = property(lambda self: self.layer.T)
    depth = property(lambda self: self.layer.depth)
    rangeMin = property(lambda self: self.layer.rangeMin)
    rangeMax = property(lambda self: self.layer.rangeMax)
    resolution = property(lambda self: self.layer.resolution)
    distanceFromCenter = property(lambda self: self.layer.distanceFromCenter)
    yAxis = property(lambda self: np.copy(self.layer.yAxis))
    xAxis = property(lambda self: np.copy(self.layer.xAxis))


def _sum_host_arrays(arrays, n):
    """zeros(n) + a0 + a1 + ... for host arrays, summed by the device kernel."""
    ctx = _ctx()
    tmp = []
    try:
        for a in arrays:
            tmp.append(ctx.buffer(max(n, 1)).upload(np.ascontiguousarray(a, dtype=np.float64)))
        return _sum_on_device(ctx, tmp, n)
    finally:
        for b in tmp:
            b.free()


def _sum_on_device(ctx, bufs, n):
    """zeros + b0 + b1 + ... in list order (cls:567-569, 685-687), on the device."""
    out = ctx.buffer(max(n, 1))
    try:
        ctx.sum_dev(bufs, n, out)
        return out.download(n)
    finally:
        out.free()


# ----------------------------------------------------------------------------------------
# Layer (cls:645-787)
# ----------------------------------------------------------------------------------------
class Layer(_OpticalMixin, list):
    hasAtmosphere = False
    crossSection = _LazyArray("crossSection")      # sum of the molecule cross sections, formed on read

    def __init__(self, depth, T, P, rangeMin, rangeMax, atmosphere=None, name='', dynamicResolution=True):
        super().__init__()
        self.rangeMin = rangeMin
        self.rangeMax = rangeMax
        self.T = T
        self.P = P
        self.depth = depth
        self.distanceFromCenter = self.P / 1013.25 * 5
        self.effectiveRangeMin = max(self.rangeMin - self.distanceFromCenter, 0)
        self.effectiveRangeMax = self.rangeMax + self.distanceFromCenter
        self.dynamicResolution = dynamicResolution
        self._set_resolution()
        if not atmosphere:
            if not Layer.hasAtmosphere:
                self.atmosphere = Atmosphere('generic')
                Layer.hasAtmosphere = self.atmosphere
            else:
                self.atmosphere = Layer.hasAtmosphere
        else:
            self.atmosphere = atmosphere
            self.hasAtmosphere = atmosphere
        _zeros_later(Layer.crossSection, self, int((rangeMax - rangeMin) / utils.BASE_RESOLUTION))
        self.progressCrossSection = False
        if not name:
            name = 'layer %s' % self.atmosphere.nextLayerName()
        self.name = name
        self.exotic = False

    def _set_resolution(self):
        if not self.dynamicResolution:
            self.resolution = utils.BASE_RESOLUTION
        else:
            self.resolution = max(10**int(np.log10((self.P / 1013.25))) * .01, utils.BASE_RESOLUTION)

    def _grid(self):
        """Grid scalars for the C ABI from the layer's CURRENT attributes (cls:672, 700, 377).  (Remembered by the attributes
        they are computed from: len(np.arange(...)) alone is 1 us, and the getters and the column ask for the grid of every
        layer on every call.)"""
        base = utils.BASE_RESOLUTION
        key = (self.distanceFromCenter, self.effectiveRangeMin, self.effectiveRangeMax, self.resolution, base, self.rangeMin, self.rangeMax)
        hit = self.__dict__.get("_grid_cache")
        if hit is not None and hit[0] == key:
            return hit[1]
        g = dict(dfc=self.distanceFromCenter, eff_min=self.effectiveRangeMin, eff_max=self.effectiveRangeMax,
                 resolution=self.resolution, base_resolution=base,
                 n_base=int((self.rangeMax - self.rangeMin) / base),
                 n_work=int((self.rangeMax - self.rangeMin) / self.resolution),
                 W=len(np.arange(0, self.distanceFromCenter, self.resolution)),
                 range_min=self.rangeMin, range_max=self.rangeMax)
        self.__dict__["_grid_cache"] = (key, g)
        return g

    def __str__(self):
        return '%s; %s' % (self.name, '; '.join(str(m) for m in self))

    def __bool__(self):
        return True

    def _layer(self):
        return self

    def _sweep_members(self):
        # Layer.absCoef (cls:707-712) walks the molecules through getAbsCoef; dirty ones are recomputed
        return [m._members() for m in self], [m.concentration for m in self]

    def _members_ready(self):
        for m in self:
            m._mark_sum_ready()

    def _column_stamp(self):
        """(values, versions) for Atmosphere's resident column: ``values`` changes when the layer's blocks on the C side must be
        re-read (a mutator changed T, P, the range, a concentration, the depth, the line lists, or somebody installed a cross
        section), ``versions`` when any of its line lists is due (resetCrossSection, cls:38-45).  A handful of attribute
        reads: this runs for every layer on every Atmosphere.transmission before the first kernel is enqueued."""
        ver = struct = 0
        conc = []
        for m in self:
            conc.append(m.concentration)
            for iso in m:
                ver += iso._inputs_version
                struct += iso._struct_version + id(iso)          # (WHICH isotopologues: a molecule swapped for another re-reads the layer)
        return (self.T, self.P, self.rangeMin, self.rangeMax, self.resolution, len(self), struct, tuple(conc), self.depth), ver

    def _merged_step_applies(self, flat, lbl):
        """settings.LAYER_STEP "merged" (default): the layer's property chain comes from one merged accumulate job when
        any of its line lists is due (all of them line-by-line: a measured cross-section table has no lines to merge).
        With every cross section current (somebody asked for each of them) the sweep kernel over those arrays is cheaper."""
        return (settings.LAYER_STEP == "merged" and bool(lbl) and len(lbl) == len(flat)
                and not _ctx().option("sweep_ieee_divisions")            # (the reference's rounding chain: per-line-list entry points only)
                and len(flat) <= nat.limit("merged_lists_per_job")      # (more line lists: the per-line-list step, up to "arrays_per_layer")
                and not any(i._xs_installed and i.progressCrossSection for i in lbl)     # an installed array is not the lines' (advisor, round 5)
                and any(not i.progressCrossSection or i._xs_deferred for i in lbl))

    def createCrossSection(self):
        """cls:684-689: sum of the molecule cross sections.  One fused layer step brings every dirty
        line list up to date (and leaves absorption coefficient and transmittance resident for the
        getters that follow); the sum is formed on the device when ``crossSection`` is read."""
        self._ensure_swept()
        n = int((self.rangeMax - self.rangeMin) / utils.BASE_RESOLUTION)
        molecules = list(self)

        def load():
            ctx = _ctx()
            tmp = []
            try:
                for molecule in molecules:
                    xs = np.ascontiguousarray(getCrossSection(molecule), dtype=np.float64)
                    if xs.shape != (n,):        # e.g. a partially overlapping xsc table (mergeArray, cls:216-219)
                        raise ValueError("operands could not be broadcast together with shapes (%d,) %s" % (n, xs.shape))
                    tmp.append(ctx.buffer(max(n, 1)).upload(xs))
                return _sum_on_device(ctx, tmp, n)
            finally:
                for b in tmp:
                    b.free()
        Layer.crossSection.defer(self, load)
        self.progressCrossSection = True

    @property
    def lineSurvey(self):
        """cls:691-696: sum of the molecule surveys (device sum, list order)."""
        return _sum_host_arrays([molecule.lineSurvey for molecule in self],
                                int((self.rangeMax - self.rangeMin) / utils.BASE_RESOLUTION))

    @property
    def yAxis(self):
        return np.zeros(int((self.rangeMax - self.rangeMin) / self.resolution))

    @property
    def xAxis(self):
        return np.linspace(self.rangeMin, self.rangeMax, int((self.rangeMax - self.rangeMin) / utils.BASE_RESOLUTION),
                           endpoint=True)

    @property
    def title(self):
        return '%s\nP: %smBars; T: %sK; depth: %scm' % (str(self), self.P, self.T, self.depth)

    def changeRange(self, rangeMin, rangeMax):
        self.rangeMin = rangeMin
        self.rangeMax = rangeMax
        self.effectiveRangeMax = self.rangeMax + self.distanceFromCenter
        self.effectiveRangeMin = max(self.rangeMin - self.distanceFromCenter, 0)
        resetData(self)

    def changeTemperature(self, temperature):
        self.T = temperature
        resetCrossSection(self)

    def changePressure(self, pressure):
        self.P = pressure
        self.distanceFromCenter = self.P / 1013.25 * 5
        self._set_resolution()
        resetData(self)

    def changeDepth(self, depth):
        self.depth = depth

    def addMolecule(self, name, isotopeDepth=1, **abundance):
        molecule = Molecule(name, self, isotopeDepth, **abundance)
        self.append(molecule)
        if totalConcentration(self) > 1:
            print('**Warning : Concentrations exceed 1.')
        if not molecule.exotic:
            molecule.getData()
        return molecule

    def returnCopy(self):
        newCopy = Layer(self.depth, self.T, self.P, self.rangeMin, self.rangeMax,
                        self.atmosphere, name=self.atmosphere.nextLayerName(), dynamicResolution=self.dynamicResolution)
        for molecule in self:
            newCopy.append(molecule.returnCopy(newCopy))      # bound to the NEW layer (the reference binds to the old one)
        return newCopy

    def returnMoleculeObjects(self):
        return list(self)

    def planck(self, temperature):
        """pyradPlanck.planckWavenumber(self.xAxis, temperature) (cls:781-782, pl:38-44), on the device."""
        ctx = _ctx()
        n = int((self.rangeMax - self.rangeMin) / utils.BASE_RESOLUTION)
        st = self.__dict__.get("_planck_state")
        if st is None:
            st = self.__dict__["_planck_state"] = _SweepState(self)
        out = st.reserve(ctx, n).buf(ctx, "planck")
        ctx.planck_dev(self.rangeMin, self.rangeMax, n, float(temperature), out)
        return out.download(n, pinned=True)


# ----------------------------------------------------------------------------------------
# Atmosphere (cls:790-821) + the column fold this build defines on it
# ----------------------------------------------------------------------------------------
class Atmosphere(list):
    def __init__(self, name):
        super().__init__()
        self.name = name

    def __str__(self):
        return self.name

    def __bool__(self):
        return True

    def addLayer(self, depth, T, P, rangeMin, rangeMax, name=None, dynamicResolution=True):
        if not name:
            name = self.nextLayerName()
        newLayer = Layer(depth, T, P, rangeMin, rangeMax, atmosphere=self, name=name,
                         dynamicResolution=dynamicResolution)
        self.append(newLayer)
        return newLayer

    def nextLayerName(self):
        return 'Layer %s' % (len(self) + 1)

    def returnLayerNames(self):
        return [layer.name for layer in self]

    def returnLayerObjects(self):
        return list(self)

    def transmission(self, surfaceSpectrum=None, surfaceTemperature=None):
        """Fold Layer.transmission bottom to top over the layers in list order:
        I <- T_i I + (1 - T_i) B(nu, T_i), I_0 = surfaceSpectrum or B(nu, surfaceTemperature).
        (The reference announces an atmosphere path but ships no driver; this is the fold of
        cls:784-787, computed by one column-sweep kernel.)"""
        layers = list(self)
        if not layers:
            raise ValueError("atmosphere has no layers")
        first = layers[0]
        for L in layers[1:]:
            if (L.rangeMin, L.rangeMax) != (first.rangeMin, first.rangeMax):
                raise ValueError("all layers of a column must share one wavenumber range")
        ctx = _ctx()
        n = int((first.rangeMax - first.rangeMin) / utils.BASE_RESOLUTION)
        if surfaceSpectrum is None and surfaceTemperature is None:
            raise ValueError("give surfaceSpectrum or surfaceTemperature")
        merged = self._transmission_merged(ctx, layers, n, surfaceSpectrum, surfaceTemperature)
        if merged is not None:
            return merged
        _compute_cross_sections([iso for L in layers for m in L for iso in m])
        for L in layers:
            L._members_ready()
        # one pass over every layer's device-resident cross sections (lbl_column_step_dev)
        desc = []
        for L in layers:
            xs, iso_mol = [], []
            for k, m in enumerate(L):
                for iso in m._members():
                    xs.append(iso._device_xsec_current(ctx, n))
                    iso_mol.append(k)
            desc.append(dict(xsec=xs, iso_mol=iso_mol, conc=[m.concentration for m in L], P=L.P, T=L.T, depth=L.depth))
        tmp = []
        try:
            out = ctx.buffer(max(n, 1)); tmp.append(out)
            I_in = None
            if surfaceSpectrum is not None:
                I_in = ctx.buffer(max(n, 1)).upload(np.ascontiguousarray(surfaceSpectrum, dtype=np.float64))
                tmp.append(I_in)
            ctx.column_step_dev(desc, first.rangeMin, first.rangeMax, n, out, I_in=I_in,
                                surface_T=float(surfaceTemperature or 0.0))
            return out.download(n, pinned=True)
        finally:
            for b in tmp:
                b.free()


    def _transmission_merged(self, ctx, layers, n, surfaceSpectrum, surfaceTemperature):
        """settings.LAYER_STEP "merged": every layer whose inputs changed gets ONE accumulate job over its merged,
        factor-weighted line lists, all of them in one launch sequence (lbl_layers_merged_accumulate_dev: the layers'
        absorption coefficients, cls:707-712), then one pass folds transmittance and emission bottom to top over those
        arrays (lbl_column_fold_dev, cls:714-716, 784-787) and leaves every layer's transmittance resident for its own
        getters.  A layer that cannot be a merged job (no line list, a measured cross-section table among its molecules, more
        line lists than a job takes, an installed cross section) brings its absorption coefficient by its own route
        (_ensure_swept) and is folded with the others.  None only when settings.LAYER_STEP is not "merged": the caller then
        goes through the per-line-list cross sections of the whole column (lbl_column_step_dev)."""
        if settings.LAYER_STEP != "merged" or ctx.option("sweep_ieee_divisions"):
            return None
        fast = self._transmission_resident(ctx, layers, n, surfaceSpectrum, surfaceTemperature)
        if fast is not None:
            return fast
        plan = []
        for L in layers:
            members, conc = L._sweep_members()
            flat = [iso for isos in members for iso in isos]
            g = L._grid()
            if (not flat or any(i.exotic for i in flat) or len(flat) > nat.limit("merged_lists_per_job")
                    or any(i._xs_installed and i.progressCrossSection for i in flat)):
                # a layer that cannot be one merged job (no line list, a measured cross-section table, more line lists than a
                # job takes, somebody's own array installed): its own route leaves the same resident absorption coefficient
                st, _ = L._ensure_swept()
                plan.append((L, st, g, members, flat, conc, None))
                continue
            _check_window(g)
            st = L.__dict__.get("_sweep_state")
            if st is None:
                st = L.__dict__["_sweep_state"] = _SweepState(L)
            st.reserve(ctx, n)
            plan.append((L, st, g, members, flat, conc, L._merged_key(flat, conc, L, g)))
        # (a layer's absorption coefficient stands as long as everything but the depth is what it was computed from)
        todo = [p for p in plan if p[6] is not None and not (isinstance(p[1].key, tuple) and p[1].key[:-1] == p[6][:-1])]
        ctx.layers_merged_accumulate_dev(
            [dict(lines=[i._device_lines(ctx) for i in flat], iso=[_iso_params(i) for i in flat],
                  grid=_engine.native_grid(g), iso_mol=[m for m, isos in enumerate(members) for _ in isos], conc=conc,
                  abs_coef=st.buf(ctx, "abs_coef")) for (L, st, g, members, flat, conc, key) in todo])
        # (the outgoing spectrum's buffer stays with the atmosphere: a hipMalloc + hipFree pair per call is 0.2 ms of a 5 ms call)
        ast = self.__dict__.get("_toa_state")
        if ast is None:
            ast = self.__dict__["_toa_state"] = _SweepState(self)
        out = ast.reserve(ctx, n).buf(ctx, "toa")
        I_in = None
        if surfaceSpectrum is not None:
            I_in = ast.buf(ctx, "I_in").upload(np.ascontiguousarray(surfaceSpectrum, dtype=np.float64))
        first = layers[0]
        # The fold in four pieces of the grid, each piece's part of the outgoing spectrum on its way to the host while the
        # next piece is folded (19 MB at the link's rate are 0.4 ms of a 5 ms call).  No layer's transmittance is written
        # (30 x 19 MB for arrays nobody has asked for): a layer's own getter makes it from the resident absorption
        # coefficient, as after changeDepth (_ensure_swept: key equal up to its last entry).
        host = ctx.host_array(n)
        pieces = 4 if n >= (1 << 16) else 1
        step = max(((n + pieces - 1) // pieces + 3) & ~3, 4)
        try:
            for lo in range(0, n, step):
                cnt = min(step, n - lo)
                ctx.column_fold_dev([p[1].bufs["abs_coef"] for p in plan], [p[0].T for p in plan], [p[0].depth for p in plan],
                                    first.rangeMin, first.rangeMax, n, out, I_in=I_in, surface_T=float(surfaceTemperature or 0.0),
                                    first=lo, count=cnt)
                out.download_async(host, cnt, lo, lo)
            for (L, st, g, members, flat, conc, key) in plan:
                if key is not None and st.key != key:
                    st.key = key[:-1] + ("absorption coefficient only",)
            for (L, st, g, members, flat, conc, key) in todo:
                for iso in flat:
                    iso._defer_cross_section()
                L._members_ready()
        finally:
            # (also when a piece raised: copies into `host` may be in flight, and its page-locked block must not go back to
            # the pool before they have landed - advisor, round 5)
            ctx.download_wait()
        self._column_remember(ctx, layers, n, plan)
        return host

    # -- the resident column: the next call's argument blocks are already on the C side (lbl_column, ABI 5) --------------
    def _column_drop(self):
        fast = self.__dict__.pop("_column_fast", None)
        if fast is not None and fast["col"].h and fast["col"].ctx.h:
            fast["col"].free()

    def _column_remember(self, ctx, layers, n, plan):
        """After a call through the general route: if every layer was one merged job, keep the column's blocks in a C-side
        handle; the next call then only looks at every layer's stamp (Layer._column_stamp)."""
        self._column_drop()
        if not plan or any(p[6] is None for p in plan):
            return
        col = ctx.column([dict(lines=[i._device_lines(ctx) for i in flat], iso=[_iso_params(i) for i in flat],
                               grid=_engine.native_grid(g), iso_mol=[m for m, isos in enumerate(members) for _ in isos], conc=conc,
                               depth=L.depth, abs_coef=st.bufs["abs_coef"]) for (L, st, g, members, flat, conc, key) in plan])
        stamps = [L._column_stamp() for L in layers]
        self.__dict__["_column_fast"] = dict(
            col=col, glob=(id(ctx), n, utils.BASE_RESOLUTION, settings.ACCURACY), layers=list(layers),
            val=[s_[0] for s_ in stamps], done=[(s_[0][:-1], s_[1]) for s_ in stamps], kbuf=[p[1].bufs["abs_coef"] for p in plan],
            n_iso=[len(p[4]) for p in plan])
        import weakref
        weakref.finalize(self, lambda c=col: c.free() if (c.h and c.ctx.h) else None)

    def _transmission_resident(self, ctx, layers, n, surfaceSpectrum, surfaceTemperature):
        """The call through the resident column handle, or None (no handle yet, another context / grid / accuracy mode, the
        list of layers changed, a layer can no longer be one merged job: the general route then rebuilds the handle)."""
        fast = self.__dict__.get("_column_fast")
        if fast is None:
            return None
        col = fast["col"]
        if (fast["glob"] != (id(ctx), n, utils.BASE_RESOLUTION, settings.ACCURACY) or not col.h or len(layers) != len(fast["layers"])
                or any(a is not b for a, b in zip(layers, fast["layers"]))):
            return None
        due, redo = [], []
        for l, L in enumerate(layers):
            val, ver = L._column_stamp()
            st = L.__dict__.get("_sweep_state")
            if st is None or st.bufs.get("abs_coef") is not fast["kbuf"][l]:
                return None
            if val != fast["val"][l]:
                if val[-1:] != fast["val"][l][-1:] and val[:-1] == fast["val"][l][:-1]:
                    redo.append((l, L, val, True))              # only the depth (changeDepth resets nothing, cls:754-755)
                else:
                    redo.append((l, L, val, False))
            due.append((val[:-1], ver) != fast["done"][l])
        for l, L, val, depth_only in redo:
            members, conc = L._sweep_members()
            flat = [iso for isos in members for iso in isos]
            if (len(flat) != fast["n_iso"][l] or any(i.exotic for i in flat) or len(conc) != len(val[7])
                    or any(i._xs_installed and i.progressCrossSection for i in flat)):
                return None
            g = L._grid()
            if g["n_base"] != n:
                return None
            _check_window(g)
            col.set_layer(l, [i._device_lines(ctx) for i in flat], [_iso_params(i) for i in flat], _engine.native_grid(g), conc,
                          L.depth, fast["kbuf"][l])
            fast["val"][l] = val
        ast = self.__dict__.get("_toa_state")
        if ast is None:
            ast = self.__dict__["_toa_state"] = _SweepState(self)
        out = ast.reserve(ctx, n).buf(ctx, "toa")
        I_in = None
        if surfaceSpectrum is not None:
            I_in = ast.buf(ctx, "I_in").upload(np.ascontiguousarray(surfaceSpectrum, dtype=np.float64))
        host = ctx.host_array(n)
        try:
            # (four pieces: eight measured no faster, and every piece is one more argument block in the library's cache of eight)
            col.transmission(due, out, host=host, I_in=I_in, surface_T=float(surfaceTemperature or 0.0),
                             pieces=4 if n >= (1 << 16) else 1)
            # bookkeeping of the object model, while the device works: what the general route does for the layers it recomputed
            for l, L in enumerate(layers):
                if not due[l]:
                    continue
                members, conc = L._sweep_members()
                flat = [iso for isos in members for iso in isos]
                key = L._merged_key(flat, conc, L, L._grid())
                L.__dict__["_sweep_state"].key = key[:-1] + ("absorption coefficient only",)
                for iso in flat:
                    iso._defer_cross_section()
                L._members_ready()
                val, ver = L._column_stamp()                  # (_defer_cross_section bumps no input version)
                fast["done"][l] = (val[:-1], ver)
        except Exception:
            self._column_drop()
            raise
        finally:
            ctx.download_wait()
        return host


# ----------------------------------------------------------------------------------------
# plot-type dispatch (cls:824-839), plot (cls:849-873) and plotSpectrum (cls:876-944).  What the
# figures show is computed on the device path (getters, transmission, band integrals); drawing it is
# matplotlib's job, imported when a figure is asked for.
# ----------------------------------------------------------------------------------------
def returnPlot(obj, propertyToPlot):
    if propertyToPlot == "transmittance":
        return getTransmittance(obj), 1
    if propertyToPlot == 'absorption coefficient':
        return getAbsCoef(obj), 0
    if propertyToPlot == 'cross section':
        return getCrossSection(obj), 0
    if propertyToPlot == 'absorbance':
        return getAbsorbance(obj), 0
    if propertyToPlot == 'optical depth':
        return getOpticalDepth(obj), 0
    if propertyToPlot == 'line survey':
        return obj.lineSurvey, 0
    return False


def _pyplot():
    try:
        import matplotlib.pyplot as plt
    except ImportError as e:            # the numbers do not need it: returnPlot / spectrumCurves
        raise ImportError("pyrad_amd.plot / plotSpectrum draw with matplotlib, which is not installed; the curves and legend "
                          "texts themselves are available without it from returnPlot() and spectrumCurves()") from e
    return plt


def _dark_axes(plt, title):
    plt.figure(figsize=(10, 6), dpi=80)
    plt.subplot(111, facecolor='xkcd:dark grey')
    plt.margins(0.01)
    plt.subplots_adjust(left=.07, bottom=.08, right=.97, top=.90)
    plt.title('%s' % title)


def _white_legend(plt, handles):
    legend = plt.legend(handles=handles, frameon=False)
    plt.setp(legend.get_texts(), color='w')


def plot(propertyToPlot, title, plotList, fill=False):
    """cls:849-873: one curve per object of ``plotList`` for a plot-type string of returnPlot (ui:407-413)."""
    plt = _pyplot()
    _dark_axes(plt, title)
    plt.xlabel('wavenumber cm-1')
    plt.ylabel(propertyToPlot)
    if propertyToPlot == 'line survey':
        plt.yscale('log')
    plt.grid('grey', linewidth=.5, linestyle=':')
    handles = []
    style = dict(linewidth=1.2, alpha=.7)
    for singlePlot, color in zip(plotList, COLOR_LIST):
        yAxis, fillAxis = returnPlot(singlePlot, propertyToPlot)
        xAxis = singlePlot.xAxis
        curve, = plt.plot(xAxis, yAxis, color=color, label='%s' % singlePlot.name, **style)
        handles.append(curve)
        plt.fill_between(xAxis, fillAxis, yAxis, color=color, alpha=.3 * fill)
        style = dict(linewidth=.7, alpha=.5)
    _white_legend(plt, handles)
    plt.show()


def _planck_axis(planckType, rangeMin, rangeMax):
    """abscissa and Planck function of a plotSpectrum type (cls:888-903); None for an unknown type"""
    if planckType == 'wavenumber':
        n = int((rangeMax - rangeMin) / utils.BASE_RESOLUTION)
        return ('wavenumber cm-1', 'Radiance Wm-2sr-1(cm-1)-1',
                np.linspace(rangeMin, rangeMax, n), lambda T: _planck_wavenumber_axis(rangeMin, rangeMax, n, T))
    if planckType == 'Hz':
        x = np.linspace(rangeMin, rangeMax, 1000)
        return 'Hertz', 'Radiance Wm-2sr-1Hz-1', x, lambda T: planckHz(x, T)
    if planckType == 'wavelength':
        x = np.linspace(rangeMin, rangeMax, int((rangeMax - rangeMin) / utils.BASE_RESOLUTION))
        return 'wavelength um', 'Radiance Wm-2sr-1um-1', x, lambda T: planckWavelength(x, T)
    return None


def _planck_wavenumber_axis(rangeMin, rangeMax, n, temperature):
    """pyradPlanck.planckWavenumber on linspace(rangeMin, rangeMax, n) (pl:38-44), on the device (lbl_planck_dev)."""
    ctx = _ctx()
    out = ctx.buffer(max(n, 1))
    try:
        ctx.planck_dev(rangeMin, rangeMax, n, float(temperature), out)
        return out.download(n)
    finally:
        out.free()


def planckHz(Hz, temp):
    """pyradPlanck.py:18-26 (Wm-2sr-1Hz-1; plot-only, host NumPy)"""
    with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
        a = 2 * h * Hz**3 / c**2
        b = h * Hz / k / temp
        return a / (np.exp(b) - 1)


def planckWavelength(lam, temp):
    """pyradPlanck.py:29-35 (wavelength in um, Wm-2sr-1um-1; plot-only, host NumPy)"""
    with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
        a = 2.0E24 * h * c ** 2 / (lam ** 5)
        b = 10 ** 6 * h * c / lam / k / temp
        return a / (np.exp(b) - 1)


def spectrumCurves(layer=None, title=None, rangeMin=None, rangeMax=None, objList=None, surfaceSpectrum=None,
                   planckTemperatureList=None, planckType='wavenumber'):
    """Everything plotSpectrum draws (cls:876-944), without drawing it: axis labels, title, and the curves in
    the reference's order - one Planck curve per temperature, labelled '<T>K : <band integral>Wm-2' with
    integrateSpectrum(y, res=(rangeMax - rangeMin) / len(y)) (cls:914), then for every object of ``objList`` its
    ``transmission(surfaceSpectrum)`` labelled '<name> : <integrateSpectrum(y, pi)>Wm-2' (cls:933-937).  Transmission,
    Planck curves on the layer axis and the integrals run on the device."""
    if layer:
        rangeMin, rangeMax, title = layer.rangeMin, layer.rangeMax, layer.title
    axis = _planck_axis(planckType, rangeMin, rangeMax)
    if axis is None:
        raise UnboundLocalError("local variable 'xAxis' referenced before assignment")     # what cls:906-912 ends in
    xlabel, ylabel, xAxis, planckFunction = axis
    if not rangeMax:
        xAxis = layer.xAxis                                                                # cls:910-911
    curves = []
    for temperature in planckTemperatureList:
        yAxis = planckFunction(float(temperature))
        power = integrateSpectrum(yAxis, res=(rangeMax - rangeMin) / len(yAxis))
        curves.append(dict(kind='planck', x=xAxis, y=yAxis, power=power, label='%sK : %sWm-2' % (temperature, round(power, 2))))
    surfacePower = None
    if objList:
        surfacePower = integrateSpectrum(surfaceSpectrum, pi)                              # cls:933
        for obj in objList:
            yAxis = obj.transmission(surfaceSpectrum)
            power = integrateSpectrum(yAxis, pi)
            curves.append(dict(kind='object', x=layer.xAxis, y=yAxis, power=power, label='%s : %sWm-2' % (obj.name, round(power, 2))))
    return dict(title=title, xlabel=xlabel, ylabel=ylabel, curves=curves, surfacePower=surfacePower)


def plotSpectrum(layer=None, title=None, rangeMin=None, rangeMax=None, objList=None, surfaceSpectrum=None,
                 planckTemperatureList=None, planckType='wavenumber', fill=False):
    """cls:876-944 (ui:373 Planck curves, ui:399 transmission through a layer): the figure of spectrumCurves()."""
    plt = _pyplot()
    spec = spectrumCurves(layer, title, rangeMin, rangeMax, objList, surfaceSpectrum, planckTemperatureList, planckType)
    _dark_axes(plt, spec['title'])
    plt.xlabel(spec['xlabel'])
    plt.ylabel(spec['ylabel'])
    handles = []
    rgb = [1.0, .6, .3]                       # the Planck curves walk through the colours like cls:904-931
    step = [-.15, .15, .15]
    objects = iter(zip(COLOR_LIST, [dict(alpha=.7, linewidth=1.2)] + [dict(alpha=.5, linewidth=1)] * len(COLOR_LIST)))
    for c in spec['curves']:
        if c['kind'] == 'planck':
            curve, = plt.plot(c['x'], c['y'], linewidth=.75, color=tuple(rgb), linestyle=':', label=c['label'])
            for i in range(3):
                if not 0 <= rgb[i] + step[i] <= 1:
                    step[i] = -step[i]
                rgb[i] += step[i]
            if rgb[0] < .3 and rgb[1] < .3 and rgb[2] < .3:
                rgb[1] += .5
                rgb[2] += .2
            if rgb[0] < .3 and rgb[1] < .3:
                rgb[1] += .4
        else:
            color, style = next(objects)
            curve, = plt.plot(c['x'], c['y'], color=color, label=c['label'], **style)
        handles.append(curve)
    _white_legend(plt, handles)
    plt.show()
