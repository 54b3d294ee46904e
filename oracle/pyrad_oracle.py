"""CPU oracle: a NumPy restatement of PyRad's hot path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, not the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
Nothing under ``pyrad_amd/`` imports it, and the product path raises when the HIP
library is missing instead of falling back to this code.

Parity status: **pinned**.  Every function below was checked in the build
container against the real reference functions imported from /root/reference
through the offline harness ``tests/golden/make_golden.py``; the
resulting vectors are committed under ``tests/golden/*.npz`` and
``tests/test_oracle_golden.py`` re-checks this file against them on every run.

Each function cites the reference lines it restates
(cls = pyradClasses.py, ls = pyradLineshape.py, int = pyradIntensity.py,
pl = pyradPlanck.py).  Arithmetic is IEEE fp64 in the reference's operation order
so results agree to the last bit wherever NumPy evaluates the same expression.
"""
from __future__ import annotations

import math

import numpy as np

# constants: cls:15-23, ls:14-19, int:3-13, pl:4-9
k = 1.38064852E-23
c = 299792458.0
h = 6.62607004e-34
pi = 3.141592653589793
t0 = 296
p0 = 1013.25
avo = 6.022140857E23
c2 = c * h * 100 / k  # int:13


# ----------------------------------------------------------------------------
# pyradIntensity (int:16-32)
# ----------------------------------------------------------------------------
def boltzmannFactors(E, t):
    """int:16-20"""
    return np.exp(-c2 * E / t) / np.exp(-c2 * E / t0)


def stimulatedEmissions(wavenumber, t):
    """int:23-27"""
    return (1 - np.exp(-c2 * wavenumber / t)) / (1 - np.exp(-c2 * wavenumber / t0))


def intensityFactor(intensity, wavenumber, t, lowerEnergy, q, q0):
    """int:30-32"""
    return intensity * (q0 / q) * (stimulatedEmissions(wavenumber, t)) * (boltzmannFactors(lowerEnergy, t))


# ----------------------------------------------------------------------------
# pyradLineshape (ls:22-76) — without the dead str/float-keyed curve caches
# ----------------------------------------------------------------------------
def gaussianHW(wavenumber, t, m):
    """ls:22-24 (1/e Doppler half-width)"""
    return wavenumber * np.sqrt(2 * k * t / m / c**2)


def lorentzHW(airHalfWidth, selfHalfWidth, P, T, q, tExponent):
    """ls:27-29"""
    return ((1 - q) * airHalfWidth + q * selfHalfWidth) * (P / p0) * (t0 / T)**tExponent


def gaussianLineShape(halfWidth, xValue):
    """ls:39"""
    return np.exp(-xValue**2 / halfWidth**2) / halfWidth / np.sqrt(pi)


def lorentzLineShape(halfWidth, xValue):
    """ls:52"""
    return halfWidth / pi / (xValue**2 + halfWidth**2)


def pseudoVoigtParams(gHW, lHW):
    """ls:59-71: returns (fValue, nValue)."""
    gFW = 2 * gHW
    lFW = 2 * lHW
    fValue = (gFW**5 + 2.69269 * gFW**4 * lFW +
              2.42843 * gFW**3 * lFW**2 +
              4.47163 * gFW**2 * lFW**3 +
              .07842 * gFW * lFW**4 + lFW**5)**.2
    nValue = 1.36603 * (lFW / fValue) - .47719 * (lFW / fValue)**2 + .11116 * (lFW / fValue)**3
    return fValue, nValue


def pseudoVoigtShape(gHW, lHW, xValue):
    """ls:58-76"""
    fValue, nValue = pseudoVoigtParams(gHW, lHW)
    gCurve = gaussianLineShape(fValue / 2, xValue)
    lCurve = lorentzLineShape(fValue / 2, xValue)
    return nValue * lCurve + (1 - nValue) * gCurve


# ----------------------------------------------------------------------------
# pyradPlanck (pl:12-44)
# ----------------------------------------------------------------------------
def planckWavenumber(n, temp):
    """pl:38-44 (0/0 at n = 0 yields NaN exactly as the reference does under pl:2)"""
    n = np.asarray(n, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        a = 2E8 * h * c**2 * n**3
        b = 100 * h * c * n / k / float(temp)
        return a / (np.exp(b) - 1)


def planckHz(Hz, temp):
    """pl:18-25"""
    Hz = np.asarray(Hz, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        return (2 * h * Hz**3 / c**2) / (np.exp(h * Hz / k / temp) - 1)


def planckWavelength(lam, temp):
    """pl:28-35"""
    lam = np.asarray(lam, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        return (2.0E24 * h * c ** 2 / (lam ** 5)) / (np.exp(10 ** 6 * h * c / lam / k / temp) - 1)


# ----------------------------------------------------------------------------
# Layer grid definition (cls:648-676, 698-705, 745-752)
# ----------------------------------------------------------------------------
def layer_grid(P, range_min, range_max, base_resolution, dynamic_resolution=True):
    """Scalars of cls:655-662, 672, 700 plus the window length W = len(arange(0,dfc,res))
    used at cls:377."""
    dfc = P / 1013.25 * 5                                   # cls:655
    eff_min = max(range_min - dfc, 0)                       # cls:656
    eff_max = range_max + dfc                               # cls:657
    if not dynamic_resolution:
        res = base_resolution                               # cls:660
    else:
        res = max(10**int(np.log10((P / 1013.25))) * .01, base_resolution)   # cls:662
    n_base = int((range_max - range_min) / base_resolution)  # cls:672
    n_work = int((range_max - range_min) / res)              # cls:700
    W = len(np.arange(0, dfc, res))                          # cls:377
    return dict(dfc=dfc, eff_min=eff_min, eff_max=eff_max, resolution=res,
                n_base=n_base, n_work=n_work, W=W, base_resolution=base_resolution,
                range_min=range_min, range_max=range_max)


def x_axis(range_min, range_max, base_resolution):
    """cls:702-705 with the float ``num`` truncated (what pre-1.18 NumPy did)."""
    return np.linspace(range_min, range_max, int((range_max - range_min) / base_resolution), endpoint=True)


# ----------------------------------------------------------------------------
# Per-line derived quantities (cls:252-263, 294-296, 378-390)
# ----------------------------------------------------------------------------
def line_quantities(lines, T, P, conc, molmass, range_min, resolution):
    """Vectorised cls:252-263 + cls:378-390 for a whole SoA line list.

    Returns dict(broadened, lhw, ghw, ratio, regime, index).  regime: 0 Gaussian
    (ratio < .01), 1 Lorentz (ratio > 100), 2 pseudo-Voigt (cls:379-387).
    """
    nu = lines["nu"]
    broadened = nu + lines["delta_air"] * P / p0                                     # cls:254
    lhw = ((1 - conc) * lines["gamma_air"] + conc * lines["gamma_self"]) \
        * (P / p0) * (t0 / T) ** lines["n_air"]                                      # cls:258-259
    m = molmass / 1000 / avo                                                          # cls:296
    ghw = broadened * np.sqrt(2 * k * T / m / c ** 2)                                 # cls:263
    ratio = lhw / ghw                                                                 # cls:378
    regime = np.where(ratio < .01, 0, np.where(ratio > 100, 1, 2)).astype(np.int32)   # cls:379-387
    index = ((nu - range_min) / resolution).astype(np.int64)                          # cls:390 (trunc toward 0)
    return dict(broadened=broadened, lhw=lhw, ghw=ghw, ratio=ratio, regime=regime, index=index)


def _right_curve(regime, ghw, lhw, xValues):
    if regime == 0:
        return gaussianLineShape(ghw, xValues)                # cls:380
    if regime == 1:
        return lorentzLineShape(lhw, xValues)                 # cls:383
    return pseudoVoigtShape(ghw, lhw, xValues)                # cls:386


# ----------------------------------------------------------------------------
# The hot loop: Isotope.createCrossSection (cls:361-407)
# ----------------------------------------------------------------------------
def create_cross_section_scalar(lines, T, P, conc, molmass, q_T, q296, grid, regrid=True):
    """Faithful restatement of cls:361-405: same per-line, per-point loop order, a
    Python ``for`` over dx with per-point bounds tests.  This is the CPU baseline
    that is timed (the reference itself cannot travel to the GPU box)."""
    res = grid["resolution"]
    crossSection = np.zeros(grid["n_work"])                    # cls:367 (yAxis, cls:700)
    lq = line_quantities(lines, T, P, conc, molmass, grid["range_min"], res)
    counts = [0, 0, 0]
    nu = lines["nu"]
    for i in range(len(nu)):
        xValues = np.arange(0, grid["dfc"], res)               # cls:377
        regime = int(lq["regime"][i])
        rightCurve = _right_curve(regime, lq["ghw"][i], lq["lhw"][i], xValues)
        counts[regime] += 1
        intensity = intensityFactor(lines["sw"][i], lq["broadened"][i], T,
                                    lines["elower"][i], q_T, q296)   # cls:388-389
        arrayIndex = int((nu[i] - grid["range_min"]) / res)    # cls:390
        arrayLength = len(crossSection) - 1                    # cls:391
        if 0 <= arrayIndex <= arrayLength:                     # cls:392
            crossSection[arrayIndex] = crossSection[arrayIndex] + rightCurve[0] * intensity
        for dx in range(1, len(rightCurve) - 1):               # cls:394
            rightIndex = arrayIndex + dx
            leftIndex = arrayIndex - dx
            if 0 <= rightIndex <= arrayLength:                 # cls:397
                crossSection[rightIndex] += rightCurve[dx] * intensity
            if 0 <= leftIndex <= arrayLength:                  # cls:399
                crossSection[leftIndex] += rightCurve[dx] * intensity
    out = regrid_to_base(crossSection, grid) if regrid else crossSection
    return out, tuple(counts)


def create_cross_section(lines, T, P, conc, molmass, q_T, q296, grid, regrid=True):
    """Vectorised restatement of cls:361-405 (slice adds instead of per-point Python).
    Per grid point the summation order over lines is the reference's (line order),
    so it matches the scalar form bit for bit."""
    res = grid["resolution"]
    n = grid["n_work"]
    crossSection = np.zeros(n)
    lq = line_quantities(lines, T, P, conc, molmass, grid["range_min"], res)
    xValues = np.arange(0, grid["dfc"], res)                   # cls:377
    W = len(xValues)
    nu = lines["nu"]
    inten = intensityFactor(lines["sw"], lq["broadened"], T, lines["elower"], q_T, q296)
    counts = np.bincount(lq["regime"], minlength=3)
    if W == 0:
        raise IndexError("index 0 is out of bounds for axis 0 with size 0")   # rightCurve[0], cls:393
    for i in range(len(nu)):
        cidx = int(lq["index"][i])
        H = max(W - 2, 0)
        if cidx + H < 0 or cidx - H > n - 1:
            continue
        curve = _right_curve(int(lq["regime"][i]), lq["ghw"][i], lq["lhw"][i], xValues) * inten[i]
        if 0 <= cidx <= n - 1:
            crossSection[cidx] += curve[0]
        if W > 2:
            # right wing: indices cidx+1 .. cidx+W-2
            lo = max(cidx + 1, 0)
            hi = min(cidx + W - 2, n - 1)
            if hi >= lo:
                crossSection[lo:hi + 1] += curve[lo - cidx:hi - cidx + 1]
            # left wing: indices cidx-1 .. cidx-(W-2)
            lo = max(cidx - (W - 2), 0)
            hi = min(cidx - 1, n - 1)
            if hi >= lo:
                crossSection[lo:hi + 1] += curve[cidx - hi:cidx - lo + 1][::-1]
    out = regrid_to_base(crossSection, grid) if regrid else crossSection
    return out, (int(counts[0]), int(counts[1]), int(counts[2]))


def regrid_to_base(work, grid):
    """cls:401-405: np.interp from linspace(min,max,N_work) onto xAxis (N_base)."""
    xa = x_axis(grid["range_min"], grid["range_max"], grid["base_resolution"])
    xw = np.linspace(grid["range_min"], grid["range_max"], grid["n_work"], endpoint=True)
    return np.interp(xa, xw, work)


def eval_count(index, W, n_work):
    """Exact number of (line, grid point) contributions (SURVEY.md §8d 'unit of work')."""
    index = np.asarray(index, dtype=np.int64)
    if W <= 0:
        return 0
    H = max(W - 2, 0)
    lo = np.maximum(index - H, 0)
    hi = np.minimum(index + H, n_work - 1)
    cnt = np.maximum(hi - lo + 1, 0)      # W <= 2: range(1, W-1) is empty, centre only
    return int(cnt.sum())


def line_survey(nu_in_reader_order, sw, range_min, range_max, resolution, base_resolution):
    """cls:409-428 createLineSurvey: S added into the bin int((nu - rangeMin) / layer.resolution) of an
    array of int((max - min) / BASE) bins, in the order the lines were appended (cls:352-357: the
    reader's dict order); bins outside [0, len-1] are dropped (isBetween, cls:842-846).  With
    resolution != BASE the index and the array length use different steps - kept as is."""
    survey = np.zeros(int((range_max - range_min) / base_resolution))       # cls:416
    last = len(survey) - 1
    for v, s in zip(nu_in_reader_order, sw):
        idx = int((v - range_min) / resolution)                             # cls:423
        if 0 <= idx <= last:
            survey[idx] = survey[idx] + s
    return survey


# ----------------------------------------------------------------------------
# Concentration setters (cls:543-560)
# ----------------------------------------------------------------------------
def concentration(**abundance):
    """cls:453-463 + cls:543-560, including the ppb x 1e-8 quirk (cls:554)."""
    conc = 0
    for key, v in abundance.items():
        if key == 'ppm':
            conc = v * 10**-6
        elif key == 'ppb':
            conc = v * 10**-8
        elif key in ('percentage', 'perc', '%'):
            conc = v / 100
        elif key == 'concentration':
            conc = (v * 1E6) * 10**-6                     # cls:559 -> cls:549
    return conc


# ----------------------------------------------------------------------------
# Aggregation + optical properties (cls:322-340, 566-606, 707-732, 784-787, 26-29)
# ----------------------------------------------------------------------------
def abs_coef(xsec, conc, P, T):
    """cls:324 / cls:583"""
    return xsec * conc * P / 1E4 / k / T


def transmittance(absCoef, depth):
    """cls:328 / 587 / 716"""
    return np.exp(-absCoef * depth)


def absorbance(trans):
    """cls:340 / 598 / 720"""
    with np.errstate(divide="ignore"):
        return np.log10(1 / trans)


def optical_depth(trans):
    """cls:76"""
    with np.errstate(divide="ignore"):
        return -np.log(trans)


def transmission(trans, surfaceSpectrum, planck_layer):
    """cls:784-787"""
    transmitted = trans * surfaceSpectrum
    emitted = (1 - trans) * planck_layer
    return transmitted + emitted


def integrateSpectrum(spectrum, unitAngle=pi, res=0.01):
    """cls:26-29"""
    value = np.sum(np.nan_to_num(spectrum))
    return value * unitAngle * res


def layer_properties(cfg, scalar=False):
    """Run one gas cell (a dict from pyrad_amd.synthetic.config_*) end to end:
    per-molecule xsec -> absCoef -> layer absCoef (cls:707-712) -> transmittance.
    ``molecules[i]`` may carry ``q_T``/``q296``/``molmass`` directly; otherwise they are
    looked up in pyrad_amd.synthetic."""
    from pyrad_amd import synthetic
    grid = layer_grid(cfg["P"], cfg["range_min"], cfg["range_max"], cfg["base_resolution"],
                      cfg.get("dynamic_resolution", True))
    fn = create_cross_section_scalar if scalar else create_cross_section
    k_layer = np.zeros(grid["n_base"])
    xsecs, counts = [], []
    for mol in cfg["molecules"]:
        sp = synthetic.SPECIES[mol["species"]]
        conc = concentration(**mol["conc"])
        lines = select_window(mol["lines"], grid["eff_min"], grid["eff_max"])
        xs, cnt = fn(lines, cfg["T"], cfg["P"], conc, mol.get("molmass", sp["molmass"]),
                     mol.get("q_T", synthetic.q_value(mol["species"], cfg["T"])),
                     mol.get("q296", sp["q296"]), grid)
        xsecs.append(xs)
        counts.append(cnt)
        k_layer = k_layer + abs_coef(xs, conc, cfg["P"], cfg["T"])      # cls:709-712
    trans = transmittance(k_layer, cfg["depth"])
    return dict(grid=grid, xsec=xsecs, counts=counts, abs_coef=k_layer, transmittance=trans)


def select_window(lines, lo, hi):
    """strict lo < nu < hi selection of ut:437-438."""
    m = (lines["nu"] > lo) & (lines["nu"] < hi)
    if m.all():
        return lines
    return {k_: v[m] for k_, v in lines.items()}


def column_transmission(layer_trans, layer_T, xaxis, surface_T):
    """Fold of cls:784-787 bottom-to-top: I0 = B(nu, T_surface);
    I <- T_i I + (1 - T_i) B(nu, T_i)."""
    I = planckWavenumber(xaxis, surface_T)
    for tr, T in zip(layer_trans, layer_T):
        I = transmission(tr, I, planckWavenumber(xaxis, T))
    return I


def cross_section_at_points(lines, T, P, conc, molmass, q_T, q296, grid, points):
    """Work-grid cross section at a few selected grid indices only (full-size parity checks):
    for each point j the lines with |j - index| <= W-2 contribute rightCurve[|j - index|] * intensity
    exactly as cls:392-400 deposits them (same profile sample x = |dx| * res, cls:377), summed in
    line order."""
    res = grid["resolution"]
    W = grid["W"]
    H = max(W - 2, 0)
    lq = line_quantities(lines, T, P, conc, molmass, grid["range_min"], res)
    inten = intensityFactor(lines["sw"], lq["broadened"], T, lines["elower"], q_T, q296)
    out = np.zeros(len(points))
    idx = lq["index"]
    for n, j in enumerate(points):
        sel = np.nonzero(np.abs(idx - j) <= H)[0]
        if W < 1 or sel.size == 0:
            continue
        dx = np.abs(idx[sel] - j)
        x = 0 + dx * res                                    # element dx of np.arange(0, dfc, res)
        reg = lq["regime"][sel]
        g, l = lq["ghw"][sel], lq["lhw"][sel]
        f, eta = pseudoVoigtParams(g, l)
        with np.errstate(under="ignore"):
            curve = np.where(reg == 0, gaussianLineShape(g, x),
                             np.where(reg == 1, lorentzLineShape(l, x),
                                      eta * lorentzLineShape(f / 2, x) + (1 - eta) * gaussianLineShape(f / 2, x)))
        terms = curve * inten[sel]
        s = 0.0
        for t in terms:                                      # line order, like the reference's loop
            s = s + t
        out[n] = s
    return out


# ----------------------------------------------------------------------------
# measured cross sections ("xsc" molecules): cls:165-233, cls:466-505
# ----------------------------------------------------------------------------
def merge_array(newX, oldX, oldY):
    """cls:165-233 restated with offsets instead of list building.  Abscissae are compared after
    Python ``round(x, 2)``; the table is copied index-for-index from the first shared value and
    its last overlapping sample is left out; ``list.index`` failures (ValueError) and running off
    the table (IndexError) surface as in the reference."""
    nx = [round(float(v), 2) for v in np.asarray(newX).tolist()]
    ox = [round(float(v), 2) for v in np.asarray(oldX).tolist()]
    y = list(np.asarray(oldY).tolist()) if not isinstance(oldY, list) else oldY
    if max(nx) < min(ox) or min(nx) > max(ox):
        return np.zeros(len(nx))
    starts_before = min(nx) <= min(ox)
    ends_after = max(nx) >= max(ox)
    dst0 = nx.index(min(ox)) if starts_before else 0
    src0 = 0 if starts_before else ox.index(min(nx))
    src1 = (len(ox) - 1) if ends_after else (src0 + len(nx) - 1)
    tail = max(len(nx) - ((dst0 + len(ox) - 1) if ends_after else (len(nx) - 1)), 0)   # [0] * negative == [] (cls:229)
    body = []
    for i in range(src0, src1):
        body.append(y[i])                       # IndexError if the table is too short
    out = np.zeros(dst0 + len(body) + tail)
    if body:
        out[dst0:dst0 + len(body)] = body
    return out


def xsc_layer_state(temp_text, pressure_torr_text):
    """cls:479-481: the layer temperature (int K) and pressure (mbar) an xsc file imposes."""
    return int(float(temp_text)), float(pressure_torr_text) / 0.75006


def xsc_cross_section(layer_x_axis, range_min, range_max, res, wavenumber, intensity):
    """cls:492-499: table -> 0.01 cm^-1 axis (np.interp only if the file is coarser) -> layer axis."""
    axis = np.arange(range_min, range_max, .01)
    table = np.interp(axis, wavenumber, intensity) if res > .01 else intensity
    return merge_array(layer_x_axis, axis, table)
