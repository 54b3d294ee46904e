"""Seeded synthetic HITRAN-shaped inputs.

No HITRAN data exists offline, so "identical HITRAN inputs" means identical
synthetic line lists in the reference's own schema: the per-line dict that
``pyradUtilities.gatherData`` returns (ut:421-448), the ``{int T: Q}`` table of
``readQFile`` (ut:451-461) and the 8-field list of ``readMolParams``
(ut:464-477).  Everything is drawn from ``numpy.random.default_rng(seed)`` so the
golden generator, the oracle, the HIP path and the bench all see the same bytes.

This module is data generation only; it contains no line-shape arithmetic.
"""
from __future__ import annotations

import numpy as np

# global-iso id, short name, HITRAN molecule number, molar mass [g/mol], Q(296 K),
# exponent of the synthetic partition function Q(T) = Q296 (T/296)^beta.
# Nominal values: parity only needs every consumer to share them.
SPECIES = {
    "co2": dict(global_iso=7, mol_num=2, molmass=43.98983, q296=286.09, beta=1.0),
    "h2o": dict(global_iso=1, mol_num=1, molmass=18.010565, q296=174.58, beta=1.5),
    "ch4": dict(global_iso=32, mol_num=6, molmass=16.0313, q296=590.48, beta=1.5),
    "o3": dict(global_iso=16, mol_num=3, molmass=47.984745, q296=3483.71, beta=1.5),
    # second CO2 isotopologue (636) for the isotopeDepth=2 composition goldens
    "co2_636": dict(global_iso=8, mol_num=2, molmass=44.993185, q296=576.64, beta=1.0),
}
BY_GLOBAL_ISO = {v["global_iso"]: (k, v) for k, v in SPECIES.items()}

FIELDS = ("nu", "sw", "a", "elower", "gamma_air", "gamma_self", "delta_air", "n_air")


def q_table(species: str, t_max: int = 3000) -> dict:
    """``{int T: Q(T)}`` for T = 1..t_max, the shape of ut:451-461."""
    sp = SPECIES[species]
    t = np.arange(1, t_max + 1, dtype=np.float64)
    q = sp["q296"] * (t / 296.0) ** sp["beta"]
    return {int(k): float(v) for k, v in zip(t, q)}


def q_value(species: str, T: int) -> float:
    sp = SPECIES[species]
    return float(sp["q296"] * (np.float64(int(T)) / 296.0) ** sp["beta"])


def mol_params(species: str) -> list:
    """The list ``readMolParams`` returns (ut:464-477):
    [globalIso, shortName, molNum, isoN, abundance, q296, gj, molMass]."""
    sp = SPECIES[species]
    return [sp["global_iso"], species.split("_")[0].upper(), sp["mol_num"], 1, 1.0,
            sp["q296"], 1, sp["molmass"]]


def make_lines(seed: int, n_lines: int, eff_min: float, eff_max: float,
               decimals: int | None = 6) -> dict:
    """Structure-of-arrays line list, sorted by wavenumber, unique.

    70 % of the centres are drawn from Gaussian band clusters (centres every
    100 cm^-1, sigma 15) and 30 % uniformly over (eff_min, eff_max); values that
    fall outside the open interval are redrawn uniformly (the reference's reader
    keeps only min < nu < max, ut:437-438).  ``decimals`` rounds nu the way a
    HITRAN file would (6 decimals) so the CSV round trip is exact.
    """
    rng = np.random.default_rng(seed)
    n_band = int(round(0.7 * n_lines))
    centres = np.arange(np.floor(eff_min / 100.0) * 100.0 + 50.0, eff_max + 100.0, 100.0)
    nu_band = rng.choice(centres, size=n_band) + rng.normal(0.0, 15.0, size=n_band)
    nu_uni = rng.uniform(eff_min, eff_max, size=n_lines - n_band)
    nu = np.concatenate([nu_band, nu_uni])
    if decimals is not None:
        nu = np.round(nu, decimals)
    for _ in range(64):
        bad = ~((nu > eff_min) & (nu < eff_max))
        # duplicates collapse in the reference's dict (ut:447): make them unique here
        order = np.argsort(nu, kind="stable")
        dup = np.zeros(n_lines, dtype=bool)
        dup[order[1:]] = np.diff(nu[order]) == 0.0
        bad |= dup
        if not bad.any():
            break
        redraw = rng.uniform(eff_min, eff_max, size=int(bad.sum()))
        nu[bad] = np.round(redraw, decimals) if decimals is not None else redraw
    else:  # pragma: no cover
        raise RuntimeError("could not draw unique in-range wavenumbers")
    order = np.argsort(nu, kind="stable")
    nu = nu[order]
    out = {
        "nu": nu,
        "sw": 10.0 ** rng.uniform(-28.0, -19.0, size=n_lines),
        "a": rng.uniform(0.01, 10.0, size=n_lines),          # Einstein A: carried, never used (cls:245)
        "gamma_air": rng.uniform(0.05, 0.10, size=n_lines),
        "gamma_self": rng.uniform(0.06, 0.12, size=n_lines),
        "n_air": rng.uniform(0.5, 0.8, size=n_lines),
        "delta_air": rng.uniform(-0.01, 0.0, size=n_lines),
        "elower": rng.uniform(0.0, 5000.0, size=n_lines),
    }
    return {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in out.items()}


def lines_to_reference_dict(lines: dict, lo: float | None = None, hi: float | None = None) -> dict:
    """SoA -> the ``{nu: {...}}`` dict of ut:421-448 (strict bounds, last duplicate wins)."""
    d = {}
    for i in range(len(lines["nu"])):
        nu = float(lines["nu"][i])
        if lo is not None and not (lo < nu):
            continue
        if hi is not None and not (nu < hi):
            continue
        d[nu] = {
            "isotope": 1,
            "intensity": float(lines["sw"][i]),
            "einsteinA": float(lines["a"][i]),
            "airHalfWidth": float(lines["gamma_air"][i]),
            "selfHalfWidth": float(lines["gamma_self"][i]),
            "lowerEnergy": float(lines["elower"][i]),
            "tempExponent": float(lines["n_air"][i]),
            "pressureShift": float(lines["delta_air"][i]),
        }
    return d


# --------------------------------------------------------------------------- #
# BASELINE.json configurations.  Each returns a plain dict that
# both the oracle and the HIP path consume.
# --------------------------------------------------------------------------- #

def layer_window(P: float, range_min: float, range_max: float):
    """effective range of cls:655-657."""
    dfc = P / 1013.25 * 5
    return max(range_min - dfc, 0), range_max + dfc


def config_c1(n_lines: int = 4096):
    lo, hi = layer_window(1013.25, 600, 700)
    return dict(name="C1", depth=10.0, T=296, P=1013.25, range_min=600, range_max=700,
                base_resolution=0.01, dynamic_resolution=True,
                molecules=[dict(species="co2", conc=dict(ppm=400), lines=make_lines(1, n_lines, lo, hi))])


def config_c2(n_lines: int = 65536, range_min=500, range_max=900, seed=2):
    lo, hi = layer_window(1013.25, range_min, range_max)
    return dict(name="C2", depth=10.0, T=296, P=1013.25, range_min=range_min, range_max=range_max,
                base_resolution=0.001, dynamic_resolution=False,
                molecules=[dict(species="co2", conc=dict(ppm=400), lines=make_lines(seed, n_lines, lo, hi))])


def config_c3(n_lines: int = 131072, range_min=100, range_max=2500, seeds=(3, 4, 5)):
    lo, hi = layer_window(1013.25, range_min, range_max)
    mols = [("co2", dict(ppm=400)), ("h2o", dict(percentage=1)), ("ch4", dict(ppm=1.8))]
    return dict(name="C3", depth=10.0, T=296, P=1013.25, range_min=range_min, range_max=range_max,
                base_resolution=0.001, dynamic_resolution=False,
                molecules=[dict(species=s, conc=c, lines=make_lines(seed, n_lines, lo, hi))
                           for (s, c), seed in zip(mols, seeds)])


def config_c5(n_layers: int = 30, n_lines: int = 131072, range_min=100, range_max=2500,
              seeds=(3, 4, 5)):
    """30-layer column: P log-spaced 1013.25 -> 10 mbar, T integer K from 288 down to
    217 (linear in log P, rounded), hydrostatic thickness, H2O 1 % -> 5 ppm,
    CO2 400 ppm, O3 0.03 -> 8 ppm with height.  The same three line sets are used in
    every layer, drawn over the widest (surface) window."""
    lo, hi = layer_window(1013.25, range_min, range_max)
    line_sets = {s: make_lines(seed, n_lines, lo, hi) for s, seed in zip(("h2o", "co2", "o3"), seeds)}
    P = np.exp(np.linspace(np.log(1013.25), np.log(10.0), n_layers))
    P[0], P[-1] = 1013.25, 10.0                       # exact end points (exp(log(x)) is off by an ulp)
    frac = (np.log(1013.25) - np.log(P)) / (np.log(1013.25) - np.log(10.0))
    T = np.rint(288.0 - frac * (288.0 - 217.0)).astype(int)
    # hydrostatic thickness of a layer centred on P_i with scale height R T / (M g)
    edges = np.exp(np.linspace(np.log(1013.25), np.log(10.0), n_layers + 1) if n_layers > 1
                   else np.log([1013.25, 10.0]))
    H_cm = 8.314462618 * T / (0.0289644 * 9.80665) * 100.0
    depth = H_cm * np.log(edges[:-1] / edges[1:])
    h2o = np.exp(np.log(1e-2) + frac * (np.log(5e-6) - np.log(1e-2)))
    o3 = np.exp(np.log(0.03e-6) + frac * (np.log(8e-6) - np.log(0.03e-6)))
    layers = []
    for i in range(n_layers):
        layers.append(dict(name="C5.%d" % i, depth=float(depth[i]), T=int(T[i]), P=float(P[i]),
                           range_min=range_min, range_max=range_max,
                           base_resolution=0.001, dynamic_resolution=False,
                           molecules=[
                               dict(species="h2o", conc=dict(concentration=float(h2o[i])), lines=line_sets["h2o"]),
                               dict(species="co2", conc=dict(ppm=400), lines=line_sets["co2"]),
                               dict(species="o3", conc=dict(concentration=float(o3[i])), lines=line_sets["o3"]),
                           ]))
    return dict(name="C5", surface_T=288, layers=layers)
