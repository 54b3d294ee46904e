"""One process per GPU: rank discovery, the out-of-band exchange of the RCCL unique id, and
the host-side shard arithmetic of the contiguous-range grid sharding.

The data path has exactly one collective, the RCCL all-gather of the per-rank spectrum
shards (``lbl_allgather_dev``).  Everything here is control plane: it moves 128 bytes once.
The launcher contract is torchrun's (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
in the environment); the ranks of one node share a filesystem, so the id travels through a
file — no second HIP runtime (PyTorch's bundled one) is pulled into the process.
"""
from __future__ import annotations

import os
import time

import numpy as np


def env_world():
    """(rank, local_rank, world_size) from the launcher's environment (1 process if unset)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    return rank, local, world


class FileRendezvous:
    """Single-node rendezvous through a directory.  The key is unique per launch: all workers
    of one torchrun share their parent (the elastic agent) and the master port."""

    def __init__(self, rank: int, world: int, key: str | None = None, root: str | None = None, timeout: float = 120.0):
        self.rank, self.world, self.timeout = rank, world, timeout
        if key is None:
            key = os.environ.get("PYRAD_RENDEZVOUS_KEY")
        if key is None:
            key = "%s_%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"),
                                os.getppid())
        root = root or os.environ.get("PYRAD_RENDEZVOUS_DIR", "/tmp")
        self.dir = os.path.join(root, "pyrad_amd_rdzv_%s" % key)
        os.makedirs(self.dir, exist_ok=True)

    def broadcast(self, name: str, data: bytes | None) -> bytes:
        """Rank 0 publishes ``data``; every rank returns it."""
        path = os.path.join(self.dir, name)
        if self.rank == 0:
            tmp = path + ".tmp"
            with open(tmp, "wb") as f:
                f.write(data)
            os.replace(tmp, path)          # atomic publish
            return data
        t0 = time.time()
        while not os.path.isfile(path):
            if time.time() - t0 > self.timeout:
                raise TimeoutError("rendezvous: %s not published within %.0f s" % (path, self.timeout))
            time.sleep(0.005)
        with open(path, "rb") as f:
            return f.read()

    def arrive(self, name: str):
        """File barrier (control plane only; the timed region uses an RCCL barrier)."""
        open(os.path.join(self.dir, "%s.%d" % (name, self.rank)), "wb").close()
        t0 = time.time()
        while True:
            if all(os.path.isfile(os.path.join(self.dir, "%s.%d" % (name, r))) for r in range(self.world)):
                return
            if time.time() - t0 > self.timeout:
                raise TimeoutError("rendezvous barrier '%s' timed out" % name)
            time.sleep(0.005)

    def cleanup(self):
        """Every rank acknowledges that it is past its last wait; rank 0 removes the directory
        only after all acknowledgements (otherwise it could delete files a slower rank still polls)."""
        open(os.path.join(self.dir, "ack.%d" % self.rank), "wb").close()
        if self.rank != 0:
            return
        t0 = time.time()
        while not all(os.path.isfile(os.path.join(self.dir, "ack.%d" % r)) for r in range(self.world)):
            if time.time() - t0 > self.timeout:
                return                      # leave the files rather than hang
            time.sleep(0.005)
        try:
            for f in os.listdir(self.dir):
                os.remove(os.path.join(self.dir, f))
            os.rmdir(self.dir)
        except OSError:
            pass


# ----------------------------------------------------------------------------------------
# shard arithmetic (pure host logic; exercised with gloo on CPU in tests/test_sharding.py)
# ----------------------------------------------------------------------------------------
def shard_bounds(n: int, world_size: int, rank: int):
    """Equal contiguous shards of an n-point grid padded to world_size * S.
    Rank r owns [r*S, min((r+1)*S, n)).  Returns (S, first, count)."""
    S = -(-int(n) // int(world_size))
    first = min(rank * S, n)
    count = max(min((rank + 1) * S, n) - first, 0)
    return S, first, count


class ShardPlan:
    """Contiguous grid-range shards of one work grid: ``bounds[r] = (first, count)`` for every
    rank, ``S`` = the per-rank slot of the all-gather (max count; every rank sends S doubles).
    ``in_place``: all shards are S long and start at r*S (the last may be short), so the
    all-gather can run in place on a buffer of world*S doubles; otherwise every rank sends S
    doubles starting at its ``first`` (its buffer is S longer than the grid) into a separate
    gathered buffer of world*S doubles whose slot r holds rank r's shard in its first count_r
    entries (``assemble`` / lbl_gather_compact_dev bring it back to grid order)."""

    def __init__(self, n: int, bounds, rank: int):
        self.n = int(n)
        self.bounds = [(int(f), int(c)) for f, c in bounds]
        self.world = len(self.bounds)
        self.rank = int(rank)
        self.S = max(max(c for _, c in self.bounds), 1)
        self.in_place = all(f == min(r * self.S, self.n) for r, (f, _) in enumerate(self.bounds))
        pos = 0
        for f, c in self.bounds:
            if f != pos or c < 0:
                raise ValueError("shards must tile the grid contiguously: %r" % (self.bounds,))
            pos += c
        if pos != self.n:
            raise ValueError("shards cover %d of %d grid points" % (pos, self.n))

    @property
    def first(self):
        return self.bounds[self.rank][0]

    @property
    def count(self):
        return self.bounds[self.rank][1]

    def assemble(self, gathered):
        """padded gathered buffer (world*S) -> the n-point spectrum in grid order"""
        g = np.asarray(gathered)
        return np.concatenate([g[r * self.S: r * self.S + c] for r, (_, c) in enumerate(self.bounds)])[:self.n]


def equal_plan(n: int, world_size: int, rank: int) -> ShardPlan:
    """Equal-width shards (the in-place all-gather layout)."""
    return ShardPlan(n, [shard_bounds(n, world_size, r)[1:] for r in range(world_size)], rank)


SPAN = 256          # grid points a wavefront owns at R = 4 (cost model granularity)
ALIGN = 1024        # shard boundaries are multiples of one workgroup's points (4 spans)


def span_costs(centre_index, H: int, n: int, has_gaussian=None) -> np.ndarray:
    """Estimated K2 wave-instructions per span of 256 grid points for one line list, the host
    model of lbl_api.hip's group_schedule: near lines (within 4 half-spans of the span) are
    evaluated point by point, 5R instructions, plus a Gaussian pass of ~25 on the spans their
    Gaussian part reaches (~0.6 of the near spans) when the line has one (pseudo-Voigt or Gaussian
    regime); lines reached through the far-field series ~1.6; a fixed part per span.
    ``centre_index`` = sorted int centre indices (cls:390); ``has_gaussian`` = per-line bool in the
    same order (None: every line is taken to have a Gaussian part).  The regime matters for the
    balance: below ~630 cm^-1 at 1 atm CO2 lines are pure Lorentz (cls:382) and cost 40 % of a
    pseudo-Voigt line."""
    c = np.asarray(centre_index, dtype=np.int64)
    n_spans = -(-int(n) // SPAN)
    lo = np.arange(n_spans, dtype=np.int64) * SPAN
    hi = np.minimum(lo + SPAN, n) - 1
    H = int(H)
    reach = np.searchsorted(c, hi + H, "right") - np.searchsorted(c, lo - H, "left")
    near_half = min(H, 4 * (SPAN // 2))
    a = np.searchsorted(c, lo - near_half, "left")
    b = np.searchsorted(c, hi + near_half, "right")
    near = np.minimum(b - a, reach)
    if has_gaussian is None:
        gauss = near
    else:
        pre = np.concatenate([[0], np.cumsum(np.asarray(has_gaussian, dtype=np.int64))])
        gauss = np.minimum(pre[b] - pre[a], near)
    return near * (5.0 * 4) + gauss * 15.0 + (reach - near) * 1.6 + 600.0


FAR_HALF_SPANS = 4           # lbl_kernels.hip FF_FAR: a line is "far" from a span when its centre is >= 4 half-spans from the span centre


def pair_split(centre_index, H: int, n: int, first: int = 0, count: int | None = None) -> dict:
    """How the default accumulate kernel treats the (line, span) pairs of one line list on grid points
    [first, first + count): the same classes lbl_api.hip's group_schedule tabulates per span of 256 points
    (edge: the line's support ends inside the span; near: every point inside the support, centre within
    4 half-spans of the span centre; far: the rest, evaluated through the 30-term series about the span centre).
    Windows too narrow for any far line (H < 640) run all-direct kernels: everything is direct there.
    Returns the pair counts and the (line, grid point) evaluations behind them:
    evals_series = 256 per far pair (a far line covers the whole span), evals_direct = the rest of the exact
    eval count.  Host bookkeeping for bench.py; never used for results."""
    c = np.asarray(centre_index, dtype=np.int64)
    count = int(n) - int(first) if count is None else int(count)
    H = int(H)
    n_spans = -(-count // SPAN)
    lo = int(first) + np.arange(n_spans, dtype=np.int64) * SPAN
    hi = np.minimum(lo + SPAN, int(first) + count) - 1
    iA = np.searchsorted(c, lo - H, "left")
    iB = np.searchsorted(c, hi - H, "left")
    iC = np.searchsorted(c, lo + H + 1, "left")
    iD = np.searchsorted(c, hi + H + 1, "left")
    none = hi - H >= lo + H + 1                      # span wider than the support: no interior line
    iB = np.where(none, iD, iB)
    iC = np.where(none, iD, iC)
    pairs = int((iD - iA).sum())
    lo_c = np.maximum(c - H, int(first))
    hi_c = np.minimum(c + H, int(first) + count - 1)
    evals = int(np.maximum(hi_c - lo_c + 1, 0).sum())
    if H < (SPAN // 2) * (FAR_HALF_SPANS + 1):
        return dict(pairs=pairs, pairs_series=0, pairs_direct=pairs, evals=evals, evals_series=0, evals_direct=evals)
    reach = FAR_HALF_SPANS * (SPAN // 2)
    iF1 = np.minimum(np.maximum(np.searchsorted(c, lo + SPAN // 2 - reach, "left"), iB), iC)
    iF2 = np.minimum(np.maximum(np.searchsorted(c, lo + SPAN // 2 + reach, "left"), iF1), iC)
    far = int(((iF1 - iB) + (iC - iF2)).sum())
    full = (hi - lo + 1 == SPAN)
    far_evals = int((((iF1 - iB) + (iC - iF2)) * (hi - lo + 1)).sum())
    return dict(pairs=pairs, pairs_series=far, pairs_direct=pairs - far, evals=evals, evals_series=far_evals,
                evals_direct=evals - far_evals)


def gaussian_part(lines: dict, T, P, conc, molmass) -> np.ndarray:
    """Which lines carry a Gaussian term (regime select of cls:378-387: not 'Lorentz only'), from the
    reference's own half-width expressions (cls:252-263) - a host estimate for the shard cost model
    only; the device decides the regime itself in K1."""
    nu = np.asarray(lines["nu"], dtype=np.float64)
    broadened = nu + np.asarray(lines["delta_air"]) * P / 1013.25
    lhw = ((1 - conc) * np.asarray(lines["gamma_air"]) + conc * np.asarray(lines["gamma_self"])) * (P / 1013.25) * \
        (296.0 / T) ** np.asarray(lines["n_air"])
    m = molmass / 1000.0 / 6.022140857E23
    ghw = broadened * np.sqrt(2 * 1.38064852E-23 * T / m / 299792458.0 ** 2)
    with np.errstate(divide="ignore", invalid="ignore"):
        return ~(lhw / ghw > 100.0)


MAX_SHARD_RATIO = 1.125      # a shard is at most this much longer than n / world_size


def balanced_plan(n: int, world_size: int, rank: int, cost_per_span: np.ndarray,
                  max_ratio: float = MAX_SHARD_RATIO) -> ShardPlan:
    """Contiguous shards with (nearly) equal summed cost; boundaries at multiples of ALIGN grid
    points.  ``cost_per_span``: summed over every job (isotopologue x layer) that runs on the grid.
    Every rank sends as many doubles as the LONGEST shard holds (the all-gather's slot), so a shard
    may not grow beyond ``max_ratio`` x the equal share: the cost balance is traded against gathered
    bytes (at 8 ranks the all-gather of a 2.4e6-point spectrum takes about as long as the step)."""
    n = int(n)
    per_block = ALIGN // SPAN
    cost = np.asarray(cost_per_span, dtype=np.float64)
    n_blocks = -(-n // ALIGN)
    pad = n_blocks * per_block - cost.size
    block = np.concatenate([cost, np.zeros(max(pad, 0))])[:n_blocks * per_block].reshape(n_blocks, per_block).sum(axis=1)
    prefix = np.concatenate([[0.0], np.cumsum(block)])
    cap = max(int(np.ceil(max_ratio * n_blocks / world_size)), 1)
    cuts = [0]
    for r in range(1, world_size):
        # an equal share of what is left for the ranks that are left (a capped shard hands its surplus
        # to all later shards, not to its neighbour alone)
        target = prefix[cuts[-1]] + (prefix[-1] - prefix[cuts[-1]]) / (world_size - r + 1)
        b = int(np.searchsorted(prefix, target, "left"))
        # the boundary nearer to the target of the two that bracket it
        if b > 0 and abs(prefix[b - 1] - target) <= abs(prefix[min(b, n_blocks)] - target):
            b -= 1
        b = min(b, cuts[-1] + cap)                                   # this shard is at most cap blocks long ...
        b = max(b, n_blocks - (world_size - r) * cap)                # ... and so can every shard after it be
        cuts.append(min(max(b, cuts[-1]), n_blocks))
    cuts.append(n_blocks)
    bounds = []
    for r in range(world_size):
        f = min(cuts[r] * ALIGN, n)
        e = min(cuts[r + 1] * ALIGN, n)
        bounds.append((f, e - f))
    return ShardPlan(n, bounds, rank)


def halo_select(lines: dict, range_min, resolution, W, first, count):
    """Lines whose wing support (centre +- (W-2) points, cls:394) can reach grid points
    [first, first+count).  Selected by wavenumber with two grid steps of slack, so the
    truncation of cls:390 can never drop a contributing line; extra lines contribute nothing
    because the kernel clips every line to its exact support."""
    H = max(int(W) - 2, 0)
    lo = range_min + (first - H - 2) * resolution
    hi = range_min + (first + count + H + 2) * resolution
    nu = np.asarray(lines["nu"])
    m = (nu > lo) & (nu < hi)
    return {k: np.asarray(v)[m] for k, v in lines.items()}


def assemble_gathered(parts, n: int):
    """What the in-place all-gather leaves in the padded buffer: rank-ordered shards of S
    points each; the first n points are the spectrum."""
    return np.concatenate([np.asarray(p) for p in parts])[:n]
