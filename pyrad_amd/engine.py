"""Host-side engine: the layer grid of the reference computed with the reference's own
expressions, and device-resident gas cells / columns that keep line lists and spectra in
HBM and only enqueue kernels (``lbl_*_dev`` entry points of include/pyrad_hip.h).

This is product code: it never imports ``oracle`` and it raises if the HIP library or a
GPU is missing.
"""
from __future__ import annotations

import numpy as np

from . import _native as nat
from . import settings
from .dist import ShardPlan, equal_plan, halo_select, shard_bounds  # noqa: F401 (re-exported)

K_BOLTZMANN = 1.38064852E-23     # pyradClasses.py:16
PI = 3.141592653589793           # pyradClasses.py:19


# ----------------------------------------------------------------------------------------
# Layer grid definition (pyradClasses.py:648-676, 698-705, 745-752)
# ----------------------------------------------------------------------------------------
def layer_grid(P, range_min, range_max, base_resolution=None, dynamic_resolution=True) -> dict:
    """Scalars of Layer.__init__ / changePressure with the reference's expressions, so that
    every int() truncation and the arange length agree with PyRad bit for bit."""
    if base_resolution is None:
        base_resolution = settings.BASE_RESOLUTION
    dfc = P / 1013.25 * 5                                              # cls:655
    eff_min = max(range_min - dfc, 0)                                  # cls:656
    eff_max = range_max + dfc                                          # cls:657
    if not dynamic_resolution:
        res = base_resolution                                          # cls:660
    else:
        res = max(10**int(np.log10((P / 1013.25))) * .01, base_resolution)   # cls:662
    n_base = int((range_max - range_min) / base_resolution)            # cls:672
    n_work = int((range_max - range_min) / res)                        # cls:700
    W = len(np.arange(0, dfc, res))                                    # cls:377
    return dict(dfc=dfc, eff_min=eff_min, eff_max=eff_max, resolution=res, base_resolution=base_resolution,
                n_base=n_base, n_work=n_work, W=W, range_min=range_min, range_max=range_max)


def native_grid(g: dict, shard=None) -> nat.Grid:
    first, count = (0, 0) if shard is None else shard
    return nat.Grid(float(g["range_min"]), float(g["range_max"]), float(g["resolution"]),
                    float(g["base_resolution"]), int(g["n_work"]), int(g["n_base"]), int(g["W"]),
                    int(first), int(count))


def x_axis(range_min, range_max, base_resolution=None) -> np.ndarray:
    """Layer.xAxis (cls:702-705) with the float ``num`` truncated as pre-1.18 NumPy did."""
    if base_resolution is None:
        base_resolution = settings.BASE_RESOLUTION
    return np.linspace(range_min, range_max, int((range_max - range_min) / base_resolution), endpoint=True)


def select_window(lines: dict, lo: float, hi: float) -> dict:
    """Strict lo < nu < hi of readHitranOnlineFile (ut:437-438)."""
    nu = np.asarray(lines["nu"])
    m = (nu > lo) & (nu < hi)
    if m.all():
        return lines
    return {k: np.asarray(v)[m] for k, v in lines.items()}


def eval_count(nu, range_min, resolution, W, n_work, shard=None) -> int:
    """Exact number of (line, grid point) contributions of the scatter loop cls:392-400
    restricted to grid points [first, first+count) (SURVEY.md §8d unit of work)."""
    idx = ((np.asarray(nu, dtype=np.float64) - range_min) / resolution).astype(np.int64)   # cls:390
    H = max(int(W) - 2, 0)
    first, count = (0, n_work) if shard is None or shard[1] == 0 else shard
    lo = np.maximum(idx - H, first)
    hi = np.minimum(idx + H, first + count - 1)
    return int(np.maximum(hi - lo + 1, 0).sum())


# ----------------------------------------------------------------------------------------
# the engine singleton
# ----------------------------------------------------------------------------------------
_engine = None


class Engine:
    """One context on one GPU.  Raises nat.NoDeviceError when there is no GPU."""

    def __init__(self, device: int = 0):
        self.ctx = nat.Context(device)
        if settings.ACCURACY == "budget":
            self.ctx.set_option("accuracy", 1)
        # page-locked staging for the arrays getters hand back (hipHostMalloc of 32 MiB takes ~6 ms: paid with the
        # engine, once per process, instead of inside the first getter; the pool grows on demand)
        # (two blocks: a caller usually still holds the previous spectrum when it asks for the next)
        # Each block takes one device-to-host copy right away: the first DMA into a fresh page-locked block was measured at
        # 8 ms for 19 MB (its pages are mapped for the device on first use), the following ones at 0.75 ms.
        if settings.PINNED_POOL_BYTES > 0:
            n = settings.PINNED_POOL_BYTES // 8
            warm = self.ctx.buffer(n).fill(0.0)
            keep = [warm.download(n, pinned=True) for _ in range(2)]
            del keep
            warm.free()
        self._line_masters = {}          # id(master wavenumber array) -> (resident nat.Lines, master dict)

    def pooled_lines(self, lines: dict):
        """A view (lbl_lines_view) of the resident copy of the registered line list that ``lines`` is a slice of
        (data.master_slice), uploading that list on first use; None if ``lines`` is not such a slice."""
        from . import data
        hit = data.master_slice(lines, nat.Lines.ORDER)
        if hit is None:
            return None
        master, first, count = hit
        key = id(master["nu"])
        entry = self._line_masters.get(key)
        if entry is None or entry[1]() is not master or entry[0].h is None:
            import weakref
            self._prune_masters(keep=key)
            entry = self._line_masters[key] = (self.ctx.lines(master), weakref.ref(master))
        return entry[0].view(first, count)

    MASTER_POOL_BYTES = 8 << 30      # resident line lists nobody windows any more are dropped beyond this (56 B per line)

    def _prune_masters(self, keep=None):
        """Free resident copies that nothing needs any more: lists whose host master is gone (a data source rebuilt it:
        PyradDataDir makes a new list whenever it parses a segment it had not seen) as soon as no view of them is left,
        and, oldest first, idle lists beyond MASTER_POOL_BYTES.  The host masters are held weakly: the pool never keeps
        a dropped source's arrays alive."""
        idle = [k for k, (dev, ref) in self._line_masters.items()
                if k != keep and (dev.h is None or not dev.has_views())]
        for k in idle:
            dev, ref = self._line_masters[k]
            if dev.h is None or ref() is None:
                self._line_masters.pop(k)
                if dev.h is not None:
                    dev.free()
        total = sum(dev.n * 56 for dev, _ in self._line_masters.values() if dev.h is not None)
        for k in idle:                                            # (dict order = insertion order: oldest first)
            if total <= self.MASTER_POOL_BYTES:
                break
            if k in self._line_masters:
                dev, _ = self._line_masters.pop(k)
                total -= dev.n * 56
                dev.free()

    def close(self):
        self._line_masters = {}
        self.ctx.close()


def get_engine(device: int | None = None) -> Engine:
    global _engine
    if _engine is None or _engine.ctx.h is None:
        _engine = Engine(settings.DEVICE if device is None else device)
    return _engine


def shutdown():
    global _engine
    if _engine is not None:
        _engine.close()
        _engine = None


# ----------------------------------------------------------------------------------------
# device-resident gas cell / column (what bench.py and the sharded path drive)
# ----------------------------------------------------------------------------------------
def choose_shards(layer_cfgs, world: int, rank: int, mode: str = "auto"):
    """(plan, what was chosen).  ``mode``: "equal", "balanced", or "auto": cost-balanced bounds only where the
    host cost model expects them to shorten the slowest shard by clearly more than the longer all-gather
    slot costs (every rank sends as many doubles as the LONGEST shard holds): model gain minus half the
    relative growth of the slot must reach 2 %.  Measured on MI355X: balancing the
    100-2500 cm^-1 cell gains 3.5 % at 2 shards and nothing at 8, where the slot grows 12.6 %."""
    if mode == "equal" or world <= 1:
        n = layer_grid(layer_cfgs[0]["P"], layer_cfgs[0]["range_min"], layer_cfgs[0]["range_max"],
                       layer_cfgs[0].get("base_resolution"), layer_cfgs[0].get("dynamic_resolution", True))["n_work"]
        return equal_plan(n, world, rank), "equal"
    bal, cost = balanced_shards(layer_cfgs, world, rank, return_cost=True)
    if mode == "balanced":
        return bal, "balanced"
    eq = equal_plan(bal.n, world, rank)
    prefix = np.concatenate([[0.0], np.cumsum(cost)])
    from .dist import SPAN
    load = lambda plan: max(prefix[-(-(f + k) // SPAN)] - prefix[f // SPAN] for f, k in plan.bounds)
    gain = load(eq) / max(load(bal), 1e-300) - 1.0
    growth = bal.S / max(eq.S, 1) - 1.0
    if gain - 0.5 * growth >= 0.02:
        return bal, "balanced (auto: model gain %.1f %%, slot +%.1f %%)" % (100 * gain, 100 * growth)
    return eq, "equal (auto: model gain of balancing %.1f %%, slot +%.1f %%)" % (100 * gain, 100 * growth)


def balanced_shards(layer_cfgs, world: int, rank: int, return_cost: bool = False):
    """Cost-balanced contiguous shards for a layer or a column: ``layer_cfgs`` = list of dicts with
    P, range_min, range_max, base_resolution, dynamic_resolution and molecules (each with
    isotopologues = [dict(lines=...)]); all layers must share one work grid.  The cost of a span is
    the host model of K2 summed over every line list of every layer (dist.span_costs)."""
    from .dist import span_costs, balanced_plan, gaussian_part
    cost, n_work = None, None
    for c in layer_cfgs:
        g = layer_grid(c["P"], c["range_min"], c["range_max"], c.get("base_resolution"), c.get("dynamic_resolution", True))
        if n_work is None:
            n_work = g["n_work"]
        elif n_work != g["n_work"]:
            raise ValueError("all layers of a column must share one work grid to be sharded together")
        H = max(int(g["W"]) - 2, 0)
        for mol in c["molecules"]:
            for iso in mol["isotopologues"]:
                lines = select_window(iso["lines"], g["eff_min"], g["eff_max"])
                nu = np.asarray(lines["nu"], dtype=np.float64)
                order = np.argsort(nu, kind="stable")
                idx = ((nu[order] - g["range_min"]) / g["resolution"]).astype(np.int64)
                has_g = gaussian_part(lines, c["T"], c["P"], mol["conc"], iso["molmass"])[order]
                sc = span_costs(idx, H, n_work, has_g)
                cost = sc if cost is None else cost + sc
    if cost is None:
        plan = equal_plan(n_work or 0, world, rank)
        return (plan, np.zeros(0)) if return_cost else plan
    plan = balanced_plan(n_work, world, rank, cost)
    return (plan, cost) if return_cost else plan


def as_plan(shard, n_work):
    """None | ShardPlan | (world, rank) [equal-width shards] -> ShardPlan or None (one rank)."""
    if shard is None:
        return None
    if isinstance(shard, ShardPlan):
        return shard if shard.world > 1 else None
    world, rank = shard
    return equal_plan(n_work, world, rank) if world > 1 else None


class StepGraph:
    """A resident step as one hipGraph launch, captured again by itself when the library reports the graph
    stale (LBL_ERR_STATE: a scratch buffer grew, a descriptor slot was rewritten, or ANY buffer or line list of
    the context was destroyed since the capture - e.g. a temporary of pyrad_amd.model sharing the context).
    The stale launch is not lost: the step is enqueued kernel by kernel once, which also re-creates whatever the
    capture needs, and recorded for the launches that follow."""

    def __init__(self, owner, enqueue_kwargs):
        self.owner, self.kwargs = owner, dict(enqueue_kwargs)
        self.recaptures = 0
        owner.enqueue(**self.kwargs)               # scratch, schedules and descriptors exist before the capture
        self.g = owner.ctx.capture(lambda: owner.enqueue(**self.kwargs))

    def launch(self):
        if self.g is not None:
            try:
                self.g.launch()
                return
            except nat.LblError as e:
                if e.code != -6:
                    raise
            self.g.free()
            self.g = None                          # (a failed recapture below leaves a consistent object: the next launch retries)
        self.owner.enqueue(**self.kwargs)          # this step, kernel by kernel
        self.g = self.owner.ctx.capture(lambda: self.owner.enqueue(**self.kwargs))
        self.recaptures += 1

    def free(self):
        if self.g is not None:
            self.g.free()
            self.g = None


class ResidentLayer:
    """One layer (gas cell) whose line lists, cross sections and spectra live in HBM.

    ``molecules``: list of dict(species params) each with
        conc (volume fraction), isotopologues = [dict(lines=SoA, molmass, q_T, q296), ...]
    The layer's scalars follow Layer.__init__ (cls:648-676).  ``shard=(world, rank)`` keeps only
    this rank's contiguous range of the grid (lines are pre-selected with the halo they need).
    """

    def __init__(self, ctx: nat.Context, depth, T, P, range_min, range_max, molecules,
                 base_resolution=None, dynamic_resolution=True, shard=None, keep_host_lines=False, line_pool=None):
        self.ctx = ctx
        self.depth, self.T, self.P = depth, T, P
        self.range_min, self.range_max = range_min, range_max
        self.g = layer_grid(P, range_min, range_max, base_resolution, dynamic_resolution)
        g = self.g
        n = g["n_base"]
        self.n = n
        plan = as_plan(shard, g["n_work"])
        self.plan = plan
        if plan is not None:
            self.world, self.rank = plan.world, plan.rank
            self.S, self.first, self.count = plan.S, plan.first, plan.count
        else:
            self.world, self.rank = 1, 0
            self.S, self.first, self.count = n, 0, n
        # a sharded layer's spectra are S longer than the grid when the all-gather is out of place
        # (every rank sends S doubles starting at its own first point); in place they are world*S long
        if plan is None:
            self.padded_n = n
        elif plan.in_place:
            self.padded_n = self.S * self.world
        else:
            self.padded_n = n + self.S
        self.empty = plan is not None and self.count == 0      # more ranks than tiles: nothing to do here
        self.jobs = []
        self.iso_mol, self.conc = [], []
        self.evals = 0
        self.n_lines = 0
        self.pairs = dict(pairs=0, pairs_series=0, pairs_direct=0, evals=0, evals_series=0, evals_direct=0)
        self._keep = []
        sh = None if plan is None else (self.first, self.count)
        self.grid_native = native_grid(g, sh)
        self.gathered = {}
        for m, mol in enumerate(molecules):
            self.conc.append(float(mol["conc"]))
            for iso in mol["isotopologues"]:
                lines = select_window(iso["lines"], g["eff_min"], g["eff_max"])
                if plan is not None:
                    lines = self._halo_select(lines)
                dev_lines = None
                if line_pool is not None:
                    # the window (and the shard's halo) is a wavenumber range: a VIEW of the column's one resident
                    # copy of this line list (lbl_lines_view), not another upload
                    lo, hi = g["eff_min"], g["eff_max"]
                    if plan is not None:
                        H = max(int(g["W"]) - 2, 0)
                        lo = max(lo, g["range_min"] + (self.first - H - 2) * g["resolution"])
                        hi = min(hi, g["range_min"] + (self.first + self.count + H + 2) * g["resolution"])
                    dev_lines = line_pool.view(iso["lines"], lo, hi, expect=len(lines["nu"]))
                if dev_lines is None:
                    dev_lines = ctx.lines(lines)
                out = ctx.buffer(max(self.padded_n, 1))
                out.fill(0.0)
                ip = nat.IsoParams(float(T), float(P), float(mol["conc"]), float(iso["molmass"]),
                                   float(iso["q_T"]), float(iso["q296"]))
                self.jobs.append((dev_lines, ip, self.grid_native, out))
                self.iso_mol.append(m)
                self.n_lines += dev_lines.n
                if not self.empty:
                    self.evals += eval_count(lines["nu"], g["range_min"], g["resolution"], g["W"], g["n_work"], sh)
                    from .dist import pair_split
                    idx = np.sort(((np.asarray(lines["nu"], dtype=np.float64) - g["range_min"]) / g["resolution"]).astype(np.int64))
                    ps = pair_split(idx, max(int(g["W"]) - 2, 0), g["n_work"], *((0, None) if sh is None else sh))
                    for k_ in self.pairs:
                        self.pairs[k_] += ps[k_]
                if keep_host_lines:
                    self._keep.append(lines)
        self.abs_coef = ctx.buffer(max(self.padded_n, 1))
        self.trans = ctx.buffer(max(self.padded_n, 1))
        self.I_out = ctx.buffer(max(self.padded_n, 1))
        for b in (self.abs_coef, self.trans, self.I_out):
            b.fill(0.0)

    def _halo_select(self, lines):
        g = self.g
        return halo_select(lines, g["range_min"], g["resolution"], g["W"], self.first, self.count)

    # -- enqueue ------------------------------------------------------------------------
    def _range(self):
        return (0, 0) if self.plan is None else (self.first, self.count)

    def enqueue_xsec(self):
        if not self.empty:
            self.ctx.xsec_accumulate_dev(self.jobs)

    def enqueue_sweep(self, I_in=None, surface_T=0.0, want_I=True):
        if self.empty:
            return
        first, count = self._range()
        self.ctx.layer_sweep_dev([j[3] for j in self.jobs], self.iso_mol, self.conc, self.P, self.T, self.depth,
                                 self.range_min, self.range_max, self.n, I_in=I_in, surface_T=surface_T,
                                 abs_coef=self.abs_coef, trans=self.trans,
                                 I_out=self.I_out if want_I else None, first=first, count=count)

    def enqueue(self, surface_T=288.0, I_in=None, fused=True, merged=False):
        """One layer step: line prep, accumulate, sweep.  ``merged``: ONE accumulate job over the layer's merged,
        factor-weighted line lists with the sweep in its output stage (lbl_layer_merged_step_dev): the absorption
        coefficient is accumulated directly and the per-line-list cross sections are not written (``enqueue_xsec``
        produces them on demand).  Otherwise one job per line list: ``fused`` (default) through lbl_layer_step_dev
        (the sweep rides in the accumulate kernel's output stage when the layer has one line list); False: the
        accumulate launch followed by the separate sweep launch (bit-identical, kept for A/B tests)."""
        if self.empty:
            return
        if merged and self.jobs:
            self.ctx.layer_merged_step_dev([j[0] for j in self.jobs], [j[1] for j in self.jobs], self.grid_native,
                                           self.iso_mol, self.conc, self.depth, I_in=I_in, surface_T=surface_T,
                                           abs_coef=self.abs_coef, trans=self.trans, I_out=self.I_out)
            return
        if fused and self.jobs:
            self.ctx.layer_step_dev([j[0] for j in self.jobs], [j[1] for j in self.jobs], self.grid_native,
                                    [j[3] for j in self.jobs], self.iso_mol, self.conc, self.depth, I_in=I_in,
                                    surface_T=surface_T, abs_coef=self.abs_coef, trans=self.trans, I_out=self.I_out)
            return
        self.enqueue_xsec()
        self.enqueue_sweep(I_in=I_in, surface_T=surface_T)

    def capture_step(self, **enqueue_kwargs) -> StepGraph:
        """Capture this layer's step (``enqueue(**enqueue_kwargs)``) into a graph: run once so that scratch,
        schedule and descriptors exist, then record.  ``graph.launch()`` replays line prep, accumulate
        and sweep with one host call."""
        if self.empty:
            return None
        return StepGraph(self, enqueue_kwargs)

    def send_range(self):
        """(offset, S): the S doubles of a spectrum buffer this rank contributes to the all-gather"""
        if self.plan is None or self.plan.in_place:
            return self.rank * self.S, self.S
        return self.first, self.S

    def enqueue_allgather(self, comm: nat.Comm, buffers=None, overlap_slot=None):
        """The single RCCL all-gather of the path.  Equal shards: in place on the padded buffers.
        Cost-balanced (unequal) shards: every rank sends S doubles from its own first point into a
        gathered buffer of world*S doubles (slot r = rank r's shard; ``results`` puts it back in
        grid order).  With ``overlap_slot`` the context stream does not wait for it: call
        ``comm.fence_dev(overlap_slot)`` before this layer's buffers are touched again."""
        for b in (buffers if buffers is not None else (self.abs_coef,)):
            if self.plan is None or self.plan.in_place:
                comm.allgather_dev(b, self.rank * self.S, self.S, b, overlap_slot=overlap_slot)
            else:
                comm.allgather_dev(b, self.first, self.S, self._gathered(b), overlap_slot=overlap_slot)

    def _gathered(self, b):
        key = id(b)
        if key not in self.gathered:
            self.gathered[key] = self.ctx.buffer(self.S * self.world).fill(0.0)
        return self.gathered[key]

    # -- results ------------------------------------------------------------------------
    def xsec_host(self, i=0):
        return self.jobs[i][3].download(self.n)

    def spectrum_host(self, b):
        """Host copy of a spectrum buffer in grid order (after the all-gather when sharded)."""
        if self.plan is not None and not self.plan.in_place and id(b) in self.gathered:
            return self.plan.assemble(self.gathered[id(b)].download(self.S * self.world))
        return b.download(self.n)

    def results(self):
        return dict(abs_coef=self.spectrum_host(self.abs_coef), transmittance=self.spectrum_host(self.trans),
                    transmission=self.spectrum_host(self.I_out))

    def free(self):
        for j in self.jobs:
            j[0].free(); j[3].free()
        for b in (self.abs_coef, self.trans, self.I_out):
            b.free()
        for b in self.gathered.values():
            b.free()
        self.gathered = {}
        self.jobs = []


class LinePool:
    """One resident copy of every distinct line list of a column (identity of the host dict), handed out as views:
    the 30 layers of a column select wavenumber windows of the same three lists."""

    def __init__(self, ctx: nat.Context):
        self.ctx = ctx
        self.masters = {}

    def view(self, lines: dict, lo: float, hi: float, expect: int):
        """the lines with lo < nu < hi (ut:437-438 is strict) as a view of the resident copy; None if that is not
        the selection the caller made (then it uploads its own)"""
        key = id(lines["nu"])               # (layer configs are often fresh dicts around the same arrays)
        if key not in self.masters or self.masters[key][2]["nu"] is not lines["nu"]:
            nu = np.asarray(lines["nu"], dtype=np.float64)
            order = np.argsort(nu, kind="stable")
            self.masters[key] = (self.ctx.lines(lines), nu[order], lines)       # (Lines sorts the same way)
        master, nu, _ = self.masters[key]
        first = int(np.searchsorted(nu, lo, "right"))
        end = int(np.searchsorted(nu, hi, "left"))
        if max(end - first, 0) != expect:
            return None
        return master.view(first, max(end - first, 0))

    def free(self):
        for master, _, _ in self.masters.values():
            master.free()
        self.masters = {}


class ResidentColumn:
    """A column of layers (bottom to top) resident in HBM: all layers' isotopologue jobs go
    through ONE batched K1/K2 launch sequence, every layer gets its fused sweep, and the
    column-sweep kernel folds Layer.transmission (cls:784-787) over the layers:
    I <- T_i I + (1 - T_i) B(nu, T_i), I_0 = B(nu, surface_T).  All layers share one wavenumber
    range and base grid.  With ``shard=(world, rank)`` every rank keeps ALL layers for its own
    contiguous grid range (the fold is independent per grid point), so one all-gather of the
    outgoing spectrum suffices.  Line lists that several layers share (the same host arrays) are uploaded
    once; every layer works on a view of its wavenumber window (LinePool)."""

    def __init__(self, ctx: nat.Context, layer_cfgs, surface_T, shard=None):
        self.ctx = ctx
        self.surface_T = float(surface_T)
        self.pool = LinePool(ctx)
        self.layers = [ResidentLayer(ctx, c["depth"], c["T"], c["P"], c["range_min"], c["range_max"], c["molecules"],
                                     c.get("base_resolution"), c.get("dynamic_resolution", True), shard=shard,
                                     line_pool=self.pool)
                       for c in layer_cfgs]
        first = self.layers[0]
        for L in self.layers[1:]:
            if (L.range_min, L.range_max, L.n) != (first.range_min, first.range_max, first.n):
                raise ValueError("all layers of a column must share one wavenumber range and base grid")
        self.n = first.n
        self.plan = first.plan
        self.world, self.rank, self.S = first.world, first.rank, first.S
        self.first, self.count = first.first, first.count
        self.empty = first.empty
        self.I_toa = ctx.buffer(max(first.padded_n, 1)).fill(0.0)
        self.I_gathered = None
        self.jobs = [j for L in self.layers for j in L.jobs]
        self.evals = sum(L.evals for L in self.layers)
        self.n_lines = sum(L.n_lines for L in self.layers)
        self.pairs = {k: sum(L.pairs[k] for L in self.layers) for k in self.layers[0].pairs}

    def enqueue(self, layer_arrays=True, fused=True, merged=False):
        """One column step.  ``merged``: one accumulate job per LAYER over its merged, factor-weighted line lists
        (lbl_layers_merged_accumulate_dev: 30 absorption-coefficient arrays instead of 90 cross sections), then the
        fold over them (lbl_column_fold_dev; ``layer_arrays`` also writes the transmittances).  Otherwise one job per
        line list and, ``fused``: a single pass over all cross sections (lbl_column_step_dev) instead of one sweep per
        layer plus the fold; ``layer_arrays`` False skips writing the per-layer absorption coefficient /
        transmittance arrays (only the outgoing spectrum)."""
        if self.empty:
            return
        first, count = (0, 0) if self.plan is None else (self.first, self.count)
        if merged and all(L.jobs for L in self.layers):
            self.ctx.layers_merged_accumulate_dev(
                [dict(lines=[j[0] for j in L.jobs], iso=[j[1] for j in L.jobs], grid=L.grid_native, iso_mol=L.iso_mol,
                      conc=L.conc, abs_coef=L.abs_coef) for L in self.layers])
            self.ctx.column_fold_dev([L.abs_coef for L in self.layers], [L.T for L in self.layers],
                                     [L.depth for L in self.layers], self.layers[0].range_min, self.layers[0].range_max,
                                     self.n, self.I_toa, surface_T=self.surface_T,
                                     trans=[L.trans for L in self.layers] if layer_arrays else None, first=first, count=count)
            return
        self.ctx.xsec_accumulate_dev(self.jobs)
        if fused:
            desc = [dict(xsec=[j[3] for j in L.jobs], iso_mol=L.iso_mol, conc=L.conc, P=L.P, T=L.T, depth=L.depth,
                         trans=L.trans if layer_arrays else None, abs_coef=L.abs_coef if layer_arrays else None)
                    for L in self.layers]
            self.ctx.column_step_dev(desc, self.layers[0].range_min, self.layers[0].range_max, self.n, self.I_toa,
                                     surface_T=self.surface_T, first=first, count=count)
            return
        for L in self.layers:
            L.enqueue_sweep(want_I=False)
        self.ctx.column_sweep_dev([L.trans for L in self.layers], [L.T for L in self.layers],
                                  self.layers[0].range_min, self.layers[0].range_max, self.n, self.I_toa,
                                  surface_T=self.surface_T, first=first, count=count)

    def capture_step(self, **enqueue_kwargs) -> StepGraph:
        """The column step (all layers' line prep + accumulate launches + the column fold) as one graph."""
        if self.empty:
            return None
        return StepGraph(self, enqueue_kwargs)

    def send_range(self):
        """(offset, S): the S doubles of the outgoing spectrum this rank contributes to the all-gather"""
        if self.plan is None or self.plan.in_place:
            return self.rank * self.S, self.S
        return self.first, self.S

    def enqueue_allgather(self, comm: nat.Comm, overlap_slot=None):
        if self.plan is None or self.plan.in_place:
            comm.allgather_dev(self.I_toa, self.rank * self.S, self.S, self.I_toa, overlap_slot=overlap_slot)
            return
        if self.I_gathered is None:
            self.I_gathered = self.ctx.buffer(self.S * self.world).fill(0.0)
        comm.allgather_dev(self.I_toa, self.first, self.S, self.I_gathered, overlap_slot=overlap_slot)

    def results(self):
        if self.plan is not None and not self.plan.in_place and self.I_gathered is not None:
            toa = self.plan.assemble(self.I_gathered.download(self.S * self.world))
        else:
            toa = self.I_toa.download(self.n)
        return dict(toa=toa, transmittance=[L.trans.download(self.n) for L in self.layers])

    def free(self):
        for L in self.layers:
            L.free()
        self.pool.free()                    # after the layers' views
        self.I_toa.free()
        if self.I_gathered is not None:
            self.I_gathered.free()
            self.I_gathered = None
