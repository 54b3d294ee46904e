"""CPU, world_size 2 over gloo: the N > 1 host path — contiguous grid shards with a line halo,
per-rank compute, one all-gather into the padded buffer — reproduces the unsharded spectrum.
The per-rank compute is the CPU oracle here (tests may use it as the checker); on the GPU box
the same shard arithmetic drives the HIP kernels and RCCL (tests/test_gpu_parity.py)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import REPO


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, REPO)
    import torch
    import torch.distributed as dist
    from pyrad_amd import synthetic
    from pyrad_amd import dist as pdist
    from oracle import pyrad_oracle as orc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        r, lr, w = pdist.env_world()
        assert (r, w) == (rank, world)
        cfg = synthetic.config_c2(n_lines=700, range_min=640, range_max=652, seed=21)
        mol = cfg["molecules"][0]
        sp = synthetic.SPECIES["co2"]
        grid = orc.layer_grid(cfg["P"], cfg["range_min"], cfg["range_max"], cfg["base_resolution"], False)
        lines = orc.select_window(mol["lines"], grid["eff_min"], grid["eff_max"])
        S, first, count = pdist.shard_bounds(grid["n_work"], world, rank)
        mine = pdist.halo_select(lines, grid["range_min"], grid["resolution"], grid["W"], first, count)
        assert len(mine["nu"]) < len(lines["nu"])              # the halo really prunes
        xs, _ = orc.create_cross_section(mine, cfg["T"], cfg["P"], 4e-4, sp["molmass"],
                                         synthetic.q_value("co2", cfg["T"]), sp["q296"], grid, regrid=False)
        k_local = orc.abs_coef(xs, 4e-4, cfg["P"], cfg["T"])
        # in-place all-gather layout: padded buffer of world*S, own shard at rank*S
        send = np.zeros(S)
        send[:count] = k_local[first:first + count]
        recv = [torch.zeros(S, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(recv, torch.from_numpy(send))
        full = pdist.assemble_gathered([t.numpy() for t in recv], grid["n_work"])
        # max-over-ranks timing and summed eval counts travel the same way in bench.py
        from pyrad_amd import engine
        ev = torch.tensor([float(engine.eval_count(mine["nu"], grid["range_min"], grid["resolution"], grid["W"],
                                                   grid["n_work"], (first, count)))], dtype=torch.float64)
        dist.all_reduce(ev)
        if rank == 0:
            ref_xs, _ = orc.create_cross_section(lines, cfg["T"], cfg["P"], 4e-4, sp["molmass"],
                                                 synthetic.q_value("co2", cfg["T"]), sp["q296"], grid, regrid=False)
            ref = orc.abs_coef(ref_xs, 4e-4, cfg["P"], cfg["T"])
            lq = orc.line_quantities(lines, cfg["T"], cfg["P"], 4e-4, sp["molmass"], grid["range_min"], grid["resolution"])
            np.savez(os.path.join(out_dir, "result.npz"), full=full, ref=ref, evals=ev.numpy(),
                     evals_ref=orc.eval_count(lq["index"], grid["W"], grid["n_work"]))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_spectrum_matches_unsharded(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    z = np.load(tmp_path / "result.npz")
    assert np.array_equal(z["full"], z["ref"])          # same lines per point, same order: bit exact
    assert float(z["evals"][0]) == float(z["evals_ref"])


def _worker_balanced(rank, world, port, out_dir):
    """The cost-balanced form: unequal contiguous shards, every rank sends S (= longest shard) doubles
    starting at its own first point into slot `rank` of a world*S gathered buffer; `assemble` restores
    grid order.  Same control flow as engine.ResidentLayer.enqueue_allgather with a balanced plan."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, REPO)
    import torch
    import torch.distributed as dist
    from pyrad_amd import synthetic, engine
    from pyrad_amd import dist as pdist
    from oracle import pyrad_oracle as orc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = synthetic.config_c2(n_lines=900, range_min=640, range_max=656, seed=22)
        # all the lines in the first third of the range: equal-width shards would be badly unbalanced
        lines = {k: v.copy() for k, v in cfg["molecules"][0]["lines"].items()}
        lines["nu"] = np.sort(640.0 + (lines["nu"] - lines["nu"].min()) / 3.0)
        sp = synthetic.SPECIES["co2"]
        mols = [dict(conc=4e-4, isotopologues=[dict(lines=lines, molmass=sp["molmass"], q_T=1.0, q296=1.0)])]
        plan = engine.balanced_shards([dict(cfg, molecules=mols)], world, rank)
        grid = orc.layer_grid(cfg["P"], cfg["range_min"], cfg["range_max"], cfg["base_resolution"], False)
        assert plan.n == grid["n_work"] and not plan.in_place and plan.bounds[0][1] < plan.bounds[1][1]
        sel = orc.select_window(lines, grid["eff_min"], grid["eff_max"])
        mine = pdist.halo_select(sel, grid["range_min"], grid["resolution"], grid["W"], plan.first, plan.count)
        xs, _ = orc.create_cross_section(mine, cfg["T"], cfg["P"], 4e-4, sp["molmass"],
                                         synthetic.q_value("co2", cfg["T"]), sp["q296"], grid, regrid=False)
        k_local = np.zeros(plan.n + plan.S)                      # the rank's spectrum buffer is S longer than the grid
        k_local[plan.first:plan.first + plan.count] = orc.abs_coef(xs, 4e-4, cfg["P"], cfg["T"])[plan.first:plan.first + plan.count]
        send = torch.from_numpy(k_local[plan.first:plan.first + plan.S].copy())
        recv = [torch.zeros(plan.S, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(recv, send)
        full = plan.assemble(np.concatenate([t.numpy() for t in recv]))
        if rank == 0:
            ref_xs, _ = orc.create_cross_section(sel, cfg["T"], cfg["P"], 4e-4, sp["molmass"],
                                                 synthetic.q_value("co2", cfg["T"]), sp["q296"], grid, regrid=False)
            np.savez(os.path.join(out_dir, "balanced.npz"), full=full, ref=orc.abs_coef(ref_xs, 4e-4, cfg["P"], cfg["T"]),
                     bounds=np.array(plan.bounds))
    finally:
        dist.destroy_process_group()


def test_two_rank_balanced_shards_padded_gather(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker_balanced, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    z = np.load(tmp_path / "balanced.npz")
    assert np.array_equal(z["full"], z["ref"])
    assert z["bounds"][0][1] % 1024 == 0 and z["bounds"][0][1] + z["bounds"][1][1] == z["full"].size


def test_file_rendezvous_two_ranks(tmp_path):
    """The control plane bench.py uses to hand the 128-byte RCCL unique id to every rank."""
    import multiprocessing as mp
    from pyrad_amd import dist as pdist

    def rank1(q):
        r = pdist.FileRendezvous(1, 2, key="t", root=str(tmp_path), timeout=30)
        q.put(r.broadcast("uid", None))
        r.arrive("done")
        r.cleanup()

    q = mp.get_context("fork").Queue()
    p = mp.get_context("fork").Process(target=rank1, args=(q,))
    p.start()
    r0 = pdist.FileRendezvous(0, 2, key="t", root=str(tmp_path), timeout=30)
    payload = bytes(range(128))
    assert r0.broadcast("uid", payload) == payload
    assert q.get(timeout=30) == payload
    r0.arrive("done")
    p.join(30)
    assert p.exitcode == 0
    r0.cleanup()
    assert not os.path.isdir(r0.dir)


def _worker_eight(rank, world, port, out_dir):
    """World size 8 (the node the driver scales to) over gloo, on the MERGED step's arrays: every rank computes the
    layer's absorption coefficient sum_m f_m xs_m (what lbl_layer_merged_step_dev accumulates directly) on its own
    shard from the lines its halo keeps, and the three gather layouts of bench.py / engine.py move it:
    equal shards in place, cost-balanced shards through the padded out-of-place gather, and B steps batched into
    ONE collective (rank r, step b at [(r B + b) S, +S))."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, REPO)
    import torch
    import torch.distributed as dist
    from pyrad_amd import synthetic, engine
    from pyrad_amd import dist as pdist
    from oracle import pyrad_oracle as orc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        base = synthetic.config_c2(n_lines=10, range_min=640, range_max=660, seed=31)
        species = (("co2", 4e-4, 31), ("h2o", 1e-2, 32), ("ch4", 1.8e-6, 33))
        grid = orc.layer_grid(base["P"], base["range_min"], base["range_max"], base["base_resolution"], False)
        mols = []
        for sp_name, conc, seed in species:
            lines = synthetic.make_lines(seed, 260, grid["eff_min"], grid["eff_max"])
            if sp_name == "co2":        # a dense cluster in the first quarter: equal-width shards are unbalanced
                lines["nu"] = np.sort(np.concatenate([lines["nu"][:60], 641.0 + 3.0 * np.random.default_rng(7).random(200)]))
            sp = synthetic.SPECIES[sp_name]
            mols.append(dict(conc=conc, species=sp_name, isotopologues=[dict(lines=lines, molmass=sp["molmass"], q_T=1.0, q296=1.0)]))

        def layer_k(T, first, count):
            """the merged step's array on points [first, first + count) (zeros elsewhere), from the halo'd line lists"""
            k = np.zeros(grid["n_work"])
            for m in mols:
                iso = m["isotopologues"][0]
                sp = synthetic.SPECIES[m["species"]]
                sel = orc.select_window(iso["lines"], grid["eff_min"], grid["eff_max"])
                mine = pdist.halo_select(sel, grid["range_min"], grid["resolution"], grid["W"], first, count)
                xs, _ = orc.create_cross_section(mine, T, base["P"], m["conc"], sp["molmass"], synthetic.q_value(m["species"], T),
                                                 sp["q296"], grid, regrid=False)
                k = k + orc.abs_coef(xs, m["conc"], base["P"], T)
            out = np.zeros(grid["n_work"])
            out[first:first + count] = k[first:first + count]
            return out

        out = {}
        # (1) equal shards: in place on a buffer padded to world * S
        eq = engine.equal_plan(grid["n_work"], world, rank)
        assert eq.in_place and eq.S * world >= grid["n_work"] and eq.bounds[rank] == (eq.first, eq.count)
        buf = np.zeros(eq.S * world)
        buf[:grid["n_work"]] = layer_k(296, eq.first, eq.count)
        recv = [torch.zeros(eq.S, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(recv, torch.from_numpy(buf[rank * eq.S:(rank + 1) * eq.S].copy()))
        out["equal"] = np.concatenate([t.numpy() for t in recv])[:grid["n_work"]]
        # (2) cost-balanced shards: every rank sends S (the longest shard) doubles from its own first point
        cfgs = [dict(base, molecules=mols)]
        bal = engine.balanced_shards(cfgs, world, rank)
        assert bal.world == world and sum(c for _, c in bal.bounds) == grid["n_work"] and len({c for _, c in bal.bounds}) > 1
        assert all(f % 1024 == 0 for f, _ in bal.bounds)
        kb = np.zeros(grid["n_work"] + bal.S)
        kb[:grid["n_work"]] = layer_k(296, bal.first, bal.count)
        recv = [torch.zeros(bal.S, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(recv, torch.from_numpy(kb[bal.first:bal.first + bal.S].copy()))
        out["balanced"] = bal.assemble(np.concatenate([t.numpy() for t in recv]))
        # (3) B = 3 steps (three temperatures) staged side by side and sent as ONE collective of B * S doubles per rank
        B, temps = 3, (296, 250, 310)
        stage = np.zeros(world * B * eq.S)
        for b, T in enumerate(temps):
            kT = np.zeros(eq.S * world)
            kT[:grid["n_work"]] = layer_k(T, eq.first, eq.count)
            stage[(rank * B + b) * eq.S:(rank * B + b + 1) * eq.S] = kT[rank * eq.S:(rank + 1) * eq.S]
        recv = [torch.zeros(B * eq.S, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(recv, torch.from_numpy(stage[rank * B * eq.S:(rank + 1) * B * eq.S].copy()))
        g = np.concatenate([t.numpy() for t in recv]).reshape(world, B, eq.S)
        for b in range(B):
            out["batch%d" % b] = g[:, b, :].reshape(-1)[:grid["n_work"]]
        # the choice bench.py makes by default, and the timing / eval reductions
        plan, why = engine.choose_shards(cfgs, world, rank, "auto")
        assert plan.world == world and ("equal" in why or "balanced" in why)
        ev = torch.tensor([float(sum(engine.eval_count(pdist.halo_select(orc.select_window(m["isotopologues"][0]["lines"], grid["eff_min"], grid["eff_max"]),
                                                                          grid["range_min"], grid["resolution"], grid["W"], eq.first, eq.count)["nu"],
                                                       grid["range_min"], grid["resolution"], grid["W"], grid["n_work"], (eq.first, eq.count))
                                     for m in mols))], dtype=torch.float64)
        dist.all_reduce(ev)
        t = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            assert float(t[0]) == 0.008
            ref = {("ref%d" % T): layer_k(T, 0, grid["n_work"]) for T in temps}
            e_ref = sum(engine.eval_count(orc.select_window(m["isotopologues"][0]["lines"], grid["eff_min"], grid["eff_max"])["nu"],
                                          grid["range_min"], grid["resolution"], grid["W"], grid["n_work"]) for m in mols)
            np.savez(os.path.join(out_dir, "eight.npz"), evals=ev.numpy(), evals_ref=float(e_ref),
                     bounds=np.array(bal.bounds), **out, **ref)
    finally:
        dist.destroy_process_group()


def test_eight_rank_gather_layouts_on_the_merged_step(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker_eight, args=(r, 8, port, str(tmp_path))) for r in range(8)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    z = np.load(tmp_path / "eight.npz")
    assert np.any(z["ref296"] > 0)
    assert np.array_equal(z["equal"], z["ref296"]) and np.array_equal(z["balanced"], z["ref296"])
    for b, T in enumerate((296, 250, 310)):
        assert np.array_equal(z["batch%d" % b], z["ref%d" % T])
    assert float(z["evals"][0]) == float(z["evals_ref"])
    assert z["bounds"][:, 1].sum() == z["equal"].size
