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
