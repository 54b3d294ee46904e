"""CPU: `python bench.py --gpus N` without a launcher starts its own N ranks (before anything
touches the GPU) with the torchrun environment contract, relays rank 0's single line and
propagates a failing rank's status; shard plans (equal and cost-balanced) tile the grid."""
import os
import sys

import numpy as np
import pytest

from conftest import REPO

sys.path.insert(0, REPO)
import bench  # noqa: E402


def test_spawn_plan_env_and_argv_for_two_ranks():
    argv = ["--gpus", "2", "--steps", "7", "--warmup", "2"]
    plan = bench.spawn_plan(2, argv, {"PATH": "/usr/bin", "WORLD_SIZE_UNRELATED": "x"}, port=29517)
    assert len(plan) == 2
    keys = set()
    for r, (cmd, env) in enumerate(plan):
        assert cmd[0] == sys.executable and cmd[1] == os.path.join(REPO, "bench.py") and cmd[2:] == argv
        assert env["RANK"] == str(r) and env["LOCAL_RANK"] == str(r) and env["WORLD_SIZE"] == "2"
        assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "29517"
        assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"          # dmabuf IPC: RCCL needs it on this driver
        assert env["PATH"] == "/usr/bin"
        keys.add(env["PYRAD_RENDEZVOUS_KEY"])
    assert len(keys) == 1                                        # one rendezvous for the whole launch
    # an explicit setting of the caller is kept
    plan = bench.spawn_plan(2, argv, {"HSA_ENABLE_IPC_MODE_LEGACY": "1"}, port=1)
    assert plan[0][1]["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"


def test_the_parent_spawns_before_importing_the_gpu_library():
    """The self-launch branch sits before the first import of pyrad_amd in main()."""
    src = open(os.path.join(REPO, "bench.py")).read()
    body = src[src.index("def main():"):]
    assert body.index("run_ranks(spawn_plan(") < body.index("from pyrad_amd import _native")
    head = src[:src.index("def main():")]
    assert "import pyrad_amd" not in head and "from pyrad_amd" not in head.replace("    from pyrad_amd", "")


def test_run_ranks_relays_rank0_and_propagates_failure(capfd):
    ok = [([sys.executable, "-c", "import os; print('{\"rank\": %s}' % os.environ['RANK'])"], dict(os.environ, RANK="0")),
          ([sys.executable, "-c", "print('noise from rank 1')"], dict(os.environ, RANK="1"))]
    assert bench.run_ranks(ok, timeout_s=60) == 0
    out, err = capfd.readouterr()
    assert out.strip() == '{"rank": 0}'                           # exactly rank 0's line on stdout
    assert "noise from rank 1" in err                             # other ranks' stdout goes to stderr
    bad = [([sys.executable, "-c", "import time; time.sleep(30)"], dict(os.environ)),
           ([sys.executable, "-c", "raise SystemExit(3)"], dict(os.environ))]
    assert bench.run_ranks(bad, timeout_s=60) == 3                # and the sleeping rank was ended, not waited for


def test_gpus_mismatch_and_missing_device_are_errors(monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE=4" in str(e.value)


def test_shard_plans_tile_the_grid():
    from pyrad_amd import dist
    rng = np.random.default_rng(5)
    # banded line density: most lines in one eighth of the grid
    c = np.sort(np.concatenate([rng.integers(0, 2_400_000, 40_000), rng.integers(900_000, 1_200_000, 260_000)]))
    cost = dist.span_costs(c, 4998, 2_400_000)
    for world in (2, 3, 4, 8):
        eq = dist.equal_plan(2_400_000, world, 0)
        assert eq.in_place and sum(k for _, k in eq.bounds) == 2_400_000
        bal = [dist.balanced_plan(2_400_000, world, r, cost, max_ratio=8.0) for r in range(world)]
        assert all(b.bounds == bal[0].bounds for b in bal)        # every rank derives the same plan
        b = bal[0]
        assert b.bounds[0][0] == 0 and sum(k for _, k in b.bounds) == 2_400_000
        assert all(f % dist.ALIGN == 0 for f, _ in b.bounds)
        # summed cost per shard: balanced is much flatter than equal-width on this density
        prefix = np.concatenate([[0.0], np.cumsum(cost)])
        load = lambda bounds: [prefix[-(-(f + k) // dist.SPAN)] - prefix[f // dist.SPAN] for f, k in bounds]
        lb, le = load(b.bounds), load(eq.bounds)
        assert max(lb) / (sum(lb) / world) < 1.1 < max(le) / (sum(le) / world)
        # the padded all-gather layout round-trips
        spec = rng.random(2_400_000)
        gathered = np.zeros(world * b.S)
        for r, (f, k) in enumerate(b.bounds):
            gathered[r * b.S:r * b.S + k] = spec[f:f + k]
        assert np.array_equal(b.assemble(gathered), spec)
    # the default cap on a shard's length (every rank sends the longest shard's size into the all-gather)
    capped = dist.balanced_plan(2_400_000, 8, 0, cost)
    assert sum(k for _, k in capped.bounds) == 2_400_000
    assert capped.S <= dist.MAX_SHARD_RATIO * 300_000 + dist.ALIGN
    # more ranks than aligned blocks: some shards are empty, nothing is lost
    tiny = dist.balanced_plan(3000, 8, 7, dist.span_costs(np.array([10, 20, 2999]), 48, 3000))
    assert sum(k for _, k in tiny.bounds) == 3000 and sum(1 for _, k in tiny.bounds if k == 0) == 5


def test_steps_in_flight_rule():
    """One step in flight by default, sharded or not (the simple path is the one that has run with a communicator);
    more are opt-in; small unsharded cells (partial round of workgroups) get the extra in-flight leg."""
    f = bench.steps_in_flight
    assert f("auto", False) == 1 and f("auto", True) == 1
    assert f("2", False) == 2 and f("3", True) == 3 and f("1", True) == 1
    assert bench.partial_round(4e5) and bench.partial_round(1e4) and bench.partial_round(0.9e6)      # C2, C1, C3 / 8
    assert not bench.partial_round(7.2e6) and not bench.partial_round(1.8e6)


def test_gather_batch_rule():
    """A collective per step by default; "fit": batches of 3..8 steps per collective, sized so that the timed region
    ends on a full batch where it can."""
    f = bench.gather_batch
    assert f("auto", False, 10) == 1 and f("4", False, 10) == 1          # in-stream or multi-array gathers: a collective per step
    assert f("auto", True, 10) == 1 and f("auto", True, 32) == 1         # default: one collective per step
    assert f("fit", True, 10) == 5                                       # 20 steps over two resident sets
    assert f("fit", True, 32) == 8 and f("fit", True, 25) == 5 and f("fit", True, 3) == 3 and f("fit", True, 2) == 2
    assert f("fit", True, 7) == 7 and f("fit", True, 11) in (4, 6)       # 11 = 2 * 6 - 1 = 3 * 4 - 1
    assert f("6", True, 10) == 6 and f("1", True, 10) == 1


class _FakeBuffer:
    def __init__(self, n):
        self.a = np.zeros(n)
        self.log = []

    def fill(self, v):
        self.a[:] = v
        return self

    def stage_from_dev(self, src, src_offset, n, dst_offset=0):
        self.a[dst_offset:dst_offset + n] = src.a[src_offset:src_offset + n]
        return self

    def free(self):
        pass


class _FakeCtx:
    def buffer(self, n):
        return _FakeBuffer(n)


class _FakeNet:
    """Lock-step stand-in for the communicator of `world` ranks: a collective completes when every rank has
    issued its call with the same sequence number; then recv[r'*count : (r'+1)*count] = rank r' send range."""

    def __init__(self, world):
        self.world, self.calls = world, {}

    def comm(self, rank):
        net = self

        class Comm:
            def __init__(self):
                self.seq, self.events = 0, []

            def fence_dev(self, slot):
                self.events.append(("fence", slot))

            def allgather_dev(self, send, off, count, recv, overlap_slot=None):
                self.events.append(("gather", overlap_slot))
                key = self.seq
                self.seq += 1
                net.calls.setdefault(key, {})[rank] = (send, off, count, recv)
                if len(net.calls[key]) == net.world:
                    parts = net.calls.pop(key)
                    assert len({c for _, _, c, _ in parts.values()}) == 1          # same count on every rank
                    data = {r: s.a[o:o + c].copy() for r, (s, o, c, _) in parts.items()}
                    for r, (_, _, c, rv) in parts.items():
                        for rr in range(net.world):
                            rv.a[rr * c:(rr + 1) * c] = data[rr]
        return Comm()


def test_batched_gather_layout_two_simulated_ranks():
    """bench._Batch on two simulated ranks: shards of B consecutive steps staged at (rank * B + b) * S leave in one
    in-place collective of B * S per rank; every rank then holds every rank's shard of every step of the batch; a
    buffer is fenced before it is refilled; a partly filled batch leaves at a flush."""
    world, S, B = 2, 7, 3
    net = _FakeNet(world)
    comms = [net.comm(r) for r in range(world)]
    batches = [bench._Batch(_FakeCtx(), comms[r], r, world, S, B, (4, 5)) for r in range(world)]
    src = [_FakeBuffer(world * S + 5) for _ in range(world)]          # a rank's padded spectrum buffer
    for step in range(2 * B + 1):                                     # two full batches and one step of a third
        for r in range(world):
            src[r].a[r * S:(r + 1) * S] = 1000.0 * r + step + np.arange(S) / 100.0
            batches[r].before_step()
            batches[r].stage(src[r], r * S)
        if step == B - 1 or step == 2 * B - 1:                        # a batch just left: check the buffer it left from
            first_step = step - (B - 1)
            for r in range(world):
                got = batches[r].bufs[batches[r].cur ^ 1].a.reshape(world, B, S)
                for rr in range(world):
                    for b in range(B):
                        assert np.array_equal(got[rr, b], 1000.0 * rr + first_step + b + np.arange(S) / 100.0)
    for r in range(world):
        assert batches[r].fill == 1 and batches[r].sent == 2
        batches[r].flush()
        assert batches[r].fill == 0 and batches[r].sent == 3
    for r in range(world):
        got = batches[r].bufs[batches[r].cur ^ 1].a.reshape(world, B, S)
        for rr in range(world):
            assert np.array_equal(got[rr, 0], 1000.0 * rr + 2 * B + np.arange(S) / 100.0)
        ev = comms[r].events
        # fence of a buffer's slot before its first copy, gathers alternate between the two slots
        assert ev[0] == ("fence", 4) and [e for e in ev if e[0] == "gather"] == [("gather", 4), ("gather", 5), ("gather", 4)]
        assert [e for e in ev if e[0] == "fence"] == [("fence", 4), ("fence", 5), ("fence", 4)]
    assert not net.calls                                              # every collective was matched by every rank


def test_cpu_baseline_legs_share_one_sample():
    """bench.cpu_baseline: the Python, vectorised and C legs are timed on the same seeded lines; C1 also at
    the whole workload, a multi-list cell at a seeded 1/16 of every list; eval counts are exact."""
    from pyrad_amd import synthetic
    cfg = synthetic.config_c1(n_lines=600)
    r = bench.cpu_baseline([dict(cfg, raw_molecules=cfg["molecules"])], "C1", seconds_target=0.3)
    assert r["kind"] == "port" and r["cores"] == 1 and r["value"] > 1e5
    assert "SAME sample" in r["c_port_sample"] and "SAME sample" in r["vectorised_sample"]
    ext = r["at_survey_extent"]
    assert ext["what"] == "the WHOLE workload" and ext["lines"] == 600 and ext["c_port_value"] > r["value"]
    # a three-list cell: the 1/16 extent
    c3 = synthetic.config_c3(n_lines=1600, range_min=600, range_max=640)
    r3 = bench.cpu_baseline([dict(c3, raw_molecules=c3["molecules"])], "C3", seconds_target=0.3)
    assert "1/16" in r3["at_survey_extent"]["what"] and r3["at_survey_extent"]["lines"] == 3 * 100
    assert r3["at_survey_extent"]["whole_workload_seconds_c_port"] > 0
