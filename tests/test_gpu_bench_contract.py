"""GPU: bench.py honours the driver's contract — one JSON line on stdout with the required
keys — on the default path, and with the RCCL communicator forced on (one rank, pipelined
all-gather over two buffer sets)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def run_bench(args, env_extra=None):
    env = dict(os.environ)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, capture_output=True, text=True, env=env,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines          # exactly ONE line on stdout
    return json.loads(lines[0])


def test_default_contract():
    d = run_bench(["--steps", "5", "--warmup", "2", "--cpu-seconds", "1"])
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] == "strong" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["unit"] == "evals/s"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert "CO2+H2O+CH4 100-2500" in d["config"]["workload"] and d["config"]["grid_points_per_gpu"] == 2400000
    assert d["value"] > 1e12 and d["ms_per_step"] > 0
    assert abs(d["value"] - d["config"]["evals_per_step"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and r["launches"] == 2      # steps 0 and 4 of 5 carry events
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 1e5 and "sample" in c
    assert c["vectorised_value"] > c["value"]
    # round 5: the step is ONE merged accumulate job per layer with the sweep in its output stage; the per-list step is a leg
    assert d["config"]["step"].startswith("merged") and "evals_per_step unchanged" in d["config"]["step"]
    assert d["config"]["evals_per_step"] == 3925123383.0
    assert r["sweep_fused_in"] is True and "traffic_stale" in r and "busy_frac" in d["valu_f64"]
    # K2 of the merged three-molecule cell: 56 B per line read; k, transmittance and radiance written, no cross section
    assert r["algorithmic_bytes_per_launch"] == 56.0 * d["config"]["lines_per_gpu"] + 8.0 * 2400000 * 3
    assert d["roofline_sweep"] == {"fused_into": "xsec_accumulate_lds_kernel (lbl_layer_merged_step_dev)"}
    assert d["kernel_ms_per_step"]["layer_sweep"] == 0.0
    pl = d["per_list_leg"]
    assert pl["ms_per_step"] > d["ms_per_step"] and pl["kernel_ms_per_step"]["layer_sweep"] > 0 and pl["xsec_accumulate_launches_per_step"] == 1
    assert "api_path" in d and d["api_path"]["ms_per_call"] > 0
    # round 4: re-windowing legs (the getter AFTER changePressure / changeRange; schedules are built on the device), the
    # steady state in blocks, where each kernel figure comes from, the accuracy mode and the other mode's leg
    a = d["api_path"]
    assert 0 < a["ms_change_pressure"] < 50 and 0 < a["ms_change_range"] < 50 and a["ms_change_pressure_mutator"] >= 0
    assert a["ms_first_call"] > 0 and a["ms_engine_create"] > 0 and "finite True" in a["rewindow_what"]
    bl = d["ms_per_step_blocks"]
    assert len(bl["blocks"]) == 5 and bl["blocks"][0] == pytest.approx(d["ms_per_step"], rel=1e-5) and bl["min"] <= bl["median"] <= bl["max"]
    assert "SEPARATE pass" in d["kernel_ms_per_step"]["source"] and "TIMED REGION" in d["kernel_ms_per_step"]["source"]
    assert d["config"]["accuracy"] == "exact" and "ablated" not in d
    bg = d["budget_leg"]
    assert 0 < bg["ms_per_step"] < d["ms_per_step"] * 1.02 and bg["kernel_ms_per_step"]["xsec_accumulate"] > 0
    # the headline cannot be read as 1e13 evaluated profiles per second: the direct / series split travels with it
    pr = d["pairs"]
    assert pr["pairs_direct"] + pr["pairs_series"] == pr["pairs"] and pr["evals_direct"] + pr["evals_series"] == pr["evals"]
    assert pr["evals"] == d["config"]["evals_per_step"] and 0.8 < pr["evals_series"] / pr["evals"] < 0.95
    assert d["value_direct_kernel"] > 1e12 and 0.2 < d["direct_frac"] < 1.0 and 0 < d["evals_direct_per_s"] < d["value"]
    assert d["in_flight_leg"]["steps_in_flight"] == 2 and d["in_flight_leg"]["evals_per_s"] > 1e12       # N = 1: the other mode, for ratios
    c = d["cpu_baseline"]
    assert "SAME sample" in c["c_port_sample"] and "SAME sample" in c["vectorised_sample"]
    assert "1/16" in c["at_survey_extent"]["what"] and c["at_survey_extent"]["c_port_value"] > 1e7
    # round 6: every other BASELINE configuration rides on the default line as a leg (the driver had only ever timed C3)
    legs = d["workload_legs"]
    assert sorted(legs) == ["C1", "C2", "C5"] and d["value_step"] == "merged" and 0 < d["value_per_list"] < d["value"]
    for w, points, step in (("C1", 10000, "merged"), ("C2", 400000, "merged"), ("C5", 2400000, "merged")):
        g = legs[w]
        assert g["steps"] == 5 and g["warmup"] == 2 and g["n_gpus"] == 1 and g["grid_points_per_gpu"] == points and g["step"] == step
        assert g["ms_per_step"] > 0 and g["evals_per_s"] == pytest.approx(g["evals_per_step"] / (g["ms_per_step"] * 1e-3))
        assert g["kernel_ms_per_step"]["xsec_accumulate"] > 0 and g["kernel_ms_per_step"]["line_prep"] > 0
    assert "600-700" in legs["C1"]["workload"] and "500-900" in legs["C2"]["workload"] and "column" in legs["C5"]["workload"]
    assert legs["C1"]["ms_per_step"] < legs["C2"]["ms_per_step"] < d["ms_per_step"] < legs["C5"]["ms_per_step"]
    c5 = legs["C5"]
    assert c5["evals_per_step"] > 2.5e10 and 0.3 < c5["fold"]["hbm_frac"] < 1.0 and 0.3 < c5["line_prep"]["hbm_frac"] < 1.0
    assert c5["fold"]["algorithmic_bytes_per_launch"] == 8.0 * 2400000 * 31 and c5["kernel_ms_per_step"]["column_sweep"] > 0
    assert c5["api_path"]["ms_per_call"] > 0 and c5["api_path"]["bytes_downloaded_per_call"] == 8 * 2400000
    assert "api_path" not in legs["C1"] and "sharded_step_breakdown" not in c5


def test_forced_single_rank_communicator_pipeline():
    # the default N > 1 path: one step in flight, one collective per step, overlapped with the next step (two buffer sets)
    d0 = run_bench(["--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--workload", "C1"], {"PYRAD_FORCE_COMM": "1"})
    assert d0["config"]["steps_in_flight"] == 1 and d0["config"]["gather_batch"] == 1 and "2 buffer sets" in d0["config"]["allgather"]
    bd = d0["sharded_step_breakdown"]       # what an N > 1 line carries: kernels without the gather, the gather alone, like-for-like legs
    assert bd["kernels_only_ms_per_step"] > 0 and bd["allgather_alone_ms_per_step"] > 0 and len(bd["kernels_only_ms_by_rank"]) == 1
    assert "allgather_alone_GBps_per_rank" in bd and bd["allgather_bytes_received_per_rank_per_step"] == 0.0      # one rank: nothing to receive
    ll = bd["like_for_like"]
    assert ll["in_stream_ms_per_step"] > 0 and ll["overlapped_ms_per_step"] > 0 and ll["step_latency_ms"] > 0
    # opt-in: the gathers of 3 steps batched into one collective
    d = run_bench(["--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--workload", "C1", "--gather-batch", "fit"],
                  {"PYRAD_FORCE_COMM": "1"})
    assert d["config"]["allgather"].startswith("one collective per 3 steps") and d["value"] > 0       # 6 steps, two sets
    assert d["config"]["gather_batch"] == 3 and d["config"]["gather_verified"] is True
    d2 = run_bench(["--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--workload", "C1", "--no-overlap"],
                   {"PYRAD_FORCE_COMM": "1"})
    assert d2["config"]["allgather"] == "in-stream" and d2["kernel_ms_per_step"]["allgather"] > 0


def test_sharded_column_leg_through_the_communicator():
    """`bench.py --gpus N` carries BASELINE config 5 beside config 4: the 30-layer column sharded N-way by contiguous grid
    range with its single all-gather per step and a breakdown - here with one rank forced through the RCCL communicator."""
    d = run_bench(["--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-direct-pass"], {"PYRAD_FORCE_COMM": "1"})
    assert "sharded_step_breakdown" in d and sorted(d["workload_legs"]) == ["C5"]
    g = d["workload_legs"]["C5"]
    assert g["n_gpus"] == 1 and g["step"] == "merged" and g["evals_per_step"] > 2.5e10 and g["ms_per_step"] > 1.0
    assert "outgoing spectrum" in g["config"]["allgather"] and g["kernel_ms_per_step"]["allgather"] > 0
    bd = g["sharded_step_breakdown"]
    assert 0 < bd["kernels_only_ms_per_step"] <= g["ms_per_step"] * 1.1 and bd["allgather_alone_ms_per_step"] > 0
    assert len(bd["kernels_only_ms_by_rank"]) == 1 and bd["allgather_bytes_received_per_rank_per_step"] == 0.0
    assert 0.3 < g["fold"]["hbm_frac"] < 1.0


def test_self_launch_on_a_one_gpu_box_fails_loudly():
    """`python bench.py --gpus 2` with no launcher starts its own two ranks; on a box with one GPU
    rank 1 has no device of its own and says so, the parent ends rank 0 and returns the failure -
    no hang, no silent sharing of a GPU.  (A real 2-rank run needs a 2-GPU node.)"""
    from pyrad_amd import _native as nat
    if nat.device_count() != 1:
        pytest.skip("needs exactly one visible GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--workload", "C1", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode != 0
    assert "one process per GPU" in p.stderr and "LOCAL_RANK 1" in p.stderr
    assert not p.stdout.strip()


@pytest.mark.parametrize("shards", ["balanced", "equal"])
def test_sharded_step_with_the_communicator_pipeline(shards):
    """The code path a rank of an 8-GPU run takes - a shard plan (cost-balanced: out-of-place padded
    gather; equal: in place), two buffer sets, the all-gather of step k overlapped with step k+1, fences -
    driven on one GPU: shard 3 of 8 of the C3 cell with the RCCL communicator forced on (one rank, so the
    collective moves this rank's slot only).  Catches host-side mistakes in the N > 1 branch that the
    gloo tests (no HIP) and the --shard-of runs (no communicator) do not reach."""
    d = run_bench(["--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-api-path", "--shard-of", "8,3",
                   "--shards", shards, "--in-flight", "2", "--gather-batch", "fit"], {"PYRAD_FORCE_COMM": "1"})
    assert d["config"]["allgather"].startswith("one collective per 3 steps") and d["value"] > 1e12
    assert d["config"]["gather_verified"] is True
    b = d["config"]["shard_bounds"]
    assert len(b) == 8 and sum(c for _, c in b) == 2400000 and d["config"]["grid_points_per_gpu"] == b[3][1]
    bd = d["sharded_step_breakdown"]
    assert 0.03 < bd["kernels_only_ms_per_step"] < 0.2 and 0 < bd["allgather_alone_ms_per_step"] < 0.2
    d2 = run_bench(["--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-api-path", "--shard-of", "8,3",
                    "--shards", shards, "--no-overlap", "--gather", "all"], {"PYRAD_FORCE_COMM": "1"})
    assert d2["config"]["allgather"] == "in-stream" and d2["kernel_ms_per_step"]["allgather"] > 0
    assert d["config"]["steps_in_flight"] == 2 and d["config"]["gather_batch"] == 3      # opt-in: two steps in flight, gathers batched
    d1 = run_bench(["--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-api-path", "--shard-of", "8,3",
                    "--shards", shards], {"PYRAD_FORCE_COMM": "1"})           # the default: one in flight, a collective per step
    assert d1["config"]["steps_in_flight"] == 1 and "2 buffer sets" in d1["config"]["allgather"] and d1["config"]["gather_batch"] == 1
    assert d1["config"]["evals_per_step"] == d["config"]["evals_per_step"]
    ll = d1["sharded_step_breakdown"]["like_for_like"]
    assert 0.03 < ll["in_stream_ms_per_step"] < 0.3 and 0.03 < ll["overlapped_ms_per_step"] < 0.3 and ll["step_latency_ms"] > 0.03
    # the column takes the same route
    d3 = run_bench(["--steps", "6", "--warmup", "1", "--no-cpu-baseline", "--workload", "C5", "--shard-of", "8,3",
                    "--shards", shards, "--gather-batch", "fit"], {"PYRAD_FORCE_COMM": "1"})
    assert d3["value"] > 1e11 and "column" in d3["config"]["workload"] and d3["config"]["gather_verified"] is True


def test_column_line_carries_the_atmosphere_api_leg():
    """C5: the line's api_path is pyrad_amd.model.Atmosphere.transmission on the same 30-layer column."""
    d = run_bench(["--workload", "C5", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-direct-pass"])
    a = d["api_path"]
    assert a["ms_per_call"] > 0 and a["bytes_downloaded_per_call"] == 8 * 2400000 and "30 layers" in a["what"] and "finite True" in a["what"]
    assert a["evals_per_s"] > 1e11 and d["roofline_sweep"]["kernel"] == "column_step_kernel"
    assert 0 < a["ms_change_pressure"] < 500 and "30 changePressure calls" in a["rewindow_what"]


def test_budget_mode_line_says_so():
    d = run_bench(["--workload", "C2", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-api-path", "--accuracy", "budget",
                   "--blocks", "2"])
    assert d["config"]["accuracy"] == "budget" and "budget_leg" not in d and len(d["ms_per_step_blocks"]["blocks"]) == 2
