"""bench.py end to end on the GPU box, small shapes: the JSON contract, the self-launched multi-rank path and the native
communicator (RCCL with one rank; two ranks over the shared-memory transport -- RCCL refuses two ranks on one GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--pairs", "60", "--height", "256", "--width", "384", "--steps", "3", "--warmup", "1", "--sustained-s", "1"]


def run_bench(args, env_extra=None):
    env = dict(os.environ, **(env_extra or {}))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, cwd=ROOT, env=env,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]            # ONE JSON line on stdout, nothing else (RCCL's banner goes to stderr)
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_json_contract_single_gpu(gpu):
    d = run_bench(SMALL + ["--cpu-pairs", "4"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["unit"] == "frame-pairs/s" and d["value"] > 0
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["higher_is_better"] is True
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert r["algorithmic_bytes_per_pair"] == 2 * 256 * 384 + 16 * d["config"]["windows_per_pair"]
    # round 5: the dominant kernel timed INSIDE the launch as issued (library events), the burst's sustained twin
    assert "kernel_timing" in r and r["kernel_ms_per_launch"] > 0 and r["kernel_ms_same_launch_rescue_off"] > 0
    assert r["kernel_ms_per_launch"] <= r["launch_ms_with_rescue_kernels"] * 1.05
    assert d["sustained_pairs_per_s"] > 0 and d["config"]["sustained"]["steps"] >= 50 and d["config"]["sustained"]["seconds"] >= 1.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    assert c["parity_nan_mismatch"] == 0 and c["parity_max_rel_err_vs_oracle"] <= 1e-4 and "parity_float64_ties" in c and c["parity_windows_ill_posed"] == 0


@pytest.mark.gpu
def test_bench_launches_its_own_ranks_and_gathers_over_the_native_comm(gpu):
    """`python bench.py --gpus 2` starts two ranks itself; on a 1-GPU box they share the device over the shared-memory
    transport (LSPIV_BENCH_SAME_DEVICE): sharding, pipelined all-gather, max-over-ranks timing and the bit check."""
    d = run_bench(SMALL + ["--gpus", "2", "--strong-pairs", "300"], {"LSPIV_BENCH_SAME_DEVICE": "1"})
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "cpu_baseline" not in d
    comm = d["config"]["comm"]
    assert comm["transport"] == "shm" and comm["ranks_reported_by_transport"] == 2
    assert comm["allgather_matches_single_launch"] is True and comm["same_device_plumbing_test"] is True
    assert comm["allgather_bytes_per_rank_per_step"] == 4 * 4 * 60 * d["config"]["windows_per_pair"]
    # round 4: the step is the library's own sharded path, and the line diagnoses itself
    assert "ShardedPivDev" in comm["path"] and comm["mode"] == "weak" and comm["pairs_total"] == 120 and comm["pairs_rank0"] == 60
    for key in ("kernel_ms_while_gather_in_flight", "gather_ms_overlapped", "exposed_comm_ms", "gather_end_after_kernel_end_ms",
                "allgather_ms_alone", "kernel_ms_alone", "allgather_bytes_received_per_rank_per_step", "rccl_env", "gather_stream_priority"):
        assert key in comm, key
    assert comm["kernel_ms_while_gather_in_flight"] > 0 and comm["gather_ms_overlapped"] > 0 and comm["gather_stream_priority"] == "high"
    assert abs(comm["exposed_comm_ms"] - (d["ms_per_step"] - comm["kernel_ms_while_gather_in_flight"])) < 1e-3
    assert d["config"]["binary"]["binary_hash_matches"] is True
    # round 6: the weak line also carries ONE strong pass -- north_star's reading -- timed after the weak region:
    # a fixed total cut over the ranks on the anchors (300 pairs of this small grid, anchors of 25: 150 + 150)
    assert comm["strong_pairs_total"] == 300 and comm["strong_pairs_rank0"] == 150 and comm["strong_ms_per_step"] > 0
    assert abs(comm["strong_pairs_per_s"] - 300 / (comm["strong_ms_per_step"] * 1e-3)) / comm["strong_pairs_per_s"] < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("n", [4, 8])
def test_bench_at_the_rank_counts_of_the_scaling_run(gpu, n):
    """The driver's N = 4 and 8, as far as one GPU goes: n self-launched ranks share device 0 over the shared-memory transport --
    block sizes, the gathered (n, 4, pairs, windows) result against ONE launch, the max-over-ranks clock and the weak-scaling value."""
    d = run_bench(["--pairs", "25", "--height", "256", "--width", "384", "--steps", "2", "--warmup", "1", "--gpus", str(n)],
                  {"LSPIV_BENCH_SAME_DEVICE": "1"})
    comm = d["config"]["comm"]
    assert d["n_gpus"] == n and d["scaling"] == "weak" and comm["ranks_reported_by_transport"] == n
    assert comm["pairs_total"] == 25 * n and comm["pairs_rank0"] == 25 and comm["allgather_matches_single_launch"] is True
    assert comm["allgather_bytes_received_per_rank_per_step"] == (n - 1) * comm["allgather_bytes_per_rank_per_step"]
    assert abs(d["value"] - 25 * n / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3


@pytest.mark.gpu
def test_bench_strong_mode_cuts_a_fixed_total(gpu):
    """`--strong`: a fixed total (here 150 pairs) cut over the ranks on the walking kernels' anchors (75 + 75), value = total / time."""
    d = run_bench(SMALL + ["--gpus", "2", "--strong", "--strong-pairs", "150"], {"LSPIV_BENCH_SAME_DEVICE": "1"})
    comm = d["config"]["comm"]
    assert d["scaling"] == "strong" and comm["mode"] == "strong" and comm["pairs_total"] == 150 and comm["pairs_rank0"] == 75
    assert comm["allgather_matches_single_launch"] is True
    assert abs(d["value"] - 150 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3


@pytest.mark.gpu
def test_bench_over_rccl_with_one_rank(gpu):
    """The RCCL transport itself (librccl.so dlopen'ed by the C library, ncclCommInitRank, ncclAllGather on a second
    stream, all-reduce for the timing), with the one rank a 1-GPU box allows."""
    d = run_bench(SMALL + ["--gpus", "1", "--cpu-pairs", "0", "--no-extras"], {"LSPIV_BENCH_FORCE_COMM": "1"})
    comm = d["config"]["comm"]
    assert comm["transport"] == "rccl" and comm["ranks_reported_by_transport"] == 1
    assert comm["allgather_matches_single_launch"] is True
    assert comm["rccl_env"].get("NCCL_MAX_NCHANNELS") == "16"       # the default cap on RCCL's CU take is in force and recorded


@pytest.mark.gpu
def test_bench_falls_back_to_shared_memory_when_rccl_cannot_be_brought_up(gpu):
    """Round 6: RCCL has never met two ranks on the boxes this was built on.  A node on which it cannot be brought up must still yield a
    line that says so: the ranks vote on their communicator and, unless all of them succeeded, all of them fall back to the shared-memory
    transport and the line carries the error.  Provoked here with two ranks on ONE device, which RCCL refuses ("Duplicate GPU")."""
    d = run_bench(SMALL + ["--gpus", "2", "--strong-pairs", "300"], {"LSPIV_BENCH_SAME_DEVICE": "1", "LSPIV_COMM": "rccl"})
    comm = d["config"]["comm"]
    assert d["n_gpus"] == 2 and comm["transport"] == "shm" and comm["ranks_reported_by_transport"] == 2
    assert "rccl_error" in comm and "rank" in comm["rccl_error"] and "diagnosis" in comm["note"]
    assert comm["allgather_matches_single_launch"] is True and d["value"] > 0
