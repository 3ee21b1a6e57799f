"""GPU tests that start OTHER processes (ranks, torchrun, bench.py): collected LAST (the file name and conftest's ordering
both see to it), so that a harness fault on some box can never hide a kernel parity test behind `pytest -x`.  No test here
forks the pytest process: ranks are fresh children (tests/_ranks.py), ports are free ports picked at run time."""
import json
import os
import subprocess
import sys

import pytest

from _ranks import ROOT, free_port, run_ranks

pytestmark = [pytest.mark.gpu, pytest.mark.multiprocess]


def test_two_rank_processes_on_the_hip_backend(oracle, tmp_path):
    """world_size 2, one PROCESS per rank (both on cuda:0): HipShard + ShardedSearch's gather layout + the HIP merge give
    the oracle's global top-k on every rank."""
    res = run_ranks("hip_two_rank", 2, [20_001, 64, 33, 10], tmp_path, timeout=600)
    assert len(res) == 2 and all(r["ok"] for r in res), res


def _bench(args, timeout, launcher=True, extra_env=None):
    env = dict(os.environ, ZVDB_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra_env or {})
    if launcher:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.join(ROOT, "bench.py")] + args
    else:
        for v in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
            env.pop(v, None)
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    # the contract line is the LAST line of stdout, small enough for a record that keeps only the tail of stdout (round 5 lost a
    # 20 KB line to that), and carries every key the driver and the judge read; the details are in the full record beside bench.py
    assert r.stdout.rstrip("\n").splitlines()[-1] == lines[-1]
    assert len(lines[-1]) < 4096, len(lines[-1])
    line = json.loads(lines[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "full_record"):
        assert k in line, k
    assert isinstance(line["config"]["workload"], str) and len(line["config"]["workload"]) <= 160 and "model" not in line["config"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    assert "(" not in line["roofline"]["kernel"] and len(line["roofline"]["kernel"]) < 40        # symbol only
    full = json.loads(open(os.path.join(ROOT, line["full_record"])).read())
    assert full["value"] == pytest.approx(line["value"], rel=1e-6) and full["roofline"]["frac"] == pytest.approx(line["roofline"]["frac"], rel=1e-4)
    if line["n_gpus"] > 1:   # the CPU leg runs at N = 1 only; the key is still there (null value) so SCALE records have the same shape
        assert line["cpu_baseline"]["value"] is None and line["cpu_baseline"]["kind"] == "port"
    full["_line"] = line
    return full


def test_bench_graph_route_sharded_rehearsal(oracle):
    """`bench.py --route graph --row-dtype bf16` (BASELINE configs[3]'s route: bf16 rows, HNSW M = 16, sharded) launched as the driver
    launches N > 1 runs, at world size 1 with the RCCL step forced on, on a small shard: one JSON line with the graph route's
    recall against the sharded brute-force route and its HBM roofline object."""
    out = _bench(["--gpus", "1", "--rows", "200000", "--dim", "128", "--steps", "3", "--warmup", "1", "--route", "graph",
                  "--row-dtype", "bf16", "--ef", "64", "--no-cpu-baseline", "--no-paths"], 900)
    assert out["config"]["route"] == "graph" and out["config"]["row_dtype"] == "bf16" and out["n_gpus"] == 1
    assert out["value"] > 0 and 0.0 < out["recall_at_10"] <= 1.0 and out["roofline"]["bound"] == "hbm" and out["roofline"]["frac"] > 0


def test_bench_under_torchrun_with_the_rccl_backend_world_size_1(oracle):
    """bench.py exactly as the driver launches it for N > 1 (`python -m torch.distributed.run ... bench.py --gpus N`), at the
    one world size a one-GPU box allows: the process group is RCCL ("nccl"), the step is search -> ONE packed
    all_gather_into_tensor -> zvdb_hip_merge_topk_packed_dev, the timing is barrier-bracketed and all-reduced (MAX).
    ZVDB_BENCH_FORCE_DIST=1 keeps that path on at world size 1."""
    out = _bench(["--gpus", "1", "--rows", "400000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-paths"], 600)
    assert out["n_gpus"] == 1 and out["dtype"] == "bf16" and out["recall_at_10"] == 1.0
    assert out["value"] > 0 and out["roofline"]["frac"] > 0
    c = out["collective"]  # what the communicator itself reports, and the all-gather timed by HIP events inside the library
    assert c["ranks"] == 1 and c["rccl_ranks_seen"] == [0] and c["distinct_devices"] == 1 and len(c["device_bus_ids"][0]) >= 7
    assert c["all_gather_samples"] >= 1 and c["all_gather_us"] > 0 and "ncclAllGather" in c["source"]


def test_bench_invoked_bare_with_the_rccl_step_on(oracle):
    """`python bench.py --gpus 1` with NO launcher around it (how the driver calls N = 1) and the N > 1 step forced on: the process
    forms its own one-rank RCCL group."""
    out = _bench(["--gpus", "1", "--rows", "400000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-paths"], 600, launcher=False)
    assert out["n_gpus"] == 1 and out["recall_at_10"] == 1.0 and out["collective"]["ranks"] == 1


def test_bench_invoked_bare_with_gpus_2_starts_its_own_ranks(oracle):
    """`python bench.py --gpus 2` with NO launcher: bench.py starts torch.distributed.run itself before touching the GPU and relays
    rank 0's line and the exit code.  On a one-GPU box both ranks share cuda:0 (ZVDB_BENCH_ONE_DEVICE=1) and the collective is
    gloo (two ranks on one device cannot form an RCCL communicator): the sharded step, the (host-staged) all-gather and the HIP
    merge of two id-range shards -- recall 1.0 against the sharded fp32 route."""
    out = _bench(["--gpus", "2", "--rows", "600000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-paths"], 900, launcher=False,
                 extra_env={"ZVDB_BENCH_ONE_DEVICE": "1", "ZVDB_BENCH_BACKEND": "gloo", "ZVDB_BENCH_FORCE_DIST": "0"})
    assert out["n_gpus"] == 2 and out["recall_at_10"] == 1.0 and out["value"] > 0
    c = out["collective"]
    assert c["ranks"] == 2 and c["ranks_reporting"] == 2 and c["rccl_ranks_seen"] == [0, 1]


def test_bench_invoked_bare_with_gpus_4_at_the_drivers_argument_shape(oracle):
    """The driver's N > 1 argument shape (`bench.py --gpus N --steps K --warmup W`, no launcher) at the largest N a one-GPU box
    allows: the pool's process guard admits 6 GPU processes, this pytest process is one of them, so N = 4 (not 8).  Four id-range
    shards of 800k rows on cuda:0, gloo collective: four ranks report, the packed all-gather carries four shards' top-k, recall 1.0,
    and the line keeps the contract's size and keys with a null-valued cpu_baseline (checked in _bench)."""
    out = _bench(["--gpus", "4", "--rows", "800000", "--steps", "3", "--warmup", "1", "--no-paths"], 900, launcher=False,
                 extra_env={"ZVDB_BENCH_ONE_DEVICE": "1", "ZVDB_BENCH_BACKEND": "gloo", "ZVDB_BENCH_FORCE_DIST": "0"})
    assert out["n_gpus"] == 4 and out["recall_at_10"] == 1.0 and out["value"] > 0
    c = out["collective"]
    assert c["ranks"] == 4 and c["ranks_reporting"] == 4 and c["rccl_ranks_seen"] == [0, 1, 2, 3]
    assert "x4" in out["config"]["parallelism"] and "id-range shards" in out["config"]["parallelism"]
    assert out["_line"]["collective"]["ranks"] == 4 and out["_line"]["config"]["parallelism"] == out["config"]["parallelism"]
