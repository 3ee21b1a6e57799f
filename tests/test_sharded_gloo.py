"""World-size-2 (and 3) gloo tests of the multi-GPU host logic on CPU: id-range shards, global id
offsets, all-gather layout, k-way merge.  The compute inside each rank is the ORACLE here (the HIP
backend needs a GPU); what is under test is zvdb_amd/sharded.py's collective + layout logic.
Every rank is a fresh child process (tests/_ranks.py -> tests/rank_worker.py), results come back as files."""
import pytest

from _ranks import run_ranks

pytestmark = pytest.mark.multiprocess


@pytest.mark.parametrize("world,n,d,nq,k", [(2, 1001, 16, 9, 10), (3, 64, 8, 4, 10), (2, 5, 4, 3, 10)])
def test_sharded_topk_allgather_merge(oracle, tmp_path, world, n, d, nq, k):
    res = run_ranks("gloo_bruteforce", world, [n, d, nq, k], tmp_path, timeout=300)
    assert len(res) == world and all(r["ok"] for r in res), res


def test_sharded_graph_route_allgather_merge(oracle, tmp_path):
    """route="graph" through ShardedSearch at world size 2: per-shard graphs over the shards' own id ranges, global id offsets,
    one packed all-gather, k-way merge -- equal to the oracle's merge of the per-shard lists on every rank."""
    res = run_ranks("gloo_graph", 2, [1500, 16, 12, 10, 64], tmp_path, timeout=600)
    assert len(res) == 2 and all(r["ok"] for r in res), res


def test_shard_ranges_partition_the_id_space():
    from zvdb_amd.sharded import shard_range
    for n in (0, 1, 7, 10_000_000, 100_000_003):
        for g in (1, 2, 4, 8):
            rs = [shard_range(n, g, r) for r in range(g)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(g - 1))
            assert max(hi - lo for lo, hi in rs) - min(hi - lo for lo, hi in rs) <= 1


def test_bench_bare_gpus_n_starts_ranks_and_relays_their_exit_code():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE (how the driver may call it): bench.py starts
    torch.distributed.run itself, before any GPU call.  Without GPUs the ranks refuse loudly (the product path has no CPU
    fallback) and the parent's exit code is the launcher's -- non-zero, no JSON line, never a usage error about launchers."""
    import os
    import subprocess
    import sys
    import zvdb_amd
    if zvdb_amd.device_count() >= 2:
        pytest.skip("a multi-GPU box would run the real benchmark here")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rows", "20000", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--no-paths"], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "starting 2 ranks" in r.stderr and "launch with torch.distributed.run" not in r.stderr + r.stdout
    assert "needs a GPU" in r.stderr or "GPU(s) visible" in r.stderr, r.stderr[-2000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
