"""bench.py's launcher on CPU: `--gpus N` starts N ranks itself (gloo self-test mode: rendezvous, rank count by all-reduce, the
reference's unit sharding, the one all-gather into the global row order), refuses to run N ranks on fewer GPUs and refuses a
--gpus / WORLD_SIZE mismatch — it can no longer print n_gpus for ranks that did not run (outpainting_updown_sampling_redux.py
:157-177,1605-1715 is the fan-out it stands in for)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_gpus_flag_spawns_that_many_ranks():
    for n in (2, 3):
        r = _run(["--gpus", str(n), "--selftest-launcher"])
        assert r.returncode == 0, r.stderr[-2000:]
        out = _json_line(r.stdout)
        assert out == {"selftest": "launcher", "n_ranks": n, "joined_ranks": n, "backend": "gloo", "ok": True}


def test_single_rank_selftest_needs_no_rendezvous():
    r = _run(["--gpus", "1", "--selftest-launcher"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_line(r.stdout)["n_ranks"] == 1


def test_more_ranks_than_gpus_fails_loudly():
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    r = _run(["--gpus", str(max(have + 1, 2)), "--no-cpu-baseline"], timeout=120)
    assert r.returncode != 0
    assert "GPU(s) are visible" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_world_size_mismatch_fails_loudly():
    r = _run(["--gpus", "2", "--selftest-launcher"], env_extra={"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"}, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr and "n_ranks" not in r.stdout


import pytest  # noqa: E402


@pytest.mark.gpu
def test_retrieval_workload_with_three_ranks_sharing_one_gpu():
    """the N > 1 control flow of the retrieval bench line on real kernels: `--debug-share-gpu` puts every rank on GPU 0 and the
    collectives on gloo (NOT a measurement: the line says so) — corpus and query sharding, the one all-gather into the global row
    order, rank 0's report; every planted neighbour must come back first on rank 0's query shard"""
    r = _run(["--gpus", "3", "--debug-share-gpu", "--workload", "retrieval", "--steps", "1", "--warmup", "0", "--corpus", "5000",
              "--queries", "48", "--no-cpu-baseline"], timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 3 and out["rccl_ranks"] == 3 and "NOT a measurement" in out["debug_share_gpu"]
    assert out["planted_neighbour_first"].startswith("16/16")
    assert out["allgather"]["bytes_received_per_gpu"] == 2 * 1667 * 512 * 4


@pytest.mark.gpu
def test_retrieval_workload_with_eight_ranks_sharing_one_gpu():
    """the REAL rank count of the `--gpus 8` line, before a SCALE record is ever taken: eight processes on GPU 0
    (gloo collectives, a reduced corpus; NOT a measurement) run the corpus / query sharding of eight ranks (500 rows and 8 queries each),
    the one all-gather of eight padded shards into the global row order (seven foreign shards received per rank) and rank 0's merge and
    report; every planted neighbour of rank 0's query shard must come back first"""
    r = _run(["--gpus", "8", "--debug-share-gpu", "--workload", "retrieval", "--steps", "1", "--warmup", "0", "--corpus", "4000",
              "--queries", "64", "--no-cpu-baseline"], timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 8 and out["rccl_ranks"] == 8 and "NOT a measurement" in out["debug_share_gpu"]
    assert out["planted_neighbour_first"].startswith("8/8")
    assert out["allgather"]["bytes_received_per_gpu"] == 7 * 500 * 512 * 4
