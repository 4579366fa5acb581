"""N>1 data path on CPU with gloo (world_size 2 and 3): shard rule, the single all-gather of embedding shards,
global index space, and result merging — the same code the GPUs run with backend="nccl" (RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from domain_rag_amd.retrieval import allgather_rows, shard_bounds
        from domain_rag_amd.hostlogic import split_samples_for_gpus
        g = torch.Generator().manual_seed(123)
        full = torch.randn(n_total, 16, generator=g)
        s, e = shard_bounds(n_total, world, rank)
        gathered = allgather_rows(full[s:e].clone(), n_total)
        ok = torch.equal(gathered, full)
        # unit sharding of generation jobs: every unit exactly once, contiguous, reference rule
        units = list(range(n_total))
        mine = split_samples_for_gpus(units, world)[rank] if world > 1 else units
        allu = [None] * world
        dist.all_gather_object(allu, mine)
        ok = ok and sorted(sum(allu, [])) == units and mine == list(range(s, e))
        # max-over-ranks timing reduction used by bench.py
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok = ok and t.item() == float(world)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total", [(2, 11), (2, 8), (3, 10)])
def test_allgather_and_sharding_gloo(world, n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(r, True) for r in range(world)]


def test_single_process_is_identity():
    from domain_rag_amd.retrieval import allgather_rows
    x = torch.randn(5, 4)
    assert allgather_rows(x, 5) is x


def _ts_worker(rank, world, port, q):
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.pop("DRAG_TIMESTAMP", None)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from domain_rag_amd.cli import stage2_generate as S2
    time.sleep(1.2 * rank)                      # the ranks reach the call in different seconds
    a = S2.run_timestamp(world)
    time.sleep(1.1)
    b = S2.run_timestamp(world)                 # a second call is a second broadcast: again rank 0's clock, again shared (main() calls it once per run)
    q.put((rank, a, b))
    if dist.is_initialized():
        dist.destroy_process_group()


def test_stage2_ranks_agree_on_rank0s_timestamp_over_gloo():
    """ranks of one launch write ONE results directory per dataset whatever started them (the parent-process
    start time only works for same-node children of one parent): rank 0's clock is broadcast over a gloo group"""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ts_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    firsts, seconds = {r[1] for r in res}, {r[2] for r in res}
    assert len(firsts) == 1 and len(seconds) == 1 and firsts != seconds, res


# ---- the eight-way stage-1 exchange of BASELINE configs[4], end to end on CPU: gloo world 8, N = 118 287 unit-norm rows ----
def _row_block(start: int, stop: int, dim: int = 512) -> torch.Tensor:
    """rows [start, stop) of the synthetic corpus; every row depends only on its global index (1024-row blocks, one seed each), so a
    shard can be produced without the others"""
    out = torch.empty(stop - start, dim)
    b0 = start // 1024
    for b in range(b0, (stop + 1023) // 1024):
        g = torch.Generator().manual_seed(9000 + b)
        blk = torch.randn(1024, dim, generator=g)
        lo, hi = max(start, b * 1024), min(stop, (b + 1) * 1024)
        out[lo - start:hi - start] = blk[lo - b * 1024:hi - b * 1024]
    return out / out.norm(dim=1, keepdim=True)


def _exchange_worker(rank, world, port, n_total, n_query, k, out_json, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import json
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from domain_rag_amd.retrieval import allgather_rows, shard_bounds
        from oracle import retrieval as oretr          # the CPU checker stands in for the HIP top-k (no GPU here)
        s, e = shard_bounds(n_total, world, rank)
        corpus = allgather_rows(_row_block(s, e), n_total).numpy()          # the ONE collective of the data path
        queries = _row_block(n_total, n_total + n_query).numpy()            # queries: rows past the corpus
        qs, qe = shard_bounds(n_query, world, rank)
        D, I = oretr.cosine_topk(corpus, queries[qs:qe], k) if qe > qs else (None, None)
        parts = [None] * world
        dist.gather_object((qs, qe, None if D is None else D.tolist(), None if I is None else I.tolist()), parts if rank == 0 else None, dst=0)
        if rank == 0:
            res = {}
            for a, b, d_, i_ in parts:
                for j in range(a, b):
                    res[str(j)] = {"scores": d_[j - a], "indices": i_[j - a]}
            with open(out_json, "w") as f:
                json.dump({"n_total": n_total, "k": k, "world": world, "results": res}, f)
        dist.barrier()
        q.put((rank, True))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_eight_way_stage1_exchange_gloo(tmp_path):
    """shard_bounds -> per-rank shard -> allgather_rows -> per-rank top-100 of its query slice -> gather to rank 0 -> JSON: (D, I) must equal the
    single-process result bit for bit (retrieval/clip100_resnet_style_all_shots.py:298, 419; outpainting_updown_sampling_redux.py:157-177)"""
    import json
    import numpy as np
    from oracle import retrieval as oretr
    world, n_total, n_query, k = 8, 118287, 16, 100
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    out_json = str(tmp_path / "stage1_exchange.json")
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, n_total, n_query, k, out_json, q)) for r in range(world)]
    for p in procs:
        p.start()
    full = _row_block(0, n_total).numpy()
    D, I = oretr.cosine_topk(full, _row_block(n_total, n_total + n_query).numpy(), k)
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(res) == [(r, True) for r in range(world)]
    got = json.load(open(out_json))
    assert got["world"] == world and len(got["results"]) == n_query
    for j in range(n_query):
        assert got["results"][str(j)]["indices"] == I[j].tolist(), j
        assert np.array_equal(np.asarray(got["results"][str(j)]["scores"], dtype=np.float32), D[j]), j
