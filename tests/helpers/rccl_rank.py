"""one rank of the RCCL-on-one-GPU test (tests/test_gpu_rccl_two_ranks.py): joins a backend-"nccl" (= RCCL) communicator through
the product's own entry (domain_rag_amd.rccl.init_rccl), runs the embedding all-gather of retrieval.allgather_rows on a row count
that does not divide by the world size (zero-padded shards) and an all-reduce, and prints one JSON line.
    RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment; every rank uses cuda:0."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main() -> int:
    from domain_rag_amd.rccl import init_rccl, prepare_env
    from domain_rag_amd.retrieval import allgather_rows, shard_bounds
    env = prepare_env()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", 0)
    try:
        dist = init_rccl(dev)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        n_total = 1003                                     # 1003 = 502 + 501: the second shard is padded by one row
        full = torch.arange(n_total * 512, dtype=torch.float32, device=dev).view(n_total, 512) / 7.0
        s, e = shard_bounds(n_total, world, rank)
        got = allgather_rows(full[s:e].clone(), n_total)
        torch.cuda.synchronize()
        ok = bool(torch.equal(got, full)) and int(round(ones.item())) == world
        print(json.dumps({"rank": rank, "world": world, "joined": int(round(ones.item())), "gathered_equal": bool(torch.equal(got, full)),
                          "backend": dist.get_backend(), "env": env, "ok": ok}), flush=True)
        dist.destroy_process_group()
        return 0 if ok else 1
    except Exception as ex:  # noqa: BLE001  (RCCL refuses two ranks on one device on some builds: the caller reports it)
        print(json.dumps({"rank": rank, "error": f"{type(ex).__name__}: {ex}"[:2000]}), flush=True)
        return 3


if __name__ == "__main__":
    sys.exit(main())
