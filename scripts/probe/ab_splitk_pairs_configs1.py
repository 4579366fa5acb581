"""configs[1] with split-K off / on (run once with DRAG_SPLITK_KEEP_PAIRS=1: pairs stay merged, once without)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from domain_rag_amd import ops
dev = torch.device("cuda:0")
res = {0: [], 1: []}
for rep in range(4):
    for v in (1, 0):
        ops.set_option("gemm_splitk", v)
        res[v].append(bench.side_config1(dev)["ms_per_image"])
ops.set_option("gemm_splitk", 0)
print(f"KEEP_PAIRS={os.environ.get('DRAG_SPLITK_KEEP_PAIRS')}: never {min(res[1]):.2f} ms | policy {min(res[0]):.2f} ms ({100 * (min(res[1]) / min(res[0]) - 1):+.1f} %)  all: {[round(x, 1) for x in res[1]]} {[round(x, 1) for x in res[0]]}")
