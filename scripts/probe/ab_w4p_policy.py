"""round 5: where gemm_bf16_w4p takes over from the 8-wave kernel — BASELINE configs[1] (512 x 512, batch 1: launches of 72 ... 504 tiles) and
the Fill batch's shapes under "gemm_w4" 0 (policy), 1 (never), 2 (every launch it can run), 3 (launches of >= 256 tiles)"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from domain_rag_amd import ops
dev = torch.device("cuda:0")
vals = [int(v) for v in os.environ.get("W4", "1,0,3,2").split(",")]
res = {v: [] for v in vals}
for rep in range(3):
    for v in vals:
        ops.set_option("gemm_w4", v)
        res[v].append(bench.side_config1(dev)["ms_per_image"])
ops.set_option("gemm_w4", 0)
for v in vals:
    print(f"configs[1], gemm_w4 = {v}: {min(res[v]):.2f} ms per image (min of 3 x 5 runs; all {[round(x, 2) for x in res[v]]})")
