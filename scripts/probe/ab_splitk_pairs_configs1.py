"""configs[1] (graph replay, as bench.py's side field runs it): split-K never | single launches only (pairs stay merged) | pairs too — interleaved in one process.
The middle mode needs a library whose drag_gemm_bf16_pair honours DRAG_SPLITK_KEEP_PAIRS (a measurement build of round 5: profiles/r05_gemm_splitk_pairs_in_configs1.log);
with the committed library it equals the third."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from domain_rag_amd import ops
dev = torch.device("cuda:0")
modes = {"never": (1, "1"), "singles": (0, "1"), "singles + pairs": (0, None)}   # (DRAG_SPLITK_KEEP_PAIRS existed in the measurement build only)
res = {m: [] for m in modes}
for rep in range(4):
    for m, (v, keep) in modes.items():
        ops.set_option("gemm_splitk", v)
        if keep: os.environ["DRAG_SPLITK_KEEP_PAIRS"] = keep
        else: os.environ.pop("DRAG_SPLITK_KEEP_PAIRS", None)
        res[m].append(bench.side_config1(dev)["ms_per_image"])
ops.set_option("gemm_splitk", 0); os.environ.pop("DRAG_SPLITK_KEEP_PAIRS", None)
print(" | ".join(f"{m}: {min(v):.2f} ms ({[round(x, 1) for x in v]})" for m, v in res.items()))
