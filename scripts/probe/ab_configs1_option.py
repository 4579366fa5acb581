"""configs[1] under one option's values:  python ab_configs1_option.py <option> <v0> <v1> ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from domain_rag_amd import ops
dev = torch.device("cuda:0")
name, vals = sys.argv[1], [int(v) for v in sys.argv[2:]]
res = {v: [] for v in vals}
for rep in range(3):
    for v in vals:
        ops.set_option(name, v)
        res[v].append(bench.side_config1(dev)["ms_per_image"])
ops.set_option(name, vals[0])
print(name, " | ".join(f"{v}: {min(res[v]):.2f} ms" for v in vals))
