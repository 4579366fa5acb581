"""same-box, same-process A/B of a library option on the whole composite batch (bench.py's job): alternating rounds, images/s.
   python scripts/ab_option.py gemm_group_m 8 0     (value 0 = the library's own policy).  Eager launches (no hipGraph replay)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from domain_rag_amd import ops
from domain_rag_amd.fill_pipeline import SyntheticFillJob
name, vals = sys.argv[1], [int(v) for v in sys.argv[2:]]
job = SyntheticFillJob(batch=8, res=1024, denoise_steps=30, device="cuda:0", seed=1)
job.fill.use_graph = False        # a captured hipGraph replays the kernel arguments of capture time: options would not take effect
job.run_batch(); torch.cuda.synchronize()
res = {v: [] for v in vals}
for rep in range(int(os.environ.get("REPS", "3"))):
    for v in vals:
        ops.set_option(name, v)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        job.run_batch(); torch.cuda.synchronize()
        res[v].append(8 / (time.perf_counter() - t0))
ops.set_option(name, 0)
for v in vals:
    print(f"{name}={v}: " + " ".join(f"{x:.4f}" for x in res[v]) + f"  median {sorted(res[v])[len(res[v]) // 2]:.4f} images/s", flush=True)
