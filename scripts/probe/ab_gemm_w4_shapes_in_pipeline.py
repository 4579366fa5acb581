"""per-shape GEMM TFLOP/s INSIDE the pipeline (events on the launch stream, 4 denoise steps) with the 4-wave kernel on ("gemm_w4" 0) and off (1)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from domain_rag_amd import ops
from domain_rag_amd.fill_pipeline import SyntheticFillJob
job = SyntheticFillJob(batch=8, res=1024, denoise_steps=4, device="cuda:0", seed=1)
job.fill.use_graph = False
job.run_batch()
res = {}
for rep in range(2):
    for v in (0, 1):
        ops.set_option("gemm_w4", v)
        rec = ops.GemmRecorder()
        job.run_batch(recorder=rec)
        for (shape, n, ms, tf) in rec.by_shape()[:14]:
            res.setdefault(shape, {}).setdefault(v, []).append((ms, tf, n))
ops.set_option("gemm_w4", 0)
print(f"{'M':>8} {'N':>8} {'K':>6} {'n':>4} | w4 on: ms TF/s | w4 off: ms TF/s | gain")
tot = {0: 0.0, 1: 0.0}
for shape, d in sorted(res.items(), key=lambda kv: -kv[1][1][0][0]):
    a = min(d[0]); b = min(d[1])
    tot[0] += a[0]; tot[1] += b[0]
    print(f"{shape[0]:8d} {shape[1]:8d} {shape[2]:6d} {a[2]:4d} | {a[0]:8.2f} {a[1]:7.0f} | {b[0]:8.2f} {b[1]:7.0f} | {100 * (b[0] / a[0] - 1):+5.1f} %")
print(f"top shapes together: {tot[0]:.1f} ms vs {tot[1]:.1f} ms = {100 * (tot[1] / tot[0] - 1):+.2f} %")
