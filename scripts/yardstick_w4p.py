"""Same-process, interleaved A/B of the product's large-Linear kernel against the vendor yardstick on the DiT's headline shapes (VERDICT round 5,
next-5a): gemm_bf16_w4p ("gemm_kernel" 3), the 8-wave gemm_bf16_t256<0> ("gemm_kernel" 2) and hipBLASLt through torch.matmul (NOT on the
product path: the yardstick only), each run back to back for ~1.2 s per turn, three turns per kernel interleaved, with the socket power and the
shader clock rocm-smi reports during the turn (the kernels sit on the 1400 W cap: a TFLOP/s figure only compares next to its clock).
    python scripts/yardstick_w4p.py > profiles/r06_yardstick_w4p.log"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402  (PowerSampler)
from domain_rag_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [(32768, 9216, 3072), (32768, 3072, 3072), (32768, 12288, 3072), (32768, 3072, 12288), (42696, 21504, 3072), (42696, 3072, 15360)]
TURN_S = 1.2


def turn(fn, flops):
    fn(); fn()
    torch.cuda.synchronize()
    # calibrate the repeat count, then one timed burst under the power sampler
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); fn(); e1.record(); torch.cuda.synchronize()
    reps = max(4, int(TURN_S / (e0.elapsed_time(e1) / 2 * 1e-3)))
    ps = bench.PowerSampler(0, period_s=0.2)
    with ps:
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    s = ps.summary() or {}
    return flops / ms / 1e9, s.get("socket_w_mean"), s.get("sclk_mhz_mean")


print(f"{'shape (M, N, K)':24s} {'gemm_bf16_w4p':>28s} {'gemm_bf16_t256<0>':>28s} {'hipBLASLt (torch.matmul)':>28s}   w4p / hipBLASLt", flush=True)
for (M, N, K) in SHAPES:
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    A = torch.randn(M, K, device=dev, generator=g).bfloat16()
    W = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * N * K

    def ours(kern):
        def f():
            ops.gemm(A, W, out=C)
        def run():
            ops.set_option("gemm_kernel", kern)
            try:
                return turn(f, fl)
            finally:
                ops.set_option("gemm_kernel", 0)
        return run
    runs = {"w4p": ours(3), "t256": ours(2), "lib": lambda: turn(lambda: torch.matmul(A, W.t(), out=C), fl)}
    res = {k: [] for k in runs}
    for _ in range(3):
        for k, r in runs.items():
            res[k].append(r())

    def fmt(v):
        v = sorted(v, key=lambda t: t[0])[1]          # the median turn
        return f"{v[0]:6.0f} TF/s {v[1] or 0:5.0f} W {((v[2] or 0) / 1000):.2f} GHz"
    med = {k: sorted(x[0] for x in v)[1] for k, v in res.items()}
    print(f"{str((M, N, K)):24s} {fmt(res['w4p']):>28s} {fmt(res['t256']):>28s} {fmt(res['lib']):>28s}   {med['w4p'] / med['lib']:.3f}", flush=True)
    del A, W, C
    time.sleep(0.5)
