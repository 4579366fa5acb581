"""interleaved A/B of a library option on the DiT's headline Linear shapes (isolated launches, events on the launch stream):
    python scripts/ab_gemm_option.py gemm_epilogue 0 2        -> TFLOP/s per shape and option value, median of 5 interleaved turns of 20 launches"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from domain_rag_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
name, vals = sys.argv[1], [int(v) for v in sys.argv[2:]]
SHAPES = [(32768, 9216, 3072), (32768, 3072, 3072), (32768, 12288, 3072), (32768, 3072, 12288), (42696, 21504, 3072), (42696, 3072, 15360)]


def turn(fn, reps=20):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


try:
    for (M, N, K) in SHAPES:
        g = torch.Generator(device=dev).manual_seed(M + N + K)
        A = torch.randn(M, K, device=dev, generator=g).bfloat16()
        W = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        res = {v: [] for v in vals}
        outs = {}
        for _ in range(5):
            for v in vals:
                ops.set_option(name, v)
                res[v].append(2.0 * M * N * K / turn(lambda: ops.gemm(A, W, out=C)) / 1e9)
                outs[v] = C.clone()
        same = all(torch.equal(outs[v], outs[vals[0]]) for v in vals)
        print(f"{str((M, N, K)):24s} " + "  ".join(f"{name}={v}: {sorted(res[v])[2]:6.0f} TF/s" for v in vals) + ("  same bits" if same else "  BITS DIFFER"), flush=True)
        del A, W, C
finally:
    ops.set_option(name, 0)
