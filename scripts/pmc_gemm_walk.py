"""Workload of the tile-walk / MFMA-shape study of the persistent 256x256 GEMM: for every DiT launch shape and
every variant (gemm_group_m = M tiles per group of the XCD's super-tile; gemm_mfma = 0 the 16x16x32 loop, 1 the 32x32x16 loop) a fixed
number of launches, in a fixed order, so that a rocprofv3 --pmc pass over this process can be cut into per-variant chunks
(scripts/rocpd_gemm_walk.py reads the plan this script writes).

    python scripts/pmc_gemm_walk.py plan.json            under rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace
    TIMED=1 python scripts/pmc_gemm_walk.py plan.json    un-profiled: interleaved event timing of the same variants (TFLOP/s)
"""
import json, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from domain_rag_amd import ops

dev = torch.device("cuda:0")
SHAPES = [tuple(int(v) for v in s.split("x")) for s in os.environ.get(
    "SHAPES", "42696x21504x3072,42696x3072x15360,32768x12288x3072,32768x3072x12288,32768x9216x3072,32768x3072x3072").split(",")]
GROUPS = [int(v) for v in os.environ.get("GROUP_MS", "1,2,4,8,16,32").split(",")]
MFMAS = [int(v) for v in os.environ.get("MFMAS", "0").split(",")]
NTS = [int(v) for v in os.environ.get("STORE_NT", "0").split(",")]
LAUNCHES = int(os.environ.get("LAUNCHES", "4"))
timed = os.environ.get("TIMED", "") not in ("", "0")
plan = []


def set_variant(g, mf, nt=0):
    ops.set_option("gemm_group_m", g)
    if MFMAS != [0]:
        ops.set_option("gemm_mfma", mf)
    if NTS != [0]:
        ops.set_option("gemm_store_nt", nt)


def bench(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for (M, N, K) in SHAPES:
    gen = torch.Generator(device=dev).manual_seed(M + N + K)
    A = torch.randn(M, K, device=dev, generator=gen).bfloat16()
    W = (torch.randn(N, K, device=dev, generator=gen) * 0.02).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    variants = [(g, mf, nt) for mf in MFMAS for nt in NTS for g in GROUPS]
    if timed:
        t = {v: [] for v in variants}
        for v in variants:
            set_variant(*v); bench(lambda: ops.gemm(A, W, out=C), 2)
        for rep in range(int(os.environ.get("REPS", "5"))):
            for v in variants:
                set_variant(*v)
                t[v].append(bench(lambda: ops.gemm(A, W, out=C), 8))
        fl = 2 * M * N * K / 1e9
        print(f"M={M} N={N} K={K}: " + " | ".join(f"mfma{mf} nt{nt} g{g} {fl / statistics.median(ms):.0f}" for (g, mf, nt), ms in t.items()) + "  TFLOP/s",
              flush=True)
    else:
        for v in variants:
            set_variant(*v)
            for _ in range(LAUNCHES):
                ops.gemm(A, W, out=C)
            torch.cuda.synchronize()
            plan.append({"M": M, "N": N, "K": K, "group_m": v[0], "mfma": v[1], "store_nt": v[2], "launches": LAUNCHES})
    del A, W, C
set_variant(0, 0)
if not timed:
    json.dump(plan, open(sys.argv[1], "w"))
