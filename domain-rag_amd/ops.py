"""Thin Python wrappers over the C ABI (include/domainrag_hip.h).

Tensors are torch CUDA(ROCm) tensors used only as device-memory handles: every wrapper hands raw
pointers + sizes to libdomainrag_hip.so on torch's current stream.  No arithmetic is done by torch
here, and nothing falls back to torch when the library is missing (``_lib.load()`` raises).
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import _lib
from ._lib import GemmArgs, check

ACT_NONE, ACT_GELU_TANH, ACT_SILU, ACT_QUICK_GELU, ACT_GELU_ERF = 0, 1, 2, 3, 4


class GemmRecorder:
    """Per-launch timing of the GEMM kernel with events on the launch stream (bench.py's roofline).  ``every`` = n
    samples the denoise loop: the GEMMs of every n-th DiT step are bracketed (all steps launch the same shapes), the
    other steps run the way the product does (hipGraph replay, no events) so the timed region is not slowed by ~16k
    event records."""

    def __init__(self, every: int = 1, f32: bool = False):
        self.every = max(1, int(every))
        self.f32 = f32              # also bracket the float32 matrix-core launches (conv2d_f32_kernel: the CLIP tower, LaMa)
        self.events = []
        self.shapes = []
        self.is_conv = []
        self.attn = []              # (start, end, flops, kernel name) of the bf16 attention launches of the sampled steps

    def totals(self):
        torch.cuda.synchronize()
        flops = sum(f for _, _, f in self.events)
        ms = sum(s.elapsed_time(e) for s, e, _ in self.events)
        return flops, ms, len(self.events)

    @staticmethod
    def kernel_of(shape, conv=False):
        """the kernel drag_gemm_bf16 / drag_gemm_bf16_pair / drag_conv3x3_bf16 dispatch this launch to (``drag_gemm_bf16_choice``:
        the library's own tile policy); ``conv`` = ("pair", M1, M2) for a merged pair launch"""
        if conv == "f32":
            return "conv2d_f32_kernel"
        M, N, K = shape
        if isinstance(conv, tuple) and conv[0] == "splitk":  # drag_gemm_bf16 as stacked K slices + the reduce pass (one recorded interval)
            return "gemm_bf16_w4p + splitk_reduce_kernel"
        M1, M2 = (conv[1], conv[2]) if isinstance(conv, tuple) else (M, 0)
        if conv is True:                                     # the 3x3 convolutions take t256 or t128 only, by their own predicate
            return "gemm_bf16_t256<1>" if _lib.load().drag_conv3x3_bf16_choice(M, N, K // 9) == 2 else "gemm_bf16_t128<1>"
        code = _lib.load().drag_gemm_bf16_choice(M1, M2, N, K)
        if code == 3:
            return "gemm_bf16_w4p"
        if code == 2:
            return "gemm_bf16_t256_pair" if M2 > 0 else "gemm_bf16_t256<0>"
        if code == 0:
            return "gemm_bf16_t128<0>"
        mi, st = (code % 100) // 10, code % 10
        return f"gemm_bf16_deep<{mi}, {st}, 6>" if code >= 100 else f"gemm_bf16_deep<{mi}, {st}>"

    def by_kernel(self):
        """{kernel name: (launches, total_ms, flops)}"""
        torch.cuda.synchronize()
        agg: dict = {}
        for (s, e, f), shp, cv in zip(self.events, self.shapes, self.is_conv):
            a = agg.setdefault(self.kernel_of(shp, cv), [0, 0.0, 0.0])
            a[0] += 1; a[1] += s.elapsed_time(e); a[2] += f
        return agg

    def attention_by_kernel(self):
        """{kernel name: (launches, total_ms, flops)} of the recorded drag_attention*_bf16 launches"""
        torch.cuda.synchronize()
        agg: dict = {}
        for s, e, f, name in self.attn:
            a = agg.setdefault(name, [0, 0.0, 0.0])
            a[0] += 1; a[1] += s.elapsed_time(e); a[2] += f
        return agg

    def by_shape(self):
        """{(M, N, K): (launches, total_ms, TFLOP/s)} sorted by total time"""
        torch.cuda.synchronize()
        agg: dict = {}
        for (s, e, f), shp in zip(self.events, self.shapes):
            a = agg.setdefault(shp, [0, 0.0, 0.0])
            a[0] += 1; a[1] += s.elapsed_time(e); a[2] += f
        return sorted(((k, v[0], v[1], v[2] / (v[1] * 1e-3) / 1e12 if v[1] > 0 else 0.0) for k, v in agg.items()), key=lambda r: -r[2])


_recorder: GemmRecorder | None = None


def set_recorder(rec: GemmRecorder | None) -> None:
    global _recorder
    _recorder = rec


def set_option(name: str, value: int) -> None:
    """measurement switch of the library (``drag_set_option``): "attn_sched", "attn_w4" """
    check(_lib.load().drag_set_option(name.encode(), int(value)), "drag_set_option")


def experiments_built() -> bool:
    """True when libdomainrag_hip.so was built with DRAG_EXPERIMENTS=1 (csrc/drag_common.h): only then does it carry the kernels behind
    "attn_persist", "attn_sched" = 3 and "topk_qt" — measured non-improvements kept for their A/B records, not product code"""
    return bool(_lib.load().drag_experiments_built())        # a pure query: no option is read or written


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _need(t: torch.Tensor, dtype, name: str):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a GPU tensor (domain-rag_amd has no CPU path)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")


def _gemm_args(a, w, out, bias, act, act_n0, gate, resid, out_f32, M, a_rows_per_batch, a_batch_stride, lda, c_rows_per_batch,
               c_batch_stride, ldc, ldg, out2, ldc2, n_split):
    """(GemmArgs, out, (M, N, K)) of one Linear (argument meaning: ``gemm``)"""
    _need(a, torch.bfloat16, "gemm.a")
    _need(w, torch.bfloat16, "gemm.w")
    N, K = w.shape
    if not w.is_contiguous():
        raise ValueError("gemm.w must be contiguous [N, K]")
    if M is None:
        M = a.numel() // K
        if lda is None:
            if a.dim() < 2 or a.stride(-1) != 1:
                raise ValueError("gemm.a must have unit inner stride")
            lda = a.stride(-2) if a.dim() >= 2 else K
    if lda is None:
        lda = K
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32 if out_f32 else torch.bfloat16, device=a.device)
    if ldc is None:
        # a 2-D destination carries its own row stride (a two-destination launch writes n_split columns into `out`, not N: defaulting to N
        # there wrote past the tensor — round 5, found as a GPU memory fault by a test that forgot ldc)
        ldc = out.stride(-2) if out.dim() >= 2 and out.stride(-1) == 1 else N
    if c_rows_per_batch == 0 and (out2 is None or (0 < n_split < N and n_split % 256 == 0)):     # (an invalid n_split is the library's to report)
        # last element written into each destination must lie inside its storage: a wrong ldc is an error here, not a memory fault there
        def _room(t):
            return t.untyped_storage().nbytes() // t.element_size() - t.storage_offset()
        cols = n_split if out2 is not None else N
        if (M - 1) * ldc + cols > _room(out):
            raise ValueError(f"gemm.out: {M} rows of stride {ldc} x {cols} columns do not fit the destination ({_room(out)} elements from its start)")
        if out2 is not None and (M - 1) * ldc2 + (N - n_split) > _room(out2):
            raise ValueError(f"gemm.out2: {M} rows of stride {ldc2} x {N - n_split} columns do not fit the destination ({_room(out2)} elements from its start)")
    args = GemmArgs()
    args.A, args.W, args.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    args.bias = bias.data_ptr() if bias is not None else None
    args.gate = gate.data_ptr() if gate is not None else None
    args.resid = resid.data_ptr() if resid is not None else None
    args.M, args.N, args.K = M, N, K
    args.lda, args.a_rows_per_batch, args.a_batch_stride = lda, a_rows_per_batch, a_batch_stride
    args.ldc, args.c_rows_per_batch, args.c_batch_stride = ldc, c_rows_per_batch, c_batch_stride
    args.ldg, args.act, args.act_n0, args.out_f32 = ldg, act, act_n0, int(out_f32)
    if out2 is not None:        # columns >= n_split go to out2 (dense rows of ldc2 elements)
        _need(out2, torch.bfloat16, "gemm.out2")
        args.C2, args.ldc2, args.n_split = out2.data_ptr(), ldc2, n_split
    return args, out, (M, N, K)


def _recorded(call, shape, kind=False):
    """run one GEMM launch, bracketed by events when a recorder is installed"""
    if _recorder is None:
        call()
        return
    s_ev, e_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s_ev.record()
    call()
    e_ev.record()
    M, N, K = shape
    _recorder.events.append((s_ev, e_ev, 2.0 * M * N * K))
    _recorder.shapes.append(shape)
    _recorder.is_conv.append(kind)


def _recorded_attention(call, B, S, H, qprep, vrow):
    """run one head_dim-128 attention launch, bracketed by events when a recorder is installed (4 S^2 128 flops per batch and head:
    q k^T and p v; the kernel name from the library's own dispatch, ``drag_attention_bf16_choice``)"""
    if _recorder is None:
        call()
        return
    s_ev, e_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s_ev.record()
    call()
    e_ev.record()
    fam = _lib.load().drag_attention_bf16_choice(S, int(vrow), int(qprep))
    q = "true" if qprep else "false"
    name = {64: f"attention_q64_kernel<{q}>", 640: f"attention_q64g_kernel<{q}, false>", 641: f"attention_q64g_kernel<{q}, true>"}.get(
        fam, f"attention_d128_kernel<{fam}, ..., {q}, ...>")
    _recorder.attn.append((s_ev, e_ev, 4.0 * S * S * 128 * H * B, name))


_gemm_workspaces: dict[int, torch.Tensor] = {}
_retired_workspaces: list[torch.Tensor] = []      # replaced workspaces stay alive: captured graphs may still hold their addresses
_workspace_stream: dict[int, int] = {}            # device -> the stream whose launches last used the workspace


def _workspace_guard(a: torch.Tensor) -> None:
    """the library picks the workspace of hipGetDevice() and one buffer serves every stream of a device: refuse a launch whose operands
    live on another device than the current one, and order a launch on a NEW stream behind the previous stream's work (split launches
    on two streams would otherwise race on the partial sums).  Nothing is recorded during capture: a graph is one stream's work."""
    idx = a.device.index
    if idx != torch.cuda.current_device():
        raise RuntimeError(f"gemm: operands on cuda:{idx} but the current device is cuda:{torch.cuda.current_device()} "
                           "(the library launches on — and takes the split-K workspace of — the current device)")
    if torch.cuda.is_current_stream_capturing():
        return
    st = torch.cuda.current_stream()
    last = _workspace_stream.get(idx)
    if last is not None and last != st.cuda_stream:
        ev = torch.cuda.Event()
        ev.record(torch.cuda.ExternalStream(last, device=a.device) if last != 0 else torch.cuda.default_stream(a.device))
        st.wait_event(ev)
    _workspace_stream[idx] = st.cuda_stream


def gemm_workspace(device: torch.device, mbytes: int | None = None) -> None:
    """register this device's split-K workspace with the library (``drag_gemm_set_workspace``): ``gemm`` does it once per device with
    $DRAG_GEMM_WORKSPACE_MB (default 64; 0 = none, no launch is split) — a Linear of few output tiles and a long K (batch-1 proj_out /
    ff down-projections) then runs as stacked K slices + one reduce pass.  Not taken during stream capture (the buffer must outlive the
    graph: call this before capturing)."""
    import os
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if mbytes is None:
        mbytes = int(os.environ.get("DRAG_GEMM_WORKSPACE_MB", "64"))
    lib = _lib.load()
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("gemm_workspace: not during stream capture (the buffer must outlive the graph: register it before capturing)")
    old = _gemm_workspaces.get(idx)
    if old is not None and old.numel() > 0:
        # a hipGraph captured while `old` was registered replays launches that hold its address (flux.forward_graphed): the buffer is
        # retired, not freed — a re-registration costs memory, never a use-after-free on replay
        _retired_workspaces.append(old)
    with torch.cuda.device(idx):
        if mbytes <= 0:
            _gemm_workspaces[idx] = torch.empty(0, dtype=torch.uint8, device=device)
            check(lib.drag_gemm_set_workspace(None, 0), "drag_gemm_set_workspace")
            return
        ws = torch.empty(mbytes << 20, dtype=torch.uint8, device=device)
        check(lib.drag_gemm_set_workspace(ctypes.c_void_p(ws.data_ptr()), ws.numel()), "drag_gemm_set_workspace")
        _gemm_workspaces[idx] = ws


def gemm(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor | None = None, *, bias=None, act: int = ACT_NONE,
         act_n0: int = 0, gate=None, resid=None, out_f32: bool = False, M: int | None = None,
         a_rows_per_batch: int = 0, a_batch_stride: int = 0, lda: int | None = None,
         c_rows_per_batch: int = 0, c_batch_stride: int = 0, ldc: int | None = None, ldg: int = 0,
         out2: torch.Tensor | None = None, ldc2: int = 0, n_split: int = 0) -> torch.Tensor:
    """out = epi(a @ w.T) (``out2``: output columns >= ``n_split`` are written there instead, as dense rows of ``ldc2``
    elements).  ``a`` / ``out`` may be views into larger buffers: pass the logical row
    count ``M`` and the batched-row addressing (rows_per_batch, batch_stride, ld) explicitly; by
    default ``a`` is a dense [M, K] matrix and ``out`` a dense [M, N] one."""
    lib = _lib.load()
    if a.is_cuda and a.device.index not in _gemm_workspaces and not torch.cuda.is_current_stream_capturing():
        gemm_workspace(a.device)
    if a.is_cuda:
        _workspace_guard(a)
    args, out, shape = _gemm_args(a, w, out, bias, act, act_n0, gate, resid, out_f32, M, a_rows_per_batch, a_batch_stride, lda,
                                  c_rows_per_batch, c_batch_stride, ldc, ldg, out2, ldc2, n_split)
    slices = lib.drag_gemm_bf16_splitk_slices(ctypes.byref(args)) if _recorder is not None else 0
    _recorded(lambda: check(lib.drag_gemm_bf16(ctypes.byref(args), _stream()), "drag_gemm_bf16"), shape, ("splitk", slices) if slices else False)
    return out


def gemm_cost(M: int, N: int, K: int, M2: int = 0) -> int:
    """the tile policy's cost of one launch (``drag_gemm_bf16_cost``: tile rounds on the busiest CU x (tile rows + columns)) —
    comparable between launches of equal K; used to choose between one fused launch and two"""
    return int(_lib.load().drag_gemm_bf16_cost(M, M2, N, K))


def gemm_pair(first: dict, second: dict):
    """Two Linears with their own operands and the same N, K, activation and output type (``drag_gemm_bf16_pair``): ``first`` /
    ``second`` are the keyword arguments of ``gemm`` (``a``, ``w``, ``out`` included).  One launch unless both problems fill the chip
    alone; bit-identical to ``gemm(**first); gemm(**second)`` either way.  A FluxTransformerBlock's image-stream / text-stream
    pairs (to_q|k|v + add_q|k|v_proj, to_out + to_add_out, ff + ff_context)."""
    lib = _lib.load()
    packed = []
    for kw in (first, second):
        kw = dict(kw)
        a, w, out = kw.pop("a"), kw.pop("w"), kw.pop("out", None)
        d = dict(bias=None, act=ACT_NONE, act_n0=0, gate=None, resid=None, out_f32=False, M=None, a_rows_per_batch=0, a_batch_stride=0,
                 lda=None, c_rows_per_batch=0, c_batch_stride=0, ldc=None, ldg=0)
        unknown = set(kw) - set(d)
        if unknown:
            raise TypeError(f"gemm_pair: unexpected arguments {sorted(unknown)}")
        d.update(kw)
        packed.append(_gemm_args(a, w, out, d["bias"], d["act"], d["act_n0"], d["gate"], d["resid"], d["out_f32"], d["M"], d["a_rows_per_batch"],
                                 d["a_batch_stride"], d["lda"], d["c_rows_per_batch"], d["c_batch_stride"], d["ldc"], d["ldg"], None, 0, 0))
    (a1, o1, s1), (a2, o2, s2) = packed
    if s1[1:] != s2[1:]:
        raise ValueError(f"gemm_pair: the two problems must share N and K (got {s1} and {s2})")
    if (a1.act, a1.act_n0, a1.out_f32) != (a2.act, a2.act_n0, a2.out_f32):
        raise ValueError("gemm_pair: the two problems must share activation, act_n0 and output type")
    if first["a"].is_cuda and first["a"].device.index not in _gemm_workspaces and not torch.cuda.is_current_stream_capturing():
        gemm_workspace(first["a"].device)
    if first["a"].is_cuda:
        _workspace_guard(first["a"])
    slices = lib.drag_gemm_bf16_pair_splitk_slices(ctypes.byref(a1), ctypes.byref(a2))
    if slices:                   # one partial launch over both problems' rows + a reduce pass each (one recorded interval)
        _recorded(lambda: check(lib.drag_gemm_bf16_pair(ctypes.byref(a1), ctypes.byref(a2), _stream()), "drag_gemm_bf16_pair"),
                  (s1[0] + s2[0], s1[1], s1[2]), ("splitk", slices))
    elif lib.drag_gemm_bf16_pair_merges(s1[0], s2[0], s1[1], s1[2]):
        _recorded(lambda: check(lib.drag_gemm_bf16_pair(ctypes.byref(a1), ctypes.byref(a2), _stream()), "drag_gemm_bf16_pair"),
                  (s1[0] + s2[0], s1[1], s1[2]), ("pair", s1[0], s2[0]))
    else:       # two launches (what the library would issue itself), accounted one by one
        for args_i, s_i in ((a1, s1), (a2, s2)):
            sl = lib.drag_gemm_bf16_splitk_slices(ctypes.byref(args_i)) if _recorder is not None else 0
            _recorded(lambda: check(lib.drag_gemm_bf16(ctypes.byref(args_i), _stream()), "drag_gemm_bf16"), s_i, ("splitk", sl) if sl else False)
    return o1, o2


def qk_norm_rope_vt(qkv: torch.Tensor, vt: torch.Tensor, wq_txt, wk_txt, wq_img, wk_img, rope_cos, rope_sin,
                    B: int, S: int, H: int, ld: int, s_txt: int, eps: float = 1e-6) -> None:
    lib = _lib.load()
    _need(qkv, torch.bfloat16, "qkv")
    _need(vt, torch.bfloat16, "vt")
    if rope_cos is not None:
        _need(rope_cos, torch.float32, "rope_cos")
    check(lib.drag_qk_norm_rope_vt_bf16(_p(qkv), _p(vt), _p(wq_txt), _p(wk_txt), _p(wq_img), _p(wk_img),
                                        _p(rope_cos), _p(rope_sin), B, S, H, ld, s_txt, eps, _stream()),
          "drag_qk_norm_rope_vt_bf16")


def attention(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, out: torch.Tensor, B: int, S: int, H: int,
              ld_qk: int, qk_batch_stride: int, ld_o: int, o_batch_stride: int, scale: float) -> None:
    lib = _lib.load()
    _need(q, torch.bfloat16, "q")
    _need(out, torch.bfloat16, "out")
    _recorded_attention(lambda: check(lib.drag_attention_bf16(_p(q), _p(k), _p(vt), _p(out), B, S, H, ld_qk, qk_batch_stride, ld_o,
                                                              o_batch_stride, scale, _stream()), "drag_attention_bf16"), B, S, H, False, False)


def k_norm_rope_vt(qkv: torch.Tensor, vt: torch.Tensor, wk_txt, wk_img, rope_cos, rope_sin, B: int, S: int, H: int, ld: int,
                   s_txt: int, eps: float = 1e-6) -> None:
    """k RMSNorm + RoPE in place and V -> V^T; q is left as projected (``attention_qprep`` prepares it on load)"""
    lib = _lib.load()
    _need(qkv, torch.bfloat16, "qkv")
    if vt is not None:              # None: k only (``attention_v`` reads v row-major)
        _need(vt, torch.bfloat16, "vt")
    check(lib.drag_k_norm_rope_vt_bf16(_p(qkv), _p(vt), _p(wk_txt), _p(wk_img), _p(rope_cos), _p(rope_sin), B, S, H, ld, s_txt,
                                       eps, _stream()), "drag_k_norm_rope_vt_bf16")


def attention_qprep(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, out: torch.Tensor, B: int, S: int, H: int, ld_qk: int,
                    qk_batch_stride: int, ld_o: int, o_batch_stride: int, scale: float, wq_txt, wq_img, rope_cos, rope_sin,
                    s_txt: int, eps: float = 1e-6) -> None:
    """attention over the RAW q projection: norm_q + RoPE are applied to each query row as its fragments are loaded"""
    lib = _lib.load()
    _need(q, torch.bfloat16, "q")
    _need(out, torch.bfloat16, "out")
    if rope_cos is not None:
        _need(rope_cos, torch.float32, "rope_cos")
    _recorded_attention(lambda: check(lib.drag_attention_qprep_bf16(_p(q), _p(k), _p(vt), _p(out), B, S, H, ld_qk, qk_batch_stride, ld_o,
                                                                    o_batch_stride, scale, _p(wq_txt), _p(wq_img), _p(rope_cos), _p(rope_sin),
                                                                    s_txt, eps, _stream()), "drag_attention_qprep_bf16"), B, S, H, True, False)


def attention_v(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, B: int, S: int, H: int, ld_qk: int,
                qk_batch_stride: int, ld_o: int, o_batch_stride: int, scale: float, wq_txt=None, wq_img=None, rope_cos=None,
                rope_sin=None, s_txt: int = 0, eps: float = 1e-6) -> None:
    """attention over q | k | v as the Linears wrote them: v is read row-major (no V^T pass / buffer); the q preparation
    (norm weights + RoPE tables) is optional"""
    lib = _lib.load()
    _need(q, torch.bfloat16, "q")
    _need(v, torch.bfloat16, "v")
    _need(out, torch.bfloat16, "out")
    if rope_cos is not None:
        _need(rope_cos, torch.float32, "rope_cos")
    _recorded_attention(lambda: check(lib.drag_attention_v_bf16(_p(q), _p(k), _p(v), _p(out), B, S, H, ld_qk, qk_batch_stride, ld_o, o_batch_stride,
                                                                scale, _p(wq_txt), _p(wq_img), _p(rope_cos), _p(rope_sin), s_txt, eps, _stream()),
                                      "drag_attention_v_bf16"), B, S, H, wq_txt is not None, True)


def layernorm(x: torch.Tensor, y: torch.Tensor, M: int, D: int, *, scale=None, shift=None, gamma=None, beta=None,
              ldx: int | None = None, rows_per_batch: int = 0, x_batch_stride: int = 0, ldy: int | None = None,
              ld_mod: int = 0, eps: float = 1e-6) -> torch.Tensor:
    lib = _lib.load()
    _need(x, torch.bfloat16, "x")
    _need(y, torch.bfloat16, "y")
    check(lib.drag_layernorm_modulate_bf16(_p(x), _p(y), _p(scale), _p(shift), _p(gamma), _p(beta), M, D,
                                           ldx if ldx is not None else D, rows_per_batch, x_batch_stride,
                                           ldy if ldy is not None else D, ld_mod, eps, _stream()),
          "drag_layernorm_modulate_bf16")
    return y


def act(x: torch.Tensor, kind: int, out: torch.Tensor | None = None) -> torch.Tensor:
    lib = _lib.load()
    _need(x, torch.bfloat16, "x")
    if out is None:
        out = torch.empty_like(x)
    check(lib.drag_act_bf16(_p(x), _p(out), x.numel(), kind, _stream()), "drag_act_bf16")
    return out


def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    lib = _lib.load()
    _need(t, torch.float32, "t")
    out = torch.empty((t.numel(), dim), dtype=torch.bfloat16, device=t.device)
    check(lib.drag_timestep_embedding_bf16(_p(t), _p(out), t.numel(), dim, _stream()), "drag_timestep_embedding_bf16")
    return out


def flow_euler_step(x: torch.Tensor, v: torch.Tensor, dt: float) -> None:
    lib = _lib.load()
    _need(x, torch.bfloat16, "x")
    _need(v, torch.bfloat16, "v")
    check(lib.drag_flow_euler_step_bf16(_p(x), _p(v), dt, x.numel(), _stream()), "drag_flow_euler_step_bf16")


def add(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    lib = _lib.load()
    _need(a, torch.bfloat16, "a")
    if out is None:
        out = torch.empty_like(a)
    check(lib.drag_add_bf16(_p(a), _p(b), _p(out), a.numel(), _stream()), "drag_add_bf16")
    return out


def to_bf16(x: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    _need(x, torch.float32, "x")
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    check(lib.drag_cast_f32_to_bf16(_p(x), _p(out), x.numel(), _stream()), "drag_cast_f32_to_bf16")
    return out


def to_f32(x: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    _need(x, torch.bfloat16, "x")
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    check(lib.drag_cast_bf16_to_f32(_p(x), _p(out), x.numel(), _stream()), "drag_cast_bf16_to_f32")
    return out


def _need_rows(t: torch.Tensor, name: str):
    """the kernels take a base pointer + (rows, d): a sliced / transposed view would be scanned with the wrong stride"""
    if t.dim() != 2 or not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous [rows, d] matrix (got shape {tuple(t.shape)}, strides {t.stride()}); "
                         "call .contiguous() on views")


def cosine_topk(corpus: torch.Tensor, queries: torch.Tensor, k: int):
    """(D, I) = exact inner-product top-k; D f32 [Q, k] descending, I int64 [Q, k]."""
    lib = _lib.load()
    _need(corpus, torch.float32, "corpus")
    _need(queries, torch.float32, "queries")
    _need_rows(corpus, "corpus"); _need_rows(queries, "queries")
    N, d = corpus.shape
    Q = queries.shape[0]
    if queries.shape[1] != d:
        raise ValueError(f"queries have d = {queries.shape[1]}, corpus has d = {d}")
    ws_bytes = lib.drag_cosine_topk_workspace_bytes(N, Q)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=corpus.device)
    D = torch.empty((Q, k), dtype=torch.float32, device=corpus.device)
    I = torch.empty((Q, k), dtype=torch.int64, device=corpus.device)
    check(lib.drag_cosine_topk_f32(_p(corpus), _p(queries), N, d, Q, k, _p(D), _p(I), _p(ws), _stream()),
          "drag_cosine_topk_f32")
    return D, I


def cosine_scores(corpus: torch.Tensor, queries: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """scan pass alone (ONE pass over the corpus): f32 [Q <= 64, ceil64(N)] inner products in the pinned summation order (entries >= N unspecified)"""
    lib = _lib.load()
    _need(corpus, torch.float32, "corpus")
    _need(queries, torch.float32, "queries")
    _need_rows(corpus, "corpus"); _need_rows(queries, "queries")
    N, d = corpus.shape
    Q = queries.shape[0]
    npad = (N + 63) // 64 * 64
    if out is None:
        out = torch.empty((Q, npad), dtype=torch.float32, device=corpus.device)
    check(lib.drag_cosine_scores_f32(_p(corpus), _p(queries), N, d, Q, _p(out), _stream()), "drag_cosine_scores_f32")
    return out


def l2_normalize_(x: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    _need(x, torch.float32, "x")
    _need_rows(x, "x")
    check(lib.drag_l2_normalize_f32(_p(x), x.shape[0], x.shape[1], _stream()), "drag_l2_normalize_f32")
    return x


# ------------------------------------------------------------------ VAE-side ops
def conv3x3(x_pad: torch.Tensor, w: torch.Tensor, y: torch.Tensor, *, B: int, Ho: int, Wo: int, Hp: int, Wp: int,
            Cin: int, Cout: int, bias=None, resid=None, ldy: int | None = None, stride: int = 1, oy: int = 0, ox: int = 0,
            act: int = ACT_NONE) -> torch.Tensor:
    """3x3 conv over a zero-haloed NHWC input as an implicit GEMM (w is [Cout, 3, 3, Cin])."""
    lib = _lib.load()
    _need(x_pad, torch.bfloat16, "conv.x")
    _need(w, torch.bfloat16, "conv.w")
    a = _lib.ConvArgs()
    a.x, a.w, a.y = x_pad.data_ptr(), w.data_ptr(), y.data_ptr()
    a.bias = bias.data_ptr() if bias is not None else None
    a.resid = resid.data_ptr() if resid is not None else None
    a.B, a.Ho, a.Wo, a.Hp, a.Wp, a.Cin, a.Cout = B, Ho, Wo, Hp, Wp, Cin, Cout
    a.ldy, a.stride, a.oy, a.ox, a.act = ldy if ldy is not None else Cout, stride, oy, ox, act
    if _recorder is not None:
        s_ev, e_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_ev.record()
        check(lib.drag_conv3x3_bf16(ctypes.byref(a), _stream()), "drag_conv3x3_bf16")
        e_ev.record()
        _recorder.events.append((s_ev, e_ev, 2.0 * B * Ho * Wo * Cout * 9 * Cin))
        _recorder.shapes.append((B * Ho * Wo, Cout, 9 * Cin))
        _recorder.is_conv.append(True)
        return y
    check(lib.drag_conv3x3_bf16(ctypes.byref(a), _stream()), "drag_conv3x3_bf16")
    return y


_gn_ws: dict = {}


def groupnorm_silu(x, y, gamma, beta, B, H, W, C, *, out_pad: int, silu: bool, eps: float = 1e-6):
    lib = _lib.load()
    _need(x, torch.bfloat16, "gn.x")
    nbytes = lib.drag_groupnorm_workspace_bytes(B, H, W, C)
    key = (x.device, nbytes)
    ws = _gn_ws.get(key)
    if ws is None:
        ws = _gn_ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    check(lib.drag_groupnorm_silu_bf16(_p(x), _p(y), _p(gamma), _p(beta), B, H, W, C, 32, out_pad, int(silu), eps,
                                       _p(ws), _stream()), "drag_groupnorm_silu_bf16")
    return y


def pad_copy(x, y, B, H, W, C, upsample: int = 1):
    lib = _lib.load()
    check(lib.drag_pad_copy_bf16(_p(x), _p(y), B, H, W, C, upsample, _stream()), "drag_pad_copy_bf16")
    return y


def softmax_rows(x: torch.Tensor, y: torch.Tensor, rows: int, cols: int, scale: float, ldy: int | None = None):
    lib = _lib.load()
    _need(x, torch.float32, "softmax.x")
    check(lib.drag_softmax_rows_f32_bf16(_p(x), _p(y), rows, cols, ldy if ldy is not None else cols, scale, _stream()),
          "drag_softmax_rows_f32_bf16")
    return y


def unpack_latents(tokens, y, B, h, w, ld, C, scaling, shift):
    lib = _lib.load()
    check(lib.drag_unpack_latents_bf16(_p(tokens), _p(y), B, h, w, ld, C, scaling, shift, _stream()), "drag_unpack_latents_bf16")
    return y


def sample_pack_latents(moments, noise, tokens, B, H, W, ldm, ld, scaling, shift):
    lib = _lib.load()
    check(lib.drag_sample_pack_latents_bf16(_p(moments), _p(noise), _p(tokens), B, H, W, ldm, ld, scaling, shift, _stream()),
          "drag_sample_pack_latents_bf16")
    return tokens


def image_preprocess(img_u8, mask_u8, y, B, H, W, C):
    lib = _lib.load()
    _need(img_u8, torch.uint8, "img")
    check(lib.drag_image_preprocess_u8(_p(img_u8), _p(mask_u8), _p(y), B, H, W, C, _stream()), "drag_image_preprocess_u8")
    return y


def image_postprocess(x, out_u8, npix, ld):
    lib = _lib.load()
    check(lib.drag_image_postprocess_u8(_p(x), _p(out_u8), npix, ld, _stream()), "drag_image_postprocess_u8")
    return out_u8


def mask_pack(mask_u8, tokens, B, H, W, ld):
    lib = _lib.load()
    _need(mask_u8, torch.uint8, "mask")
    check(lib.drag_mask_pack_u8(_p(mask_u8), _p(tokens), B, H, W, ld, _stream()), "drag_mask_pack_u8")
    return tokens


def flow_euler_rows(x, v, rows, cols, ldx, ldv, dt):
    lib = _lib.load()
    check(lib.drag_flow_euler_rows_bf16(_p(x), _p(v), rows, cols, ldx, ldv, dt, _stream()), "drag_flow_euler_rows_bf16")


def scale_noise_rows(x, noise, rows, cols, ldx, ldn, sigma):
    lib = _lib.load()
    check(lib.drag_scale_noise_rows_bf16(_p(x), _p(noise), rows, cols, ldx, ldn, sigma, _stream()), "drag_scale_noise_rows_bf16")


# ------------------------------------------------------------------ ViT front ends / Redux combine
def patchify(img_u8: torch.Tensor, out: torch.Tensor, B: int, H: int, W: int, P: int, ldo: int, mean, std):
    lib = _lib.load()
    _need(img_u8, torch.uint8, "img")
    m = (ctypes.c_float * 3)(*mean)
    s = (ctypes.c_float * 3)(*std)
    check(lib.drag_patchify_u8(_p(img_u8), _p(out), B, H, W, P, ldo, m, s, _stream()), "drag_patchify_u8")
    return out


def scale_sum(x: torch.Tensor, scales: torch.Tensor, out: torch.Tensor, G: int, N: int, elems: int):
    lib = _lib.load()
    _need(x, torch.bfloat16, "x")
    _need(scales, torch.float32, "scales")
    check(lib.drag_scale_sum_bf16(_p(x), _p(scales), _p(out), G, N, elems, _stream()), "drag_scale_sum_bf16")
    return out


def resnet_stem_style(img_f32: torch.Tensor, conv_w, bn_scale, bn_shift, eps: float = 1e-5) -> torch.Tensor:
    """img fp32 [B,3,H,W] in [0,1] -> fp32 [B,128] (channel mean | std) of the ResNet50 stem output"""
    lib = _lib.load()
    _need(img_f32, torch.float32, "img")
    B, _, H, W = img_f32.shape
    out = torch.empty((B, 128), dtype=torch.float32, device=img_f32.device)
    ws = torch.empty(lib.drag_resnet_stem_style_workspace_bytes(B, H, W), dtype=torch.uint8, device=img_f32.device)
    check(lib.drag_resnet_stem_style_f32(_p(img_f32), _p(conv_w), _p(bn_scale), _p(bn_shift), _p(out), B, H, W, eps,
                                         _p(ws), _stream()), "drag_resnet_stem_style_f32")
    return out


def patchify_f32(img: torch.Tensor, out: torch.Tensor, B: int, H: int, W: int, P: int, ldo: int):
    lib = _lib.load()
    _need(img, torch.float32, "img")
    check(lib.drag_patchify_f32_nchw(_p(img), _p(out), B, H, W, P, ldo, _stream()), "drag_patchify_f32_nchw")
    return out


# ---- LaMa stage (float32, NHWC) ------------------------------------------------------------------------------
PAD_ZERO, PAD_REFLECT = 0, 1
CONV_ACT_NONE, CONV_ACT_RELU, CONV_ACT_SIGMOID, CONV_ACT_QUICK_GELU = 0, 1, 2, 3


def conv2d_f32(x: torch.Tensor, w: torch.Tensor, y: torch.Tensor, *, B: int, Hi: int, Wi: int, Ho: int, Wo: int, Cin: int, ldx: int,
               ldy: int, stride: int = 1, pad: int = 0, pad_mode: int = PAD_ZERO, transposed: bool = False, act: int = CONV_ACT_NONE,
               scale=None, shift=None, addend=None, ld_add: int = 0, resid=None, ld_res: int = 0) -> torch.Tensor:
    """y = act((conv(x, w) + addend) * scale + shift) + resid on NHWC f32 views: ``x`` / ``y`` / ``addend`` / ``resid`` may be
    channel slices of wider buffers (1-D or offset views; ld* = floats per pixel).  ``w`` is [Cout, KH, KW, Cin]."""
    lib = _lib.load()
    for t, n in ((x, "x"), (w, "w"), (y, "y")):
        _need(t, torch.float32, "conv2d_f32." + n)
    Cout, KH, KW, wc = w.shape
    if wc != Cin or not w.is_contiguous():
        raise ValueError("conv2d_f32.w must be contiguous [Cout, KH, KW, Cin]")
    a = _lib.Conv2dF32Args()
    a.x, a.w, a.y = x.data_ptr(), w.data_ptr(), y.data_ptr()
    a.scale = scale.data_ptr() if scale is not None else None
    a.shift = shift.data_ptr() if shift is not None else None
    a.addend = addend.data_ptr() if addend is not None else None
    a.resid = resid.data_ptr() if resid is not None else None
    a.B, a.Hi, a.Wi, a.Cin, a.ldx, a.Ho, a.Wo, a.Cout, a.ldy, a.ld_add, a.ld_res = B, Hi, Wi, Cin, ldx, Ho, Wo, Cout, ldy, ld_add, ld_res
    a.KH, a.KW, a.stride, a.pad, a.pad_mode, a.transposed, a.act = KH, KW, stride, pad, pad_mode, int(transposed), act
    if _recorder is not None and _recorder.f32:
        s_ev, e_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_ev.record()
        check(lib.drag_conv2d_f32(ctypes.byref(a), _stream()), "drag_conv2d_f32")
        e_ev.record()
        _recorder.events.append((s_ev, e_ev, 2.0 * B * Ho * Wo * Cout * KH * KW * Cin))
        _recorder.shapes.append((B * Ho * Wo, Cout, KH * KW * Cin))
        _recorder.is_conv.append("f32")
        return y
    check(lib.drag_conv2d_f32(ctypes.byref(a), _stream()), "drag_conv2d_f32")
    return y


def rfft2_f32(x, tmp, y, B, H, W, C, ldx, tw_w, tw_h):
    check(_lib.load().drag_rfft2_f32(_p(x), _p(tmp), _p(y), B, H, W, C, ldx, _p(tw_w), _p(tw_h), _stream()), "drag_rfft2_f32")


def irfft2_f32(f, tmp, y, add, B, H, W, C, ldy, ld_add, tw_w, tw_h):
    check(_lib.load().drag_irfft2_f32(_p(f), _p(tmp), _p(y), _p(add), B, H, W, C, ldy, ld_add, _p(tw_w), _p(tw_h), _stream()),
          "drag_irfft2_f32")


def lama_prepare(img_u8, mask_u8, x, H, W, Hp, Wp):
    _need(img_u8, torch.uint8, "lama_prepare.img"); _need(mask_u8, torch.uint8, "lama_prepare.mask")
    check(_lib.load().drag_lama_prepare_u8(_p(img_u8), _p(mask_u8), _p(x), H, W, Hp, Wp, _stream()), "drag_lama_prepare_u8")


def lama_blend(pred, ld, img_u8, mask_u8, out_u8, H, W, Hp, Wp):
    check(_lib.load().drag_lama_blend_u8(_p(pred), ld, _p(img_u8), _p(mask_u8), _p(out_u8), H, W, Hp, Wp, _stream()),
          "drag_lama_blend_u8")


# ---- float32 CLIP tower ------------------------------------------------------------------------------------
def linear_f32(x: torch.Tensor, w: torch.Tensor, y: torch.Tensor, M: int, *, ldx: int, ldy: int, bias=None, act: int = CONV_ACT_NONE,
               resid=None, ld_res: int = 0) -> torch.Tensor:
    """y[m, :N] = act(x[m, :K] @ w.T + bias) + resid on the f32 matrix core (a 1x1 convolution over M "pixels"); w [N, K]"""
    N, K = w.shape
    return conv2d_f32(x, w.view(N, 1, 1, K), y, B=1, Hi=1, Wi=M, Ho=1, Wo=M, Cin=K, ldx=ldx, ldy=ldy, act=act, shift=bias,
                      resid=resid, ld_res=ld_res)


def vit_prepare(img: torch.Tensor, out: torch.Tensor, mean, std) -> None:
    """uint8 [B,S,S,3] or normalised float [B,3,S,S] -> NHWC f32 [B,S,S,4] (4th channel 0)"""
    lib = _lib.load()
    if img.dtype == torch.uint8:
        m, s = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
        check(lib.drag_vit_prepare_u8(_p(img), _p(out), img.numel() // 3, m, s, _stream()), "drag_vit_prepare_u8")
    else:
        _need(img, torch.float32, "vit_prepare.img")
        B = img.shape[0]
        check(lib.drag_vit_prepare_f32(_p(img), _p(out), B, img.numel() // (3 * B), _stream()), "drag_vit_prepare_f32")


def layernorm_f32(x, y, gamma, beta, rows: int, D: int, eps: float, ldx: int | None = None, ldy: int | None = None) -> None:
    check(_lib.load().drag_layernorm_f32(_p(x), _p(y), _p(gamma), _p(beta), rows, D, ldx or D, ldy or D, eps, _stream()), "drag_layernorm_f32")


def clip_embed_ln(emb, cls, pos, gamma, beta, x, B: int, T: int, D: int, eps: float) -> None:
    check(_lib.load().drag_clip_embed_ln_f32(_p(emb), _p(cls), _p(pos), _p(gamma), _p(beta), _p(x), B, T, D, eps, _stream()),
          "drag_clip_embed_ln_f32")


def attention_small_f32(qkv, out, B: int, T: int, H: int, hd: int, ld: int, ldo: int, scale: float) -> None:
    check(_lib.load().drag_attention_small_f32(_p(qkv), _p(out), B, T, H, hd, ld, ldo, scale, _stream()), "drag_attention_small_f32")
