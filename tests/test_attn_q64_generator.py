"""attention_q64g_kernel's KV loop is generated text (scripts/gen/attn_q64_tile.py -> domain-rag_amd/csrc/attn_q64_tile.h).  CPU only:
the committed header is the generator's output; the generator's own emulator runs the instruction records of a whole item for the four waves
of a workgroup (LDS image, LDS-DMA staging, barriers) and must reproduce softmax(q k^T) v — ragged last tile, rescale-heavy hot keys, several
loop iterations, both forms — while its hazard checker watches every instruction; and that checker must catch the mutations it exists for."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts", "gen"))
import attn_q64_tile as G  # noqa: E402

HEADER = os.path.join(ROOT, "domain-rag_amd", "csrc", "attn_q64_tile.h")
SCALE = 1 / np.sqrt(128)


def test_the_committed_header_is_the_generators_output():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gen", "attn_q64_tile.py")], capture_output=True, text=True, check=True)
    assert r.stdout == open(HEADER).read(), "regenerate: python scripts/gen/attn_q64_tile.py > domain-rag_amd/csrc/attn_q64_tile.h"


def _bf(x):
    return G.bf16_to_f32(G.bf16_round(np.asarray(x, np.float32)))


def _case(S, seed, hot=()):
    rng = np.random.default_rng(seed)
    q = _bf(rng.standard_normal((256, 128)))
    k = _bf(rng.standard_normal((S, 128)))
    v = _bf(rng.standard_normal((S, 128)))
    for key, qrow, mag in hot:
        k[key] = _bf(q[qrow] * mag)
    return q, k, v


def _reference(q, k, v):
    s = (q.astype(np.float64) @ k.astype(np.float64).T) * SCALE
    p = np.exp(s - s.max(1, keepdims=True))
    return (p / p.sum(1, keepdims=True)) @ v.astype(np.float64)


@pytest.mark.parametrize("fold", [False, True])
@pytest.mark.parametrize("S,hot", [(250, ()), (512, ((300, 17, 3.0), (509, 40, 6.0), (5, 200, 1.5), (450, 255, 9.0))), (193, ((192, 3, 5.0),))])
def test_emulated_item_is_softmax_attention(fold, S, hot):
    """S = 250: four tiles, 58 valid keys in the last; S = 512: eight tiles (three loop iterations), keys tens of octaves above their row's
    running maximum early, late and in the last tile (the rescale blocks of every tile variant run); S = 193: ONE valid key in the last tile,
    and it is the row's maximum"""
    q, k, v = _case(S, S + int(fold), hot)
    O, l, emu = G.emulate_item(fold, S, 384, q, k, v, SCALE)
    ref = _reference(q, k, v)
    out = O / l[:, None]
    assert np.isfinite(out).all()
    err = np.abs(out - ref).max() / np.abs(ref).max()
    # P is rounded to bf16 (2^-9 relative per probability); the fold adds one more bf16 rounding of q c in THIS harness (the kernel folds
    # it into the q preparation's single rounding)
    assert err < (1.2e-2 if fold else 4e-3), err
    w0 = emu.waves[0].counts
    nkv = (S + 63) // 64
    assert w0["v_mfma_f32_32x32x16_bf16"] == 32 + 64 * nkv + 8 and w0["s_barrier"] == nkv
    if hot:
        assert w0.get("v_accvgpr_read_b32", 0) >= 2 * 64          # some rescale block ran


@pytest.mark.parametrize("fold", [False, True])
def test_emulated_walk_second_item_runs_on_the_tiles_the_first_items_stream_staged(fold):
    """a walking workgroup: the stream's last two tiles stage the NEXT item's K(0), K(1), V^T(0) through the next item's descriptors; the next
    item then starts from the LDS as it was left.  (The one bug of the round the single-item emulation could not see: staging offsets of
    those two tiles set behind the pieces that use them — only a walking launch on the GPU showed it.  The mutation below re-creates it.)"""
    S = 250
    q1, k1, v1 = _case(S, 11)
    q2, k2, v2 = _case(S, 12)
    O1, l1, emu1 = G.emulate_item(fold, S, 384, q1, k1, v1, SCALE, next_kv=(k2, v2))
    O2, l2, emu2 = G.emulate_item(fold, S, 384, q2, k2, v2, SCALE, lds_from=emu1)
    for O, l, (q, k, v) in ((O1, l1, (q1, k1, v1)), (O2, l2, (q2, k2, v2))):
        ref = _reference(q, k, v)
        err = np.abs(O / l[:, None] - ref).max() / np.abs(ref).max()
        assert err < (1.2e-2 if fold else 4e-3), err

    # the re-created bug: the pre-last tile's "K pieces now read the next item from offset 0" moved behind that tile's pieces
    def stale_offsets(st):
        idx = [n for n, i in enumerate(st.ins) if i.op == "s_mov_b32" and i.dst == ("s", G.S_KOFF) and i.src == (0,)]
        assert len(idx) == 1
        ins = st.ins.pop(idx[0])
        last_piece = max(n for n in range(idx[0], len(st.ins)) if st.ins[n].op == "lds_dma" and n < idx[0] + 200)
        st.ins.insert(last_piece + 1, ins)
    orig, b = _mutated(stale_offsets)
    G.product = b
    try:
        _, _, bad1 = G.emulate_item(fold, S, 384, q1, k1, v1, SCALE, next_kv=(k2, v2))
        Ob, lb, _ = G.emulate_item(fold, S, 384, q2, k2, v2, SCALE, lds_from=bad1)
    finally:
        G.product = orig
    ref2 = _reference(q2, k2, v2)
    assert np.abs(Ob / lb[:, None] - ref2).max() / np.abs(ref2).max() > 5e-2          # the second item read the wrong K(0)


def test_instruction_budget_per_tile():
    """the stream's point: 420 instructions per tile and wave without the fold (the hand-placed kernel: 425 + ~60 of hipcc's), 344 with it"""
    for fold, want in ((False, 420), (True, 344)):
        st = G.product(fold)
        bars = [n for n, i in enumerate(st.ins) if i.op == "s_barrier"]
        per_tile = [b - a for a, b in zip(bars, bars[1:])]
        assert min(per_tile) <= want + 4 and max(per_tile) <= want + 4 + 6, (fold, per_tile)       # (+ loop control / the staging-offset scalars)
        tile = st.ins[bars[0]:bars[1]]
        assert sum(1 for i in tile if i.op.startswith("v_mfma")) == 64
        assert sum(1 for i in tile if i.op == "ds_read_b128") == 32 and sum(1 for i in tile if i.op == "lds_dma") == 8
        assert sum(1 for i in tile if i.op == "v_exp_f32") == 64
        assert sum(1 for i in tile if i.op == "v_fma_f32") == (0 if fold else 64)


def test_no_fold_stream_keeps_the_hand_placed_kernels_mfma_order():
    """same bits as attention_q64_kernel rest on: per score accumulator the k-slices 0..7 in order starting from C = 0, per output accumulator the
    key groups 0..3 of a tile in order (the fourth trailing into the next tile's top), the softmax's float operations in the old macros' order"""
    st = G.product(False)
    bars = [n for n, i in enumerate(st.ins) if i.op == "s_barrier"]
    tile = st.ins[bars[0]:bars[1]]
    chains = {}
    for i in tile:
        if i.op.startswith("v_mfma") and i.dst[0] == "v":
            chains.setdefault(i.dst, []).append(i)
    assert len(chains) == 4                                             # [group][key half] of the NEXT tile's score set
    for dst, seq in chains.items():
        assert len(seq) == 8 and len(seq[0].src) == 2                   # first k-slice: C = 0 (inline)
        assert all(s.src[2] == dst for s in seq[1:])
        assert [s.src[1][1] for s in seq] == [seq[0].src[1][1] + 4 * ks for ks in range(8)]       # Q fragments ks = 0..7
    pv = [i for i in tile if i.op.startswith("v_mfma") and i.dst[0] == "a"]
    assert len(pv) == 32 and all(i.src[2] == i.dst for i in pv)
    assert [i.dst for i in pv[:8]] == [G.O(g, dt) for dt in range(4) for g in range(2)]           # trailing group: dt outer, group inner
    soft = [i.op for i in tile if i.op in ("v_fma_f32", "v_exp_f32", "v_add_f32", "v_cvt_pk_bf16_f32")]
    step = ["v_fma_f32", "v_fma_f32", "v_exp_f32", "v_fma_f32", "v_exp_f32", "v_fma_f32", "v_add_f32", "v_exp_f32", "v_add_f32", "v_exp_f32",
            "v_cvt_pk_bf16_f32", "v_add_f32", "v_add_f32", "v_cvt_pk_bf16_f32"]
    assert soft[:14 * 16] == step * 16


def _mutated(mut):
    orig = G.product

    def b(f):
        st = orig(f)
        mut(st)
        return st
    return orig, b


def _drop(op, skip=0, pred=None):
    def m(st):
        n = 0
        for j, i in enumerate(st.ins):
            if i.op == op and (pred is None or pred(i)):
                if n == skip:
                    del st.ins[j]
                    return
                n += 1
        raise AssertionError("mutation found nothing to drop")
    return m


def _swap_dma_earlier(st):
    """move the first LDS-DMA piece of a tile (its m0 write, its address add, the piece) in front of that tile's barrier: it overwrites a
    buffer other waves may still read"""
    bars = [n for n, i in enumerate(st.ins) if i.op == "s_barrier"]
    j = next(n for n in range(bars[1], len(st.ins)) if st.ins[n].op == "lds_dma")
    k = max(n for n in range(bars[1], j) if st.ins[n].op == "s_add_u32" and st.ins[n].dst == ("s", "m0"))
    assert st.ins[j - 1].op == "v_add_u32"
    piece = [st.ins[k], st.ins[j - 1], st.ins[j]]
    for n in (j, j - 1, k):
        del st.ins[n]
    st.ins[bars[1] - 1: bars[1] - 1] = piece


def _read_fresh_score(st):
    """a VALU instruction reading a score accumulator two instructions behind the MFMA that wrote it (an MFMA's result takes its passes to land)"""
    bars = [n for n, i in enumerate(st.ins) if i.op == "s_barrier"]
    j = next(n for n in range(bars[1], len(st.ins)) if st.ins[n].op.startswith("v_mfma") and st.ins[n].dst[0] == "v")
    d = G.r1(st.ins[j].dst, 3)
    st.ins.insert(j + 2, G.Ins("v_mov_b32", G.TMP(23), [d], text=f"v_mov_b32 {G.rs(G.TMP(23))}, {G.rs(d)}"))


@pytest.mark.parametrize("name,mut,match", [
    ("a step's counted lgkmcnt wait", _drop("s_waitcnt_lgkmcnt", skip=4), "outstanding"),
    ("the vmcnt wait at a tile's top", _drop("s_waitcnt_vmcnt", skip=1), "before its LDS-DMA"),
    ("a tile's barrier", _drop("s_barrier", skip=1), "before its LDS-DMA"),
    ("the wait state in front of v_permlane32_swap", _drop("s_nop", pred=lambda i: i.imm == 0), "permlane32_swap"),
    ("a score register read right behind its MFMA", _read_fresh_score, "MFMA result"),
    ("a staging piece issued in front of the barrier", _swap_dma_earlier, "in an epoch in which it is read"),
])
def test_the_hazard_checker_catches(name, mut, match):
    orig, b = _mutated(mut)
    q, k, v = _case(250, 3)
    G.product = b
    try:
        with pytest.raises(G.HazardError, match=match):
            G.emulate_item(False, 250, 384, q, k, v, SCALE)
    finally:
        G.product = orig
