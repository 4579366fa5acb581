"""Generator (+ CPU emulator and hazard checker) of attention_q64g_kernel's KV loop (csrc/attention.hip): ONE asm statement per item
(256 queries of one (batch, head); this wave's 64 of them against every KV tile) — prologue scores, every tile, the final P V.

    python scripts/gen/attn_q64_tile.py > domain-rag_amd/csrc/attn_q64_tile.h

Why generated: attention_q64_kernel's tile is 425 hand-placed instructions in ~40 asm statements with ~60 of hipcc's
in between (19 s_nop, register moves, the rescale decision, LDS-DMA address arithmetic), and hipcc's allocation holds 250 of 256 VGPRs, so the one
change known to REMOVE instructions — scale * log2(e) folded into the q preparation and -m into the score MFMAs' C operand: 64 v_fma per tile
less — did not fit.  Here every register is named by this script (the K / V^T fragment rings and the trailing V^T fragments live in AGPRs: only
ds_read and MFMA touch them), the statement's operands are bound to those physical registers, and the fold fits.

Register map of a wave (64 queries = two groups of 32; query = lane & 31, hh = lane >> 5 selects the 8-element k-slice / the key rows):
  a[0:127]    O accumulators  [group][dt][16]            a[64 g + 16 dt ...]      (outputs of the statement)
  a[128:191]  Q fragments     [group][ks][4]             a[128 + 32 g + 4 ks ...] (inputs)
  a[192:203]  K fragment ring (step s uses slot s % 3)   a[204:215] V^T fragment ring   a[216:231] the 4 trailing V^T fragments
  v[0:63]     score set X     [group][key half][16]      v[32 g + 16 t ...]       v[64:127] score set Y
  v[128:159]  fold only: C tuples, 16 x (-M) per group   (the score MFMAs of a chain's first k-slice take them as C)
  v[160:175]  packed P ring: slot = key group & 1, [group][4]     v[160 + 8 slot + 4 g ...]
  v[176:183]  AKL  K fragment LDS addresses per ks       v[184:187] AVL V^T fragment LDS addresses per key group
  v[188:191]  KOFF this wave's 4 K staging pieces (global byte offsets)     v[192:195] VOFF of its 4 V^T pieces     (176..199: inputs)
  v[200:203]  softmax temporaries   v[204:205] ta tb   v[206:207] ps   v[208:209] l (outputs)   v[210:211] m / M   v[212:213] nm (no fold)
  v[214:215]  row maxima of the tile at hand   v[216:239] temporaries (maxima trees, rescale, mask, staging addresses)
  v[240:255], a[240:255] are left to the compiler.
Scalars: %[rsk] %[rsv] descriptors of this item's K / V^T, %[rskn] %[rsvn] of the next item's (zero records when there is none),
%[c] scale * log2(e) (no fold), %[npairs] (tiles - 2) / 2, %[ktile] bytes between K tiles, %[nvalid] valid keys of the last tile,
%[ldsw] LDS address of this wave's first staging piece; s[84:95] temporaries.

One tile = TOP (wait + barrier | the first two K fragment reads | the previous tile's trailing P V (8 MFMAs) with the row maxima of this
tile's scores under them | the rescale decision) + 16 STEPS (step g: softmax of two scores per group, packed; from step 4 on two P V MFMAs
of the key group finished four steps earlier; two S MFMAs of the NEXT tile's scores; the fragment reads of step g + 2; the LDS-DMA pieces
of tiles it + 2 (K) / it + 1 (V^T) in the first steps).  Per query group and accumulator the MFMA order and every float operation of
attention_q64_kernel's stream are kept in the NO-FOLD form: same bits (tests/test_gpu_kernels.py).  The FOLD form differs by design.

The emulator below executes the generated instruction records for the four waves of a workgroup on numpy (64 lanes, LDS image, LDS-DMA from
numpy "global memory") and checks while doing so: ds_read results not touched before the s_waitcnt that retires them; VALU -> MFMA operand
distance; MFMA result latency before non-MFMA use; v_exp -> consumer distance; m0 -> LDS-DMA distance; LDS regions neither read before their
DMA landed + barrier nor overwritten in the epoch they are read (tests/test_attn_q64_generator.py)."""
from __future__ import annotations

import sys

KT = 64 * 256
VT = 128 * 128
THR = 8.0

# ---------------------------------------------------------------------------------------------- register map
O = lambda g, dt: ("a", 64 * g + 16 * dt, 16)
QA = lambda g, ks: ("a", 128 + 32 * g + 4 * ks, 4)
RING = [3]       # ring depth of the K / V^T fragment rings (3: reads two steps ahead; 4: three — Stream(ahead=3))
KFR = lambda s: ("a", 192 + 4 * (s % RING[0]), 4)
VFR = lambda s: ("a", 192 + 4 * RING[0] + 4 * (s % RING[0]), 4)
VTR = lambda dt: ("a", 192 + 8 * RING[0] + 4 * dt, 4)
SX = lambda g, t: ("v", 32 * g + 16 * t, 16)
SY = lambda g, t: ("v", 64 + 32 * g + 16 * t, 16)
CT = lambda g: ("v", 128 + 16 * g, 16)
PK = lambda slot, g: ("v", 160 + 8 * (slot & 1) + 4 * g, 4)
AKL = lambda ks: ("v", 176 + ks, 1)
AVL = lambda kg: ("v", 184 + kg, 1)
KOFF = lambda i: ("v", 188 + i, 1)
VOFF = lambda i: ("v", 192 + i, 1)
Y = lambda i: ("v", 200 + i, 1)            # ya0 ya1 yb0 yb1
TA, TB = ("v", 204, 1), ("v", 205, 1)
PS = lambda g: ("v", 206 + g, 1)
LA = lambda g: ("a", 224 + 16 * g, 16)      # lsum form: the row sums as accumulators of ONES x P MFMAs (rings of 2: fragments end at a223)
ONES = ("v", 200, 4)                        # lsum form: an A fragment of bf16 ones (the softmax temporaries' registers: the fold does not use them)
PS2 = lambda g: ("v", 204 + 2 * g, 2)       # pk_add form: (sum of the even, sum of the odd elements) per group; TA / TB are not used then
L = lambda g: ("v", 208 + g, 1)
M = lambda g: ("v", 210 + g, 1)
NM = lambda g: ("v", 212 + g, 1)
MT = lambda g: ("v", 214 + g, 1)
TMP = lambda i: ("v", 216 + i, 1)          # 24 temporaries
S_KOFF, S_VSOFF, S_CNT, S_THR, S_FLOOR, S_T0, S_ONES = "s84", "s85", "s86", "s87", "s88", "s89", "s90"
NEG_INF = 0xff800000


def r1(reg, i=0):
    return (reg[0], reg[1] + i, 1)


def rs(reg):
    f, lo, n = reg
    return f"{f}{lo}" if n == 1 else f"{f}[{lo}:{lo + n - 1}]"


class Ins:
    __slots__ = ("op", "dst", "src", "imm", "text", "tag")

    def __init__(self, op, dst=None, src=(), imm=None, text="", tag=""):
        self.op, self.dst, self.src, self.imm, self.text, self.tag = op, dst, tuple(src), imm, text, tag


class Stream:
    """instruction records + their assembly text"""

    def __init__(self, fold: bool, pieces_at=None, no_dma=False, no_barrier=False, no_reads=False, no_exp=False, no_softmax=False, no_maxima=False,
                 pk_add=False, ahead=2, dot2=False, lsum=False, early_max=False):
        self.fold = fold
        self.ins: list[Ins] = []
        self.nlabel = 0
        # slot -> list of ("k" | "v", piece index): where the 8 LDS-DMA pieces of a tile go.  slot = a step (behind its first S MFMA) or
        # ("top", j) (behind the j-th trailing P V MFMA of the tile's top block; fold form only: the address temporaries are the softmax's)
        self.pieces_at = pieces_at or {g: [("k", g)] if g < 4 else [("v", g - 4)] for g in range(8)}
        self.no_dma, self.no_barrier = no_dma, no_barrier      # timing ablations (wrong results)
        self.no_reads, self.no_exp, self.no_softmax, self.no_maxima = no_reads, no_exp, no_softmax, no_maxima
        # pk_add (fold form): the row sums as v_pk_add_f32 of the exponentials' register pairs — one instruction per pair instead of two
        # (sum of the even + sum of the odd elements: another summation order)
        assert not pk_add or fold
        self.pk_add = pk_add
        # dot2 (fold form): the row sums from the PACKED probabilities, ps += p0 + p1 as one v_dot2c_f32_bf16 with (1, 1) — one instruction
        # per pair instead of two adds, and the normaliser is the sum of exactly the bf16 values the P V MFMAs multiply
        assert not dot2 or (fold and not pk_add)
        self.dot2 = dot2
        # lsum (fold form): the row sums on the matrix core — per 16-key group and query group ONE more MFMA, ones[32 x 16] x P^T, whose
        # accumulator holds the sum of the packed probabilities in every row (both lane halves: no cross-lane add at the end) — instead of
        # 64 v_add per tile: the stream is bound by a lone wave's instruction issue, not by the matrix pipe.  Needs 32 AGPRs: fragment rings of 2
        assert not lsum or (fold and not pk_add and not dot2 and ahead == 1)
        self.lsum = lsum
        # early_max: the maxima of the NEXT tile's first key half (complete after step 7) are taken in steps 9..15, whose MFMAs leave issue slots
        # free, instead of in the next tile's issue-bound top block; MT(g) carries the partial maximum across the barrier
        self.early_max = early_max
        self.ahead = ahead          # fragment reads run `ahead` steps in front of their MFMAs (rings of ahead + 1 fragments)
        RING[0] = ahead + 1
        self.queue = []             # the wave's outstanding LDS reads, in issue order (tags): the counted waits come from here

    def emit(self, op, dst=None, src=(), imm=None, text=None, tag=""):
        if text is None:
            ops = ([rs(dst)] if dst is not None else []) + [rs(s) if isinstance(s, tuple) else str(s) for s in src]
            text = f"{op} " + ", ".join(ops)
        self.ins.append(Ins(op, dst, src, imm, text, tag))

    # ---- primitives
    def mfma(self, d, a, b, c):
        ctext = "0" if c is None else rs(c)
        self.emit("v_mfma_f32_32x32x16_bf16", d, [a, b] + ([c] if c is not None else []), text=f"v_mfma_f32_32x32x16_bf16 {rs(d)}, {rs(a)}, {rs(b)}, {ctext}")

    def ds_read(self, d, addr, off, tag=None):
        if self.no_reads and getattr(self, "in_tile", False):
            return
        self.queue.append(tag)
        self.emit("ds_read_b128", d, [addr], imm=off, text=f"ds_read_b128 {rs(d)}, {rs(addr)} offset:{off}")

    def wait_lgkm(self, n):
        del self.queue[:max(len(self.queue) - n, 0)]
        self.emit("s_waitcnt_lgkmcnt", imm=n, text=f"s_waitcnt lgkmcnt({n})")

    def wait_for(self, *tags):
        """the counted wait that retires the reads tagged `tags` (and everything older), leaving the younger ones in flight; nothing when
        none of them is outstanding"""
        idx = [n for n, t in enumerate(self.queue) if t in tags]
        if idx:
            self.wait_lgkm(len(self.queue) - max(idx) - 1)

    def wait_vm(self, n):
        self.emit("s_waitcnt_vmcnt", imm=n, text=f"s_waitcnt vmcnt({n})")

    def valu(self, op, d, *src, text=None):
        self.emit(op, d, src, text=text)

    def salu(self, op, d, *src):
        """scalar op; `d` is the first operand of the text (the destination, except for s_cmp_*)"""
        self.emit(op, ("s", d), src, text=f"{op} {d}, " + ", ".join(str(x) for x in src) if src else f"{op} {d}")

    def label(self, name):
        self.emit("label", imm=name, text=f"{name}:")

    def new_label(self, stem):
        self.nlabel += 1
        return f"L_aq64_{stem}_{self.nlabel}_%="

    def nop(self, n):
        self.emit("s_nop", imm=n, text=f"s_nop {n}")

    def dma_m0(self, kind, i, buf):
        base = buf * KT + i * 1024 if kind == "k" else 2 * KT + buf * VT + i * 1024
        self.salu("s_add_u32", "m0", "%[ldsw]", base)
        return base

    def dma_piece(self, kind, i, buf, nxt, base, filler=True):
        """one LDS-DMA piece (its m0 is written: dma_m0): K piece i of the tile two ahead into K buffer `buf`, or V^T piece i of the next tile
        into V^T buffer `buf`.  m0 is not interlocked: one instruction between its write and the piece (K: the address add; V^T: `filler`)"""
        if kind == "k":
            tmp = (("v", 204 + (i & 3), 1) if self.lsum else Y(i & 3)) if self.fold else TMP(20 + (i & 3))
            self.valu("v_add_u32", tmp, S_KOFF, KOFF(i), text=f"v_add_u32 {rs(tmp)}, {S_KOFF}, {rs(KOFF(i))}")
            rsrc = "%[rskn]" if nxt else "%[rsk]"
            self.emit("lds_dma", None, [tmp], imm=("k", base, rsrc, 0), text=f"buffer_load_dwordx4 {rs(tmp)}, {rsrc}, 0 offen lds")
        else:
            rsrc = "%[rsvn]" if nxt else "%[rsv]"
            if filler:
                self.nop(0)
            self.emit("lds_dma", None, [VOFF(i)], imm=("v", base, rsrc, S_VSOFF), text=f"buffer_load_dwordx4 {rs(VOFF(i))}, {rsrc}, {S_VSOFF} offen lds")

    def pieces(self, slot, par, nxt_k, nxt_v, around=None):
        """the LDS-DMA pieces of `slot`; `around` (a callable emitting one instruction, e.g. an MFMA) is placed between the first piece's
        m0 write and the piece itself (so a V^T piece needs no s_nop); returns whether `around` was emitted"""
        todo = [] if self.no_dma else self.pieces_at.get(slot, [])
        if not todo:
            if around:
                around()
            return
        for n, (kind, i) in enumerate(todo):
            buf = par if kind == "k" else par ^ 1
            base = self.dma_m0(kind, i, buf)
            if n == 0 and around:
                around()
            self.dma_piece(kind, i, buf, nxt_k if kind == "k" else nxt_v, base, filler=not (n == 0 and around))

    # ---- softmax of step g (elements E0, E0 + 1 of key half T of both groups), in the old stream's operation order
    def softmax_ops(self, sc, g):
        T, E0 = g >> 3, (2 * g) & 15
        sa0, sa1 = r1(sc(0, T), E0), r1(sc(0, T), E0 + 1)
        sb0, sb1 = r1(sc(1, T), E0), r1(sc(1, T), E0 + 1)
        wa, wb = r1(PK(g >> 2, 0), g & 3), r1(PK(g >> 2, 1), g & 3)
        ya0, ya1, yb0, yb1 = Y(0), Y(1), Y(2), Y(3)
        ops = {}
        if self.fold:       # the scores ARE s c - M: exponentials in place
            ops["V1"] = ops["V2"] = ops["V4"] = ops["V6"] = None
            ops["V3"] = ("v_exp_f32", sa0, sa0)
            ops["V5"] = ("v_exp_f32", sa1, sa1)
            ops["V8"] = ("v_exp_f32", sb0, sb0)
            ops["V10"] = ("v_exp_f32", sb1, sb1)
            ya0, ya1, yb0, yb1 = sa0, sa1, sb0, sb1
        else:
            ops["V1"] = ("v_fma_f32", ya0, sa0, "%[c]", NM(0))
            ops["V2"] = ("v_fma_f32", ya1, sa1, "%[c]", NM(0))
            ops["V3"] = ("v_exp_f32", ya0, ya0)
            ops["V4"] = ("v_fma_f32", yb0, sb0, "%[c]", NM(1))
            ops["V5"] = ("v_exp_f32", ya1, ya1)
            ops["V6"] = ("v_fma_f32", yb1, sb1, "%[c]", NM(1))
            ops["V8"] = ("v_exp_f32", yb0, yb0)
            ops["V10"] = ("v_exp_f32", yb1, yb1)
        ops["V7"] = ("v_add_f32", TA, ya0, ya1)
        ops["V9"] = ("v_add_f32", PS(0), PS(0), TA)
        ops["V11"] = ("v_cvt_pk_bf16_f32", wa, ya0, ya1)
        ops["V12"] = ("v_add_f32", TB, yb0, yb1)
        ops["V13"] = ("v_add_f32", PS(1), PS(1), TB)
        ops["V14"] = ("v_cvt_pk_bf16_f32", wb, yb0, yb1)
        if self.lsum:
            ops["V7"] = ops["V9"] = ops["V12"] = ops["V13"] = None
        if self.dot2:
            ops["V7"] = ops["V9"] = ops["V12"] = ops["V13"] = None
            ops["D_a"] = ("v_dot2c_f32_bf16", PS(0), S_ONES, wa)
            ops["D_b"] = ("v_dot2c_f32_bf16", PS(1), S_ONES, wb)
        if self.pk_add:
            ops["V7"] = ("v_pk_add_f32", PS2(0), PS2(0), (sa0[0], sa0[1], 2))
            ops["V12"] = ("v_pk_add_f32", PS2(1), PS2(1), (sb0[0], sb0[1], 2))
            ops["V9"] = ops["V13"] = None
        return ops

    def V(self, ops, *names):
        if self.no_softmax:
            return
        for n in names:
            o = ops[n]
            if o is not None:
                if self.no_exp and o[0] == "v_exp_f32":
                    self.valu("v_mov_b32", o[1], o[2])
                else:
                    self.valu(o[0], o[1], *o[2:])

    def half_tree(self, regs16, tmp, out, extra=None):
        """the maximum of 16 registers (+ `extra`) into `out` as v_max3 trees: returns the instructions as thunks (8, or 9 with `extra`)"""
        e = regs16
        t = tmp
        ops = [lambda i=i: self.valu("v_max3_f32", t[i], e[3 * i], e[3 * i + 1], e[3 * i + 2]) for i in range(5)]
        ops.append(lambda: self.valu("v_max3_f32", t[5], t[0], t[1], t[2]))
        ops.append(lambda: self.valu("v_max3_f32", t[6], t[3], t[4], e[15]))
        if extra is None:
            ops.append(lambda: self.valu("v_max_f32", out, t[5], t[6]))
        else:
            ops.append(lambda: self.valu("v_max3_f32", out, t[5], t[6], extra))
        return ops

    # ---- one tile
    def tile(self, par, sc, sn, variant="steady"):
        """par: parity of the tile (its V^T buffer; the next tile's K is in K buffer par ^ 1); sc / sn: score sets of this / the next tile;
        variant: steady | prelast (K pieces = the NEXT item's K(0)) | last (mask; K pieces = next K(1), V^T pieces = next V^T(0))"""
        KN = (par ^ 1) * KT
        VB = 2 * KT + par * VT
        self.in_tile = True
        nxt_k = variant in ("prelast", "last")
        nxt_v = variant == "last"
        # staging offsets of this tile's pieces (K of tile it + 2, V^T of tile it + 1): the item's last two tiles stage the NEXT item's
        # K(0) | K(1), V^T(0).  (Set before the top block: a first form set them behind the rescale decision, and the variants that
        # stage under the trailing P V used the previous tile's offsets in these two tiles — the next item's first tiles were wrong,
        # which only a walking launch on the GPU showed: the emulator runs one item.)
        if variant == "prelast":
            self.salu("s_mov_b32", S_KOFF, 0)
        elif variant == "last":
            self.salu("s_mov_b32", S_KOFF, "%[ktile]")
            self.salu("s_mov_b32", S_VSOFF, 0)
        self.wait_vm(0)
        if not self.no_barrier:
            self.emit("s_barrier", text="s_barrier")
        for g0 in range(self.ahead):
            self.ds_read(KFR(g0), AKL(g0), KN, tag=("k", g0))
        if variant == "last":
            self.mask_block(sc)
        self.top_block(sc, par, nxt_k, nxt_v, full=not self.early_max or variant == "last")
        self.wait_lgkm(0)
        self.rescale_decision(sc)
        for g in range(2):
            if self.lsum:
                pass
            elif self.pk_add:
                for h in range(2):
                    self.valu("v_mov_b32", r1(PS2(g), h), 0, text=f"v_mov_b32 {rs(r1(PS2(g), h))}, 0")
            else:
                self.valu("v_mov_b32", PS(g), 0, text=f"v_mov_b32 {rs(PS(g))}, 0")
        self.early = []
        if self.early_max and not self.no_maxima:       # the next tile's first-key-half maxima, for steps 9..15
            for grp in range(2):
                self.early += self.half_tree([r1(sn(grp, 0), i) for i in range(16)], [TMP(8 * grp + i) for i in range(7)], MT(grp))
        for g in range(16):
            self.step(g, par, sc, sn, KN, VB, nxt_k, nxt_v)
        assert not self.early
        self.wait_lgkm(0)
        for g in range(2):
            if self.lsum:
                pass
            elif self.pk_add:
                self.valu("v_add_f32", L(g), L(g), r1(PS2(g), 0))
                self.valu("v_add_f32", L(g), L(g), r1(PS2(g), 1))
            else:
                self.valu("v_add_f32", L(g), L(g), PS(g))
        if variant == "steady":
            self.salu("s_add_u32", S_KOFF, S_KOFF, "%[ktile]")
            self.salu("s_add_u32", S_VSOFF, S_VSOFF, 128)

    def step(self, g, par, sc, sn, KN, VB, nxt_k, nxt_v):
        T = g >> 3
        A = self.ahead
        ops = self.softmax_ops(sc, g)
        kf = KFR(g)
        c0 = None if g in (0, 8) else sn(0, T)
        c1 = None if g in (0, 8) else sn(1, T)
        if self.fold and g in (0, 8):
            c0, c1 = CT(0), CT(1)

        # reads issued in step g (in order): K(g + A) if it exists; V^T(g + A) if 4 <= g + A <= 15; the 4 trailing V^T fragments in steps 14 / 15
        rd = []
        if g + A <= 15:
            rd.append((KFR(g + A), AKL((g + A) & 7), KN + ((g + A) >> 3) * (32 * 256), ("k", g + A)))
        if 4 <= g + A <= 15:
            rd.append((VFR(g + A), AVL(((g + A) >> 2) - 1), VB + ((g + A) & 3) * (32 * 128), ("v", g + A)))
        if g >= 14:
            f1 = VB + (2 * (g - 14)) * (32 * 128)
            rd.append((VTR(2 * (g - 14)), AVL(3), f1, ("t", 2 * (g - 14))))
            rd.append((VTR(2 * (g - 14) + 1), AVL(3), f1 + 32 * 128, ("t", 2 * (g - 14) + 1)))

        def read(n):          # (n = 1: the second read and whatever else the step has — with reads one step ahead, step 14 has four)
            for d, a, off, tag in (rd[n:n + 1] if n == 0 else rd[1:]):
                self.ds_read(d, a, off, tag=tag)

        # the step's wait: K(g) (and V^T(g)) landed; younger reads stay in flight
        self.wait_for(("k", g), ("v", g))
        if g >= 4:
            kgc = (g >> 2) - 1
            self.V(ops, "V1", "V2")
            self.mfma(O(0, g & 3), VFR(g), PK(kgc, 0), O(0, g & 3))
            read(0)
            self.V(ops, "V3")
            self.mfma(O(1, g & 3), VFR(g), PK(kgc, 1), O(1, g & 3))
            read(1)
            self.V(ops, "V4", "V5", "V6")
            self.pieces(g, par, nxt_k, nxt_v, around=lambda: self.mfma(sn(0, T), kf, QA(0, g & 7), c0))
            if self.dot2:
                self.V(ops, "V8", "V11", "V10")
                self.mfma(sn(1, T), kf, QA(1, g & 7), c1)
                self.V(ops, "D_a", "V14", "D_b")
            else:
                self.V(ops, "V7", "V8", "V9", "V10")
                self.mfma(sn(1, T), kf, QA(1, g & 7), c1)
                self.V(ops, "V11", "V12", "V13", "V14")
            if g >= 9 and self.early:            # (steps 9..15: 16 instructions, 3 3 2 2 2 2 2)
                for _ in range(3 if g <= 10 else 2):
                    if self.early:
                        self.early.pop(0)()
            if self.lsum and (g & 3) < 2:        # the row sums of key group kgc, query group g & 1 (its packed words are a step or more old)
                self.mfma(LA(g & 1), ONES, PK(kgc, g & 1), LA(g & 1))
        else:
            self.V(ops, "V1", "V2")
            self.mfma(sn(0, T), kf, QA(0, g), c0)
            read(0); read(1)
            if self.fold:       # (no v_fma between the exponentials: keep one instruction between a v_exp and the first reader of its result)
                self.V(ops, "V3", "V5", "V8")
                self.pieces(g, par, nxt_k, nxt_v, around=lambda: self.mfma(sn(1, T), kf, QA(1, g), c1))
                if self.dot2:
                    self.V(ops, "V11", "V10", "D_a", "V14", "D_b")
                else:
                    self.V(ops, "V7", "V10", "V9", "V11", "V12", "V13", "V14")
            else:
                self.V(ops, "V3", "V4", "V5", "V6", "V7")
                self.pieces(g, par, nxt_k, nxt_v, around=lambda: self.mfma(sn(1, T), kf, QA(1, g), c1))
                self.V(ops, "V8", "V9", "V10", "V11", "V12", "V13", "V14")

    def top_block(self, sc, par=0, nxt_k=False, nxt_v=False, full=True):
        """the previous tile's last key group P V (8 MFMAs) with the row maxima of `sc` (two v_max3 trees) under them; the lane halves are
        joined with v_permlane32_swap (no LDS round trip): MT(0) / MT(1) = the groups' maxima in both lane halves"""
        el = lambda g, i: r1(sc(g, i >> 4), i & 15)
        ta = [TMP(i) for i in range(10)]
        tb = [TMP(10 + i) for i in range(10)]
        ra, rb = MT(0), MT(1)

        def L1(t, g, i):
            if not self.no_maxima:
                self.valu("v_max3_f32", t[i], el(g, 3 * i), el(g, 3 * i + 1), el(g, 3 * i + 2))

        def L2(t, g, i, into):
            if self.no_maxima:
                return
            if i < 3:
                self.valu("v_max3_f32", into, t[3 * i], t[3 * i + 1], t[3 * i + 2])
            else:
                self.valu("v_max3_f32", into, t[9], el(g, 30), el(g, 31))

        npv = [0]

        def PV(dt, g):
            self.pieces(("top", npv[0]), par, nxt_k, nxt_v, around=lambda: self.mfma(O(g, dt), VTR(dt), PK(3, g), O(g, dt)))
            npv[0] += 1

        if not full:
            # early_max: MT(g) holds the first key half's maximum (taken in the previous tile's steps 9..15): the second half's tree, joined with it
            todo = []
            if not self.no_maxima:
                for grp in range(2):
                    todo += self.half_tree([el(grp, 16 + i) for i in range(16)], [TMP(8 * grp + i) for i in range(7)], MT(grp), extra=MT(grp))
            for j, (dt, g) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1), (2, 0))):
                PV(dt, g)
                for _ in range(4 if j < 3 else 3):
                    if todo:
                        todo.pop(0)()
            assert not todo
            PV(2, 1)
            self.nop(0)
            self.emit("v_permlane32_swap_b32", None, [ra, rb], text=f"v_permlane32_swap_b32 {rs(ra)}, {rs(rb)}")
            self.valu("v_max_f32", ra, ra, rb)
            self.valu("v_mov_b32", rb, ra)
            PV(3, 0)
            self.nop(0)
            self.emit("v_permlane32_swap_b32", None, [ra, rb], text=f"v_permlane32_swap_b32 {rs(ra)}, {rs(rb)}")
            PV(3, 1)
            if self.lsum:
                for g in range(2):
                    self.mfma(LA(g), ONES, PK(3, g), LA(g))
            return
        # second-level results: group A's into TMP 20..23 (the staging-address temporaries, idle here), group B's into group A's dead first level
        u2a = [TMP(20), TMP(21), TMP(22), TMP(23)]
        u2b = [TMP(0), TMP(1), TMP(2), TMP(3)]
        PV(0, 0)
        for i in range(6):
            L1(ta, 0, i)
        PV(0, 1)
        for i in range(6, 10):
            L1(ta, 0, i)
        L1(tb, 1, 0); L1(tb, 1, 1)
        PV(1, 0)
        for i in range(2, 8):
            L1(tb, 1, i)
        PV(1, 1)
        L1(tb, 1, 8); L1(tb, 1, 9)
        for i in range(4):
            L2(ta, 0, i, u2a[i])
        PV(2, 0)
        for i in range(4):
            L2(tb, 1, i, u2b[i])
        self.valu("v_max3_f32", ra, u2a[0], u2a[1], u2a[2]); self.valu("v_max3_f32", rb, u2b[0], u2b[1], u2b[2])
        self.valu("v_max_f32", ra, ra, u2a[3]); self.valu("v_max_f32", rb, rb, u2b[3])
        PV(2, 1)
        self.nop(0)
        self.emit("v_permlane32_swap_b32", None, [ra, rb], text=f"v_permlane32_swap_b32 {rs(ra)}, {rs(rb)}")
        self.valu("v_max_f32", ra, ra, rb)
        self.valu("v_mov_b32", rb, ra)
        PV(3, 0)
        self.nop(0)
        self.emit("v_permlane32_swap_b32", None, [ra, rb], text=f"v_permlane32_swap_b32 {rs(ra)}, {rs(rb)}")
        PV(3, 1)
        if self.lsum:
            for g in range(2):
                self.mfma(LA(g), ONES, PK(3, g), LA(g))

    def mask_block(self, sc):
        """the ragged last tile: keys >= %[nvalid] do not exist (staged as zero rows): their scores become -inf before the maxima"""
        skip = self.new_label("nomask")
        self.salu("s_cmp_ge_u32", "%[nvalid]", 64)
        self.emit("s_cbranch_scc1", imm=skip, text=f"s_cbranch_scc1 {skip}")
        # TMP(16) = 4 * hh = (lane >> 5) * 4 ; key of register r of half t = 32 t + (r & 3) + 8 (r >> 2) + 4 hh
        self.emit("v_mbcnt_lo_u32_b32", TMP(16), [], text=f"v_mbcnt_lo_u32_b32 {rs(TMP(16))}, -1, 0")
        self.emit("v_mbcnt_hi_u32_b32", TMP(16), [TMP(16)], text=f"v_mbcnt_hi_u32_b32 {rs(TMP(16))}, -1, {rs(TMP(16))}")
        self.valu("v_lshrrev_b32", TMP(16), 5, TMP(16))
        self.valu("v_lshlrev_b32", TMP(16), 2, TMP(16))
        self.valu("v_mov_b32", TMP(17), NEG_INF, text=f"v_mov_b32 {rs(TMP(17))}, 0x{NEG_INF:08x}")
        for t in range(2):
            for r in range(16):
                kconst = 32 * t + (r & 3) + 8 * (r >> 2)
                self.salu("s_sub_i32", S_T0, "%[nvalid]", kconst)          # masked iff 4 hh >= nvalid - kconst  (signed)
                self.emit("v_cmp_ge_i32", None, [TMP(16), S_T0], text=f"v_cmp_ge_i32 vcc, {rs(TMP(16))}, {S_T0}", tag="vcc")
                for g in range(2):
                    d = r1(sc(g, t), r)
                    self.emit("v_cndmask_b32", d, [d, TMP(17)], text=f"v_cndmask_b32 {rs(d)}, {rs(d)}, {rs(TMP(17))}, vcc")
        self.label(skip)

    def rescale_decision(self, sc):
        for g in range(2):
            resc, back = self.new_label(f"resc{g}"), self.new_label(f"back{g}")
            if self.fold:
                # the scores are y = s c - M already: rescale when some row's y maximum exceeds the threshold (the first tile: always)
                self.emit("v_cmp_nge_f32", None, [S_THR, MT(g)], text=f"v_cmp_nge_f32 vcc, {S_THR}, {rs(MT(g))}", tag="vcc")
            else:
                self.valu("v_sub_f32", TMP(18), MT(g), M(g))
                self.valu("v_mul_f32", TMP(18), "%[c]", TMP(18))
                # !((mt - m) c <= 8)  ==  8 nge (mt - m) c   (true for NaN, like the C++ it replaces)
                self.emit("v_cmp_nge_f32", None, [THR, TMP(18)], text=f"v_cmp_nge_f32 vcc, 0x41000000, {rs(TMP(18))}", tag="vcc")
            self.emit("s_cbranch_vccnz", imm=resc, text=f"s_cbranch_vccnz {resc}")
            self.label(back)
            self.pending_resc = getattr(self, "pending_resc", []) + [(g, resc, back, sc)]
            if not self.fold:
                self.valu("v_mul_f32", NM(g), "%[c]", M(g), text=f"v_mul_f32 {rs(NM(g))}, %[c], {rs(M(g))}")
                self.valu("v_xor_b32", NM(g), 0x80000000, NM(g), text=f"v_xor_b32 {rs(NM(g))}, 0x80000000, {rs(NM(g))}")

    def rescale_blocks(self):
        """out of line: the rare path (a row maximum grew by more than the deferral threshold)"""
        for g, resc, back, sc in getattr(self, "pending_resc", []):
            self.label(resc)
            self.nop(15); self.nop(7)           # the trailing P V MFMAs' results settle before v_accvgpr_read
            alpha = TMP(19)
            if self.fold:
                # delta = max(ymax, floor)  (floor: -inf in the first tile — M := the row maximum whatever its sign — then 0)
                self.valu("v_max_f32", TMP(18), S_FLOOR, MT(g), text=f"v_max_f32 {rs(TMP(18))}, {S_FLOOR}, {rs(MT(g))}")
                self.valu("v_sub_f32", alpha, 0, TMP(18), text=f"v_sub_f32 {rs(alpha)}, 0, {rs(TMP(18))}")
                self.valu("v_min_f32", alpha, 0, alpha, text=f"v_min_f32 {rs(alpha)}, 0, {rs(alpha)}")
                self.valu("v_exp_f32", alpha, alpha)
                self.valu("v_add_f32", M(g), M(g), TMP(18))
                if self.lsum:
                    for i in range(16):
                        self.emit("v_accvgpr_read_b32", TMP(i), [r1(LA(g), i)], text=f"v_accvgpr_read_b32 {rs(TMP(i))}, {rs(r1(LA(g), i))}")
                    for i in range(16):
                        self.valu("v_mul_f32", TMP(i), TMP(i), alpha)
                    for i in range(16):
                        self.emit("v_accvgpr_write_b32", r1(LA(g), i), [TMP(i)], text=f"v_accvgpr_write_b32 {rs(r1(LA(g), i))}, {rs(TMP(i))}")
                else:
                    self.valu("v_mul_f32", L(g), L(g), alpha)
                for i in range(16):
                    self.valu("v_sub_f32", r1(CT(g), i), r1(CT(g), i), TMP(18))
                for t in range(2):
                    for i in range(16):
                        self.valu("v_sub_f32", r1(sc(g, t), i), r1(sc(g, t), i), TMP(18))
                self.salu("s_mov_b32", S_THR, "0x41000000")
                self.salu("s_mov_b32", S_FLOOR, 0)
            else:
                self.valu("v_max_f32", TMP(18), M(g), MT(g))
                self.valu("v_sub_f32", alpha, M(g), TMP(18))
                self.valu("v_mul_f32", alpha, "%[c]", alpha, text=f"v_mul_f32 {rs(alpha)}, %[c], {rs(alpha)}")
                self.valu("v_exp_f32", alpha, alpha)
                self.valu("v_mov_b32", M(g), TMP(18))
                self.valu("v_mul_f32", L(g), L(g), alpha)
            for dt in range(4):
                for i in range(16):
                    self.emit("v_accvgpr_read_b32", TMP(i), [r1(O(g, dt), i)], text=f"v_accvgpr_read_b32 {rs(TMP(i))}, {rs(r1(O(g, dt), i))}")
                for i in range(16):
                    self.valu("v_mul_f32", TMP(i), TMP(i), alpha)
                for i in range(16):
                    self.emit("v_accvgpr_write_b32", r1(O(g, dt), i), [TMP(i)], text=f"v_accvgpr_write_b32 {rs(r1(O(g, dt), i))}, {rs(TMP(i))}")
            self.emit("s_branch", imm=back, text=f"s_branch {back}")
        self.pending_resc = []

    # ---- the whole item
    def prologue(self):
        """state, then the scores of tile 0 into set X (K(0) is in K buffer 0: the caller waited and passed the barrier)"""
        for g in range(2):
            for dt in range(4):
                for i in range(16):
                    self.emit("v_accvgpr_write_b32", r1(O(g, dt), i), [0], text=f"v_accvgpr_write_b32 {rs(r1(O(g, dt), i))}, 0")
        for g in range(2):
            for i in range(4):
                self.valu("v_mov_b32", r1(PK(3, g), i), 0, text=f"v_mov_b32 {rs(r1(PK(3, g), i))}, 0")
            if self.lsum:
                for i in range(16):
                    self.emit("v_accvgpr_write_b32", r1(LA(g), i), [0], text=f"v_accvgpr_write_b32 {rs(r1(LA(g), i))}, 0")
            else:
                self.valu("v_mov_b32", L(g), 0, text=f"v_mov_b32 {rs(L(g))}, 0")
            if self.fold:
                self.valu("v_mov_b32", M(g), 0, text=f"v_mov_b32 {rs(M(g))}, 0")
                for i in range(16):
                    self.valu("v_mov_b32", r1(CT(g), i), 0, text=f"v_mov_b32 {rs(r1(CT(g), i))}, 0")
            else:
                self.valu("v_mov_b32", M(g), NEG_INF, text=f"v_mov_b32 {rs(M(g))}, 0x{NEG_INF:08x}")
        if self.fold:
            self.salu("s_mov_b32", S_THR, f"0x{NEG_INF:08x}")
            self.salu("s_mov_b32", S_FLOOR, f"0x{NEG_INF:08x}")
        # K pieces of tile 0 stage K(2): offset 2 tiles; V^T pieces stage V^T(1): 64 keys = 128 bytes into a row
        self.salu("s_lshl_b32", S_KOFF, "%[ktile]", 1)
        self.salu("s_movk_i32", S_VSOFF, 128)
        self.salu("s_mov_b32", S_CNT, "%[npairs]")
        if self.lsum:
            for i in range(4):
                self.valu("v_mov_b32", r1(ONES, i), "0x3f803f80", text=f"v_mov_b32 {rs(r1(ONES, i))}, 0x3f803f80")
        if self.dot2:
            self.salu("s_mov_b32", S_ONES, "0x3f803f80")          # (1.0, 1.0) as a bf16 pair
        for half in range(2):                 # ks 0..3, then 4..7: 8 fragments in flight (the rings' and the trailing fragments' registers)
            frags = {}
            slots = [("a", 192 + 4 * i, 4) for i in range(8)]
            n = 0
            for t in range(2):
                for ks in range(4 * half, 4 * half + 4):
                    frags[(t, ks)] = slots[n]; n += 1
                    self.ds_read(frags[(t, ks)], AKL(ks), t * (32 * 256))
            self.wait_lgkm(0)
            for ks in range(4 * half, 4 * half + 4):      # four chains in turn: an accumulator is touched by every fourth MFMA, k-slices in order
                for t in range(2):
                    for g in range(2):
                        c = None if ks == 0 else SX(g, t)
                        self.mfma(SX(g, t), frags[(t, ks)], QA(g, ks), c)
            if half == 0:
                self.nop(7)                  # the last MFMAs have read their fragments before the next batch is requested into the same registers
        self.nop(7); self.nop(7)             # ... and before the trailing fragments' registers are cleared; the scores settle before the maxima read them
        for dt in range(4):
            for i in range(4):
                self.emit("v_accvgpr_write_b32", r1(VTR(dt), i), [0], text=f"v_accvgpr_write_b32 {rs(r1(VTR(dt), i))}, 0")
        if self.early_max and not self.no_maxima:      # tile 0's first-key-half maxima (every later tile's come from its predecessor's steps 9..15)
            for grp in range(2):
                for op in self.half_tree([r1(SX(grp, 0), i) for i in range(16)], [TMP(8 * grp + i) for i in range(7)], MT(grp)):
                    op()

    def item(self):
        self.prologue()
        loop = self.new_label("loop")
        self.label(loop)
        self.tile(0, SX, SY)
        self.tile(1, SY, SX)
        self.salu("s_sub_u32", S_CNT, S_CNT, 1)
        self.salu("s_cmp_lg_u32", S_CNT, 0)
        self.emit("s_cbranch_scc1", imm=loop, text=f"s_cbranch_scc1 {loop}")
        self.tile(0, SX, SY, "prelast")
        self.tile(1, SY, SX, "last")
        for dt in range(4):
            for g in range(2):
                self.mfma(O(g, dt), VTR(dt), PK(3, g), O(g, dt))
        if self.lsum:
            for g in range(2):
                self.mfma(LA(g), ONES, PK(3, g), LA(g))
        done = self.new_label("done")
        self.emit("s_branch", imm=done, text=f"s_branch {done}")
        self.rescale_blocks()
        self.label(done)
        self.nop(15); self.nop(7)                # O settles before compiler code reads it


def clobbers(lsum=False):
    v = [f"v{i}" for i in range(240) if not (176 <= i <= 199) and (lsum or i not in (208, 209))]
    a = [f"a{i}" for i in range(192, 224 if lsum else 240)]          # (rings of 3: up to a231; of 4: a239 — one list for both; lsum: a[224:255] are outputs)
    s = [f"s{i}" for i in range(84, 96)]
    return v + a + s + ["vcc", "scc", "m0", "memory"]


def build(fold: bool, **kw) -> Stream:
    s = Stream(fold, **kw)
    s.item()
    return s


def product(fold: bool) -> Stream:
    """the stream the product kernel carries (AQ64_ITEM_FOLD / AQ64_ITEM_NOFOLD)"""
    return build(fold, **(FOLD_PRODUCT if fold else {}))


def emit_macro(name, stream, out):
    lines = [i.text for i in stream.ins]
    out.write(f"#define {name} \\\n")
    out.write(" \\\n".join('  "' + ln + '\\n\\t"' for ln in lines))
    out.write("\n\n")


# the product's fold stream: the staging pieces under the trailing P V MFMAs of the tile's top block (+0.2 ... +0.8 % over one piece per step in
# steps 0..7, three boxes: profiles/r06_attn_gen_variants_and_ablations.log, r06_attn_gen_pk_add_and_read_ahead_nulls.log)
FOLD_PRODUCT = dict(pieces_at={("top", j): [("k", j)] if j < 4 else [("v", j - 4)] for j in range(8)})
# schedule variants of the fold form for the A/B records (DRAG_EXPERIMENTS builds: "attn_gen" = 10 + index)
VARIANTS = {
    0: dict(),                                                                                                                   # one piece per step, steps 0..7 (the first product form)
    1: dict(pieces_at={0: [("k", 0), ("k", 1)], 1: [("k", 2), ("k", 3)], 2: [("v", 0), ("v", 1)], 3: [("v", 2), ("v", 3)]}),     # two pieces per step
    2: dict(pieces_at={4 + g: [("k", g)] if g < 4 else [("v", g - 4)] for g in range(8)}),                                        # steps 4..11
    3: dict(pieces_at={("top", j): [("k", j)] if j < 4 else [("v", j - 4)] for j in range(8)}),                                   # under the trailing P V
    4: dict(pieces_at={**{("top", 2 * j): [("k", j)] for j in range(4)}, **{g: [("v", g)] for g in range(4)}}),                    # K at the top, V^T in steps 0..3
    5: dict(pieces_at={**{("top", j): [("k", j)] for j in range(4)}, **{4 + g: [("v", g)] for g in range(4)}}),                    # K at the top, V^T in steps 4..7
    6: dict(no_dma=True),                                                                                                        # ablation: no staging
    7: dict(no_barrier=True),                                                                                                    # ablation: no barrier
    8: dict(no_reads=True),                                                                                                      # ablation: no fragment reads
    9: dict(no_exp=True),                                                                                                        # ablation: v_mov for v_exp
    10: dict(no_softmax=True),                                                                                                   # ablation: no softmax VALU
    11: dict(no_maxima=True),                                                                                                    # ablation: no maxima trees
    12: dict(no_dma=True, no_reads=True, no_softmax=True, no_maxima=True, no_barrier=True),                                      # ablation: MFMAs (+ waits) only
    13: dict(pk_add=True),                                                                                                       # row sums by v_pk_add_f32
    14: dict(ahead=3),                                                                                                           # fragment reads three steps ahead
    15: dict(pk_add=True, ahead=3),
    17: dict(ahead=1, **FOLD_PRODUCT),                                                                                           # fragment reads ONE step ahead (rings of 2)
    19: dict(early_max=True, **FOLD_PRODUCT),                                                                                    # first-key-half maxima in steps 9..15 of the previous tile
    18: dict(lsum=True, ahead=1, **FOLD_PRODUCT),                                                                                # row sums by ones x P MFMAs
    16: dict(dot2=True, **FOLD_PRODUCT),                                                                                         # row sums by v_dot2c_f32_bf16 of the packed P
}


def main(out=sys.stdout, experiments=False):
    """the product header (attn_q64_tile.h: committed), or with --experiments the schedule variants' header (attn_q64_tile_exp.h: generated
    by build.py for DRAG_EXPERIMENTS builds, 3 MB of text, not committed)"""
    out.write("// generated by scripts/gen/attn_q64_tile.py — do not edit; the generator holds the register map, the schedule and the emulator\n")
    if experiments:
        for k, kw in VARIANTS.items():
            if k:
                emit_macro(f"AQ64_ITEM_FOLD_V{k}", build(True, **kw), out)
        return
    out.write("#define AQ64_CLOBBERS " + ", ".join(f'"{c}"' for c in clobbers()) + "\n")
    out.write("#define AQ64_CLOBBERS_LSUM " + ", ".join(f'"{c}"' for c in clobbers(True)) + "\n\n")
    emit_macro("AQ64_ITEM_NOFOLD", product(False), out)
    emit_macro("AQ64_ITEM_FOLD", product(True), out)


if __name__ == "__main__":
    main(experiments="--experiments" in sys.argv)


# ================================================================================================ emulator + hazard checker
# (test infrastructure for the generated text: tests/test_attn_q64_generator.py; nothing here runs on a GPU)
class HazardError(AssertionError):
    pass


def _np():
    import numpy as np
    return np


def bf16_round(x):
    """float32 array -> bf16 bits (round to nearest even), as uint32 in the low 16 bits"""
    np = _np()
    u = np.asarray(x, dtype=np.float32).view(np.uint32)
    r = (u + (0x7FFF + ((u >> 16) & 1))) >> 16
    nan = np.isnan(np.asarray(x, dtype=np.float32))
    return np.where(nan, 0x7FC0, r & 0xFFFF).astype(np.uint32)


def bf16_to_f32(h):
    np = _np()
    return (np.asarray(h, dtype=np.uint32) << 16).view(np.float32)


PERM16 = [0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15]     # stored position within a 16-key group -> key offset (qk_norm_rope_vt_kernel)
MFMA_LATENCY = 19        # wait states before a non-MFMA instruction may touch an MFMA result (ISA table, 16-pass figure: conservative)


class Wave:
    def __init__(self, w):
        np = _np()
        self.w = w
        self.v = np.zeros((256, 64), np.uint32)
        self.a = np.zeros((256, 64), np.uint32)
        self.s = {}
        self.vcc = np.zeros(64, bool)
        self.scc = False
        self.m0 = 0
        self.pc = 0
        self.epoch = 0
        self.ws = 0                       # wait states issued so far
        self.lgkm = []                    # in-order queue of outstanding ds_read destinations
        self.vm = []                      # in-order queue of outstanding LDS-DMA pieces (region ids)
        self.pending = {}                 # register -> True while a ds_read result is outstanding
        self.valu_w = {}                  # register -> ws of the last VALU write
        self.mfma_w = {}                  # register -> ws of the last MFMA write
        self.trans_w = {}                 # register -> ws of the last transcendental write
        self.m0_w = -10
        self.done = False
        self.counts = {}


class Emu:
    """four waves of one workgroup; `glob` maps a descriptor operand ("%[rsk]" ...) to (uint8 array, num_records)"""

    def __init__(self, stream: Stream, glob, scalars, lane_inputs, q_frags, strict=True):
        np = _np()
        self.np = np
        self.ins = stream.ins
        self.labels = {i.imm: n for n, i in enumerate(self.ins) if i.op == "label"}
        self.glob = glob
        self.lds = np.zeros(2 * KT + 2 * VT, np.uint8)
        self.region_write = {}            # 1 KiB region -> dict(epoch, landed_epoch or None, barrier_ok)
        self.region_read_epoch = {}       # region -> last epoch in which some wave read it
        self.strict = strict
        self.waves = []
        for w in range(4):
            wv = Wave(w)
            wv.s.update(scalars(w))
            li = lane_inputs(w)
            for reg, val in li.items():
                getattr(wv, reg[0])[reg[1]] = val
            for reg, val in q_frags(w).items():
                wv.a[reg[1]:reg[1] + reg[2]] = val
            self.waves.append(wv)

    # ---- operand access
    def rd(self, wv, x, n=None):
        np = self.np
        if isinstance(x, tuple):
            f, lo, cnt = x
            arr = wv.v if f == "v" else wv.a
            return arr[lo] if cnt == 1 else arr[lo:lo + cnt]
        if isinstance(x, str):
            if x.startswith("0x"):
                return np.full(64, int(x, 16), np.uint32)
            return np.full(64, wv.s[x] & 0xFFFFFFFF, np.uint32)
        if isinstance(x, float):
            return np.full(64, np.float32(x)).view(np.uint32)
        return np.full(64, int(x) & 0xFFFFFFFF, np.uint32)

    def f(self, wv, x):
        np = self.np
        if isinstance(x, (int,)) and not isinstance(x, bool) and not isinstance(x, tuple):
            # integer source of a float instruction: inline constant 0 only
            assert x == 0, x
            return np.zeros(64, np.float32)
        return self.rd(wv, x).view(np.float32)

    def regs_of(self, x):
        if isinstance(x, tuple) and x[0] in ("v", "a"):
            return [(x[0], x[1] + i) for i in range(x[2])]
        return []

    # ---- hazard bookkeeping
    def check_sources(self, wv, i: Ins):
        is_mfma = i.op.startswith("v_mfma")
        srcs = list(i.src)
        for si, sx in enumerate(srcs):
            for r in self.regs_of(sx):
                if r in wv.pending:
                    raise HazardError(f"wave {wv.w} pc {wv.pc}: {i.text} reads {r} while a ds_read into it is outstanding")
                if is_mfma:
                    if wv.ws - wv.valu_w.get(r, -100) < 2 + 1:
                        raise HazardError(f"wave {wv.w} pc {wv.pc}: {i.text}: operand {r} was written by a VALU instruction {wv.ws - wv.valu_w[r] - 1} wait states before")
                    is_c = si == 2
                    if r in wv.mfma_w and not (is_c and sx == i.dst):
                        if wv.ws - wv.mfma_w[r] < MFMA_LATENCY + 1:
                            raise HazardError(f"wave {wv.w} pc {wv.pc}: {i.text}: source {r} is a {wv.ws - wv.mfma_w[r] - 1}-wait-state-old MFMA result (not the accumulator chain)")
                else:
                    if r in wv.mfma_w and wv.ws - wv.mfma_w[r] < MFMA_LATENCY + 1:
                        raise HazardError(f"wave {wv.w} pc {wv.pc}: {i.text} reads {r}, an MFMA result {wv.ws - wv.mfma_w[r] - 1} wait states old")
                    if r in wv.trans_w and wv.ws - wv.trans_w[r] < 1 + 1 and not i.op == "v_exp_f32":
                        raise HazardError(f"wave {wv.w} pc {wv.pc}: {i.text} reads {r} right behind the v_exp that wrote it")
        if i.op == "v_permlane32_swap_b32":
            for r in self.regs_of(i.src[0]) + self.regs_of(i.src[1]):
                if wv.ws - wv.valu_w.get(r, -100) < 2 + 1:
                    raise HazardError(f"wave {wv.w} pc {wv.pc}: permlane32_swap of {r} {wv.ws - wv.valu_w[r] - 1} wait states behind its VALU write")

    def note_write(self, wv, i: Ins, dsts):
        for d in dsts:
            for r in self.regs_of(d):
                if r in wv.pending:
                    raise HazardError(f"wave {wv.w} pc {wv.pc}: {i.text} writes {r} while a ds_read into it is outstanding")
                if i.op.startswith("v_mfma"):
                    wv.mfma_w[r] = wv.ws
                    wv.valu_w.pop(r, None)
                else:
                    if r in wv.mfma_w and wv.ws - wv.mfma_w[r] < MFMA_LATENCY + 1 and i.op != "ds_read_b128":
                        raise HazardError(f"wave {wv.w} pc {wv.pc}: {i.text} overwrites {r}, an MFMA result {wv.ws - wv.mfma_w[r] - 1} wait states old")
                    wv.mfma_w.pop(r, None)
                    if i.op != "ds_read_b128":
                        wv.valu_w[r] = wv.ws
                    if i.op == "v_exp_f32":
                        wv.trans_w[r] = wv.ws
                    else:
                        wv.trans_w.pop(r, None)

    # ---- LDS protocol
    def lds_read_check(self, wv, lo, hi, text):
        for reg in range(lo // 1024, (hi - 1) // 1024 + 1):
            st = self.region_write.get(reg)
            if st is None:
                raise HazardError(f"wave {wv.w}: {text} reads LDS region {reg} that was never staged")
            if st["landed_epoch"] is None or not wv.epoch > st["landed_epoch"]:
                raise HazardError(f"wave {wv.w} epoch {wv.epoch}: {text} reads LDS region {reg} before its LDS-DMA (issued in epoch {st['epoch']}) landed and a barrier passed")
            self.region_read_epoch[reg] = max(self.region_read_epoch.get(reg, -1), wv.epoch)

    def step_wave(self, wv):
        """run one wave until it reaches a barrier (returns 'barrier') or the end of the stream ('done')"""
        np = self.np
        while True:
            if wv.pc >= len(self.ins):
                wv.done = True
                return "done"
            i = self.ins[wv.pc]
            op = i.op
            wv.counts[op] = wv.counts.get(op, 0) + 1
            if op == "label":
                wv.pc += 1
                continue
            self.check_sources(wv, i)
            nxt = wv.pc + 1
            if op == "s_nop":
                wv.ws += i.imm
            elif op == "s_waitcnt_lgkmcnt":
                while len(wv.lgkm) > i.imm:
                    for r in wv.lgkm.pop(0):
                        wv.pending.pop(r, None)
            elif op == "s_waitcnt_vmcnt":
                while len(wv.vm) > i.imm:
                    for reg in wv.vm.pop(0):
                        self.region_write[reg]["landed_epoch"] = wv.epoch
            elif op == "s_barrier":
                if self.strict and wv.lgkm:
                    raise HazardError(f"wave {wv.w} pc {wv.pc}: s_barrier with {len(wv.lgkm)} LDS reads outstanding")
                wv.pc = nxt
                wv.ws += 1
                wv.epoch += 1
                return "barrier"
            elif op == "ds_read_b128":
                addr = self.rd(wv, i.src[0]).astype(np.int64) + i.imm
                lo, hi = int(addr.min()), int(addr.max()) + 16
                self.lds_read_check(wv, lo, hi, i.text)
                data = np.stack([self.lds[a:a + 16].view(np.uint32) for a in addr], 1)       # [4][64]
                self.note_write(wv, i, [i.dst])
                arr = wv.v if i.dst[0] == "v" else wv.a
                arr[i.dst[1]:i.dst[1] + 4] = data
                regs = self.regs_of(i.dst)
                for r in regs:
                    wv.pending[r] = True
                wv.lgkm.append(regs)
            elif op == "lds_dma":
                kind, base, rsrc, soff = i.imm
                if wv.ws - wv.m0_w < 1 + 1:
                    raise HazardError(f"wave {wv.w} pc {wv.pc}: LDS-DMA directly behind the write of m0")
                mem, nrec = self.glob[rsrc]
                so = 0 if soff == 0 else wv.s[soff]
                off = self.rd(wv, i.src[0]).astype(np.int64) + so
                dst0 = wv.m0
                regions = sorted({(dst0 + 16 * l) // 1024 for l in range(64)})
                for reg in regions:
                    if self.region_read_epoch.get(reg, -1) >= wv.epoch:
                        raise HazardError(f"wave {wv.w} epoch {wv.epoch}: {i.text} overwrites LDS region {reg} in an epoch in which it is read")
                    self.region_write[reg] = dict(epoch=wv.epoch, landed_epoch=None)
                for l in range(64):
                    o = int(off[l])
                    chunk = np.zeros(16, np.uint8)
                    if 0 <= o and o + 16 <= nrec:
                        chunk = mem[o:o + 16]
                    self.lds[dst0 + 16 * l: dst0 + 16 * l + 16] = chunk
                wv.vm.append(regions)
            elif op.startswith("v_mfma"):
                A = self.rd(wv, i.src[0]); B = self.rd(wv, i.src[1])
                C = self.rd(wv, i.src[2]).view(np.float32) if len(i.src) > 2 else np.zeros((16, 64), np.float32)
                # A[row = lane & 31][k = 8 (lane >> 5) + e], e = 0..7 over 4 registers x 2 halves; B[k][col = lane & 31]
                def unpack(X):
                    e = np.empty((8, 64), np.float32)
                    for r in range(4):
                        e[2 * r] = bf16_to_f32(X[r] & 0xFFFF); e[2 * r + 1] = bf16_to_f32(X[r] >> 16)
                    m = np.empty((32, 16), np.float64)
                    for l in range(64):
                        m[l & 31, 8 * (l >> 5): 8 * (l >> 5) + 8] = e[:, l]
                    return m
                Am, Bm = unpack(A), unpack(B)                 # [32 rows][16 k], [32 cols][16 k]
                Dm = Am @ Bm.T                                # [row][col]
                D = np.empty((16, 64), np.float32)
                for l in range(64):
                    for r in range(16):
                        row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)
                        D[r, l] = np.float32(np.float64(C[r, l]) + Dm[row, l & 31])
                self.note_write(wv, i, [i.dst])
                arr = wv.v if i.dst[0] == "v" else wv.a
                arr[i.dst[1]:i.dst[1] + 16] = D.view(np.uint32)
            elif op in ("s_add_u32", "s_sub_u32", "s_sub_i32", "s_mov_b32", "s_movk_i32", "s_lshl_b32", "s_cmp_ge_u32", "s_cmp_lg_u32"):
                d = i.dst[1]
                val = [wv.s[x] if isinstance(x, str) and not x.startswith("0x") else (int(x, 16) if isinstance(x, str) else int(x)) for x in i.src]
                if op == "s_add_u32":
                    res = (val[0] + val[1]) & 0xFFFFFFFF
                elif op in ("s_sub_u32", "s_sub_i32"):
                    res = (val[0] - val[1]) & 0xFFFFFFFF
                elif op in ("s_mov_b32", "s_movk_i32"):
                    res = val[0] & 0xFFFFFFFF
                elif op == "s_lshl_b32":
                    res = (val[0] << val[1]) & 0xFFFFFFFF
                elif op == "s_cmp_ge_u32":
                    wv.scc = wv.s[d] >= val[0]; res = None
                elif op == "s_cmp_lg_u32":
                    wv.scc = wv.s[d] != val[0]; res = None
                if res is not None:
                    if d == "m0":
                        wv.m0 = res; wv.m0_w = wv.ws
                    else:
                        wv.s[d] = res
            elif op == "s_cbranch_scc1":
                if wv.scc:
                    nxt = self.labels[i.imm]
            elif op == "s_cbranch_vccnz":
                if wv.vcc.any():
                    nxt = self.labels[i.imm]
            elif op == "s_branch":
                nxt = self.labels[i.imm]
            else:
                self.valu_exec(wv, i)
            wv.ws += 1
            wv.pc = nxt

    def valu_exec(self, wv, i: Ins):
        np = self.np
        op = i.op
        F = lambda k: self.f(wv, i.src[k])
        U = lambda k: self.rd(wv, i.src[k])
        with np.errstate(all="ignore"):
            if op == "v_fma_f32":
                res = (F(0).astype(np.float64) * F(1).astype(np.float64) + F(2).astype(np.float64)).astype(np.float32).view(np.uint32)
            elif op == "v_exp_f32":
                res = np.exp2(F(0)).astype(np.float32).view(np.uint32)
            elif op == "v_add_f32":
                res = (F(0) + F(1)).astype(np.float32).view(np.uint32)
            elif op == "v_dot2c_f32_bf16":
                a, b = U(0), U(1)
                acc = self.rd(wv, i.dst).view(np.float32).astype(np.float64)
                res = (acc + bf16_to_f32(a & 0xFFFF).astype(np.float64) * bf16_to_f32(b & 0xFFFF) +
                       bf16_to_f32(a >> 16).astype(np.float64) * bf16_to_f32(b >> 16)).astype(np.float32).view(np.uint32)
            elif op == "v_pk_add_f32":
                d = i.dst
                x, y = self.rd(wv, i.src[0]).view(np.float32), self.rd(wv, i.src[1]).view(np.float32)
                self.note_write(wv, i, [d])
                wv.v[d[1]:d[1] + 2] = (x + y).astype(np.float32).view(np.uint32)
                return
            elif op == "v_sub_f32":
                res = (F(0) - F(1)).astype(np.float32).view(np.uint32)
            elif op == "v_mul_f32":
                res = (F(0) * F(1)).astype(np.float32).view(np.uint32)
            elif op == "v_max_f32":
                res = np.fmax(F(0), F(1)).view(np.uint32)
            elif op == "v_min_f32":
                res = np.fmin(F(0), F(1)).view(np.uint32)
            elif op == "v_max3_f32":
                res = np.fmax(np.fmax(F(0), F(1)), F(2)).view(np.uint32)
            elif op == "v_cvt_pk_bf16_f32":
                res = bf16_round(F(0)) | (bf16_round(F(1)) << 16)
            elif op in ("v_mov_b32", "v_accvgpr_read_b32", "v_accvgpr_write_b32"):
                res = U(0).copy()
            elif op == "v_xor_b32":
                res = U(0) ^ U(1)
            elif op == "v_add_u32":
                res = (U(0).astype(np.uint64) + U(1)).astype(np.uint32)
            elif op == "v_lshrrev_b32":
                res = U(1) >> U(0)
            elif op == "v_lshlrev_b32":
                res = (U(1) << U(0)).astype(np.uint32)
            elif op == "v_mbcnt_lo_u32_b32":
                res = np.minimum(np.arange(64), 32).astype(np.uint32)
            elif op == "v_mbcnt_hi_u32_b32":
                res = (np.maximum(np.arange(64) - 32, 0) + U(0)).astype(np.uint32)
            elif op == "v_cmp_nge_f32":
                a, b = F(0), F(1)
                wv.vcc = ~(a >= b)
                return
            elif op == "v_cmp_ge_i32":
                wv.vcc = U(0).view(np.int32) >= U(1).view(np.int32)
                return
            elif op == "v_cndmask_b32":
                res = np.where(wv.vcc, U(1), U(0))
            elif op == "v_permlane32_swap_b32":
                x, y = self.rd(wv, i.src[0]).copy(), self.rd(wv, i.src[1]).copy()
                # swap(vdst.hi half, src.lo half): vdst lanes 32..63 <-> src lanes 0..31
                nx, ny = x.copy(), y.copy()
                nx[32:] = y[:32]
                ny[:32] = x[32:]
                self.note_write(wv, i, [i.src[0], i.src[1]])
                getattr(wv, i.src[0][0])[i.src[0][1]] = nx
                getattr(wv, i.src[1][0])[i.src[1][1]] = ny
                return
            else:
                raise NotImplementedError(op)
        self.note_write(wv, i, [i.dst])
        (wv.v if i.dst[0] == "v" else wv.a)[i.dst[1]] = res

    def run(self, max_rounds=100000):
        for _ in range(max_rounds):
            states = [self.step_wave(wv) if not wv.done else "done" for wv in self.waves]
            if all(s == "done" for s in states):
                return
            if any(s == "done" for s in states) and any(s == "barrier" for s in states):
                raise HazardError(f"waves disagree on the barrier count: {states}")
        raise RuntimeError("emulator did not finish")


def emulate_item(fold, S, ld_qk, q, k, v, scale, seed_pieces=None, stream_kw=None, next_kv=None, lds_from=None):
    """run the generated stream for ONE item (256 queries q[256, 128], keys k[S, 128], values v[S, 128]; bf16-representable float32 arrays)
    on the emulator; returns (O [256, 128] float32 unnormalised, l [256] float32, emulator).  The first three tiles' staging (K(0), V^T(0),
    K(1)) is done here the way the kernel's stage_first does it."""
    np = _np()
    s_pad = (S + 63) // 64 * 64
    nkv = s_pad // 64
    assert nkv % 2 == 0 and nkv >= 4
    c = np.float32(scale * 1.4426950408889634)
    # global images
    kbytes = ((S - 1) * ld_qk + 128) * 2
    kmem = np.zeros(kbytes, np.uint8)
    kb = bf16_round(k).astype(np.uint16)
    for r in range(S):
        kmem[r * ld_qk * 2: r * ld_qk * 2 + 256] = kb[r].view(np.uint8)
    vt = np.zeros((128, s_pad), np.uint16)
    vb = bf16_round(v).astype(np.uint16)
    for key in range(S):
        grp, within = key & ~15, key & 15
        pos = grp + PERM16.index(within)
        vt[:, pos] = vb[key]
    vmem = vt.reshape(-1).view(np.uint8).copy()
    glob = {"%[rsk]": (kmem, kbytes), "%[rsv]": (vmem, vmem.size), "%[rskn]": (kmem, 0), "%[rsvn]": (vmem, 0)}
    if next_kv is not None:          # a walking workgroup: this item's last two tiles stage the NEXT item's K(0), K(1), V^T(0)
        nk, nv = next_kv
        nkmem = np.zeros(kbytes, np.uint8)
        nkb = bf16_round(nk).astype(np.uint16)
        for r in range(S):
            nkmem[r * ld_qk * 2: r * ld_qk * 2 + 256] = nkb[r].view(np.uint8)
        nvt = np.zeros((128, s_pad), np.uint16)
        nvb = bf16_round(nv).astype(np.uint16)
        for key in range(S):
            nvt[:, (key & ~15) + PERM16.index(key & 15)] = nvb[key]
        nvmem = nvt.reshape(-1).view(np.uint8).copy()
        glob["%[rskn]"], glob["%[rsvn]"] = (nkmem, kbytes), (nvmem, nvmem.size)
    st = build(fold, **stream_kw) if stream_kw is not None else product(fold)
    qs = np.asarray(q, np.float32)
    if fold:
        qs = bf16_to_f32(bf16_round(qs * c))          # scale * log2(e) folded into the q fragments (one more bf16 rounding here; the
                                                      # kernel folds it into the q preparation's single rounding)
    qb = bf16_round(qs)

    def scalars(w):
        return {"%[c]": int(np.float32(c).view(np.uint32)), "%[npairs]": (nkv - 2) // 2, "%[ktile]": 64 * ld_qk * 2,
                "%[nvalid]": S - (nkv - 1) * 64, "%[ldsw]": w * 4096}

    def lane_inputs(w):
        l = np.arange(64)
        hh = l >> 5
        out = {}
        for ks in range(8):
            out[AKL(ks)] = ((l & 31) * 256 + (((2 * ks + hh) ^ (l & 15)) << 4)).astype(np.uint32)
        for kg in range(4):
            out[AVL(kg)] = ((l & 31) * 128 + (((2 * kg + hh) ^ (((l & 31) >> 1) & 7)) << 4)).astype(np.uint32)
        for i in range(4):
            cc = w * 4 + i
            krow = cc * 4 + (l >> 4)
            out[KOFF(i)] = (krow * (ld_qk * 2) + (((l & 15) ^ (krow & 15)) * 16)).astype(np.uint32)
            vrow = cc * 8 + (l >> 3)
            vslot = (l & 7) ^ ((vrow >> 1) & 7)
            out[VOFF(i)] = ((vrow * s_pad + vslot * 8) * 2).astype(np.uint32)
        return out

    def q_frags(w):
        l = np.arange(64)
        out = {}
        for g in range(2):
            row = w * 64 + 32 * g + (l & 31)
            for ks in range(8):
                col = 16 * ks + 8 * (l >> 5)
                words = np.empty((4, 64), np.uint32)
                for r in range(4):
                    words[r] = qb[row, col + 2 * r] | (qb[row, col + 2 * r + 1] << 16)
                out[QA(g, ks)] = words
        return out

    emu = Emu(st, glob, scalars, lane_inputs, q_frags)
    if lds_from is not None:
        # the previous item of a walking workgroup left this item's first tiles in the LDS (the kernel waits for them and passes a barrier
        # in front of the statement): take the LDS image as it is
        emu.lds[:] = lds_from.lds
        for reg in range((2 * KT + 2 * VT) // 1024):
            emu.region_write[reg] = dict(epoch=-1, landed_epoch=-1)
    # stage_first: K(0) -> K buffer 0, V^T(0) -> V^T buffer 0, K(1) -> K buffer 1 (every wave its four pieces each), landed + barrier
    for w in range(4 if lds_from is None else 0):
        li = lane_inputs(w)
        for i in range(4):
            for (mem, nrec, base, off) in ((kmem, kbytes, 0 * KT + (w * 4 + i) * 1024, li[KOFF(i)].astype(np.int64)),
                                           (vmem, vmem.size, 2 * KT + (w * 4 + i) * 1024, li[VOFF(i)].astype(np.int64)),
                                           (kmem, kbytes, 1 * KT + (w * 4 + i) * 1024, li[KOFF(i)].astype(np.int64) + 64 * ld_qk * 2)):
                for l in range(64):
                    o = int(off[l])
                    chunk = mem[o:o + 16] if o + 16 <= nrec else np.zeros(16, np.uint8)
                    emu.lds[base + 16 * l: base + 16 * l + 16] = chunk
                emu.region_write[base // 1024] = dict(epoch=-1, landed_epoch=-1)
    emu.run()
    Oq = np.zeros((256, 128), np.float32)
    lq = np.zeros(256, np.float32)
    for w, wv in enumerate(emu.waves):
        for g in range(2):
            lsum = wv.v[L(g)[1]].view(np.float32)
            if getattr(st, "lsum", False):          # the accumulator of the ones x P MFMAs: the whole row's sum in every lane half
                lsum = wv.a[LA(g)[1]].view(np.float32) * 0.5
            for l in range(64):
                qrow = w * 64 + 32 * g + (l & 31)
                if l < 32:
                    lq[qrow] = lsum[l] + lsum[l + 32]
                for dt in range(4):
                    acc = wv.a[O(g, dt)[1]: O(g, dt)[1] + 16, l].view(np.float32)
                    for r in range(16):
                        d = 32 * dt + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)
                        Oq[qrow, d] = acc[r]
    return Oq, lq, emu
