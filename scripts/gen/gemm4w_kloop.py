"""Generator of the hand-placed K loop of gemm_bf16_w4 (csrc/gemm_bf16.hip): the 256 x 256 x 64 tile as FOUR waves x (128 x 128), one wave per
SIMD with the whole 512-entry register file — a third less LDS -> register traffic per flop than the 8-wave kernel (the vendor library's
MT256x256x64 kernel on this chip has the same shape: 256 threads, 130 KiB of LDS; profiles/r05_yardstick.log).

    python scripts/gen/gemm4w_kloop.py > domain-rag_amd/csrc/gemm4w_kloop.h

Register map of a wave (physical names are written into the text; the kernel binds its operands to them with {v[..]} / {a[..]} constraints):
  a[0:255]    accumulators, tile (mi, ni) at a[(8 mi + ni) 4 ...]            mi, ni = 0..7 (16 rows x 16 columns each)
  v[0:31]     X fragments (A operand rows) of k-half 0, mi at v[4 mi ...]    v[32:63]   W fragments of k-half 0
  v[64:95]    X fragments of k-half 1                                       v[96:127]  W fragments of k-half 1
  v[128:135]  global byte offsets of this wave's 8 X chunks (8 rows x 128 B each)     v[136:143] of its 8 W chunks
  v[144:151]  LDS read addresses [stage buffer][X k-half 0, X k-half 1, W k-half 0, W k-half 1]
  v[152:153]  LDS write address of this lane's 16 bytes of chunk 0, stage buffer 0 | 1      (form R only)
  v[160:223]  the wave's 16 chunks of ONE K-step in flight from global memory, chunk i at v[160 + 4 i ...]   (form R only)
One K-step (64 k) = 128 MFMAs of 16 cycles.  k-half 0: its 64 MFMAs with the 16 fragment reads of k-half 1 behind every second one; then
lgkmcnt(0) (+ vmcnt(0) in form D) and s_barrier; k-half 1: its 64 MFMAs with the 16 reads of the next K-step's k-half 0 (other stage buffer)
behind the first 32, and the stage buffer just left refilled:
  form D (LDS-DMA): 16 buffer_load ... lds of K-step t + 2, one behind every fourth MFMA — in flight for half to one K-step;
  form R (registers): chunk i of K-step t + 2 (requested a whole K-step ago) is written to LDS from its registers (ds_write_b128) and the same
         registers are requested again for K-step t + 3: every chunk is in flight for a full K-step whatever its place in the k-half, and what
         is in flight lives in registers the 8-wave kernel does not have (64 KiB per CU on top of the two 64 KiB stage buffers).
Same MFMA order per accumulator as gemm_bf16_t256 (k-half 0, then 1): same bits."""
import sys

A_BYTES = 256 * 128          # X region of a stage buffer
STAGE = 2 * A_BYTES
G0 = 160


def mfma(mi, ni, half, zero=False):
    acc = (8 * mi + ni) * 4
    x = 64 * half + 4 * mi
    w = 64 * half + 32 + 4 * ni
    c = "0" if zero else f"a[{acc}:{acc + 3}]"          # the first K-step of a tile starts its accumulators from the inline constant 0
    return f"v_mfma_f32_16x16x32_bf16 a[{acc}:{acc + 3}], v[{w}:{w + 3}], v[{x}:{x + 3}], {c}"


def reads(half, buf):
    """the 16 fragment reads of one k-half from stage buffer `buf`, W first (the half's first MFMAs need all W fragments and X[0])"""
    out = []
    addr_x, addr_w = 144 + half + 4 * buf, 146 + half + 4 * buf
    for ni in range(8):
        d = 64 * half + 32 + 4 * ni
        out.append(f"ds_read_b128 v[{d}:{d + 3}], v{addr_w} offset:{2048 * ni}")
    for mi in range(8):
        d = 64 * half + 4 * mi
        out.append(f"ds_read_b128 v[{d}:{d + 3}], v{addr_x} offset:{2048 * mi}")
    return out


def dmas(buf, nxt=False):
    """form D: this wave's 16 chunks of one K-step into stage buffer `buf`: m0 = the chunk's LDS address (one instruction between the write
    of m0 and its use: a wait state the hardware does not interlock), soffset = the K-step's byte offset (%[soff]).  nxt: the chunks of the
    NEXT tile of a persistent workgroup (its descriptors %[rsa2] / %[rsw2], its offsets in v[224:239])"""
    out = []
    for i in range(16):
        x = i < 8
        rs = ('%[rsa2]' if x else '%[rsw2]') if nxt else ('%[rsa]' if x else '%[rsw]')
        vo = (224 if nxt else 128) + (0 if x else 8) + (i & 7)
        out.append((f"s_add_u32 m0, %[ldsw], {buf * STAGE + (0 if x else A_BYTES) + (i & 7) * 1024}",
                    f"buffer_load_dwordx4 v{vo}, {rs}, %[soff] offen lds"))
    return out


def gload(i):
    x = i < 8
    return f"buffer_load_dwordx4 v[{G0 + 4 * i}:{G0 + 4 * i + 3}], v{(128 if x else 136) + (i & 7)}, {'%[rsa]' if x else '%[rsw]'}, %[soff] offen"


def lwrite(i, buf):
    x = i < 8
    return f"ds_write_b128 v{152 + buf}, v[{G0 + 4 * i}:{G0 + 4 * i + 3}] offset:{(0 if x else A_BYTES) + (i & 7) * 1024}"


def kstep(buf, form, refill=True, request=True, next_reads=True, barrier=True, every=4, start=3, write=True):
    t = []
    order = [(mi, ni) for mi in range(8) for ni in range(8)]
    r1 = reads(1, buf)
    for j, (mi, ni) in enumerate(order):          # k-half 0
        t.append(mfma(mi, ni, 0))
        if j % 2 == 0 and j // 2 < 16:
            t.append(r1[j // 2])
    t.append("s_waitcnt vmcnt(0) lgkmcnt(0)" if form == "D" else "s_waitcnt lgkmcnt(0)")
    if barrier:
        t.append("s_barrier")
    r0 = reads(0, 1 - buf) if next_reads else []
    d = dmas(buf) if (form == "D" and refill) else []
    n = 0
    for j, (mi, ni) in enumerate(order):          # k-half 1
        if form == "D" and j >= start - 1 and (j - (start - 1)) % every == 0 and n < len(d):
            t.append(d[n][0])                      # m0 one MFMA ahead of the piece that uses it
        t.append(mfma(mi, ni, 1))
        if j % 2 == 0 and j // 2 < len(r0):
            t.append(r0[j // 2])
        if j >= start and (j - start) % every == 0 and n < 16:
            if form == "D" and refill:
                t.append(d[n][1])
            elif form == "R" and refill:
                # chunk n of K-step t + 2 has landed when at most 15 younger requests are outstanding (requests return in order); its
                # registers go to LDS and are requested again at once: LDS instructions read their data registers at issue
                t.append(f"s_waitcnt vmcnt({15 if request else 15 - n})")
                if write:
                    t.append(lwrite(n, buf))
                if request:
                    t.append(gload(n))
            n += 1
    if refill and (form == "D" or request):
        t.append("s_add_u32 %[soff], %[soff], 128")
    t.append("s_waitcnt lgkmcnt(0)")
    return t


def kstep_d2(buf, refill=True, next_reads=True, b1=36, every=5, reads_every=2, last=None, nxt=False, advance=True, b2=True, zero=False):
    """form D2: TWO barriers per K-step.  B1 (behind MFMA b1 of k-half 0, once this wave's k-half-1 fragment reads have landed): every wave
    is done reading this stage buffer, so its refill (LDS-DMA of K-step t + 2) starts there, one piece behind every `every`-th MFMA through
    the rest of the K-step — a lower request rate and a longer flight than form D's burst inside k-half 1.  B2 (between the k-halves, as
    before): the pieces of K-step t + 1 have landed everywhere (a COUNTED vmcnt: the pieces of t + 2 issued since B1 stay in flight)."""
    t = []
    order = [(mi, ni) for mi in range(8) for ni in range(8)]
    r1 = reads(1, buf)
    r0 = reads(0, 1 - buf) if next_reads else []
    d = dmas(buf, nxt) if refill else []
    n = 0
    slots = [b1 + 2 + every * i for i in range(16)]          # global MFMA slots (0..127) behind which piece i goes
    if last is not None:                                      # ... or spread evenly up to slot `last`
        slots = [b1 + 2 + round(i * (last - b1 - 2) / 15) for i in range(16)]
    assert slots[-1] <= 126 and len(set(slots)) == 16, slots
    for g in range(128):
        half, j = divmod(g, 64)
        mi, ni = order[j]
        if refill and n < 16 and g == slots[n] - 1:
            t.append(d[n][0])                      # m0 one MFMA ahead
        t.append(mfma(mi, ni, half, zero and half == 0))
        if half == 0 and j % reads_every == 0 and j // reads_every < 16:
            t.append(r1[j // reads_every])
        if half == 1 and j % 2 == 0 and j // 2 < len(r0):
            t.append(r0[j // 2])
        if g == b1:
            t.append("s_waitcnt lgkmcnt(0)")
            t.append("s_barrier")
        if refill and n < 16 and g == slots[n]:
            t.append(d[n][1]); n += 1
        if g == 63 and b2:
            issued = n                              # pieces of t + 2 issued so far stay outstanding
            t.append(f"s_waitcnt vmcnt({issued})")
            t.append("s_barrier")
    if refill and advance:
        t.append("s_add_u32 %[soff], %[soff], 128")
    t.append("s_waitcnt lgkmcnt(0)")
    return t


def emit(name, lines):
    print(f"#define {name} \\")
    print(" \\\n".join('  "' + ln + '\\n\\t"' for ln in lines))
    print()


def loop_of(form, **kw):
    return ["s_cmp_eq_u32 %[n2], 0", "s_cbranch_scc1 L_g4w_tail_%=", "L_g4w_loop_%=:"] + kstep(0, form, **kw) + kstep(1, form, **kw) + \
           ["s_sub_u32 %[n2], %[n2], 1", "s_cmp_lg_u32 %[n2], 0", "s_cbranch_scc1 L_g4w_loop_%=", "L_g4w_tail_%=:"]


print("// generated by scripts/gen/gemm4w_kloop.py — do not edit; see the generator for the register map and the schedule")
# m0 is written before every LDS-DMA piece: nothing hipcc emits around the statements uses it today, but an LDS-DMA builtin, a readlane or a
# movrel next to them would otherwise be free to assume its own m0 value survives
print("#define G4W_CLOBBERS " + ", ".join(f'"v{i}"' for i in range(128)) + ', "m0"')
print("#define G4W_CLOBBERS_R " + ", ".join(f'"v{i}"' for i in list(range(128)) + list(range(G0, G0 + 64))) + ', "m0"')
print()
# ---- form D: prologue = K-steps 0, 1 by LDS-DMA (soffset 0, 128; %[soff] leaves at 256), steady loop over pairs of K-steps (%[n2] pairs),
# tail = the last two K-steps without refill
flat = lambda d: [x for pair in d for x in pair]
emit("G4W_D_STAGE0", flat(dmas(0)) + ["s_add_u32 %[soff], %[soff], 128"] + flat(dmas(1)) + ["s_add_u32 %[soff], %[soff], 128",
                                                                                             "s_waitcnt vmcnt(16)", "s_barrier"])
emit("G4W_FIRST_READS", reads(0, 0) + ["s_waitcnt lgkmcnt(0)"])
emit("G4W_D_LOOP", loop_of("D"))
emit("G4W_D_TAIL", kstep(0, "D", refill=False) + kstep(1, "D", refill=False, next_reads=False))
# ---- form R: prologue = K-step 0 through the registers into stage buffer 0, K-step 1 into buffer 1, K-step 2 requested; the loop's K-step t
# writes K-step t + 2 and requests t + 3 (the last iterations request up to one K-step past K: the descriptors are bounded, such reads
# return zeros or the next row's start and are never used); the tail's first K-step writes the last requested K-step without requesting
pro = []
for s in range(2):
    pro += [gload(i) for i in range(16)] + ["s_add_u32 %[soff], %[soff], 128", "s_waitcnt vmcnt(0)"] + [lwrite(i, s) for i in range(16)]
pro += [gload(i) for i in range(16)] + ["s_add_u32 %[soff], %[soff], 128", "s_waitcnt lgkmcnt(0)", "s_barrier"]
emit("G4W_R_STAGE0", pro)
emit("G4W_R_LOOP", loop_of("R"))
emit("G4W_R_TAIL", kstep(0, "R", request=False) + kstep(1, "R", refill=False, next_reads=False))
# timing ablations of form R (wrong values): no barrier | no refill at all | every chunk from the same, L2-resident K-step
def d2_loop(**kw):
    return ["s_cmp_eq_u32 %[n2], 0", "s_cbranch_scc1 L_g4w_tail_%=", "L_g4w_loop_%=:"] + kstep_d2(0, **kw) + kstep_d2(1, **kw) + \
           ["s_sub_u32 %[n2], %[n2], 1", "s_cmp_lg_u32 %[n2], 0", "s_cbranch_scc1 L_g4w_loop_%=", "L_g4w_tail_%=:"]
emit("G4W_D2_LOOP", d2_loop())
emit("G4W_D2_LOOP_A", d2_loop(b1=33, every=6))
emit("G4W_D2_LOOP_B", d2_loop(b1=22, reads_every=1, last=124))
emit("G4W_D2_LOOP_C", d2_loop(b1=26, reads_every=1, every=6))
emit("G4W_R_LOOP_NOBAR", loop_of("R", write=False))            # V2: requests and waits, no LDS writes
emit("G4W_R_LOOP_NOREFILL", loop_of("R", refill=False))

# ---- the persistent kernel (gemm_bf16_w4p): one asm statement per tile.  K-steps 0 and 1 of the tile are in flight / in LDS when it starts
# (staged by G4W_D_STAGE0 for the workgroup's first tile, by the previous tile's tail otherwise); the steady loop is form D2 (first barrier
# behind MFMA 33, a piece behind every 6th MFMA from there); the TAIL's two K-steps refill their stage buffers with K-steps 0 and 1 of the
# workgroup's NEXT tile (descriptors with zero records when there is none: no traffic), so the epilogue overlaps the next tile's loads.
BEST = dict(b1=33, every=6)
# (the wait for the tile's K-steps 0 and 1 is a statement of its own in the kernel: counted when the epilogue before it is the fast one)
emit("G4W_P_FIRST", ["s_barrier"] + reads(0, 0) + ["s_waitcnt lgkmcnt(0)", "s_movk_i32 %[soff], 256"])
# the tile's first pair of K-steps (accumulators start from 0 in the first one), then %[n2] more pairs in the loop
emit("G4W_P_PAIR0", kstep_d2(0, zero=True, **BEST) + kstep_d2(1, **BEST))
emit("G4W_P_LOOP", d2_loop(**BEST))
emit("G4W_P_TAIL", ["s_mov_b32 %[soff], 0"] + kstep_d2(0, nxt=True, advance=False, **BEST) +
     ["s_movk_i32 %[soff], 128"] + kstep_d2(1, nxt=True, advance=False, next_reads=False, b2=False, **BEST))
emit("G4W_D_STAGE0_NOWAIT", flat(dmas(0)) + ["s_add_u32 %[soff], %[soff], 128"] + flat(dmas(1)))
