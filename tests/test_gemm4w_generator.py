"""The K loop of gemm_bf16_w4p is generated text (scripts/gen/gemm4w_kloop.py -> domain-rag_amd/csrc/gemm4w_kloop.h): the committed header must be
what the generator prints, and the product pieces must have the structure the kernel's correctness argument rests on (CPU only: text checks)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "domain-rag_amd", "csrc", "gemm4w_kloop.h")


def _macros():
    text = open(HEADER).read()
    out = {}
    for m in re.finditer(r"#define (\w+) \\\n((?:  \".*\" ?\\?\n)+)", text):
        body = "".join(re.findall(r'"(.*)"', line)[0] for line in m.group(2).splitlines())
        out[m.group(1)] = [ln for ln in body.replace("\\t", "").split("\\n") if ln]
    return out


def test_the_committed_header_is_the_generators_output():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gen", "gemm4w_kloop.py")], capture_output=True, text=True, check=True)
    assert r.stdout == open(HEADER).read(), "regenerate: python scripts/gen/gemm4w_kloop.py > domain-rag_amd/csrc/gemm4w_kloop.h"


def _ksteps(lines):
    """split a piece of the stream into K-steps (128 MFMAs each)"""
    steps, cur, n = [], [], 0
    for ln in lines:
        cur.append(ln)
        if ln.startswith("v_mfma"):
            n += 1
            if n % 128 == 0:
                steps.append(cur); cur = []
    if cur and steps:
        steps[-1] += cur
    return steps


def test_structure_of_the_product_stream():
    m = _macros()
    for name in ("G4W_P_FIRST", "G4W_P_PAIR0", "G4W_P_LOOP", "G4W_P_TAIL", "G4W_D_STAGE0_NOWAIT"):
        assert name in m, name
    for piece, nsteps in (("G4W_P_PAIR0", 2), ("G4W_P_LOOP", 2), ("G4W_P_TAIL", 2)):
        steps = _ksteps(m[piece])
        assert len(steps) == nsteps, (piece, len(steps))
        for si, st in enumerate(steps):
            mf = [ln for ln in st if ln.startswith("v_mfma")]
            assert len(mf) == 128
            # every accumulator tile exactly once per k-half, k-half 0 (X in v[0:31], W in v[32:63]) before k-half 1 (v[64:127]): the order of
            # gemm_bf16_t256 per output element, i.e. the same bits
            for half in (0, 1):
                accs = []
                for ln in mf[64 * half:64 * half + 64]:
                    d, w, x, c = [t.strip() for t in ln.split(None, 1)[1].split(",")]
                    accs.append(d)
                    wlo, xlo = int(re.match(r"v\[(\d+):", w).group(1)), int(re.match(r"v\[(\d+):", x).group(1))
                    assert 64 * half + 32 <= wlo < 64 * half + 64 and 64 * half <= xlo < 64 * half + 32, ln
                    zero_start = piece == "G4W_P_PAIR0" and si == 0 and half == 0
                    assert c == ("0" if zero_start else d), ln          # only the tile's first k-half starts from the constant 0
                assert sorted(accs) == sorted(f"a[{4 * i}:{4 * i + 3}]" for i in range(64))
            reads = [ln for ln in st if ln.startswith("ds_read_b128")]
            dmas = [i for i, ln in enumerate(st) if ln.startswith("buffer_load_dwordx4")]
            last = piece == "G4W_P_TAIL" and si == 1
            assert len(reads) == (16 if last else 32) and len(dmas) == 16
            for i in dmas:              # m0 is written at least one instruction before the piece that uses it (no hardware interlock)
                assert " lds" in st[i] and not st[i - 1].startswith("s_add_u32 m0")
                assert any(ln.startswith("s_add_u32 m0") for ln in st[max(0, i - 6):i])
            # two barriers per K-step: the first behind the k-half-1 fragment reads (lgkmcnt(0) right before it), the second between the
            # k-halves behind a COUNTED vmcnt that leaves the pieces issued since the first barrier in flight
            bars = [i for i, ln in enumerate(st) if ln == "s_barrier"]
            assert len(bars) == (1 if last else 2)
            assert st[bars[0] - 1] == "s_waitcnt lgkmcnt(0)"
            issued_before_b1 = sum(1 for i in dmas if i < bars[0])
            assert issued_before_b1 == 0, "the stage buffer is refilled only after every wave has read it"
            if not last:
                mm = re.match(r"s_waitcnt vmcnt\((\d+)\)", st[bars[1] - 1])
                assert mm and int(mm.group(1)) == sum(1 for i in dmas if i < bars[1])
            # fragment reads never target the registers the MFMAs of the same k-half read
            nmf = 0
            for ln in st:
                if ln.startswith("v_mfma"):
                    nmf += 1
                elif ln.startswith("ds_read_b128"):
                    lo = int(re.match(r"ds_read_b128 v\[(\d+):", ln).group(1))
                    half_now = 0 if nmf <= 64 else 1
                    assert (lo >= 64) == (half_now == 0), (piece, si, ln)
    # the tail refills with the NEXT tile's operands, the loop with this tile's
    assert all("%[rsa2]" in ln or "%[rsw2]" in ln for ln in m["G4W_P_TAIL"] if ln.startswith("buffer_load"))
    assert all("%[rsa]" in ln or "%[rsw]" in ln for ln in m["G4W_P_LOOP"] if ln.startswith("buffer_load"))


def test_every_register_the_product_stream_writes_is_an_output_or_a_clobber():
    """ the statement of gemm_bf16_w4p declares a[0:255] as outputs, %[n2] / %[soff] as read-write operands and G4W_CLOBBERS +
    "scc" as clobbers: every destination register named in the text must be one of those — m0 (written before each LDS-DMA piece) included"""
    m = _macros()
    text = open(HEADER).read()
    clob = set(re.findall(r'"(\w+)"', re.search(r"#define G4W_CLOBBERS (.*)", text).group(1)))
    assert "m0" in clob and {f"v{i}" for i in range(128)} <= clob
    src = open(os.path.join(ROOT, "domain-rag_amd", "csrc", "gemm_bf16.hip")).read()
    for stmt in re.findall(r"asm volatile\((G4W_D_STAGE0_NOWAIT|G4W_P_FIRST G4W_P_PAIR0 G4W_P_LOOP G4W_P_TAIL)(.*?)\);", src, re.S):
        assert '"m0"' in stmt[1] or "G4W_CLOBBERS" in stmt[1], stmt[0]
    written = set()
    for piece in ("G4W_P_FIRST", "G4W_P_PAIR0", "G4W_P_LOOP", "G4W_P_TAIL", "G4W_D_STAGE0_NOWAIT"):
        for ln in m[piece]:
            op = ln.split()[0]
            if op.endswith(":") or op in ("s_waitcnt", "s_barrier", "s_cbranch_scc1", "s_cmp_eq_u32", "s_cmp_lg_u32"):
                continue
            if op == "buffer_load_dwordx4":
                assert ln.rstrip().endswith(" lds"), ln          # LDS-DMA: no register destination
                continue
            dst = ln.split(None, 1)[1].split(",")[0].strip()
            written.add(dst)
    for d in written:
        mm = re.match(r"([av])\[(\d+):(\d+)\]$", d)
        if mm:
            lo, hi = int(mm.group(2)), int(mm.group(3))
            assert (mm.group(1) == "a" and hi <= 255) or (mm.group(1) == "v" and hi <= 127), d       # accumulators (outputs) | fragments (clobbers)
        else:
            assert d in ("m0", "%[soff]", "%[n2]"), d
