"""Walk the compiled gfx950 code of the kernels that issue LDS reads from inline asm and check the one rule those kernels live by:

    between an asm `ds_read` and the `s_waitcnt lgkmcnt(n)` that retires it, no instruction may touch the read's destination registers.

hipcc treats the destination of an `asm volatile("ds_read_b128 %0, ...")` as defined when the statement ends, so any code IT generates after the
statement (a register copy for a live-range split, a v_accvgpr_write to park the value across a branch) may read registers the LDS has not
written yet.  Round 4 hit exactly that in attention_q64_kernel (a K fragment requested above the rescale branch was parked in AGPRs one
instruction later: NaNs that came and went with register allocation).  The kernels are written so that only asm statements sit between a read
and its wait; this script checks the compiler's output instead of trusting the source.  Second rule (same origin: code hipcc puts next to an asm statement):
no VALU instruction writes an A / B operand of an asm MFMA less than two wait states before it (check_mfma_operands).

    python scripts/check_asm_loads.py [source.hip [kernel-name-substring ...]]      default: csrc/attention.hip attention_q64, then csrc/gemm_bf16.hip gemm_bf16_deep

Model: the LGKM queue of one wave, in order (ds_read / ds_write / ds_bpermute / ds_swizzle / s_load each add one entry).  `s_waitcnt lgkmcnt(n)`
retires the oldest entries until n remain.  Scalar loads share the counter and retire out of order, which only makes a counted wait more
conservative for the LDS reads (outstanding LDS + outstanding SMEM <= n implies outstanding LDS <= n), so a kernel-argument s_load that hipcc
sinks between two asm statements is harmless; it is modelled as an entry without destination registers.  The walk is linear over each
function's text and forgets the queue at an unconditional branch: the checked regions are straight-line, compiler-managed kernels with
branchy LDS code can show false positives and are not what this is for.  Exit status 1 when a violation is found."""
import os, re, subprocess, sys, tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "domain-rag_amd", "csrc")
REG = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])")


def regs(text):
    out = set()
    for kind, one, lo, hi in REG.findall(text):
        if one:
            out.add((kind, int(one)))
        else:
            out.update((kind, i) for i in range(int(lo), int(hi) + 1))
    return out


def compile_asm(src):
    hipcc = os.environ.get("HIPCC") or "/opt/rocm/bin/hipcc"
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=on", "--cuda-device-only", "-S", src, "-o", out]
        if os.environ.get("DRAG_EXPERIMENTS", "") not in ("", "0"):
            cmd.insert(1, "-DDRAG_EXPERIMENTS")
        subprocess.run(cmd, check=True, capture_output=True)
        return open(out).read()


def functions(asm):
    name, body = None, []
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", line)
        if m and not line.startswith(".") and not m.group(1).startswith("BB"):
            if name:
                yield name, body
            name, body = m.group(1), []
        elif line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            if name:
                yield name, body
            name, body = None, []
        elif name is not None:
            body.append(line)
    if name:
        yield name, body


def check(name, body):
    queue = []                    # [(line number, text, destination registers)]
    bad = []
    n_reads = 0
    for ln, raw in enumerate(body):
        text = raw.split(";")[0].strip()
        if not text or text.startswith(".") or text.endswith(":"):
            continue
        op = text.split()[0]
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", text)
            if m:
                keep = int(m.group(1))
                while len(queue) > keep:
                    queue.pop(0)
            continue
        touched = regs(text[len(op):])
        for qln, qtext, dst in queue:
            hit = touched & dst
            if hit:
                bad.append((ln, text, qln, qtext, sorted(hit)))
        if op.startswith("ds_"):
            dst = set()
            if op.startswith(("ds_read", "ds_bpermute", "ds_permute", "ds_swizzle")):
                dst = regs(text[len(op):].split(",")[0])
                n_reads += op.startswith("ds_read")
            queue.append((ln, text, dst))
        elif op.startswith(("s_load", "s_buffer_load")):
            queue.append((ln, text, set()))
        elif op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            queue = []            # what follows is reached from elsewhere
    return n_reads, bad


def check_mfma_operands(name, body):
    """second rule: an MFMA reads a VALU-written A / B operand correctly only two wait states after the write.  hipcc pads that for its own
    MFMAs; for one inside an asm statement it does not, and it may well restore a parked fragment (v_accvgpr_read) in the instruction right
    before the statement.  Every instruction counts one wait state, s_nop N counts N + 1."""
    ins = []
    for ln, raw in enumerate(body):
        text = raw.split(";")[0].strip()
        if text and not text.startswith(".") and not text.endswith(":"):
            ins.append((ln, text))
    bad, n_mfma = [], 0
    for k, (ln, text) in enumerate(ins):
        if not text.startswith("v_mfma"):
            continue
        n_mfma += 1
        ops = [o.strip() for o in text[len(text.split()[0]):].split(",")]
        src = regs(ops[1]) | regs(ops[2])
        waited, j = 0, k - 1
        while j >= 0 and waited < 2:
            pln, prev = ins[j]
            op = prev.split()[0]
            if op.startswith("v_") and not op.startswith("v_mfma"):
                hit = regs(prev[len(op):].split(",")[0]) & src
                if hit:
                    bad.append((ln, text, pln, prev, sorted(hit)))
            m = re.match(r"s_nop\s+(\d+)", prev)
            waited += int(m.group(1)) + 1 if m else 1
            j -= 1
    return n_mfma, bad


def run(src, wanted):
    asm = compile_asm(src)
    status, seen = 0, 0
    for name, body in functions(asm):
        if not any(w in name for w in wanted):
            continue
        seen += 1
        n_reads, bad = check(name, body)
        print(f"{name[:110]}: {n_reads} LDS reads, {len(bad)} violation(s)")
        for ln, text, qln, qtext, hit in bad[:12]:
            print(f"    +{ln}: `{text}` touches {hit} of the outstanding `{qtext}` (+{qln})")
        status |= bool(bad)
        n_mfma, bad = check_mfma_operands(name, body)
        print(f"{' ' * min(len(name), 110)}  {n_mfma} MFMAs, {len(bad)} operand(s) written less than two wait states before")
        for ln, text, pln, prev, hit in bad[:12]:
            print(f"    +{ln}: `{text[:80]}` reads {hit} written by `{prev}` (+{pln})")
        status |= bool(bad)
    if not seen:
        print(f"no kernel matching {wanted} in {src}")
        return 2
    return status


def main(argv):
    if len(argv) > 1:
        return run(os.path.abspath(argv[1]), argv[2:] or [""])
    return run(os.path.join(CSRC, "attention.hip"), ["attention_q64"]) | run(os.path.join(CSRC, "gemm_bf16.hip"), ["gemm_bf16_deep"])


if __name__ == "__main__":
    sys.exit(main(sys.argv))
