"""ISA-level guard for the kernels whose inner loops are inline-asm instruction streams (attention_q64_kernel, gemm_bf16_deep ...).

hipcc treats the destination of an ``asm volatile("ds_read_b128 %0, ...")`` as defined when the statement ends, and pads VALU -> MFMA-operand
hazards only for MFMAs it emitted itself.  Both facts produced a silently wrong product kernel in round 4 (DESIGN.md "Round 4"), so the rules
these kernels live by are checked on the COMPILED code, by ``build.py`` on every object it ships (the ``.s`` the object is assembled from,
``-save-temps``), not on the source:

  1. between an asm ``ds_read`` and the ``s_waitcnt lgkmcnt(n)`` that retires it no instruction touches the read's destination registers
     (``check``);
  2. no VALU instruction writes an A / B operand of an MFMA less than two wait states before it (``check_mfma_operands``).

Model of rule 1: the LGKM queue of one wave, in order (ds_read / ds_write / ds_bpermute / ds_swizzle / s_load each add one entry);
``s_waitcnt lgkmcnt(n)`` retires the oldest entries until n remain.  Scalar loads share the counter and retire out of order, which only makes
a counted wait more conservative for the LDS reads, so a kernel-argument s_load that hipcc sinks between two asm statements is modelled as an
entry without destination registers.  The walk is linear over each function's text and forgets the queue at an unconditional branch: the
checked regions are straight-line; compiler-managed kernels with branchy LDS code can show false positives and are not what this is for.

``scripts/check_asm_loads.py`` is the command-line front end; ``CHECKED`` lists what build.py checks."""

import os, re, subprocess, sys, tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# source -> substrings of the kernel names build.py checks in the object it ships
CHECKED = {"attention.hip": ["attention_q64"], "gemm_bf16.hip": ["gemm_bf16_deep", "gemm_bf16_w4"]}
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


def check_asm_text(asm, wanted, out=print):
    """both rules over every function of `asm` whose name contains one of `wanted`: (exit status, kernels seen)"""
    status, seen = 0, 0
    for name, body in functions(asm):
        if not any(w in name for w in wanted):
            continue
        seen += 1
        n_reads, bad = check(name, body)
        out(f"{name[:110]}: {n_reads} LDS reads, {len(bad)} violation(s)")
        for ln, text, qln, qtext, hit in bad[:12]:
            out(f"    +{ln}: `{text}` touches {hit} of the outstanding `{qtext}` (+{qln})")
        status |= bool(bad)
        n_mfma, bad = check_mfma_operands(name, body)
        out(f"{' ' * min(len(name), 110)}  {n_mfma} MFMAs, {len(bad)} operand(s) written less than two wait states before")
        for ln, text, pln, prev, hit in bad[:12]:
            out(f"    +{ln}: `{text[:80]}` reads {hit} written by `{prev}` (+{pln})")
        status |= bool(bad)
    return status, seen


def run(src, wanted):
    status, seen = check_asm_text(compile_asm(src), wanted)
    if not seen:
        print(f"no kernel matching {wanted} in {src}")
        return 2
    return status


def hipcc_version(hipcc=None):
    """the compiler the check passed with (recorded next to the objects and in their digests)"""
    r = subprocess.run([hipcc or os.environ.get("HIPCC") or "/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True)
    return " | ".join(l.strip() for l in r.stdout.splitlines() if l.strip())[:400]
