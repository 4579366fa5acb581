"""The kernels that issue LDS reads from inline asm (attention_q64_kernel, gemm_bf16_deep) rely on one rule: nothing but asm statements sits
between a read and the s_waitcnt that retires it, because hipcc takes the destination for defined as soon as the statement ends and may copy
it.  scripts/check_asm_loads.py compiles the two sources to gfx950 assembly (no GPU needed, about a minute) and walks every such kernel for an
instruction that touches the destination of an outstanding read."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="hipcc not found")
def test_no_instruction_touches_an_outstanding_asm_lds_read():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_asm_loads.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "attention_q64_kernel" in r.stdout and "gemm_bf16_deep" in r.stdout
    assert "violation" in r.stdout and ", 0 violation(s)" in r.stdout


def test_the_checker_sees_a_copy_of_an_outstanding_read():
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import check_asm_loads as chk
    body = ["\tds_read_b128 v[68:71], v252 offset:0x4000", "\tv_accvgpr_write_b32 a83, v71", "\ts_waitcnt lgkmcnt(0)", "\tv_mov_b32_e32 v1, v70"]
    n, bad = chk.check("k", body)
    assert n == 1 and len(bad) == 1 and bad[0][1].startswith("v_accvgpr_write_b32")
    # a counted wait retires the oldest read only; a scalar load in the queue does not count as a destination
    body = ["\tds_read_b128 v[0:3], v9", "\ts_load_dwordx2 s[0:1], s[2:3], 0x0", "\tds_read_b128 v[4:7], v9", "\ts_waitcnt lgkmcnt(1)",
            "\tv_mov_b32_e32 v10, v0", "\tv_mov_b32_e32 v11, v4"]
    n, bad = chk.check("k", body)
    assert n == 2 and [b[1] for b in bad] == ["v_mov_b32_e32 v11, v4"]


def test_the_checker_sees_an_operand_restored_right_before_an_mfma():
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import check_asm_loads as chk
    mfma = "\tv_mfma_f32_32x32x16_bf16 v[82:97], v[124:127], a[132:135], v[82:97]"
    # round 4: hipcc restored a parked K fragment in the instruction before the statement; one s_waitcnt between is one wait state, not two
    n, bad = chk.check_mfma_operands("k", ["\tv_accvgpr_read_b32 v124, a80", "\ts_waitcnt lgkmcnt(1)", mfma])
    assert n == 1 and len(bad) == 1 and bad[0][3].startswith("v_accvgpr_read_b32 v124")
    n, bad = chk.check_mfma_operands("k", ["\tv_accvgpr_read_b32 v124, a80", "\ts_waitcnt lgkmcnt(1)", "\tv_fma_f32 v1, v2, s3, v4", mfma])
    assert n == 1 and not bad
    n, bad = chk.check_mfma_operands("k", ["\tv_accvgpr_write_b32 a133, v1", "\ts_nop 1", mfma])
    assert n == 1 and not bad
    n, bad = chk.check_mfma_operands("k", ["\tv_accvgpr_write_b32 a133, v1", "\ts_nop 0", mfma])
    assert n == 1 and len(bad) == 1
