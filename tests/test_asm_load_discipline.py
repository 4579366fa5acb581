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


BROKEN_KERNEL = r"""
#include <hip/hip_runtime.h>
// an asm LDS read whose destination the compiler's own code consumes before the wait that retires it: rule 1 of isa_check.py
extern "C" __global__ void broken_asm_read_kernel(float* out) {
  __shared__ float s[64];
  s[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  float v;
  const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) float*)s + threadIdx.x * 4;
  asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr));
  out[threadIdx.x] = v * 2.0f;
  asm volatile("s_waitcnt lgkmcnt(0)");
}
"""


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="hipcc not found")
def test_build_refuses_an_object_that_breaks_the_rules(tmp_path):
    """build.py runs the rules on the assembly every checked object is assembled from: a deliberately broken kernel makes build_library raise,
    leaves no object, no stamp and no (stale) library behind, and the report names the compiler it ran under"""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("_drag_build_under_test", os.path.join(ROOT, "domain-rag_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    csrc, build, libdir = tmp_path / "csrc", tmp_path / "build", tmp_path / "lib"
    csrc.mkdir(); libdir.mkdir()
    (csrc / "broken.hip").write_text(BROKEN_KERNEL)
    (libdir / "libdomainrag_hip.so").write_bytes(b"a library linked from an earlier state of the sources")
    b.CSRC, b.BUILD, b.LIBDIR, b.LIB = str(csrc), str(build), str(libdir), str(libdir / "libdomainrag_hip.so")
    b.ISA.CHECKED = {"broken.hip": ["broken_asm_read_kernel"]}
    with pytest.raises(b.IsaCheckError) as e:
        b.build_library(verbose=False)
    assert "touches" in str(e.value) and "ds_read_b32" in str(e.value) and "object NOT built" in str(e.value)
    assert not (build / "broken.o").exists() and not (build / "broken.o.sha").exists()
    assert not (libdir / "libdomainrag_hip.so").exists()
    rep = json.loads((build / "broken.isa_check.json").read_text())
    assert rep["status"] == 1 and rep["kernels_seen"] == 1 and "clang version" in rep["hipcc"]
    # a checked source without any kernel of the expected name is an error too (a renamed kernel must not drop out of the check silently)
    b.ISA.CHECKED = {"broken.hip": ["no_such_kernel"]}
    with pytest.raises(b.IsaCheckError):
        b.build_library(verbose=False)
    # with the wait in front of the use the same source builds and links
    (csrc / "broken.hip").write_text(BROKEN_KERNEL.replace('  out[threadIdx.x] = v * 2.0f;\n  asm volatile("s_waitcnt lgkmcnt(0)");',
                                                             '  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v));\n  out[threadIdx.x] = v * 2.0f;'))
    b.ISA.CHECKED = {"broken.hip": ["broken_asm_read_kernel"]}
    b.build_library(verbose=False)
    assert (build / "broken.o").exists() and (libdir / "libdomainrag_hip.so").exists()
    assert json.loads((build / "broken.isa_check.json").read_text())["status"] == 0


def test_the_product_objects_carry_a_clean_report():
    """what build() left next to the shipped objects: every checked source has a report with status 0 from the compiler that built it"""
    import json
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import check_asm_loads as chk
    build = os.path.join(ROOT, "domain-rag_amd", "build")
    if not os.path.exists(os.path.join(build, "attention.o")):
        pytest.skip("the library has not been built in this tree")
    for src, wanted in chk.CHECKED.items():
        rep = json.load(open(os.path.join(build, src[:-4] + ".isa_check.json")))
        assert rep["status"] == 0 and rep["kernels_seen"] >= 1 and rep["kernels"] == wanted, rep
        assert ", 0 violation(s)" in "\n".join(rep["report"])
