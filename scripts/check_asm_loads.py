"""Command-line front end of domain-rag_amd/isa_check.py (the rules, the model and what build.py does with them are described there):

    python scripts/check_asm_loads.py [source.hip [kernel-name-substring ...]]      default: everything in isa_check.CHECKED

compiles the source to gfx950 assembly and walks the named kernels for (1) an instruction touching the destination of an outstanding asm
`ds_read`, (2) a VALU write to an MFMA operand less than two wait states before the MFMA.  Exit status 1 when a violation is found."""
import importlib.util, os, sys

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("_drag_isa_check", os.path.join(HERE, "..", "domain-rag_amd", "isa_check.py"))
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
regs, functions, check, check_mfma_operands, check_asm_text, compile_asm, run = (
    _mod.regs, _mod.functions, _mod.check, _mod.check_mfma_operands, _mod.check_asm_text, _mod.compile_asm, _mod.run)
CSRC, CHECKED = _mod.CSRC, _mod.CHECKED


def main(argv):
    if len(argv) > 1:
        return run(os.path.abspath(argv[1]), argv[2:] or [""])
    status = 0
    for src, wanted in CHECKED.items():
        status |= run(os.path.join(CSRC, src), wanted)
    return status


if __name__ == "__main__":
    sys.exit(main(sys.argv))
