"""CPU (needs hipcc): the register-resident Cholesky solver (csrc/chol_reg.h) issues its matrix instructions from inline asm so that the
accumulator stays the tile's register (through the builtin hipcc moves the result elsewhere and spills).  hipcc's hazard recogniser does
not look inside inline asm: the wait states between a matrix instruction's write of its accumulator and the first OTHER instruction that
reads or writes that register (11 for the 8-pass v_mfma_f32_16x16x4_f32 on gfx940-class parts: vector ALU, LDS and memory instructions
alike) are the source's own responsibility -- an `s_nop` pair behind the update loop.  This test reads the generated ISA of the two
kernels that instantiate the solver and counts them."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "distant_speech_recognition_amd", "csrc")
NEED = 11                                                    # wait states, XDL write VGPR -> any other access, 8 passes


def _regs(code):
    out = set()
    for a, b in re.findall(r"v\[(\d+):(\d+)\]", code):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", code):
        out.add(int(a))
    return out


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
@pytest.mark.parametrize("src,kernel", [("wpe_kernels.hip", "wpe_solve_reg_kernel"), ("mvdr_kernels.hip", "mvdr_solve_reg_kernel")])
def test_wait_states_behind_inline_asm_matrix_instructions(tmp_path, src, kernel):
    out = tmp_path / "k.s"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
                    "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", str(out)], check=True, capture_output=True, timeout=900)
    lines, on = [], False
    for line in open(out):
        if re.match(r"^_Z\w*%s\w*:" % kernel, line):
            on = True
        elif on and line.startswith(".Lfunc_end"):
            break
        elif on:
            lines.append(line)
    assert lines, "kernel %s not found in the ISA of %s" % (kernel, src)
    pending = {}                                             # accumulator register -> wait states since the matrix instruction that wrote it
    in_asm, n_mfma, n_checked = False, 0, 0
    for line in lines:
        if "#ASMSTART" in line:
            in_asm = True
            continue
        if "#ASMEND" in line:
            in_asm = False
            continue
        code = line.split(";")[0].strip()
        if not code or code.endswith(":") or code.startswith("."):
            continue
        if code.startswith("v_mfma"):
            assert in_asm, "a matrix instruction outside inline asm: the solver pins its accumulators through asm"
            ops = [o.strip() for o in code.split(None, 1)[1].split(",")]
            dst = _regs(ops[0])
            assert dst == _regs(ops[3]), "accumulator in != out: %s" % code
            for r in dst:
                pending[r] = 0
            for r in list(pending):                         # (a matrix instruction is itself one wait state for the others)
                if r not in dst:
                    pending[r] += 1
            n_mfma += 1
            continue
        m = re.match(r"s_nop\s+(\d+)", code)
        ws = int(m.group(1)) + 1 if m else 1
        if not m:
            for r in _regs(code):
                if r in pending:
                    assert pending[r] >= NEED, "%s touches v%d %d wait states after the matrix instruction that wrote it" % (code, r, pending[r])
                    n_checked += 1
        for r in list(pending):
            pending[r] += ws
            if pending[r] >= 64:
                del pending[r]
    assert n_mfma >= 16, n_mfma
