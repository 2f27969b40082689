"""CPU (needs hipcc): the matrix-core post-filter statistics kernel issues its snapshot loads from inline asm
(`global_load_dwordx2 v, v_off, s[base]`) and waits for them with a hand-placed `s_waitcnt vmcnt(0)` -- hipcc's own wait-count
insertion does not know that these loads are outstanding.  The source ties every loaded register to the wait (in/out operand of the
volatile asm that carries it), which keeps the compiler from *using* a value early, but a register copy inserted between a load and
the wait would silently move a stale value.  This test reads the generated ISA and asserts that no instruction between such a load and
the next vmcnt(0) wait touches the load's destination registers."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "distant_speech_recognition_amd", "csrc", "pf_kernels.hip")


def _regs(tok):
    """registers named by an operand token: v12 -> {12}; v[12:15] -> {12, 13, 14, 15}"""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_no_instruction_touches_an_asm_load_destination_before_its_wait(tmp_path):
    out = tmp_path / "pf.s"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                    "-S", "--cuda-device-only", SRC, "-o", str(out)], check=True, capture_output=True, timeout=900)
    kernels, cur, name = {}, None, None
    for line in open(out):
        m = re.match(r"^(_Z\w*bf_apply_stats2_mfma_kernel\w*):", line)
        if m:
            name, cur = m.group(1), []
            kernels[name] = cur
        elif cur is not None:
            if line.strip().startswith("s_endpgm"):
                cur = None
            else:
                cur.append(line)
    assert kernels, "no bf_apply_stats2_mfma_kernel instantiation in the generated ISA"
    checked = 0
    for name, lines in kernels.items():
        pending = set()                                     # destination registers of asm loads not yet waited for
        in_asm = False
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
            toks = re.findall(r"v\[\d+:\d+\]|v\d+", code)
            if in_asm and code.startswith("global_load_dwordx2"):
                # an inline-asm load (hipcc brackets inline asm with ;;#ASMSTART / ;;#ASMEND): hipcc's wait-count pass does not see it
                dst = _regs(toks[0])
                assert not (dst & pending), "%s: load into a register with a load still pending: %s" % (name, code)
                used = set().union(*[_regs(t) for t in toks[1:]]) if len(toks) > 1 else set()
                assert not (used & pending), "%s: %s reads a pending destination" % (name, code)
                pending |= dst
                checked += 1
                continue
            if re.match(r"s_waitcnt\b", code) and "vmcnt(0)" in code:
                pending.clear()
                continue
            touched = set().union(*[_regs(t) for t in toks]) if toks else set()
            assert not (touched & pending), "%s: `%s` touches v%s before the vmcnt(0) wait of its load" % (name, code, sorted(touched & pending))
    assert checked >= 16, "expected the asm loads of at least one instantiation, saw %d" % checked
