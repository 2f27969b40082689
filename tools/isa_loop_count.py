"""Instruction histogram of the loops of one kernel in the device assembly hipcc emits (-S --cuda-device-only).

    python tools/isa_loop_count.py fb512.s analysis512_bfz_kernelILi2ELi33231ELi16EfE

A loop is a backward branch (s_cbranch_* / s_branch to a label that lies above it); its body is the text between the label and
the branch.  bench.py's FUSED_ISA (the packed float32 instructions per wavefront and channel of the fused kernel's interior loop)
is the row of the loop that holds 15 global_load_dwordx2 -- the window loads of one channel."""
import collections
import re
import sys


def kernel_text(lines, name):
    start = None
    for i, l in enumerate(lines):
        if start is None and re.match(r"^_Z\w*%s\w*:" % re.escape(name), l):
            start = i
        elif start is not None and l.startswith("\t.end_amdhsa_kernel") or (start is not None and l.startswith(".Lfunc_end")):
            return lines[start:i]
    raise SystemExit("kernel %s not found" % name)


def loops(text):
    labels = {}
    for i, l in enumerate(text):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    out = []
    for i, l in enumerate(text):
        m = re.match(r"^\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            out.append((labels[m.group(1)], i))
    return out


def histogram(text, a, b):
    h = collections.Counter()
    for l in text[a:b + 1]:
        m = re.match(r"^\s+([a-z][a-z0-9_]+)", l)
        if m and not l.lstrip().startswith("."):
            h[m.group(1)] += 1
    return h


if __name__ == "__main__":
    lines = open(sys.argv[1]).read().split("\n")
    text = kernel_text(lines, sys.argv[2])
    keys = ["v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "global_load_dwordx2", "buffer_load_format_xy", "ds_read_b64", "ds_read_b128",
            "ds_write2_b64", "s_barrier"]
    print("loop(lines)  total  " + "  ".join(keys))
    for a, b in sorted(loops(text), key=lambda ab: ab[0] - ab[1]):
        h = histogram(text, a, b)
        if sum(h.values()) < 50:
            continue
        print("%5d-%5d  %5d  " % (a, b, sum(h.values())) + "  ".join("%*d" % (len(k), h[k]) for k in keys))
