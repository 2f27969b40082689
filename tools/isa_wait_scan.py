"""Scan the device assembly hipcc emits for global loads that are waited for one at a time.

    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -I include -I distant_speech_recognition_amd/csrc \
          distant_speech_recognition_amd/csrc/wpe_kernels.hip -o wpe.s
    python tools/isa_wait_scan.py wpe.s [kernel-name-fragment]

A "serial load" is a global / buffer / flat load that is followed, before the next load and within a few instructions, by
`s_waitcnt vmcnt(0)`: the wavefront pays a full memory round trip for that one load.  Guarded loads (`in_range ? p[i] : 0`) whose
result is consumed at once compile to exactly that -- a branch around every load, one destination register, a wait behind each.
The WPE lag-product kernel spent 40 % of its cycles in eight such round trips per tile until its prefetch was rewritten to load raw
values from clamped addresses into registers of their own (round 6, profiles/r06_wpe_lagprod_phases.txt); the register solver's
tile loads were one round trip per slot for the same reason.  Read-modify-write epilogues (load, add, store) show up here too and
are harmless when they run once per workgroup.  Per kernel the script prints the number of serial loads, the longest RUN of
consecutive serial loads (what a hot loop must not have) and the number of loads altogether."""
import re
import sys

LOAD = ("global_load", "buffer_load", "flat_load", "scratch_load")


def kernels(text):
    parts = re.split(r"\n(_Z\w+):", text)
    for k in range(1, len(parts), 2):
        body = parts[k + 1]
        end = body.find(".Lfunc_end")
        yield parts[k], (body[:end] if end > 0 else body)


def scan(body, window=5):
    ins = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith((";", "."))]
    serial = total = run = longest = 0
    for i, l in enumerate(ins):
        if not l.startswith(LOAD):
            continue
        total += 1
        hit = False
        for j in range(i + 1, min(i + 1 + window, len(ins))):
            if ins[j].startswith(LOAD):
                break
            if ins[j].startswith("s_waitcnt") and "vmcnt(0)" in ins[j]:
                hit = True
                break
        if hit:
            serial += 1
            run += 1
            longest = max(longest, run)
        else:
            run = 0
    return serial, longest, total


def batches(body, min_len, opcode=None):
    """number of maximal runs of >= min_len loads (optionally of one opcode) issued with no `s_waitcnt vmcnt(..)` between them"""
    ins = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith((";", "."))]
    n = run = 0
    for l in ins:
        if l.startswith(LOAD):
            if opcode is None or re.match(opcode + r"\s", l):
                run += 1
        elif l.startswith("s_waitcnt") and "vmcnt" in l:
            n += run >= min_len
            run = 0
    return n + (run >= min_len)


def serial_of(body, opcode, window=5):
    """serial loads of one opcode (e.g. 'global_load_dword')"""
    ins = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith((";", "."))]
    n = 0
    for i, l in enumerate(ins):
        if re.match(opcode + r"\s", l):
            for j in range(i + 1, min(i + 1 + window, len(ins))):
                if ins[j].startswith(LOAD):
                    break
                if ins[j].startswith("s_waitcnt") and "vmcnt(0)" in ins[j]:
                    n += 1
                    break
    return n


if __name__ == "__main__":
    text = open(sys.argv[1]).read()
    frag = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = [(scan(body), name) for name, body in kernels(text) if frag in name]
    print("serial  longest-run  loads  kernel")
    for (serial, longest, total), name in sorted(rows, reverse=True):
        if serial:
            print("%6d  %11d  %5d  %s" % (serial, longest, total, name[:120]))
