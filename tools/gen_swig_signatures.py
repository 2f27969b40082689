#!/usr/bin/env python
"""Generate distant_speech_recognition_amd/btk20cpp/_signatures.py from the reference's SWIG interface files.

The reference builds its Python wrappers with %feature("kwargs") (btk20_src/*/*.i), so scripts may call every method with the C++
parameter NAMES the .i files declare.  This script reads those declarations (names and literal defaults only -- interface data, no
code) for the classes the engine binds and writes them as a table; run it in the build container, where /root/reference exists:

    python tools/gen_swig_signatures.py > distant_speech_recognition_amd/btk20cpp/_signatures.py
"""
import os
import re
import sys

REF = "/root/reference/btk20_src"
FILES = ["feature/feature.i", "modulated/modulated.i", "beamformer/beamformer.i", "postfilter/postfilter.i",
         "dereverberation/dereverberation.i", "stream/stream.i"]
CLASSES = ["SampleFeature", "OverSampledDFTAnalysisBank", "OverSampledDFTSynthesisBank", "SnapShotArray", "SpectralMatrixArray",
           "SubbandBeamformer", "SubbandDS", "SubbandGSC", "SubbandGSCRLS", "SubbandMVDR", "SubbandMVDRGSC", "ZelinskiPostFilter",
           "McCowanPostFilter", "LefkimmiatisPostFilter", "MultiChannelWPEDereverberation", "MultiChannelWPEDereverberationFeature",
           "SingleChannelWPEDereverberationFeature", "PyVectorFloatFeatureStream", "PyVectorComplexFeatureStream"]
SKIP = {"next", "reset", "is_end", "isEnd", "size", "frame_no", "operator->", "__iter__", "name", "current"}


def strip_comments(t):
    t = re.sub(r"/\*.*?\*/", " ", t, flags=re.S)
    t = re.sub(r"//[^\n]*", " ", t)
    return re.sub(r"^\s*#.*$", " ", t, flags=re.M)


def block_after(t, i):
    """text between the brace that opens at or after i and its partner"""
    a = t.index("{", i)
    d, j = 0, a
    while True:
        d += t[j] == "{"
        d -= t[j] == "}"
        if d == 0:
            return t[a + 1:j], j
        j += 1


def split_top(s, sep=","):
    out, d, cur = [], 0, ""
    for ch in s:
        if ch in "(<[{":
            d += 1
        elif ch in ")>]}":
            d -= 1
        if ch == sep and d == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def literal(d):
    d = d.strip()
    if d in ("true", "false"):
        return d == "true"
    if d == "NULL":
        return None
    if re.fullmatch(r'"[^"]*"', d):
        return d[1:-1]
    try:
        return int(d, 0)
    except ValueError:
        pass
    try:
        return float(d.rstrip("fF"))
    except ValueError:
        return Ellipsis            # an expression (e.g. sndfile::SF_FORMAT_WAV|...): optional, but its value is the binding's business


def params(plist):
    res = []
    for p in split_top(plist):
        p = p.strip()
        if not p or p == "void":
            continue
        dflt = "required"
        if "=" in p:
            p, d = p.split("=", 1)
            dflt = literal(d)
        m = re.search(r"([A-Za-z_]\w*)\s*(\[\s*\])?\s*$", p.strip())
        toks = re.findall(r"[A-Za-z_]\w*", p)
        name = m.group(1) if m and len(toks) > 1 else None      # a lone type name is an unnamed parameter
        res.append((name, dflt))
    return res


def main():
    methods, ctors = {}, {}
    for f in FILES:
        t = strip_comments(open(os.path.join(REF, f)).read())
        for m in re.finditer(r"\bclass\s+(\w+)\s*(?::[^{;]*)?\{", t):
            cname = m.group(1)
            base = cname[:-3] if cname.endswith("Ptr") else cname
            if base not in CLASSES:
                continue
            body, _ = block_after(t, m.start())
            if cname.endswith("Ptr"):
                for e in re.finditer(r"%extend\s*\{", body):
                    ext, _ = block_after(body, e.start())
                    c = re.search(r"\b" + cname + r"\s*\(", ext)
                    if c:
                        depth, j = 0, c.end() - 1
                        k = j
                        while True:
                            depth += ext[k] == "("
                            depth -= ext[k] == ")"
                            if depth == 0:
                                break
                            k += 1
                        ctors.setdefault(cname, params(ext[j + 1:k]))
                continue
            while re.search(r"\{[^{}]*\}", body):                               # inline bodies `{ return ...; }` end a declaration
                body = re.sub(r"\{[^{}]*\}", ";", body)
            body = re.sub(r"%\w+\s*\([^)]*\)\s*\w+\s*;", " ", body)           # %feature("kwargs") name;
            body = re.sub(r"\b(public|private|protected)\s*:", " ", body)
            for st in split_top(body, ";"):
                st = " ".join(st.split())
                mm = re.match(r"(?:virtual\s+|static\s+)?(?:[\w:<>]+[\s\*&]+)+?(\w+)\s*\((.*)\)\s*(?:const)?$", st)
                if not mm or mm.group(1) in SKIP or mm.group(1) == base or mm.group(1).startswith("~"):
                    continue
                methods.setdefault(cname + "Ptr", {}).setdefault(mm.group(1), params(mm.group(2)))
    out = ['"""Parameter names of the reference\'s SWIG interface, GENERATED by tools/gen_swig_signatures.py from btk20_src/*/*.i',
           '(%feature("kwargs"): scripts may call every method with these names).  (name, default) per parameter; default "required" = no',
           'default, Ellipsis = optional with a default the .i file writes as an expression (left to the binding).  Do not edit."""', "",
           "METHODS = {"]
    for c in sorted(methods):
        out.append("    %r: {" % c)
        for name in sorted(methods[c]):
            if methods[c][name]:
                out.append("        %r: %r," % (name, methods[c][name]))
        out.append("    },")
    out += ["}", "", "CTORS = {"]
    for c in sorted(ctors):
        out.append("    %r: %r," % (c, ctors[c]))
    out += ["}"]
    sys.stdout.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
