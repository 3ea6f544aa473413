#!/usr/bin/env python3
"""The Rust `extern "C"` block for include/amdzk.h, generated — every function the header declares, in header order,
so that INTEGRATION.md's FFI section cannot drift from the ABI (tests/test_capi_symbols.py checks that the block in
INTEGRATION.md is exactly this output, and that libamdzk.so exports each name).
Usage: python tools/gen_rust_ffi.py            print the block
       python tools/gen_rust_ffi.py --update   rewrite the block between the markers in INTEGRATION.md"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "amdzk.h")
DOC = os.path.join(ROOT, "INTEGRATION.md")
BEGIN, END = "<!-- BEGIN GENERATED FFI (tools/gen_rust_ffi.py) -->", "<!-- END GENERATED FFI -->"

OPAQUE = {"amdzk_ctx": "Ctx", "amdzk_srs": "Srs", "amdzk_domain": "Domain", "amdzk_pk": "Pk", "amdzk_circuit": "AmdzkCircuit"}
SCALAR = {"int": "c_int", "uint32_t": "u32", "int32_t": "i32", "uint64_t": "u64", "size_t": "usize", "uint8_t": "u8", "float": "f32",
          "double": "f64", "char": "c_char", "void": "c_void"}


def strip(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = "\n".join(ln for ln in text.splitlines() if not ln.lstrip().startswith("#"))
    text = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)  # struct bodies
    text = re.sub(r'extern\s+"C"\s*\{', " ", text)
    return text


def rust_type(ctype):
    """`const uint64_t* const*` -> `*const *const u64`; the base type first, then one pointer level per `*`."""
    toks = re.findall(r"\w+|\*", ctype.replace("struct ", ""))
    base = [t for t in toks if t not in ("const", "*")][0]
    rb = OPAQUE.get(base) or SCALAR[base]
    # constness of each level: walk left to right; `const` binds to what precedes it, or to the base when it leads
    levels = []  # True = const, for the base then each '*'
    cur_const = False
    seen_base = False
    for t in toks:
        if t == "const":
            if not seen_base:
                cur_const = True
            else:
                levels[-1] = True
        elif t == "*":
            levels.append(False)
        else:
            seen_base = True
            levels.append(cur_const)
    out = rb
    # levels[0] = base constness, levels[i] = constness of pointer i itself; pointer i points at level i-1
    for i in range(1, len(levels)):
        out = ("*const " if levels[i - 1] else "*mut ") + out
    return out


def functions():
    text = strip(open(HEADER).read())
    out = []
    for stmt in text.split(";"):
        stmt = " ".join(stmt.split())
        m = re.match(r"^(.*?)\b(amdzk_\w+)\s*\((.*)\)$", stmt)
        if not m or "typedef" in stmt:
            continue
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3).strip()
        args = []
        if params and params != "void":
            for p in params.split(","):
                p = p.strip()
                arr = re.search(r"\[\s*\d*\s*\]$", p)
                if arr:
                    p = p[:arr.start()].strip()
                mm = re.match(r"^(.*?)(\w+)$", p)
                ctype, pname = mm.group(1).strip(), mm.group(2)
                if arr:
                    ctype += "*"
                args.append((pname if pname not in ("in", "type", "fn") else pname + "_", rust_type(ctype)))
        out.append((name, args, None if ret == "void" else rust_type(ret)))
    return out


def block():
    lines = ["extern \"C\" {"]
    for name, args, ret in functions():
        sig = "    pub fn %s(%s)%s;" % (name, ", ".join("%s: %s" % a for a in args), " -> " + ret if ret else "")
        if len(sig) > 150:  # wrap long signatures at argument boundaries
            head = "    pub fn %s(" % name
            cur, rows = head, []
            for i, a in enumerate(args):
                piece = "%s: %s%s" % (a[0], a[1], ", " if i + 1 < len(args) else "")
                if len(cur) + len(piece) > 148:
                    rows.append(cur.rstrip())
                    cur = " " * len(head)
                cur += piece
            rows.append(cur + ")" + (" -> " + ret if ret else "") + ";")
            sig = "\n".join(rows)
        lines.append(sig)
    lines.append("}")
    return "\n".join(lines)


def main():
    b = block()
    if "--update" in sys.argv:
        doc = open(DOC).read()
        i, j = doc.index(BEGIN), doc.index(END)
        doc = doc[:i + len(BEGIN)] + "\n```rust\n" + b + "\n```\n" + doc[j:]
        open(DOC, "w").write(doc)
        print("INTEGRATION.md updated: %d functions" % len(functions()))
    else:
        print(b)


if __name__ == "__main__":
    main()
