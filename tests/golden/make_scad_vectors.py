#!/usr/bin/env python3
"""Extracts the known-answer vectors of the reference's OpenSCAD interpreter tests
(crates/openscad/src/interpreter/tests.rs: `assert_output_trim(source, expected)` and `assert_output(source,
expected)` calls) into tests/golden/scad_vectors.json as data: {"source", "expected", "trim", "line"}.
Needs /root/reference; the JSON it writes is what the tests read."""
import json
import os
import re
import sys

REF = "/root/reference/crates/openscad/src/interpreter/tests.rs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scad_vectors.json")

_LIT = r'(r#"(?:.*?)"#|r"(?:[^"]*)"|"(?:[^"\\]|\\.)*")'
_CALL = re.compile(r"assert_output(_trim)?\(\s*" + _LIT + r"\s*,\s*" + _LIT + r"\s*,?\s*\)", re.S)


def unquote(lit: str) -> str:
    if lit.startswith('r#"'):
        return lit[3:-2]
    if lit.startswith('r"'):
        return lit[2:-1]
    body = lit[1:-1]
    out, i = [], 0
    while i < len(body):
        ch = body[i]
        if ch == "\\":
            nxt = body[i + 1]
            if nxt == "\n":  # line continuation: skips the newline and the next line's leading whitespace
                i += 2
                while i < len(body) and body[i] in " \t\n":
                    i += 1
                continue
            out.append({"n": "\n", "t": "\t", "r": "\r", "0": "\0", '"': '"', "'": "'", "\\": "\\"}[nxt])
            i += 2
        else:
            out.append(ch)
            i += 1
    return "".join(out)


def main():
    text = open(REF).read()
    vectors = []
    for m in _CALL.finditer(text):
        if text[text.rfind("\n", 0, m.start()) + 1:m.start()].strip().startswith("//"):
            continue  # commented out in the reference
        vectors.append({"source": unquote(m.group(2)), "expected": unquote(m.group(3)), "trim": m.group(1) is not None,
                        "line": text.count("\n", 0, m.start()) + 1})
    json.dump(vectors, open(OUT, "w"), indent=1)
    print(len(vectors), "vectors ->", OUT)


if __name__ == "__main__":
    sys.exit(main())
