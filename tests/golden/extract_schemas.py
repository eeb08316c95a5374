"""Extract every `m.def("<schema>")` string of the reference's TORCH_LIBRARY_FRAGMENT(hpc, m) blocks
(/root/reference/src/**/entry.cc) into tests/golden/ref_schemas.json.  Run in the build container (the reference
tree does not exist on the GPU box); tests/test_schemas.py compares every torch.ops.hpc.* schema with this fixture.

    python tests/golden/extract_schemas.py
"""
import json
import re
import sys
from pathlib import Path

REF = Path("/root/reference/src")
OUT = Path(__file__).resolve().parent / "ref_schemas.json"


def defs_of(text: str):
    """Yield (schema string, line number) of every m.def( "..." "..." ) call; adjacent C string literals are joined."""
    for m in re.finditer(r"\bm\.def\(", text):
        i = m.end()
        parts = []
        while True:
            while text[i] in " \t\r\n":
                i += 1
            if text[i] != '"':
                break
            j = i + 1
            while text[j] != '"' or text[j - 1] == "\\":
                j += 1
            parts.append(text[i + 1:j])
            i = j + 1
        if parts:
            yield "".join(parts), text.count("\n", 0, m.start()) + 1


def main():
    if not REF.exists():
        sys.exit("reference tree not present")
    out = {}
    for f in sorted(list(REF.rglob("*.cc")) + list(REF.rglob("*.cu"))):
        t = f.read_text()
        if "TORCH_LIBRARY" not in t:
            continue
        for schema, line in defs_of(t):
            name = schema.split("(", 1)[0].strip()
            assert name not in out, name
            # m.def("name", &fn): the schema is inferred from the C++ signature - recorded without a string
            out[name] = {"schema": schema if "(" in schema else None,
                         "at": "%s:%d" % (f.relative_to(REF.parent), line)}
    OUT.write_text(json.dumps(out, indent=1, sort_keys=True) + "\n")
    print(len(out), "schemas ->", OUT)


if __name__ == "__main__":
    main()
