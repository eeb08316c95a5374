"""Extract every `m.def("<schema>")` string of the reference's TORCH_LIBRARY_FRAGMENT(hpc, m) blocks
(/root/reference/src/**/entry.cc) into tests/golden/ref_schemas.json, and the signature (argument names + defaults) of
every public function / method of the reference's Python package (/root/reference/hpc/*.py) into
tests/golden/ref_py_signatures.json.  Run in the build container (the reference tree does not exist on the GPU box);
tests/test_schemas.py compares every torch.ops.hpc.* schema and every public hpc.* signature with these fixtures.

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


def py_signatures(path):
    """{name or Class.method: [[argument, default source or None], ...]} of the public callables defined in a module"""
    import ast

    def args_of(fn):
        a = fn.args
        names = [x.arg for x in a.posonlyargs + a.args]
        defaults = [ast.unparse(d) for d in a.defaults]
        nd = len(names) - len(defaults)
        out = [[nm, defaults[i - nd] if i >= nd else None] for i, nm in enumerate(names)]
        if a.vararg:
            out.append(["*" + a.vararg.arg, None])
        out += [[x.arg, ast.unparse(d) if d is not None else None] for x, d in zip(a.kwonlyargs, a.kw_defaults)]
        return out

    out = {}
    for node in ast.parse(Path(path).read_text()).body:
        if isinstance(node, ast.FunctionDef) and not node.name.startswith("_") and not node.name.endswith("fake"):
            out[node.name] = args_of(node)
        if isinstance(node, ast.ClassDef) and not node.name.startswith("_"):
            for m in node.body:
                if isinstance(m, ast.FunctionDef) and (not m.name.startswith("_") or m.name == "__init__"):
                    out[node.name + "." + m.name] = args_of(m)
    return out


def main():
    if not REF.exists():
        sys.exit("reference tree not present")
    sigs = {}
    for f in sorted((REF.parent / "hpc").glob("*.py")):
        for name, sig in py_signatures(f).items():
            sigs[name] = {"module": f.name, "args": sig}
    (OUT.parent / "ref_py_signatures.json").write_text(json.dumps(sigs, indent=1, sort_keys=True) + "\n")
    print(len(sigs), "python signatures")
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
