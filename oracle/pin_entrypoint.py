#!/usr/bin/env python
"""Pins the SURFACE the reference's entry points use (test infrastructure; run in the dev container where /root/reference exists).

    python oracle/pin_entrypoint.py            -> tests/golden/entrypoint_surface.json

For CIRS-RL-kuaishou.py and CIRS-RL-taobao.py the script AST-parses the file (nothing is executed, no source text is stored) and
records NAMES ONLY:
  * imports       every (module, name) of `import x` / `from x import y` at module level, with the line it is on
  * calls         for every call inside the file whose callee resolves to an imported name (or an attribute of one, e.g.
                  `KuaishouEnv.load_mat`, `gym.make`, `torch.optim.Adam`): the dotted callee, the number of positional
                  arguments, the keyword names, whether *args / **kwargs are forwarded, and the line
  * cli           the argparse option strings with their defaults (the configuration the paper's runs use; literals only)
tests/test_entrypoint_surface.py then asserts that every recorded import resolves against cirs-codes_amd/ and that every recorded
call's positional count / keyword set binds to the mirror's callable.  North-star: "CIRS-RL-kuaishou.py runs unmodified"."""
import ast
import json
import os
import sys

REF = os.environ.get("CIRS_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPTS = ["CIRS-RL-kuaishou.py", "CIRS-RL-taobao.py"]


def dotted(node):
    parts = []
    while isinstance(node, ast.Attribute):
        parts.append(node.attr)
        node = node.value
    if isinstance(node, ast.Name):
        parts.append(node.id)
        return ".".join(reversed(parts))
    return None


def literal(node):
    try:
        return ast.literal_eval(node)
    except Exception:  # noqa: BLE001
        return None


def pin(script):
    tree = ast.parse(open(os.path.join(REF, script), encoding="utf-8").read())
    imports, bound = [], {}
    for node in tree.body:
        if isinstance(node, ast.Import):
            for a in node.names:
                imports.append({"module": a.name, "name": None, "line": node.lineno})
                bound[(a.asname or a.name).split(".")[0]] = a.name if a.asname else a.name.split(".")[0]
        elif isinstance(node, ast.ImportFrom):
            for a in node.names:
                imports.append({"module": node.module, "name": a.name, "line": node.lineno})
                bound[a.asname or a.name] = f"{node.module}:{a.name}"
    calls, cli = [], []
    for node in ast.walk(tree):
        if not isinstance(node, ast.Call):
            continue
        name = dotted(node.func)
        if name is None:
            continue
        head = name.split(".")[0]
        if name.endswith(".add_argument"):
            opts = [literal(a) for a in node.args]
            kw = {k.arg: literal(k.value) for k in node.keywords if k.arg in ("default", "dest", "action", "nargs")}
            cli.append({"options": opts, **kw})
            continue
        if name.endswith(".set_defaults"):
            cli.append({"set_defaults": {k.arg: literal(k.value) for k in node.keywords}})
            continue
        if head not in bound:
            continue
        calls.append({"callee": name, "origin": bound[head], "n_positional": sum(not isinstance(a, ast.Starred) for a in node.args),
                      "keywords": [k.arg for k in node.keywords if k.arg is not None],
                      "star_args": any(isinstance(a, ast.Starred) for a in node.args),
                      "star_kwargs": any(k.arg is None for k in node.keywords), "line": node.lineno})
    calls.sort(key=lambda c: (c["line"], c["callee"]))
    return {"imports": imports, "calls": calls, "cli": cli}


def main():
    out = {"source": "AST of the reference's entry points (names only; oracle/pin_entrypoint.py)", "scripts": {s: pin(s) for s in SCRIPTS}}
    path = os.path.join(ROOT, "tests", "golden", "entrypoint_surface.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    n_imp = sum(len(v["imports"]) for v in out["scripts"].values())
    n_call = sum(len(v["calls"]) for v in out["scripts"].values())
    print(f"wrote {path}: {n_imp} imports, {n_call} calls")


if __name__ == "__main__":
    sys.exit(main())
