"""gpurun_out/parity_margins.json (written by the test session: tests/conftest.py close()) -> a markdown table of the observed errors of the
golden comparisons next to their bars.     python tools/parity_margins.py [in.json] [out.md]"""
import json
import os
import sys
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity_margins.json")
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "parity_margins.md")
rows = OrderedDict()
for r in json.load(open(src)):
    # parameter tensors of one comparison family fold into one row
    key = r["what"]
    for fam in ("learn_opts: ", "learn: ", "full update, tracker post-Adam: ", "two updates r2_", "two updates "):
        if key.startswith(fam) and ("." in key[len(fam):]):
            key = fam + "post-update parameters (all tensors)"
            break
    a = rows.setdefault(key, dict(n=0, abs=0.0, rel=0.0, bar=0.0, rtol=r["rtol"], atol=r["atol"]))
    a["n"] += r["n"]; a["abs"] = max(a["abs"], r["max_abs_err"]); a["rel"] = max(a["rel"], r["max_rel_err"] or 0.0)
    a["bar"] = max(a["bar"], r["bar_used"])
with open(dst, "w") as f:
    f.write("Observed error of every golden / restatement comparison that goes through `tests/conftest.py: close()` (both head paths), one GPU session.\n\n")
    f.write("| comparison | elements | max abs err | max rel err | bar (rtol, atol) | largest share of the bar used |\n|---|---|---|---|---|---|\n")
    for k, a in rows.items():
        f.write(f"| {k} | {a['n']} | {a['abs']:.2e} | {a['rel']:.2e} | {a['rtol']:g}, {a['atol']:g} | {a['bar']:.3f} |\n")
print(open(dst).read())
