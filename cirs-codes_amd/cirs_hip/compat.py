"""Fallbacks for third-party packages the reference's entry points import but this image does not ship (gym, logzero, tensorboard).

The mirror keeps `CIRS-RL-kuaishou.py`'s import lines resolvable WITHOUT any explicit install() call:
  * `gym`, `logzero` are small top-level packages inside cirs-codes_amd/ (found through the same PYTHONPATH entry as `core`,
    `tianshou`, ...); each first looks for a REAL distribution of its name elsewhere on sys.path and, if there is one, hands the
    import over to it (`defer_to_real`), so an installed gym / logzero always wins.
  * `torch.utils.tensorboard` exists in torch but raises ImportError when the `tensorboard` package is missing.  The entry points
    import `gym` (line 10) and `core.*` (line 15) before `torch.utils.tensorboard` (line 18), so `ensure_tensorboard()` -- called
    from those packages' __init__ -- can publish a minimal SummaryWriter stand-in first.  It writes every scalar as one JSON line
    to <log_dir>/scalars.jsonl (enough for tianshou.utils.BasicLogger's write / restore_data protocol)."""
import importlib.machinery
import importlib.util
import json
import os
import sys
import types


def defer_to_real(name, own_dir):
    """If a real distribution of `name` is importable from a sys.path entry other than the mirror's, load it in place of the
    stand-in and return the module; else None.  own_dir: the directory that holds the stand-in package."""
    own_parent = os.path.realpath(os.path.dirname(own_dir))
    paths = [p for p in sys.path if os.path.realpath(p or os.getcwd()) != own_parent]
    spec = importlib.machinery.PathFinder.find_spec(name, paths)
    if spec is None or spec.loader is None:
        return None
    for k in [k for k in sys.modules if k == name or k.startswith(name + ".")]:
        del sys.modules[k]
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


class SummaryWriter:
    """Stand-in for torch.utils.tensorboard.SummaryWriter when `tensorboard` is not installed: scalars -> <log_dir>/scalars.jsonl."""

    def __init__(self, log_dir=None, **_):
        self.log_dir = log_dir or "runs"
        os.makedirs(self.log_dir, exist_ok=True)
        self._path = os.path.join(self.log_dir, "scalars.jsonl")
        self._fh = open(self._path, "a", buffering=1)   # line-buffered: readers (restore_data, tests) see every record

    def add_scalar(self, tag, scalar_value, global_step=None, walltime=None, **_):
        self._fh.write(json.dumps({"tag": tag, "value": float(scalar_value), "step": None if global_step is None else int(global_step)}) + "\n")

    def add_text(self, tag, text_string, global_step=None, **_):
        self._fh.write(json.dumps({"tag": tag, "text": str(text_string), "step": global_step}) + "\n")

    def scalars(self, tag):
        """[(step, value)] logged under `tag` so far (what BasicLogger.restore_data reads back)."""
        self.flush()
        out = []
        with open(self._path) as fh:
            for line in fh:
                rec = json.loads(line)
                if rec.get("tag") == tag and "value" in rec:
                    out.append((rec["step"], rec["value"]))
        return out

    def flush(self):
        self._fh.flush()

    def close(self):
        self._fh.close()


def ensure_tensorboard():
    """Publish the SummaryWriter stand-in as torch.utils.tensorboard iff the real one cannot be imported."""
    if "torch.utils.tensorboard" in sys.modules:
        return sys.modules["torch.utils.tensorboard"]
    if importlib.util.find_spec("tensorboard") is not None:
        return None   # the real package is there: leave torch's module alone
    mod = types.ModuleType("torch.utils.tensorboard")
    mod.SummaryWriter = SummaryWriter
    mod.__cirs_stand_in__ = True
    sys.modules["torch.utils.tensorboard"] = mod
    try:
        import torch.utils
        torch.utils.tensorboard = mod
    except ImportError:
        pass
    return mod
