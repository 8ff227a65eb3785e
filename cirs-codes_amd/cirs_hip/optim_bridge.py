"""Bridge between the flat device Adam state (exp_avg / exp_avg_sq / step counters of csrc/ppo.hip and cirs_adam_step) and
torch.optim.Adam's state_dict format, so the reference's checkpoint code

    torch.save({'policy': policy.state_dict(), 'optim_RL': optim[0].state_dict(),
                'optim_state': optim[1].state_dict(), 'state_tracker': state_tracker.state_dict()}, path)

(CIRS-RL-kuaishou.py:340-358) writes -- and `optim.load_state_dict` restores -- the moments the device actually uses.
The torch optimisers themselves never step; they carry hyper-parameters and, through this bridge, the state."""
import types

import torch


def _slices(params, flat: torch.Tensor):
    """[(param, offset, numel)] of module parameters that are views into `flat`."""
    out = []
    base, n = flat.data_ptr(), flat.numel()
    for p in params:
        off = (p.data_ptr() - base) // 4
        assert 0 <= off and off + p.numel() <= n, "parameter is not a view of the flat device buffer"
        out.append((p, int(off), p.numel()))
    return out


def bind(optimizer, get_state, flat=None):
    """get_state() -> None (nothing trained yet) or dict(flat=, m=, v=, steps=callable(param_offset) -> int, set_steps=callable(list))
    Patches optimizer.state_dict / load_state_dict on the INSTANCE.
    The (parameter -> offset in the flat buffer) map is taken while the parameters still ARE views of it -- at bind time when `flat`
    is given, else at the first state access -- and kept: the reference's checkpoint line evaluates `policy.cpu().state_dict()`
    BEFORE `optim[0].state_dict()` (CIRS-RL-kuaishou.py:340-343), i.e. the modules' parameters have left the device by then."""
    cls = type(optimizer)
    cache = {}

    def slices_of(params, flat_buf):
        if "sl" not in cache:
            cache["sl"] = _slices(params, flat_buf)
        return cache["sl"]

    def unique_params():
        seen, out = set(), []
        for g in optimizer.param_groups:
            for p in g["params"]:
                if id(p) not in seen:
                    seen.add(id(p))
                    out.append(p)
        return out

    def state_dict(self):
        st = get_state()
        if st is not None:
            for p, off, n in slices_of(unique_params(), st["flat"]):
                self.state[p] = {"step": torch.tensor(float(st["steps"](off))),
                                 "exp_avg": st["m"][off:off + n].view(p.shape).clone(),
                                 "exp_avg_sq": st["v"][off:off + n].view(p.shape).clone()}
        return cls.state_dict(self)

    def load_state_dict(self, sd):
        cls.load_state_dict(self, sd)
        st = get_state(create=True)
        if st is None:
            return
        steps = []
        for p, off, n in slices_of(unique_params(), st["flat"]):
            s = self.state.get(p)
            if not s:
                continue
            st["m"][off:off + n].copy_(s["exp_avg"].reshape(-1).to(st["m"].device))
            st["v"][off:off + n].copy_(s["exp_avg_sq"].reshape(-1).to(st["v"].device))
            steps.append((off, int(float(s["step"]))))
        st["set_steps"](steps)

    if flat is not None:
        slices_of(unique_params(), flat)
    optimizer.state_dict = types.MethodType(state_dict, optimizer)
    optimizer.load_state_dict = types.MethodType(load_state_dict, optimizer)
    return optimizer
