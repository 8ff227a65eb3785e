"""The logging protocol of the trainer (`log_train_data`, `log_test_data`, `log_update_data`, `save_data`, `restore_data`; contract: reference
tianshou/tianshou/utils/log_tools.py:84-189) over any writer with `add_scalar(tag, value, global_step=)`.  Written for this repository: one
`_Every` gate per channel decides whether a step is due; resuming reads the three "save/*" tags back from the writer."""
from typing import Callable, Optional, Tuple


class _Every:
    """Lets a step through when at least `gap` steps passed since the last one it let through."""

    def __init__(self, gap: int):
        self.gap, self.last = gap, -1

    def due(self, step: int) -> bool:
        if step - self.last < self.gap:
            return False
        self.last = step
        return True


class BaseLogger:
    def __init__(self, writer):
        self.writer = writer

    def write(self, key: str, x: int, y, **kwargs) -> None:
        self.writer.add_scalar(key, y, global_step=x)


class BasicLogger(BaseLogger):
    def __init__(self, writer, train_interval: int = 1000, test_interval: int = 1, update_interval: int = 1000, save_interval: int = 1):
        super().__init__(writer)
        self.gate = {"train": _Every(train_interval), "test": _Every(test_interval), "update": _Every(update_interval), "save": _Every(save_interval)}

    def _emit(self, channel: str, step: int, scalars: dict, prefix: str = ""):
        if self.gate[channel].due(step):
            for tag, value in scalars.items():
                self.write(prefix + tag, step, value)

    def log_train_data(self, collect_result: dict, step: int) -> None:
        if collect_result["n/ep"] > 0:       # the caller reads "rew" / "len" back from its own dict
            collect_result.update(rew=collect_result["rews"].mean(), len=collect_result["lens"].mean())
            self._emit("train", step, {k: collect_result[k] for k in ("n/ep", "rew", "len")}, "train/")

    def log_test_data(self, collect_result: dict, step: int) -> None:
        assert collect_result["n/ep"] > 0
        spread = {name: fn(collect_result[src]) for name, src, fn in (("rew", "rews", lambda a: a.mean()), ("len", "lens", lambda a: a.mean()),
                                                                      ("rew_std", "rews", lambda a: a.std()), ("len_std", "lens", lambda a: a.std()))}
        collect_result.update(spread)
        self._emit("test", step, spread, "test/")

    def log_update_data(self, update_result: dict, step: int) -> None:
        self._emit("update", step, update_result)

    def save_data(self, epoch: int, env_step: int, gradient_step: int, save_checkpoint_fn: Optional[Callable[[int, int, int], None]] = None) -> None:
        if save_checkpoint_fn and self.gate["save"].due(epoch):
            save_checkpoint_fn(epoch, env_step, gradient_step)
            for tag, n in (("save/epoch", epoch), ("save/env_step", env_step), ("save/gradient_step", gradient_step)):
                self.write(tag, n, n)

    def _recorded(self, tag: str) -> Optional[int]:
        """Step of the last scalar written under `tag`, or None."""
        if hasattr(self.writer, "scalars"):          # the stand-in writer (cirs_hip.compat.SummaryWriter) keeps its own record
            rows = self.writer.scalars(tag)
            return rows[-1][0] if rows else None
        from tensorboard.backend.event_processing import event_accumulator
        acc = event_accumulator.EventAccumulator(self.writer.log_dir)
        acc.Reload()
        return acc.scalars.Items(tag)[-1].step if tag in acc.Tags().get("scalars", ()) else None

    def restore_data(self) -> Tuple[int, int, int]:
        epoch, env_step, gradient_step = (self._recorded("save/" + t) for t in ("epoch", "env_step", "gradient_step"))
        if epoch is None or gradient_step is None:
            epoch = gradient_step = 0
        else:
            self.gate["save"].last = self.gate["test"].last = epoch
            self.gate["update"].last = gradient_step
        if env_step is not None:
            self.gate["train"].last = env_step
        return epoch, env_step or 0, gradient_step
