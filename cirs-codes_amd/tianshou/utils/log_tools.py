"""BaseLogger / BasicLogger (reference tianshou/tianshou/utils/log_tools.py:11-189): the trainer's logging protocol
(core/trainer/onpolicy.py: log_train_data / log_test_data / log_update_data / save_data / restore_data).

`write` goes to the writer's add_scalar(key, y, global_step=x) -- a real torch.utils.tensorboard.SummaryWriter or the stand-in of
cirs_hip.compat.  restore_data reads the last "save/*" steps back: from the tensorboard event files when the tensorboard package is
there, from the stand-in writer's own record otherwise."""
from abc import ABC, abstractmethod
from typing import Any, Callable, Optional, Tuple


class BaseLogger(ABC):
    def __init__(self, writer: Any) -> None:
        super().__init__()
        self.writer = writer

    @abstractmethod
    def write(self, key: str, x: int, y, **kwargs: Any) -> None:
        pass

    def log_train_data(self, collect_result: dict, step: int) -> None:
        pass

    def log_update_data(self, update_result: dict, step: int) -> None:
        pass

    def log_test_data(self, collect_result: dict, step: int) -> None:
        pass

    def save_data(self, epoch: int, env_step: int, gradient_step: int,
                  save_checkpoint_fn: Optional[Callable[[int, int, int], None]] = None) -> None:
        pass

    def restore_data(self) -> Tuple[int, int, int]:
        pass


class BasicLogger(BaseLogger):
    def __init__(self, writer, train_interval: int = 1000, test_interval: int = 1, update_interval: int = 1000,
                 save_interval: int = 1) -> None:
        super().__init__(writer)
        self.train_interval, self.test_interval = train_interval, test_interval
        self.update_interval, self.save_interval = update_interval, save_interval
        self.last_log_train_step = self.last_log_test_step = self.last_log_update_step = self.last_save_step = -1

    def write(self, key: str, x: int, y, **kwargs: Any) -> None:
        self.writer.add_scalar(key, y, global_step=x)

    def log_train_data(self, collect_result: dict, step: int) -> None:
        """`collect_result` gains "rew" / "len" in place (log_tools.py:124-141)."""
        if collect_result["n/ep"] > 0:
            collect_result["rew"] = collect_result["rews"].mean()
            collect_result["len"] = collect_result["lens"].mean()
            if step - self.last_log_train_step >= self.train_interval:
                for key in ("n/ep", "rew", "len"):
                    self.write("train/" + key, step, collect_result[key])
                self.last_log_train_step = step

    def log_test_data(self, collect_result: dict, step: int) -> None:
        """`collect_result` gains "rew", "rew_std", "len", "len_std" in place (log_tools.py:143-164)."""
        assert collect_result["n/ep"] > 0
        rews, lens = collect_result["rews"], collect_result["lens"]
        stats = dict(rew=rews.mean(), rew_std=rews.std(), len=lens.mean(), len_std=lens.std())
        collect_result.update(stats)
        if step - self.last_log_test_step >= self.test_interval:
            for key in ("rew", "len", "rew_std", "len_std"):
                self.write("test/" + key, step, stats[key])
            self.last_log_test_step = step

    def log_update_data(self, update_result: dict, step: int) -> None:
        if step - self.last_log_update_step >= self.update_interval:
            for k, v in update_result.items():
                self.write(k, step, v)
            self.last_log_update_step = step

    def save_data(self, epoch: int, env_step: int, gradient_step: int,
                  save_checkpoint_fn: Optional[Callable[[int, int, int], None]] = None) -> None:
        if save_checkpoint_fn and epoch - self.last_save_step >= self.save_interval:
            self.last_save_step = epoch
            save_checkpoint_fn(epoch, env_step, gradient_step)
            self.write("save/epoch", epoch, epoch)
            self.write("save/env_step", env_step, env_step)
            self.write("save/gradient_step", gradient_step, gradient_step)

    def _last_step(self, tag):
        if hasattr(self.writer, "scalars"):          # stand-in writer (cirs_hip.compat.SummaryWriter)
            items = self.writer.scalars(tag)
            if not items:
                raise KeyError(tag)
            return items[-1][0]
        from tensorboard.backend.event_processing import event_accumulator
        ea = event_accumulator.EventAccumulator(self.writer.log_dir)
        ea.Reload()
        return ea.scalars.Items(tag)[-1].step

    def restore_data(self) -> Tuple[int, int, int]:
        try:
            epoch = self._last_step("save/epoch")
            self.last_save_step = self.last_log_test_step = epoch
            gradient_step = self._last_step("save/gradient_step")
            self.last_log_update_step = gradient_step
        except KeyError:
            epoch, gradient_step = 0, 0
        try:
            env_step = self._last_step("save/env_step")
            self.last_log_train_step = env_step
        except KeyError:
            env_step = 0
        return epoch, env_step, gradient_step
