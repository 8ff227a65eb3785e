"""Epoch log line of the RL training scripts (behaviour of reference util/utils.py:83-137, LoggerCallback_Policy).

After every epoch the trainer hands the merged result dict of the three test collectors (free browsing "", "NX_0_" and
"NX_<k>_") to `on_epoch_end`; the callback derives, per collector, trajectory length, trajectory reward and their ratio
(the "ctr" of the paper's tables), formats coverage with five decimals, carries the ifeat_* feature-domination entries over
and logs one line "Epoch: [e], Info: [{...}]" through logzero's logger (the script's logzero.logfile receives it); the dict is also
returned / kept in `last_results`.  create_dir, LoggerCallback_Update and LoggerCallback_RL (reference util/utils.py:14-80) live
here as well."""
import os

from logzero import logger   # the real logzero, or the stand-in package of this mirror (same `logger` / `logfile` protocol)


def create_dir(create_dirs):
    """Create the listed directories, parents first as listed (reference util/utils.py:14-24: os.mkdir per entry, an existing
    directory is left alone)."""
    for d in create_dirs:
        if not os.path.exists(d):
            logger.info("Create dir: %s" % d)
            try:
                os.mkdir(d)
            except FileExistsError:
                print("The dir [{}] already existed".format(d))


class LoggerCallback_Update:
    """Trainer callback that writes one log line per epoch (reference util/utils.py:30-56; the NAS upload is not part of the path)."""

    def __init__(self, logger_path):
        self.LOCAL_PATH = logger_path

    def on_epoch_begin(self, epoch, **kwargs):
        pass

    def on_train_begin(self, **kwargs):
        pass

    def on_train_end(self, **kwargs):
        pass

    def on_epoch_end(self, epoch, logs=None, **kwargs):
        logger.info("Epoch: [{}], Info: [{}]".format(epoch, logs))


class LoggerCallback_RL(LoggerCallback_Update):
    """Epoch line of CIRS-RL-taobao.py (reference util/utils.py:60-80): trajectory length, trajectory reward and their ratio."""

    def on_epoch_end(self, epoch, logs=None, **kwargs):
        logs = logs if logs is not None else kwargs.get("results")
        episodes = logs["n/ep"]
        length = logs["n/st"] / episodes
        reward = logs["rew"]
        result = {"num_test": episodes, "len_tra": length, "R_tra": reward, "ctr": f"{reward / length:.5f}"}
        self.last_results = result
        logger.info("Epoch: [{}], Info: [{}]".format(epoch, result))
        return result


class LoggerCallback_Policy:
    def __init__(self, logger_path, force_length):
        self.LOCAL_PATH = logger_path
        self.force_length = force_length
        self.last_results = None

    # the trainer's callback protocol (core/trainer/onpolicy.py)
    def on_train_begin(self, **kwargs):
        pass

    def on_train_end(self, **kwargs):
        pass

    def on_epoch_begin(self, epoch, **kwargs):
        pass

    def _collector_summary(self, results, prefix):
        episodes = results["n/ep"]
        length = results[prefix + "n/st"] / episodes
        reward = results[prefix + "rew"]
        summary = {"num_test": episodes,
                   prefix + "CV": "%.5f" % results[prefix + "CV"],
                   prefix + "CV_turn": "%.5f" % results[prefix + "CV_turn"],
                   prefix + "ctr": "%.5f" % (reward / length),
                   prefix + "len_tra": length,
                   prefix + "R_tra": reward}
        marker = prefix + "ifeat_"
        summary.update({k: v for k, v in results.items() if k.startswith(marker)})
        return summary

    def on_epoch_end(self, epoch, results=None, **kwargs):
        line = {}
        for prefix in ("", "NX_0_", "NX_%s_" % self.force_length):
            line.update(self._collector_summary(results, prefix))
        self.last_results = line
        logger.info("Epoch: [{}], Info: [{}]".format(epoch, line))
        return line
