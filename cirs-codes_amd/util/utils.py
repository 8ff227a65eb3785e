"""Epoch log line of the RL training scripts (behaviour of reference util/utils.py:83-137, LoggerCallback_Policy).

After every epoch the trainer hands the merged result dict of the three test collectors (free browsing "", "NX_0_" and
"NX_<k>_") to `on_epoch_end`; the callback derives, per collector, trajectory length, trajectory reward and their ratio
(the "ctr" of the paper's tables), formats coverage with five decimals, carries the ifeat_* feature-domination entries over
and logs one line "Epoch: [e], Info: [{...}]".  logzero is not a dependency here: the line goes to the standard `logging`
logger named "cirs" and the dict is also returned / kept in `last_results`."""
import logging

logger = logging.getLogger("cirs")


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
