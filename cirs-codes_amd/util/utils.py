"""Log callbacks of the training scripts (reference util/utils.py:83-137 LoggerCallback_Policy): formats the per-epoch
result dict into the paper's log line (ctr = R_tra / len_tra, CV, CV_turn, ifeat_*) for the FB / NX_0 / NX_k collectors.
logzero is not a dependency here; a standard `logging` logger named "cirs" receives the same message text."""
import logging
import re

logger = logging.getLogger("cirs")


class LoggerCallback_Policy:
    def __init__(self, logger_path, force_length):
        self.LOCAL_PATH = logger_path
        self.force_length = force_length
        self.last_results = None

    def on_epoch_begin(self, epoch, **kwargs):
        pass

    def on_train_begin(self, **kwargs):
        pass

    def on_train_end(self, **kwargs):
        pass

    def on_epoch_end(self, epoch, results=None, **kwargs):
        def find_item_domination_results(prefix):
            pattern = re.compile(prefix + "ifeat_")
            return {k: v for k, v in results.items() if re.match(pattern, k)}

        def get_one_result(prefix):
            num_test = results["n/ep"]
            len_tra = results[prefix + "n/st"] / num_test
            R_tra = results[prefix + "rew"]
            ctr = R_tra / len_tra
            res = dict()
            res['num_test'] = num_test
            res[prefix + 'CV'] = f"{results[prefix + 'CV']:.5f}"
            res[prefix + 'CV_turn'] = f"{results[prefix + 'CV_turn']:.5f}"
            res[prefix + 'ctr'] = f"{ctr:.5f}"
            res[prefix + 'len_tra'] = len_tra
            res[prefix + 'R_tra'] = R_tra
            return res

        results_all = {}
        for prefix in ["", "NX_0_", f"NX_{self.force_length}_"]:
            results_all.update(get_one_result(prefix))
            results_all.update(find_item_domination_results(prefix))
        self.last_results = results_all
        logger.info("Epoch: [{}], Info: [{}]".format(epoch, results_all))
        return results_all
