"""`import logzero` / `from logzero import logger` for the reference's entry points and util/utils.py when the real package is not
installed: `logger` is a standard logging.Logger with a stderr handler in logzero's line format, `logfile(path)` adds a file
handler (CIRS-RL-kuaishou.py:131).  An installed logzero found elsewhere on sys.path replaces this package at import time."""
import logging
import os as _os

from cirs_hip import compat as _compat

_real = _compat.defer_to_real("logzero", _os.path.dirname(_os.path.abspath(__file__)))
if _real is None:
    __cirs_stand_in__ = True
    DEFAULT_FORMAT = "[%(levelname)1.1s %(asctime)s %(module)s:%(lineno)d] %(message)s"
    DEFAULT_DATE_FORMAT = "%y%m%d %H:%M:%S"
    logger = logging.getLogger("logzero_default")
    logger.setLevel(logging.DEBUG)
    logger.propagate = False
    if not logger.handlers:
        _h = logging.StreamHandler()
        _h.setFormatter(logging.Formatter(DEFAULT_FORMAT, DEFAULT_DATE_FORMAT))
        logger.addHandler(_h)
    _file_handler = None

    def logfile(filename, formatter=None, mode="a", maxBytes=0, backupCount=0, encoding=None, loglevel=None, disableStderrLogger=False):
        """Attach (or, with filename=None, detach) the log file of the default logger."""
        global _file_handler
        if _file_handler is not None:
            logger.removeHandler(_file_handler)
            _file_handler.close()
            _file_handler = None
        if filename:
            _file_handler = logging.FileHandler(filename, mode=mode, encoding=encoding)
            _file_handler.setFormatter(formatter or logging.Formatter(DEFAULT_FORMAT, DEFAULT_DATE_FORMAT))
            if loglevel is not None:
                _file_handler.setLevel(loglevel)
            logger.addHandler(_file_handler)
        if disableStderrLogger:
            for h in list(logger.handlers):
                if isinstance(h, logging.StreamHandler) and not isinstance(h, logging.FileHandler):
                    logger.removeHandler(h)

    def loglevel(level=logging.DEBUG, update_custom_handlers=False):
        logger.setLevel(level)

    def setup_logger(name=None, logfile=None, level=logging.DEBUG, formatter=None, **_):
        lg = logging.getLogger(name or "logzero")
        lg.setLevel(level)
        if logfile:
            fh = logging.FileHandler(logfile)
            fh.setFormatter(formatter or logging.Formatter(DEFAULT_FORMAT, DEFAULT_DATE_FORMAT))
            lg.addHandler(fh)
        return lg
