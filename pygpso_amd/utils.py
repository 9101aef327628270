"""Small shared definitions (labels, file extensions, logger set-up)."""
import logging
from enum import Enum, unique

JSON_EXT = ".json"
PKL_EXT = ".pkl"
LOG_DATETIME_FORMAT = "%Y-%m-%d %H:%M:%S"


@unique
class PointLabels(Enum):
    """Where a score came from -- same names/values as the reference (gpso/utils.py:17-25)."""

    not_assigned = 0
    evaluated = 1  # the objective function was run at this point
    gp_based = 2  # UCB estimate from the GP surrogate


def set_logger(log_level=logging.INFO):
    """Root logger in the reference's line format (the golden traces are in this format)."""
    root = logging.getLogger()
    root.setLevel(log_level)
    for h in list(root.handlers):
        root.removeHandler(h)
    handler = logging.StreamHandler()
    handler.setFormatter(logging.Formatter("[%(asctime)s] %(levelname)s: %(message)s", LOG_DATETIME_FORMAT))
    root.addHandler(handler)
